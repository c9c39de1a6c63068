"""Dataset readers: the `ipts` contract of SuRF.forward from MVSNet-format scenes."""
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, DistributedSampler, RandomSampler, SequentialSampler

from .dtu import DTUDataset, TanksDataset

DATASETS = {"DTUDataset": DTUDataset, "TanksDataset": TanksDataset}


def collect_fn(data):
    return data[0]


def get_loader(conf, mode, distributed, num_workers=8):
    """datasets/__init__.py:16-43: batch size 1 (one scene + reference view per item), DistributedSampler under DDP."""
    name = conf.get_string("dataset_name")
    if name not in DATASETS:
        raise NotImplementedError(f"dataset_name {name!r}: surf_amd ships {sorted(DATASETS)} "
                                  "(BlendedMVS / ETH3D / the finetune variants of the reference are not built)")
    dataset = DATASETS[name](conf, mode)
    if mode == "finetune":
        return dataset
    if distributed:
        sampler = DistributedSampler(dataset, num_replicas=dist.get_world_size(), rank=dist.get_rank())
    else:
        sampler = RandomSampler(dataset) if mode == "train" else SequentialSampler(dataset)
    loader = DataLoader(dataset, 1, sampler=sampler, num_workers=num_workers, drop_last=(mode == "train"), pin_memory=False,
                        collate_fn=collect_fn)
    return loader, sampler, dataset
