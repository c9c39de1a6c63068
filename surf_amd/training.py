"""One optimisation step of the reference's runner (runner.py:152-165) as a function.

A train-mode `SuRF.forward` is differentiable (surf_amd.autograd: the HIP backward kernels behind two autograd nodes), so the
step is the reference's own sequence - forward, `Loss`, `loss.backward()`, `optimizer.step()` - and runner.py needs no
change to train with this package, with or without `DistributedDataParallel` (runner.py:102).  These helpers exist for
callers that do not use the runner (bench.py, the tests):

`train_step`     a volume-building model (generalisation training, configs[3]): every parameter group of surf.py:36-45.
`finetune_step`  a has_vol model (per-scene finetuning: implicit surface + the per-scene feature volumes).

Both average the gradients over the ranks themselves when a process group is up and the model is NOT wrapped in
DistributedDataParallel (`dist.all_reduce_gradients`: one flat bucket); under DDP its hooks have already done it.
Differentiated terms: every term of losses/loss.py - colour, eikonal, sparse-SDF, smooth (H.1), rendered depth, pseudo-SDF,
the patch-NCC term (`mfc_loss`), the per-stage photometric and pseudo-depth terms.
"""
import torch

from . import dist

# the differentiable per-ray outputs of a train-mode forward (tests pin the backward kernels through leaf copies of these)
LEAVES = ("color_fine", "render_depth", "gradient_error", "sparse_sdf", "ncc", "smooth_error")


def _is_ddp(model):
    return isinstance(model, torch.nn.parallel.DistributedDataParallel)


def _sync_gradients(model, optimizer):
    """Data-parallel step without DDP: average the gradients over the ranks (RCCL on GPUs, gloo in the CPU tests)."""
    if _is_ddp(model):
        return
    if dist._active():          # a group with peers, or one forced at world size 1 (dist.init_from_env(force=True))
        dist.all_reduce_gradients([p for g in optimizer.param_groups for p in g["params"]])


def _step(model, ipts, targets, loss_fn, optimizer, cos_anneal_ratio, step, mode, sync=True):
    outputs = model("train", ipts, cos_anneal_ratio=cos_anneal_ratio, step=step)          # runner.py:155
    out = loss_fn(outputs, targets, step=step, mode=mode)                                  # runner.py:159
    optimizer.zero_grad(set_to_none=True)
    out["loss"].backward()                                                                 # runner.py:163
    if sync:
        _sync_gradients(model, optimizer)
    optimizer.step()
    return {k: (float(v.detach()) if torch.is_tensor(v) else float(v)) for k, v in out.items()}


def finetune_step(model, ipts, targets, loss_fn, optimizer, cos_anneal_ratio=1.0, step=0):
    """runner.py:300-330 (finetune): any loss mode but "train" - no per-stage depth terms, the volumes are frozen structure."""
    core = model.module if _is_ddp(model) else model
    if not core.has_vol:
        raise ValueError("finetune_step drives a has_vol model (SuRF.init_volumes / load_params_vol first)")
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        # the per-scene volumes are (N_s, 7) with N_s depending on the scene: ranks holding different scenes cannot average
        # them (the reference finetunes in a single process, runner.py:62)
        raise RuntimeError("finetune_step: per-scene volumes cannot be averaged over ranks; finetune in a single process")
    return _step(model, ipts, targets, loss_fn, optimizer, cos_anneal_ratio, step, "finetune")


def train_step(model, ipts, targets, loss_fn, optimizer, cos_anneal_ratio=1.0, step=0, sync=True):
    """runner.py:152-165 for a volume-building model in train mode (model.train(): BatchNorm batch statistics, matching-field
    jitter).  targets: what losses/loss.py reads in mode "train" (color, imgs, intrs, c2ws, src_idx, mask_ref / mask_src,
    pseudo_depth_ref / pseudo_depth_src, depth_ref / depth_src, ...).  sync=False: no gradient averaging even when a process
    group is up (a deliberately local step: bench.py's single-GPU `training_step` beside the forced world-1 group)."""
    core = model.module if _is_ddp(model) else model
    if core.has_vol:
        raise ValueError("train_step drives a volume-building model; use finetune_step for has_vol models")
    if not core.training:
        raise RuntimeError("train_step: call model.train() first (the sparse U-Net's backward is that of batch-statistics BatchNorm)")
    return _step(model, ipts, targets, loss_fn, optimizer, cos_anneal_ratio, step, "train", sync)
