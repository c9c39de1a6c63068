#!/usr/bin/env python3
"""Golden `ipts` dictionaries of the REFERENCE's DTU reader (row f3): generated in the build container by importing
/root/reference/datasets/dtu.py / tanks.py and running THEIR `DTUDataset.__getitem__` / `TanksDataset.__getitem__` (val and train
mode) on the synthetic scenes of tests/golden/dtu_scene.py.  Only data is committed (tests/golden/dataset_items.npz).

Three third-party modules of the reader are absent from this image and are stood in for, for exactly the calls it makes:
  cv2.resize(img, (w, h), interpolation=cv2.INTER_NEAREST)   -> nearest-neighbour resampling (floor(dst * src / dst))
  cv2.decomposeProjectionMatrix(P)                             -> RQ decomposition returning (K, R, homogeneous camera centre)
  plyfile.PlyData.read(path)["vertex"]["x" | "y" | "z"]        -> the vertex columns of an ascii / binary PLY
(`lmdb` is imported by the reader and never used).  Everything else that runs is the reference's own code: the camera-file
parser, the pair file, PFM reader, the reference-view-relative poses, `get_scale_mat`, the K / R / t re-decomposition glue of
`load_K_Rt_from_P`, near / far, the pixel draws and their RNG order, the ray generation, the pseudo points, the dictionary.
What this fixture therefore does NOT pin is the three stand-ins themselves (tests/test_datasets.py checks them on their own)."""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from surf_amd.datasets import mvs_io  # noqa: E402
from tests.golden import dtu_scene  # noqa: E402
from tests.golden import make_golden as G  # noqa: E402


def _cv2_stub():
    m = types.ModuleType("cv2")
    m.INTER_NEAREST = 0

    def resize(img, dsize, fx=None, fy=None, interpolation=0):
        assert interpolation == m.INTER_NEAREST and dsize is not None
        return mvs_io.resize_nearest(img, (dsize[1], dsize[0]))

    def decompose(P):
        intr, pose = mvs_io.decompose_projection(P)          # K / K[2,2] and (R^T | C)
        K = intr[:3, :3].copy()
        R = pose[:3, :3].astype(np.float64).T
        c = np.concatenate([pose[:3, 3].astype(np.float64), [1.0]])[:, None]
        return K, R, c
    m.resize, m.decomposeProjectionMatrix = resize, decompose
    return m


def _plyfile_stub():
    m = types.ModuleType("plyfile")

    class PlyData:
        @staticmethod
        def read(path):
            pts = mvs_io.read_ply_points(path)
            return {"vertex": {"x": pts[:, 0], "y": pts[:, 1], "z": pts[:, 2]}}
    m.PlyData, m.PlyElement = PlyData, object
    return m


def main():
    sys.modules["cv2"] = _cv2_stub()
    sys.modules["plyfile"] = _plyfile_stub()
    sys.modules["lmdb"] = types.ModuleType("lmdb")
    sys.path.insert(0, G.REF)
    from datasets.dtu import DTUDataset
    out = {}

    def store(tag, item):
        for k, v in item.items():
            if torch.is_tensor(v):
                out[f"{tag}/{k}"] = v
            elif isinstance(v, (int, np.integer)):
                out[f"{tag}/{k}"] = torch.tensor(int(v))
            elif isinstance(v, str):
                out[f"{tag}/str/{k}/{v}"] = torch.zeros(1)
            else:
                raise TypeError((k, type(v)))

    with tempfile.TemporaryDirectory() as tmp:
        root = os.path.join(tmp, "dtu")
        dtu_scene.write_dtu_scene(root)
        for mode, extra in (("val", {}), ("train", {"n_rays": 96})):
            ds = DTUDataset(G.Conf(dict(dtu_scene.DATASET_CONF, data_dir=root, **extra)), mode)
            assert len(ds) == 1
            np.random.seed(dtu_scene.SEEDS["numpy"])
            torch.manual_seed(dtu_scene.SEEDS["torch"])
            store(mode, ds[0])
        from datasets.tanks import TanksDataset
        troot = os.path.join(tmp, "tnt")
        dtu_scene.write_tanks_scene(troot)
        for mode, extra in (("val", {}), ("train", {"n_rays": 64})):
            ds = TanksDataset(G.Conf(dict(dtu_scene.TANKS_CONF, data_dir=troot, **extra)), mode)
            assert len(ds) == 1
            np.random.seed(dtu_scene.SEEDS["numpy"])
            torch.manual_seed(dtu_scene.SEEDS["torch"])
            store("tanks_" + mode, ds[0])
    G.ONLY.clear()
    G.npz("dataset_items.npz", **out)


if __name__ == "__main__":
    main()
