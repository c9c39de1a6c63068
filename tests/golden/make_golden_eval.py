#!/usr/bin/env python3
"""Golden Chamfer numbers of the REFERENCE's DTU evaluator (row f4): /root/reference/evaluation/dtu_eval.py is run as the script
it is (`runpy`, its own argument parser, its hard-coded list of the 15 test scans, its multiprocessing pool) on the synthetic
evaluation directory of tests/golden/eval_scene.py; the results.json it writes is committed as tests/golden/dtu_eval_results.json.

Stood in for, because the image lacks them: `open3d` - only its three I/O calls (`o3d.io.read_triangle_mesh`, `o3d.io.read_point_cloud`
-> the arrays of the PLY files, float64 like open3d's) - and `tqdm` (a progress bar).  The script shuffles with an UNSEEDED
`np.random.default_rng()`; for a reproducible fixture `numpy.random.default_rng` is replaced, for the duration of the run, by a
factory of generators seeded with eval_scene.SHUFFLE_SEED; and it calls `np.mgrid` with float bounds, which numpy 2 rejects:
`np.mgrid` is given numpy 1's behaviour for the run (_MGridOfNumpy1), as is `from numpy import *` (numpy 1 did not export
min / max / round, which the script relies on).  Everything numerical - triangle sampling, the greedy thinning, the
observability mask, both nearest-neighbour passes, the ground plane, the clipping and the means - is the reference's code."""
import json
import os
import runpy
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from surf_amd import mesh_io  # noqa: E402
from surf_amd.datasets import mvs_io  # noqa: E402
from tests.golden import eval_scene  # noqa: E402

REF = "/root/reference"


def _o3d_stub():
    o3d = types.ModuleType("open3d")
    io = types.ModuleType("open3d.io")

    class _Mesh:
        def __init__(self, v, t):
            self.vertices, self.triangles = np.asarray(v, dtype=np.float64), np.asarray(t, dtype=np.int32)

    class _Cloud:
        def __init__(self, p):
            self.points = np.asarray(p, dtype=np.float64)
    io.read_triangle_mesh = lambda path: _Mesh(*mesh_io.read_ply(path))
    io.read_point_cloud = lambda path: _Cloud(mvs_io.read_ply_points(path))
    o3d.io = io
    return o3d, io


def _tqdm_stub():
    m = types.ModuleType("tqdm")

    class tqdm:
        def __init__(self, *a, **k):
            pass

        def set_description(self, *a, **k):
            pass

        def update(self, *a, **k):
            pass

        def close(self):
            pass
    m.tqdm = tqdm
    return m


class _MGridOfNumpy1:
    """`np.mgrid[:n1 + 1, :n2 + 1]` with float bounds, as the reference's sample_single_tri calls it: numpy 1.x returned FLOAT
    grids of ceil((stop - start) / step) points per axis (the script then adds 0.5 in place); numpy 2.2 raises.  The shim
    restores that behaviour for the run."""

    def __getitem__(self, key):
        key = key if isinstance(key, tuple) else (key,)
        axes, any_float = [], False
        for sl in key:
            start, stop, step = (0 if sl.start is None else sl.start), sl.stop, (1 if sl.step is None else sl.step)
            any_float |= any(isinstance(x, (float, np.floating)) for x in (start, stop, step))
            axes.append((start, stop, step))
        grids = [np.arange(a, b, c, dtype=float if any_float else int) for a, b, c in axes]
        return np.array(np.meshgrid(*grids, indexing="ij"))


def main():
    o3d, io = _o3d_stub()
    sys.modules["open3d"], sys.modules["open3d.io"] = o3d, io
    sys.modules["tqdm"] = _tqdm_stub()
    real_rng = np.random.default_rng
    with tempfile.TemporaryDirectory() as tmp:
        out_dir, data_dir = os.path.join(tmp, "exp"), os.path.join(tmp, "eval")
        eval_scene.write_eval_scene(out_dir, data_dir)
        argv = sys.argv
        sys.argv = ["dtu_eval.py", "--out_dir", out_dir, "--dataset_dir", data_dir, "--downsample_density",
                    str(eval_scene.ARGS["downsample_density"]), "--patch_size", str(eval_scene.ARGS["patch_size"]), "--max_dist",
                    str(eval_scene.ARGS["max_dist"])]
        np.random.default_rng = lambda *a: real_rng(eval_scene.SHUFFLE_SEED) if not a else real_rng(*a)
        real_mgrid, np.mgrid = np.mgrid, _MGridOfNumpy1()
        # the script does `from numpy import *` and then calls the BUILTIN max(n1, 1e-7): numpy 1.x kept min / max / round out of
        # __all__ for exactly that reason, numpy 2 exports them (np.max(n1, axis=1e-7) raises)
        real_all = list(np.__all__)
        np.__all__ = [n for n in real_all if n not in ("min", "max", "round")]
        try:
            runpy.run_path(os.path.join(REF, "evaluation", "dtu_eval.py"), run_name="__main__")
        finally:
            np.random.default_rng = real_rng
            np.mgrid = real_mgrid
            np.__all__ = real_all
            sys.argv = argv
        with open(os.path.join(out_dir, "results.json")) as f:
            res = json.load(f)
    with open(os.path.join(HERE, "dtu_eval_results.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("wrote dtu_eval_results.json:", {k: round(v["all"], 4) for k, v in res.items()})


if __name__ == "__main__":
    main()
