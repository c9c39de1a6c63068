"""Iso-surface extraction from the SDF lattice of ``ImplicitSurface.extract_geometry``.

Drop-in for ``mcubes.marching_cubes(u, threshold)`` (implicit_surface.py:353; PyMCubes 0.1.4, third party, absent here):
the classic 256-case marching cubes in HIP (csrc/mcubes.hip): `u <= threshold` inside test, one vertex per sign-changing
lattice edge by linear interpolation in double precision, vertices in lattice-index units.  The vertex SET and the
triangle set are PyMCubes'; the ORDER of the vertex list is ours (owner lattice point, axis), see oracle/mcubes_oracle.py.
There is no CPU fallback: the lattice must live on the GPU (it is produced there by ImplicitSurface.sdf_grid).
"""
import numpy as np
import torch

from . import ops


def marching_cubes(u, threshold=0.0):
    """u (X,Y,Z) device tensor (or array, moved to the current GPU); returns (vertices (V,3) float64 in index units,
    triangles (F,3) int64) as numpy arrays, like mcubes.marching_cubes."""
    if not torch.is_tensor(u):
        u = torch.from_numpy(np.ascontiguousarray(np.asarray(u, dtype=np.float32))).cuda()
    u = u.float().contiguous()
    v, t = ops.marching_cubes(u, float(threshold))
    # one device-to-host copy for both arrays (int64 triangle indices formed on the device: same bytes as float64 vertices)
    both = torch.cat([v.reshape(-1).view(torch.int64), t.reshape(-1).to(torch.int64)]).cpu()
    nv = v.numel()
    return both[:nv].view(torch.float64).numpy().reshape(-1, 3), both[nv:].numpy().reshape(-1, 3)
