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


def to_host(t):
    """Device tensor -> host tensor through a PINNED buffer of torch's caching host allocator (a pageable .cpu() of the 36 MB
    mesh is staged through one anyway and then copied again); the result owns its buffer, nothing is reused under the caller."""
    if not t.is_cuda:
        return t
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return h


def marching_cubes(u, threshold=0.0, rescale=None):
    """u (X,Y,Z) device tensor (or array, moved to the current GPU); returns (vertices (V,3) float64 in index units,
    triangles (F,3) int64) as numpy arrays, like mcubes.marching_cubes.  rescale = (resolution - 1, b_max - b_min, b_min) (ours):
    the vertices come back as `vertices / (resolution - 1) * (b_max - b_min) + b_min`, formed on the device (extract_geometry's
    rescaling, implicit_surface.py:354-356: the same float64 operations in the same order, minus three host passes over the
    vertex array)."""
    if not torch.is_tensor(u):
        u = torch.from_numpy(np.ascontiguousarray(np.asarray(u, dtype=np.float32))).cuda()
    u = u.float().contiguous()
    v, t = ops.marching_cubes(u, float(threshold))
    if rescale is not None:
        res_m1, span, lo = rescale
        v = v / float(res_m1) * torch.as_tensor(span, dtype=torch.float64, device=v.device)[None, :] \
            + torch.as_tensor(lo, dtype=torch.float64, device=v.device)[None, :]
    # one device-to-host copy for both arrays (int64 triangle indices formed on the device: same itemsize as float64 vertices)
    both = to_host(torch.cat([v.reshape(-1).view(torch.int64), t.reshape(-1).to(torch.int64)]))
    nv = v.numel()
    return both[:nv].view(torch.float64).numpy().reshape(-1, 3), both[nv:].numpy().reshape(-1, 3)
