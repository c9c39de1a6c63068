"""Iso-surface extractor (round-1 stand-in for PyMCubes): geometric properties on an analytic sphere."""
import numpy as np
import torch

from surf_amd.marching_cubes import marching_cubes


def test_sphere_mesh_is_closed_oriented_and_accurate():
    R = 40
    ax = torch.linspace(-1, 1, R)
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    u = 0.6 - torch.sqrt(x * x + y * y + z * z)          # u = -sdf: positive inside
    v, f = marching_cubes(u, 0.0)
    assert v.shape[0] > 1000 and f.shape[0] > 2000
    w = v / (R - 1) * 2 - 1
    r = np.linalg.norm(w, axis=1)
    assert abs(r.mean() - 0.6) < 2e-3 and r.std() < 2e-3
    # closed 2-manifold: every undirected edge is shared by exactly two triangles, with opposite orientation
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    und = np.sort(e, axis=1)
    _, counts = np.unique(und[:, 0] * v.shape[0] + und[:, 1], return_counts=True)
    assert (counts == 2).all()
    _, dcounts = np.unique(e[:, 0] * v.shape[0] + e[:, 1], return_counts=True)
    assert (dcounts == 1).all()
    # outward normals and area of the sphere
    a, b, c = w[f[:, 0]], w[f[:, 1]], w[f[:, 2]]
    n = np.cross(b - a, c - a)
    assert ((n * (a + b + c)).sum(1) > 0).all()
    area = 0.5 * np.linalg.norm(n, axis=1).sum()
    assert abs(area - 4 * np.pi * 0.36) / (4 * np.pi * 0.36) < 0.01
    vol = (a * np.cross(b, c)).sum() / 6.0
    assert abs(vol - 4 / 3 * np.pi * 0.216) / (4 / 3 * np.pi * 0.216) < 0.01


def test_empty_lattice():
    v, f = marching_cubes(torch.full((8, 8, 8), -1.0), 0.0)
    assert v.shape == (0, 3) and f.shape == (0, 3)
