"""Row f1: marching cubes, PLY export, scale_mat.

CPU: the 256-case table (C header == oracle copy) has the properties that make the mesh crack-free and oriented; the
oracle (restatement of PyMCubes' algorithm, parity unpinned: PyMCubes is absent) produces closed, oriented, accurate
meshes whose vertices sit on the linear roots of the sign-changing lattice edges; PLY round trip and scale_mat.
GPU (-m gpu): csrc/mcubes.hip against the oracle, vertices and triangles EXACT (same order, same double arithmetic)."""
import collections
import os
import re

import numpy as np
import pytest
import torch

from oracle import mcubes_oracle as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = [tuple(int(c) for c in row) for row in M.CORNERS]
E = M.EDGES


def _header_table():
    txt = open(os.path.join(ROOT, "surf_amd", "csrc", "mc_tables.h")).read()
    body = txt[txt.index("MC_TRI[256][16]"):]
    rows = re.findall(r"\{([^{}]*)\}", body)
    tab = [[int(v) for v in r.split(",") if v.strip()] for r in rows]
    ntri = [int(v) for v in re.findall(r"-?\d+", txt[txt.index("MC_NTRI[256] = {"):txt.index("};")].split("{")[1])]
    return tab, ntri


def test_table_in_the_c_header_equals_the_oracle_copy():
    tab, ntri = _header_table()
    assert len(tab) == 256 and len(ntri) == 256
    for c in range(256):
        row = [v for v in tab[c] if v >= 0]
        assert row == M.TRI_TABLE[c] and ntri[c] == len(row) // 3
        assert tab[c][len(row):] == [-1] * (16 - len(row))
    assert sum(ntri) == 820


def test_table_cases_are_crack_free_and_oriented():
    def mid(e):
        a, b = E[e]
        return tuple((V[a][i] + V[b][i]) / 2 for i in range(3))

    face_rule = collections.defaultdict(set)
    signs = set()
    for c, tris in enumerate(M.TRI_TABLE):
        inside = [(c >> i) & 1 for i in range(8)]
        crossing = {e for e, (a, b) in enumerate(E) if inside[a] != inside[b]}
        assert set(tris) == crossing and len(tris) % 3 == 0 and len(tris) <= 15
        cnt = collections.Counter()
        for t in range(0, len(tris), 3):
            a, b, d = tris[t:t + 3]
            assert len({a, b, d}) == 3
            for p, q in ((a, b), (b, d), (d, a)):
                cnt[frozenset((p, q))] += 1
        for k, n in cnt.items():                       # interior edges twice, boundary edges once and on a cell face
            p, q = tuple(k)
            pts = [V[i] for i in E[p] + E[q]]
            on_face = any(all(pt[ax] == val for pt in pts) for ax in range(3) for val in (0, 1))
            assert n == 2 or (n == 1 and on_face)
        for ax in range(3):                            # the segments on a face depend on that face's corner bits only
            for val in (0, 1):
                o = [i for i in range(3) if i != ax]
                key = tuple(sorted(((V[i][o[0]], V[i][o[1]]), inside[i]) for i in range(8) if V[i][ax] == val))
                segs = set()
                for k, n in cnt.items():
                    p, q = tuple(k)
                    if n == 1 and all(V[i][ax] == val for i in E[p] + E[q]):
                        segs.add(frozenset(((mid(p)[o[0]], mid(p)[o[1]]), (mid(q)[o[0]], mid(q)[o[1]]))))
                face_rule[key].add(frozenset(segs))
        f = np.array([-1.0 if b else 1.0 for b in inside])
        for t in range(0, len(tris), 3):               # orientation against the trilinear gradient at the centroid
            P = [np.array(mid(e)) for e in tris[t:t + 3]]
            x, y, z = sum(P) / 3
            g = np.zeros(3)
            for i, (vx, vy, vz) in enumerate(V):
                wx, wy, wz = (x if vx else 1 - x), (y if vy else 1 - y), (z if vz else 1 - z)
                g += f[i] * np.array([(1 if vx else -1) * wy * wz, (1 if vy else -1) * wx * wz, (1 if vz else -1) * wx * wy])
            signs.add(np.sign(round(float(np.cross(P[1] - P[0], P[2] - P[0]) @ g), 9)))
    assert len(face_rule) == 16 and all(len(v) == 1 for v in face_rule.values())
    assert signs == {-1.0}                             # normals point towards decreasing u (= out of the solid, u = -sdf)


def _field(shape, seed):
    g = np.random.default_rng(seed)
    ax = [np.linspace(-1, 1, n) for n in shape]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    u = 0.55 - np.sqrt(X * X + (Y * 1.1) ** 2 + (Z * 0.9) ** 2) + 0.15 * np.sin(3 * X + 1) * np.cos(2 * Y) * np.sin(4 * Z + seed)
    return (u + 0.002 * g.standard_normal(shape)).astype(np.float32)


def _check_mesh(u, v, t, iso):
    nv = v.shape[0]
    assert t.min() >= 0 and t.max() < nv
    # one vertex per sign-changing lattice edge, on the edge's linear root
    inside = u <= iso
    n_cross = sum(int((np.take(inside, range(0, u.shape[a] - 1), axis=a) != np.take(inside, range(1, u.shape[a]), axis=a)).sum())
                  for a in range(3))
    assert nv == n_cross
    frac = v - np.floor(v)
    on_axis = (frac > 0).sum(axis=1)
    assert (on_axis <= 1).all()
    lo = np.floor(v).astype(int)
    for a in range(3):
        sel = frac[:, a] > 0
        p0 = lo[sel]
        p1 = p0.copy()
        p1[:, a] += 1
        f1, f2 = u[tuple(p0.T)].astype(np.float64), u[tuple(p1.T)].astype(np.float64)
        assert ((f1 <= iso) != (f2 <= iso)).all()
        root = (iso - f1) / (f2 - f1)
        assert np.abs(root - frac[sel, a]).max() < 1e-12
    # edge-manifold and consistently oriented: every directed edge once; boundary only on the lattice boundary
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    key = e[:, 0].astype(np.int64) * nv + e[:, 1]
    assert np.unique(key).size == key.size
    rev = e[:, 1].astype(np.int64) * nv + e[:, 0]
    unpaired = ~np.isin(key, rev)
    pts = v[e[unpaired].reshape(-1)]
    hi = np.array(u.shape) - 1
    assert ((pts == 0) | (pts == hi[None])).any(axis=1).all()
    return int(unpaired.sum())


def test_oracle_meshes_are_closed_oriented_and_on_the_linear_roots():
    u = _field((21, 26, 19), 1)
    v, t = M.marching_cubes(u, 0.0)
    assert v.shape[0] > 500 and t.shape[0] > 1000
    assert _check_mesh(u, v, t, 0.0) == 0                    # closed: the blob does not touch the lattice boundary
    assert v.shape[0] - (3 * t.shape[0]) // 2 + t.shape[0] == 2          # one closed genus-0 component
    # a surface that leaves the lattice has its boundary on the lattice faces only
    u2 = _field((12, 9, 15), 2) + 0.5
    v2, t2 = M.marching_cubes(u2, 0.1)
    assert _check_mesh(u2, v2, t2, 0.1) > 0
    # accuracy on a sphere: radius, area, enclosed volume
    R = 40
    ax = np.linspace(-1, 1, R)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    us = (0.6 - np.sqrt(X * X + Y * Y + Z * Z)).astype(np.float32)
    vs, ts = M.marching_cubes(us, 0.0)
    w = vs / (R - 1) * 2 - 1
    assert abs(np.linalg.norm(w, axis=1).mean() - 0.6) < 2e-3
    a, b, c = w[ts[:, 0]], w[ts[:, 1]], w[ts[:, 2]]
    n = np.cross(b - a, c - a)
    assert ((n * (a + b + c)).sum(1) > 0).all()                              # outward normals
    assert abs(0.5 * np.linalg.norm(n, axis=1).sum() / (4 * np.pi * 0.36) - 1) < 0.01
    assert abs((a * np.cross(b, c)).sum() / 6.0 / (4 / 3 * np.pi * 0.216) - 1) < 0.01


def test_oracle_degenerate_lattices():
    assert M.marching_cubes(np.full((5, 5, 5), -1.0, np.float32), 0.0)[0].shape == (0, 3)
    assert M.marching_cubes(np.full((5, 5, 5), 1.0, np.float32), 0.0)[1].shape == (0, 3)
    u = np.zeros((2, 2, 2), np.float32)
    u[0, 0, 0] = 1.0                                                           # single corner outside (u > iso)
    v, t = M.marching_cubes(u, 0.5)
    assert v.shape == (3, 3) and t.shape == (1, 3)
    v, t = M.marching_cubes(np.array([[[1.0, -1.0]]], np.float32), 0.0)         # a 1 x 1 x 2 lattice: a vertex, no cell
    assert v.shape == (1, 3) and t.shape == (0, 3) and abs(v[0, 2] - 0.5) < 1e-12


def test_ply_round_trip_and_scale_mat(tmp_path):
    from surf_amd import mesh_io
    u = _field((14, 12, 13), 3)
    v, t = M.marching_cubes(u, 0.0)
    S = np.eye(4)
    S[:3, :3] *= 123.5
    S[:3, 3] = [10.0, -20.0, 5.5]
    path = tmp_path / "meshes" / "scan24_epoch0.ply"
    vw = mesh_io.export_mesh(str(path), v, t, scale_mat=torch.from_numpy(S))
    assert np.allclose(vw, v * 123.5 + S[:3, 3])                               # trimesh.apply_transform of a similarity
    rv, rt = mesh_io.read_ply(str(path))
    assert np.array_equal(rt, t.astype(np.int32)) and np.allclose(rv, vw.astype(np.float32))
    head = open(path, "rb").read(200).decode("ascii", "ignore")
    assert head.startswith("ply\nformat binary_little_endian 1.0") and "property list uchar int vertex_indices" in head
    with pytest.raises(ValueError):
        mesh_io.write_ply(str(tmp_path / "bad.ply"), v[:2], t)


# ---- GPU ---------------------------------------------------------------------------------------------------------------


@pytest.mark.gpu
@pytest.mark.parametrize("shape,iso,seed", [((21, 26, 19), 0.0, 1), ((12, 9, 15), 0.1, 2), ((33, 33, 33), -0.05, 4),
                                            ((2, 2, 2), 0.0, 5), ((1, 3, 70), 0.0, 6), ((64, 3, 2), 0.02, 7)])
def test_hip_marching_cubes_equals_the_oracle(shape, iso, seed):
    from surf_amd import ops
    u = _field(shape, seed) + (0.5 if seed in (2, 7) else 0.0)
    v_ref, t_ref = M.marching_cubes(u, iso)
    v, t = ops.marching_cubes(torch.from_numpy(u).cuda(), iso)
    torch.cuda.synchronize()
    assert v.dtype == torch.float64 and t.dtype == torch.int32
    assert tuple(v.shape) == v_ref.shape and tuple(t.shape) == t_ref.shape
    assert np.array_equal(t.cpu().numpy(), t_ref)
    assert np.array_equal(v.cpu().numpy(), v_ref)            # same double arithmetic: bit-exact


@pytest.mark.gpu
def test_hip_marching_cubes_noise_and_empty_lattices():
    from surf_amd import ops
    from surf_amd.marching_cubes import marching_cubes
    g = np.random.default_rng(0)
    u = g.standard_normal((40, 37, 41)).astype(np.float32)    # every cell active: many blocks, all 256 cases
    v_ref, t_ref = M.marching_cubes(u, 0.0)
    v, t = ops.marching_cubes(torch.from_numpy(u).cuda(), 0.0)
    assert np.array_equal(t.cpu().numpy(), t_ref) and np.array_equal(v.cpu().numpy(), v_ref)
    _check_mesh(u, v_ref, t_ref, 0.0)
    ve, te = marching_cubes(torch.full((8, 8, 8), -1.0, device="cuda"), 0.0)
    assert ve.shape == (0, 3) and te.shape == (0, 3)


@pytest.mark.gpu
def test_hip_marching_cubes_at_512_cubed_is_closed():
    """The size validate() uses: a 512^3 lattice of an analytic blob; closedness and Euler characteristic on the device."""
    from surf_amd import ops
    R = 512
    ax = torch.linspace(-1, 1, R, device="cuda")
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    u = 0.6 - torch.sqrt(X * X + (1.2 * Y) ** 2 + Z * Z) + 0.05 * torch.sin(9 * X) * torch.sin(7 * Y + 1) * torch.sin(8 * Z)
    del X, Y, Z
    v, t = ops.marching_cubes(u.contiguous(), 0.0)
    nv, nt = v.shape[0], t.shape[0]
    assert nv > 300_000 and nt == 2 * nv - 4
    tl = t.long()
    e = torch.cat([tl[:, [0, 1]], tl[:, [1, 2]], tl[:, [2, 0]]])
    key = e[:, 0] * nv + e[:, 1]
    rev = e[:, 1] * nv + e[:, 0]
    assert torch.unique(key).numel() == key.numel()                     # every directed edge once
    assert torch.equal(torch.sort(key).values, torch.sort(rev).values)  # ... and its reverse too: closed, oriented
    assert nv - (3 * nt) // 2 + nt == 2                                 # one genus-0 component
    w = v / (R - 1) * 2 - 1
    assert float(w.abs().max()) < 0.75
