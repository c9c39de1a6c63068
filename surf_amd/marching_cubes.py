"""Iso-surface extraction from the SDF lattice of ``ImplicitSurface.extract_geometry``
(reference: ``mcubes.marching_cubes(u, threshold)``, implicit_surface.py:353 -- PyMCubes, absent here).

Round-1 stand-in for SURVEY row f1: a **marching-tetrahedra** extractor written with torch tensor ops (runs on the
GPU the lattice lives on).  It produces a watertight, consistently oriented mesh of the same level set with
vertices in lattice-index units exactly like PyMCubes, but its triangulation (6 tetrahedra per cell) differs from
PyMCubes' case tables, so vertex / face counts are not identical: mesh-level parity with PyMCubes is NOT claimed.
"""
import numpy as np
import torch

# cube corners (x,y,z) and the 6 tetrahedra around the 0-6 diagonal
_CORNERS = torch.tensor([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]])
_TETS = torch.tensor([[0, 5, 1, 6], [0, 1, 2, 6], [0, 2, 3, 6], [0, 3, 7, 6], [0, 7, 4, 6], [0, 4, 5, 6]])


def marching_cubes(u, threshold=0.0, chunk=1 << 20):
    """u (X,Y,Z) tensor or array; returns (vertices (V,3) float64 in index units, triangles (F,3) int64) as numpy."""
    if not torch.is_tensor(u):
        u = torch.from_numpy(np.asarray(u))
    u = u.float()
    dev = u.device
    X, Y, Z = u.shape
    inside = u > threshold
    # cells whose 8 corners are not all on one side
    c = inside[:-1, :-1, :-1].to(torch.uint8)
    s = torch.zeros_like(c, dtype=torch.int16)
    for dx, dy, dz in _CORNERS.tolist():
        s += inside[dx:X - 1 + dx, dy:Y - 1 + dy, dz:Z - 1 + dz].to(torch.int16)
    active = ((s > 0) & (s < 8)).nonzero()                    # (M,3) cell origins
    if active.shape[0] == 0:
        return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64)
    corners = _CORNERS.to(dev)
    tets = _TETS.to(dev)
    stride = torch.tensor([Y * Z, Z, 1], device=dev)
    edge_a, edge_b, flip_ref = [], [], []

    for c0 in range(0, active.shape[0], chunk):
        cells = active[c0:c0 + chunk]
        cpos = cells[:, None, :] + corners[None]                                   # (m,8,3)
        cid = (cpos * stride).sum(-1)                                              # (m,8) linear corner ids
        tid = cid[:, tets]                                                         # (m,6,4)
        tval = u.reshape(-1)[tid] - threshold
        tin = tval > 0
        nin = tin.sum(-1)
        tid, tval, tin, nin = tid.reshape(-1, 4), tval.reshape(-1, 4), tin.reshape(-1, 4), nin.reshape(-1)
        for k in (1, 3):                                                           # one vertex isolated -> 1 triangle
            sel = nin == k
            if not bool(sel.any()):
                continue
            ids, ins = tid[sel], tin[sel]
            lone = (ins if k == 1 else ~ins).float().argmax(dim=1)                 # index of the isolated vertex
            others = torch.stack([(lone + j) % 4 for j in (1, 2, 3)], dim=1)
            a = ids.gather(1, lone[:, None]).expand(-1, 3)
            b = ids.gather(1, others)
            edge_a.append(a.reshape(-1, 3))
            edge_b.append(b.reshape(-1, 3))
            # reference direction inside -> outside: from the isolated vertex if it is inside, else towards it
            flip_ref.append(torch.full((a.shape[0],), 1 if k == 1 else -1, device=dev))
        sel = nin == 2                                                             # two / two -> quad = 2 triangles
        if bool(sel.any()):
            ids, ins = tid[sel], tin[sel]
            order = torch.argsort(ins.to(torch.int8), dim=1, descending=True, stable=True)  # inside vertices first
            p = ids.gather(1, order)                                               # p0,p1 inside; p2,p3 outside
            p0, p1, p2, p3 = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
            # quad (p0p2, p0p3, p1p3, p1p2)
            a1 = torch.stack([p0, p0, p1], 1); b1 = torch.stack([p2, p3, p3], 1)
            a2 = torch.stack([p0, p1, p1], 1); b2 = torch.stack([p2, p3, p2], 1)
            edge_a += [a1, a2]
            edge_b += [b1, b2]
            flip_ref += [torch.full((a1.shape[0],), 2, device=dev)] * 2

    A = torch.cat(edge_a)                                                          # (F,3) edge endpoints (a inside-ish)
    B = torch.cat(edge_b)
    mode = torch.cat(flip_ref)
    lo, hi = torch.minimum(A, B), torch.maximum(A, B)
    key = lo * (X * Y * Z) + hi
    uniq, inv = torch.unique(key.reshape(-1), return_inverse=True)
    tri = inv.reshape(-1, 3)
    ua, ub = uniq // (X * Y * Z), uniq % (X * Y * Z)
    flat = u.reshape(-1).double() - threshold
    va, vb = flat[ua], flat[ub]
    t = (va / (va - vb)).clamp(0.0, 1.0)

    def pos(i):
        return torch.stack([i // (Y * Z), (i // Z) % Y, i % Z], dim=1).double()

    verts = pos(ua) + (pos(ub) - pos(ua)) * t[:, None]
    # orientation: normals must point from inside (u > thr) to outside, i.e. along -grad(u)
    v0, v1, v2 = verts[tri[:, 0]], verts[tri[:, 1]], verts[tri[:, 2]]
    nrm = torch.linalg.cross(v1 - v0, v2 - v0)
    pa, pb = pos(A[:, 0]), pos(B[:, 0])
    ina = flat[A[:, 0]] > 0
    out_dir = torch.where(ina[:, None], pb - pa, pa - pb)                          # inside -> outside along one cut edge
    flip = (nrm * out_dir).sum(-1) < 0
    tri = torch.where(flip[:, None], tri[:, [0, 2, 1]], tri)
    # drop degenerate triangles (a lattice value exactly on the threshold collapses an edge)
    ok = (tri[:, 0] != tri[:, 1]) & (tri[:, 1] != tri[:, 2]) & (tri[:, 0] != tri[:, 2])
    return verts.cpu().numpy(), tri[ok].cpu().numpy().astype(np.int64)
