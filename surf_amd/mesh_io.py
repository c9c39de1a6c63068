"""Mesh export of the validation loop (runner.py:231-240): the reference builds a `trimesh.Trimesh(vertices, triangles)`,
applies the scene's `scale_mat` (normalised unit-sphere frame -> world frame, datasets/dtu.py:204-240) and exports a PLY.
trimesh is not a dependency here: `transform_vertices` is `Trimesh.apply_transform` for a point set and `write_ply` writes
the same binary little-endian PLY layout trimesh 3.22 exports (float32 x y z, faces as `list uchar int vertex_indices`)."""
import os

import numpy as np


def transform_vertices(vertices, matrix):
    """(V,3) points through a 4x4 homogeneous matrix: trimesh.Trimesh.apply_transform (runner.py:236)."""
    v = np.asarray(vertices, dtype=np.float64)
    m = np.asarray(matrix, dtype=np.float64).reshape(4, 4)
    out = v @ m[:3, :3].T + m[:3, 3][None, :]
    w = v @ m[3, :3] + m[3, 3]
    if not np.allclose(w, 1.0):
        out = out / w[:, None]
    return out


def write_ply(path, vertices, triangles):
    """Binary little-endian PLY: `element vertex N` (float x, y, z) + `element face M` (list uchar int vertex_indices)."""
    v = np.ascontiguousarray(np.asarray(vertices, dtype="<f4").reshape(-1, 3))
    t = np.ascontiguousarray(np.asarray(triangles, dtype="<i4").reshape(-1, 3))
    if t.size and (t.min() < 0 or t.max() >= max(len(v), 1)):
        raise ValueError("triangle index out of range")
    header = ("ply\nformat binary_little_endian 1.0\ncomment surf_amd mesh export\n"
              f"element vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\n"
              f"element face {len(t)}\nproperty list uchar int vertex_indices\nend_header\n")
    faces = np.empty(len(t), dtype=[("n", "u1"), ("idx", "<i4", (3,))])
    faces["n"] = 3
    faces["idx"] = t
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(v.tobytes())
        f.write(faces.tobytes())


def read_ply(path):
    """Reads back what write_ply (or trimesh's binary PLY export of a plain triangle mesh) wrote."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    head = data[:end].decode("ascii").splitlines()
    if "format binary_little_endian 1.0" not in head:
        raise ValueError("only binary little-endian PLY is supported")
    nv = nt = 0
    for l in head:
        if l.startswith("element vertex"):
            nv = int(l.split()[-1])
        if l.startswith("element face"):
            nt = int(l.split()[-1])
    v = np.frombuffer(data, dtype="<f4", count=nv * 3, offset=end).reshape(nv, 3)
    faces = np.frombuffer(data, dtype=[("n", "u1"), ("idx", "<i4", (3,))], count=nt, offset=end + nv * 12)
    if nt and not (faces["n"] == 3).all():
        raise ValueError("not a triangle mesh")
    return v.copy(), faces["idx"].copy()


def export_mesh(path, vertices, triangles, scale_mat=None):
    """runner.py:231-240: optional scale_mat transform, then PLY export.  Returns the transformed vertices."""
    v = np.asarray(vertices, dtype=np.float64)
    if scale_mat is not None:
        v = transform_vertices(v, scale_mat.detach().cpu().numpy() if hasattr(scale_mat, "detach") else scale_mat)
    write_ply(path, v, triangles)
    return v
