"""Fifteen small synthetic "scans" in the layout of the DTU evaluation data (ObsMask/ObsMask{scan}_10.mat, ObsMask/Plane{scan}.mat,
Points/stl/stl{scan:03}_total.ply) + one mesh per scan under <out>/meshes/final/scan{scan}.ply, written deterministically: the
input of tests/golden/make_golden_eval.py (which runs the REFERENCE's evaluation/dtu_eval.py on it) and of the test that compares
surf_amd.evaluation.dtu_eval with the numbers it printed."""
import os

import numpy as np
from scipy.io import savemat

SCANS = [24, 37, 40, 55, 63, 65, 69, 83, 97, 105, 106, 110, 114, 118, 122]
ARGS = {"downsample_density": 2.5, "patch_size": 60.0, "max_dist": 20.0}
SHUFFLE_SEED = 123


def _mesh(radius, centre, squash, n_lat=14, n_lon=20):
    """A latitude / longitude ellipsoid (triangles incl. a few degenerate ones at the poles, as real meshes have)."""
    th = np.linspace(0, np.pi, n_lat)
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    v = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.sin(th), np.sin(ph)) * squash, np.outer(np.cos(th), np.ones_like(ph))],
                 axis=-1).reshape(-1, 3) * radius + np.asarray(centre)[None]
    tris = []
    for i in range(n_lat - 1):
        for j in range(n_lon):
            a, b = i * n_lon + j, i * n_lon + (j + 1) % n_lon
            c, d = a + n_lon, b + n_lon
            tris += [[a, c, b], [b, c, d]]
    return v.astype(np.float32), np.asarray(tris, dtype=np.int32)


def write_eval_scene(out_dir, dataset_dir):
    from surf_amd import mesh_io
    os.makedirs(os.path.join(dataset_dir, "ObsMask"))
    os.makedirs(os.path.join(dataset_dir, "Points", "stl"))
    for k, scan in enumerate(SCANS):
        g = np.random.default_rng(1000 + scan)
        radius = 40.0 + 3.0 * k
        centre = np.array([5.0 * k, -3.0 * k, 10.0 + k])
        v, t = _mesh(radius, centre, 0.8 + 0.02 * k)
        mesh_io.write_ply(os.path.join(out_dir, "meshes", "final", f"scan{scan}.ply"), v, t)
        # the "scan": a noisy, slightly larger point set on the same ellipsoid family, part of it below the ground plane
        d = g.standard_normal((6000, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        stl = centre[None] + (radius + 2.0 + 0.3 * k) * d * np.array([1.0, 0.8 + 0.02 * k, 1.0]) + g.normal(0, 0.2, (6000, 3))
        with open(os.path.join(dataset_dir, "Points", "stl", f"stl{scan:03}_total.ply"), "wb") as f:
            f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {len(stl)}\nproperty float x\nproperty float y\n"
                     "property float z\nend_header\n").encode())
            f.write(np.ascontiguousarray(stl, dtype="<f4").tobytes())
        lo = centre - 1.6 * radius
        res = 3.2 * radius / 47
        obs = np.ones((48, 48, 48), np.uint8)
        obs[:6] = 0                                        # an unobserved slab: mesh points there do not count
        savemat(os.path.join(dataset_dir, "ObsMask", f"ObsMask{scan}_10.mat"),
                {"ObsMask": obs, "BB": np.stack([lo, lo + 3.2 * radius]).astype(np.float32), "Res": np.float32(res)})
        savemat(os.path.join(dataset_dir, "ObsMask", f"Plane{scan}.mat"),
                {"P": np.array([[0.0, 0.0, 1.0, -(centre[2] - 0.5 * radius)]])})      # z > centre_z - r/2 is "above"
