"""DTU Chamfer evaluation (evaluation/dtu_eval.py:31-190 of the reference, the protocol of the DTU / MVSNet / NeuS
evaluations): sample the mesh to a point cloud, thin it to `downsample_density`, keep the points inside the observability
mask, then mean nearest-neighbour distances mesh -> scan (d2s, accuracy) and scan (above the ground plane) -> mesh (s2d,
completeness), both clipped at `max_dist`; overall = their mean.

open3d / tqdm / multiprocessing of the reference are replaced by numpy + scikit-learn's kd-tree (the reference uses the
same `sklearn.neighbors.NearestNeighbors`); meshes are read with surf_amd.mesh_io, STL point clouds with
surf_amd.datasets.mvs_io.read_ply_points, the .mat files with scipy.io.loadmat.  The DTU evaluation data are not in the
build container: tests/test_evaluation.py checks the protocol on analytic shapes.

    python -m surf_amd.evaluation.dtu_eval --out_dir <exp dir> --dataset_dir <DTU evaluation dir>
"""
import argparse
import json
import os

import numpy as np
import sklearn.neighbors as skln

DTU_TEST_SCANS = [24, 37, 40, 55, 63, 65, 69, 83, 97, 105, 106, 110, 114, 118, 122]


def sample_mesh_points(vertices, triangles, thresh):
    """dtu_eval.py:64-85 (+ sample_single_tri :12-21): every triangle with non-zero area is covered by the lattice
    ((i + 0.5) / n1, (j + 0.5) / n2), u + v < 1, n = floor(edge length / (thresh sqrt(l1 l2 / area2))); returns the
    vertices followed by the samples, triangle by triangle."""
    vertices = np.asarray(vertices, dtype=np.float64)
    tri_vert = vertices[np.asarray(triangles)]
    v1 = tri_vert[:, 1] - tri_vert[:, 0]
    v2 = tri_vert[:, 2] - tri_vert[:, 0]
    l1 = np.linalg.norm(v1, axis=-1)
    l2 = np.linalg.norm(v2, axis=-1)
    area2 = np.linalg.norm(np.cross(v1, v2), axis=-1)
    ok = area2 > 0
    l1, l2, area2, v1, v2, origin = l1[ok], l2[ok], area2[ok], v1[ok], v2[ok], tri_vert[ok, 0]
    thr = thresh * np.sqrt(l1 * l2 / area2)
    n1 = np.floor(l1 / thr).astype(np.int64)
    n2 = np.floor(l2 / thr).astype(np.int64)
    out = [vertices]
    # triangles grouped by their lattice size: one vectorised pass per distinct (n1, n2)
    keys, inverse = np.unique(np.stack([n1, n2], axis=1), axis=0, return_inverse=True)
    pieces, order = [], []
    for k, (a, b) in enumerate(keys):
        idx = np.nonzero(inverse.reshape(-1) == k)[0]
        c = np.mgrid[:a + 1, :b + 1].astype(np.float64) + 0.5
        c[0] /= max(a, 1e-7)
        c[1] /= max(b, 1e-7)
        c = np.transpose(c, (1, 2, 0))
        kk = c[c.sum(axis=-1) < 1]                                   # (m, 2)
        if kk.shape[0] == 0:
            continue
        q = v1[idx, None, :] * kk[None, :, :1] + v2[idx, None, :] * kk[None, :, 1:] + origin[idx, None, :]
        pieces.append(q.reshape(-1, 3))
        order.append(np.repeat(idx, kk.shape[0]))
    if pieces:
        pts = np.concatenate(pieces)
        out.append(pts[np.argsort(np.concatenate(order), kind="stable")])
    return np.concatenate(out, axis=0)


def downsample_points(points, thresh, rng=None):
    """dtu_eval.py:97-112: shuffle, then greedy suppression: a kept point removes every later point within `thresh`."""
    points = np.array(points, dtype=np.float64)
    (rng or np.random.default_rng()).shuffle(points, axis=0)
    nn = skln.NearestNeighbors(n_neighbors=1, radius=thresh, algorithm="kd_tree", n_jobs=-1)
    nn.fit(points)
    rnn = nn.radius_neighbors(points, radius=thresh, return_distance=False)
    mask = np.ones(points.shape[0], dtype=bool)
    for cur, idxs in enumerate(rnn):
        if mask[cur]:
            mask[idxs] = False
            mask[cur] = True
    return points[mask]


def chamfer_dtu(data_pcd, stl, obs_mask, BB, Res, ground_plane, patch_size=60, max_dist=20, downsample_density=0.2, rng=None):
    """dtu_eval.py:97-150 for one scan.  Returns (mean_d2s, mean_s2d, overall)."""
    data_down = downsample_points(data_pcd, downsample_density, rng)
    BB = np.asarray(BB, dtype=np.float32)
    inbound = ((data_down >= BB[:1] - patch_size) & (data_down < BB[1:] + patch_size * 2)).sum(axis=-1) == 3
    data_in = data_down[inbound]
    data_grid = np.around((data_in - BB[:1]) / Res).astype(np.int32)
    grid_inbound = ((data_grid >= 0) & (data_grid < np.expand_dims(obs_mask.shape, 0))).sum(axis=-1) == 3
    g = data_grid[grid_inbound]
    in_obs = obs_mask[g[:, 0], g[:, 1], g[:, 2]].astype(bool)
    data_in_obs = data_in[grid_inbound][in_obs]
    nn = skln.NearestNeighbors(n_neighbors=1, algorithm="kd_tree", n_jobs=-1)
    nn.fit(stl)
    dist_d2s, _ = nn.kneighbors(data_in_obs, n_neighbors=1, return_distance=True)
    mean_d2s = dist_d2s[dist_d2s < max_dist].mean()
    stl_hom = np.concatenate([stl, np.ones_like(stl[:, :1])], -1)
    above = (np.asarray(ground_plane).reshape((1, 4)) * stl_hom).sum(-1) > 0
    nn.fit(data_in)
    dist_s2d, _ = nn.kneighbors(stl[above], n_neighbors=1, return_distance=True)
    mean_s2d = dist_s2d[dist_s2d < max_dist].mean()
    return float(mean_d2s), float(mean_s2d), float((mean_d2s + mean_s2d) / 2)


def evaluate_scan(mesh_file, dataset_dir, scan, **kw):
    from scipy.io import loadmat
    from .. import mesh_io
    from ..datasets import mvs_io
    vertices, triangles = mesh_io.read_ply(mesh_file)
    data_pcd = sample_mesh_points(vertices, triangles, kw.get("downsample_density", 0.2))
    m = loadmat(f"{dataset_dir}/ObsMask/ObsMask{scan}_10.mat")
    plane = loadmat(f"{dataset_dir}/ObsMask/Plane{scan}.mat")["P"]
    stl = mvs_io.read_ply_points(f"{dataset_dir}/Points/stl/stl{scan:03}_total.ply")
    return chamfer_dtu(data_pcd, stl, m["ObsMask"], m["BB"], m["Res"], plane, **kw)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out_dir", type=str, default="./outputs")
    ap.add_argument("--dataset_dir", type=str, required=True)
    ap.add_argument("--scans", type=int, nargs="*", default=DTU_TEST_SCANS)
    ap.add_argument("--downsample_density", type=float, default=0.2)
    ap.add_argument("--patch_size", type=float, default=60)
    ap.add_argument("--max_dist", type=float, default=20)
    args = ap.parse_args(argv)
    results, rows = {}, []
    for scan in args.scans:
        mesh = os.path.join(args.out_dir, "meshes", "final", f"scan{scan}.ply")
        d2s, s2d, overall = evaluate_scan(mesh, args.dataset_dir, scan, patch_size=args.patch_size, max_dist=args.max_dist,
                                          downsample_density=args.downsample_density)
        print(scan, d2s, s2d, overall)
        results[scan] = {"d2s": d2s, "s2d": s2d, "all": overall}
        rows.append((d2s, s2d, overall))
    m = np.mean(np.array(rows), axis=0)
    results["mean"] = {"d2s": float(m[0]), "s2d": float(m[1]), "all": float(m[2])}
    print("final result", *m)
    with open(os.path.join(args.out_dir, "results.json"), "w") as fp:
        json.dump(results, fp, indent=True)


if __name__ == "__main__":
    main()
