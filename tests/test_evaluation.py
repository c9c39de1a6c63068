"""Row f4: DTU Chamfer protocol and mesh cleaning.  Pinned against the reference's own code where that code can run here:
the Chamfer numbers of /root/reference/evaluation/dtu_eval.py on fifteen synthetic scans (tests/golden/dtu_eval_results.json) and
the visual-hull face filter `clean_mesh_by_mask` of utils/clean_mesh.py (tests/golden/clean_mesh.npz); the first-hit part
(trimesh + pyembree in the reference, a HIP z-buffer here) and the protocol's pieces are checked on analytic shapes."""
import numpy as np
import pytest
import torch

from oracle import mcubes_oracle as M
from surf_amd.evaluation import clean_mesh as C
from surf_amd.evaluation import dtu_eval as E


def _sphere_mesh(radius=50.0, n=36, centre=(0.0, 0.0, 0.0)):
    ax = np.linspace(-1.2, 1.2, n)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    u = (1.0 - np.sqrt(X * X + Y * Y + Z * Z)).astype(np.float32)
    v, t = M.marching_cubes(u, 0.0)
    v = (v / (n - 1) * 2.4 - 1.2) * radius + np.asarray(centre)[None]
    return v, t


def test_sample_mesh_points_covers_the_triangles():
    v = np.array([[0, 0, 0], [4.0, 0, 0], [0, 3.0, 0]])
    pts = E.sample_mesh_points(v, np.array([[0, 1, 2]]), 0.5)
    assert np.array_equal(pts[:3], v)
    s = pts[3:]
    assert len(s) == 24 and (s[:, 2] == 0).all() and (s[:, 0] > 0).all() and (s[:, 1] > 0).all()
    assert (s[:, 0] / 4 + s[:, 1] / 3 < 1).all()                      # strictly inside
    v2, t2 = _sphere_mesh(10.0, 24)
    p2 = E.sample_mesh_points(v2, t2, 0.5)
    density = (len(p2) - len(v2)) / (4 * np.pi * 100)
    assert 0.2 < density * 0.25 < 4.0                                  # of the order of one sample per thresh^2 of area (small triangles get fewer)
    degenerate = E.sample_mesh_points(v, np.array([[0, 0, 1]]), 0.5)   # zero-area triangles add nothing
    assert len(degenerate) == 3


def test_downsample_is_a_maximal_thinning():
    g = np.random.default_rng(0)
    pts = g.random((4000, 3)) * 5
    d = E.downsample_points(pts, 0.3, np.random.default_rng(1))
    from scipy.spatial import cKDTree
    tree = cKDTree(d)
    dist, _ = tree.query(d, k=2)
    assert dist[:, 1].min() > 0.3                                      # no two kept points within the radius
    assert cKDTree(d).query(pts)[0].max() <= 0.3 + 1e-12               # every input point is covered by a kept one


def test_chamfer_of_two_concentric_spheres():
    v, t = _sphere_mesh(50.0, 40)
    data = E.sample_mesh_points(v, t, 0.8)
    g = np.random.default_rng(2)
    d = g.standard_normal((60000, 3))
    stl = d / np.linalg.norm(d, axis=1, keepdims=True) * 53.0
    obs = np.ones((140, 140, 140), dtype=np.uint8)
    BB = np.array([[-70.0, -70, -70], [70, 70, 70]])
    plane = np.array([0.0, 0, 1, 0])                                   # keep the upper hemisphere of the scan for s2d
    d2s, s2d, overall = E.chamfer_dtu(data, stl, obs, BB, 1.0, plane, downsample_density=0.8, rng=np.random.default_rng(3))
    assert abs(d2s - 3.0) < 0.2 and abs(s2d - 3.0) < 0.25 and abs(overall - (d2s + s2d) / 2) < 1e-12
    # points outside the observability mask do not count for d2s: blank the lower half of the mask
    obs2 = obs.copy()
    obs2[:, :, :70] = 0
    far = np.concatenate([data, np.array([[0.0, 0.0, -69.0]] * 50)])    # outliers below, inside the masked-out region
    d2s2, _, _ = E.chamfer_dtu(far, stl, obs2, BB, 1.0, plane, downsample_density=0.8, rng=np.random.default_rng(3))
    assert abs(d2s2 - 3.0) < 0.2


def test_disk_dilation_and_components():
    m = np.zeros((40, 50), bool)
    m[20, 25] = True
    d = C.dilate_disk(m, 11)
    yy, xx = np.mgrid[:40, :50]
    assert np.array_equal(d, (yy - 20) ** 2 + (xx - 25) ** 2 <= 121)
    assert C.disk(3).shape == (7, 7) and C.disk(3).sum() == 29
    v1, t1 = _sphere_mesh(1.0, 20)
    v2, t2 = _sphere_mesh(0.2, 8, centre=(3, 0, 0))
    faces = np.concatenate([t1, t2 + len(v1)])
    keep = C.face_components(faces, min_len=500)
    assert keep[:len(t1)].all() and not keep[len(t1):].any() and len(t2) < 500 < len(t1)
    vv, ff = C.update_faces(np.concatenate([v1, v2]), faces, keep)
    assert len(vv) == len(v1) and np.array_equal(ff, t1)


def _cams(n=3, H=48, W=64):
    from surf_amd import synthetic
    intrs, c2ws, _ = synthetic.ring_cameras(n, H, W)
    return intrs, c2ws


def test_clean_mesh_by_mask_keeps_what_two_views_see():
    v, t = _sphere_mesh(0.5, 24)
    intrs, c2ws = _cams()
    masks = torch.ones(3, 48, 64)
    assert C.clean_mesh_by_mask(v, t, masks, intrs, c2ws, 1).all()
    masks[1:] = 0                                                      # visible in one view only: not "more than 1"
    assert not C.clean_mesh_by_mask(v, t, masks, intrs, c2ws, 1).any()
    half = torch.ones(3, 48, 64)
    half[:, :, :32] = 0                                                # left image half masked out in every view
    keep = C.clean_mesh_by_mask(v, t, half, intrs, c2ws, 1)
    assert 0.2 < keep.mean() < 0.8


# ---- GPU ---------------------------------------------------------------------------------------------------------------


def _first_hit_bruteforce(v, f, intr, c2w, H, W, up):
    xs = torch.linspace(0, W - 1, W * up, dtype=torch.float64)
    ys = torch.linspace(0, H - 1, H * up, dtype=torch.float64)
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    p = torch.stack([xx, yy, torch.ones_like(xx)], -1).reshape(-1, 3) @ torch.inverse(intr.double())[:3, :3].T
    d = (p / p.norm(dim=1, keepdim=True)) @ c2w.double()[:3, :3].T
    o = c2w.double()[:3, 3]
    V = torch.from_numpy(v)
    A, B, Cc = V[f[:, 0]], V[f[:, 1]], V[f[:, 2]]
    e1, e2 = B - A, Cc - A
    out = torch.full((d.shape[0],), -1, dtype=torch.long)
    for s in range(0, d.shape[0], 512):                                # Moeller-Trumbore, all faces per ray chunk
        dd = d[s:s + 512][:, None, :]
        pv = torch.cross(dd.expand(-1, len(f), -1), e2[None].expand(dd.shape[0], -1, -1), dim=-1)
        det = (e1[None] * pv).sum(-1)
        tv = (o - A)[None]
        uu = (tv * pv).sum(-1) / det
        qv = torch.cross(tv.expand(dd.shape[0], -1, -1), e1[None].expand(dd.shape[0], -1, -1), dim=-1)
        vv = (dd * qv).sum(-1) / det
        tt = (e2[None] * qv).sum(-1) / det
        ok = (det.abs() > 1e-14) & (uu >= 0) & (vv >= 0) & (uu + vv <= 1) & (tt > 0)
        tt = torch.where(ok, tt, torch.full_like(tt, float("inf")))
        best = tt.argmin(dim=1)
        out[s:s + 512] = torch.where(torch.isfinite(tt.min(dim=1).values), best, torch.full_like(best, -1))
    return out.reshape(H * up, W * up)


@pytest.mark.gpu
def test_raster_first_hit_matches_ray_casting():
    from surf_amd import ops
    v, t = _sphere_mesh(0.5, 22)
    v2, t2 = _sphere_mesh(0.15, 10, centre=(0.2, 0.1, -0.9))           # a smaller blob in front of the sphere for view 0
    v = np.concatenate([v, v2])
    t = np.concatenate([t, t2 + (len(v) - len(v2))])
    intrs, c2ws = _cams()
    dv = torch.from_numpy(v).float().cuda().contiguous()
    df = torch.from_numpy(t).int().cuda().contiguous()
    for i in range(3):
        ids = ops.raster_first_hit(dv, df, intrs[i], c2ws[i], (48, 64), 2).cpu()
        ref = _first_hit_bruteforce(v, t, intrs[i], c2ws[i], 48, 64, 2)
        assert ids.shape == ref.shape and int((ref >= 0).sum()) > 300
        agree = (ids == ref).float().mean()
        assert float(agree) > 0.995, float(agree)                       # rays grazing a shared edge may pick the neighbour
        assert torch.equal(ids >= 0, ref >= 0) or float(((ids >= 0) != (ref >= 0)).float().mean()) < 2e-3


@pytest.mark.gpu
def test_clean_mesh_drops_hidden_faces_and_small_components():
    v, t = _sphere_mesh(0.5, 40)
    v2, t2 = _sphere_mesh(0.06, 8, centre=(0.0, 0.62, 0.0))            # a floater of < 500 faces beside the object
    vv = np.concatenate([v, v2])
    ff = np.concatenate([t, t2 + len(v)])
    intrs, c2ws = _cams(3, 96, 128)
    masks = torch.ones(3, 96, 128)
    cv, cf = C.clean_mesh(vv, ff, masks, intrs, c2ws, dilation_radius=3, min_nb_visible=1, upscale=2, min_component=500)
    assert 0.2 * len(t) < len(cf) < 0.8 * len(t)                       # the far side of the sphere is nobody's first hit
    assert np.abs(np.linalg.norm(cv, axis=1) - 0.5).max() < 0.02        # the floater is gone, only sphere vertices remain
    assert cf.max() < len(cv) and len(np.unique(cf)) == len(cv)
    # camera-facing side kept: all cameras sit at z < 0
    assert cv[:, 2].mean() < -0.1


def test_dtu_eval_equals_the_reference_evaluator(tmp_path):
    """Row f4 against the REFERENCE itself: tests/golden/dtu_eval_results.json is the results.json that
    /root/reference/evaluation/dtu_eval.py wrote for the fifteen synthetic scans of tests/golden/eval_scene.py (run as the script it
    is by tests/golden/make_golden_eval.py; open3d's file readers and tqdm stood in for, its shuffle seeded).  surf_amd's
    evaluator on the same files with the same shuffle seed reproduces accuracy, completeness and their mean for every scan."""
    import json
    import os
    from tests.golden import eval_scene
    with open(os.path.join(os.path.dirname(__file__), "golden", "dtu_eval_results.json")) as f:
        gold = json.load(f)
    out_dir, data_dir = str(tmp_path / "exp"), str(tmp_path / "eval")
    eval_scene.write_eval_scene(out_dir, data_dir)
    rows = []
    for scan in eval_scene.SCANS:
        d2s, s2d, overall = E.evaluate_scan(os.path.join(out_dir, "meshes", "final", f"scan{scan}.ply"), data_dir, scan,
                                            rng=np.random.default_rng(eval_scene.SHUFFLE_SEED), **eval_scene.ARGS)
        ref = gold[str(scan)]
        assert abs(d2s - ref["d2s"]) < 1e-9 * ref["d2s"] + 1e-12, (scan, d2s, ref["d2s"])
        assert abs(s2d - ref["s2d"]) < 1e-9 * ref["s2d"] + 1e-12, (scan, s2d, ref["s2d"])
        assert abs(overall - ref["all"]) < 1e-9 * ref["all"] + 1e-12
        rows.append((d2s, s2d, overall))
    m = np.mean(np.array(rows), axis=0)
    for got, key in zip(m, ("d2s", "s2d", "all")):
        assert abs(got - gold["mean"][key]) < 1e-9 * gold["mean"][key]
    # the command-line entry writes the same file
    E.main(["--out_dir", out_dir, "--dataset_dir", data_dir, "--downsample_density", str(eval_scene.ARGS["downsample_density"]),
            "--shuffle_seed", str(eval_scene.SHUFFLE_SEED)])
    with open(os.path.join(out_dir, "results.json")) as f:
        mine = json.load(f)
    assert set(mine) == set(gold) and abs(mine["mean"]["all"] - gold["mean"]["all"]) < 1e-9 * gold["mean"]["all"]


def test_clean_mesh_by_mask_equals_the_reference(tmp_path):
    """The visual-hull face filter against the REFERENCE's own `clean_mesh_by_mask` (utils/clean_mesh.py:9-34, imported and run by
    tests/golden/make_golden_clean.py): identical keep-masks for min_nb_visible = 0, 1, 2 on a synthetic mesh whose vertices fall
    inside, outside and on the borders of three masks (one with a hole)."""
    from tests.conftest import load_npz
    g = load_npz("clean_mesh.npz")
    for m in (0, 1, 2):
        keep = C.clean_mesh_by_mask(g["vertices"].numpy(), g["faces"].numpy(), g["masks"], g["intrs"], g["c2ws"], m)
        ref = g[f"keep{m}"].numpy().astype(bool)
        assert 0 < ref.sum() < len(ref)
        assert np.array_equal(keep, ref), (m, int((keep != ref).sum()))
