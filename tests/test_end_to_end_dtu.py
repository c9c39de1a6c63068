"""GPU, rows f3 + a17 + f1 + f4 in one chain: a synthetic scene written in DTU's own file formats -> surf_amd.datasets.get_loader
-> SuRF.forward("val") (FPN, 4-stage volume build, render, 128^3 lattice + marching cubes) -> mesh_io.export_mesh with the
item's scale_mat -> evaluation.dtu_eval.evaluate_scan against an analytic ground truth with a KNOWN Chamfer distance.

The SDF network's geometric initialisation is a closed near-spherical surface around the origin of the normalised frame
whatever the feature volumes hold (sdf_network.py:62-86 zeroes the feature columns), so the exported mesh surrounds a known
world-frame centre (scale_mat's translation); the ground-truth "scan" is that surface pushed outwards by delta world units,
for which accuracy and completeness are both ~ delta."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from tests.test_datasets import _ring_cams, _write_cam

pytestmark = pytest.mark.gpu


def _write_scene(root, H=96, W=128, n_views=4):
    from surf_amd.datasets import mvs_io
    g = np.random.default_rng(11)
    for sub in ("Cameras", "Rectified_raw/scan24", "Depths_raw/scan24", "Pseudo_depths/scan24", "Pseudo_points"):
        os.makedirs(root / sub)
    K = np.array([[2892.33, 0, 823.2], [0, 2883.18, 619.07], [0, 0, 1.0]])
    for v, w2c in enumerate(_ring_cams(n_views)):
        _write_cam(root / "Cameras" / f"{v:08d}_cam.txt", w2c, K, 425.0, 2.5)
        yy, xx = np.mgrid[:H, :W]
        img = np.stack([0.5 + 0.5 * np.sin(0.11 * xx + 0.07 * yy + v), 0.5 + 0.5 * np.sin(0.05 * xx - 0.13 * yy + 2 * v),
                        0.5 + 0.5 * np.cos(0.09 * xx + 0.03 * yy)], axis=-1)
        Image.fromarray((img * 255).astype(np.uint8)).save(root / "Rectified_raw/scan24" / f"rect_{v + 1:03d}_3_r5000.png")
        Image.fromarray(np.full((H, W), 255, np.uint8)).save(root / "Depths_raw/scan24" / f"depth_visual_{v:04d}.png")
        mvs_io.write_pfm(root / "Depths_raw/scan24" / f"depth_map_{v:04d}.pfm", np.full((H, W), 600.0, np.float32))
        mvs_io.write_pfm(root / "Pseudo_depths/scan24" / f"{v:08d}.pfm", np.full((H, W), 600.0, np.float32))
    (root / "Cameras" / "pair.txt").write_text(f"{n_views}\n" + "".join(
        f"{r}\n{n_views - 1} " + " ".join(f"{s} 1.0" for s in range(n_views) if s != r) + "\n" for r in range(n_views)))
    with open(root / "Pseudo_points" / "mvsnet024_l3.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 2500\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        for p in g.standard_normal((2500, 3)) * 40:
            f.write(" ".join(f"{v:.5f}" for v in p) + "\n")


def test_dtu_files_to_chamfer(tmp_path):
    from scipy.io import savemat
    from bench import surf_conf
    from surf_amd import conf, mesh_io, synthetic
    from surf_amd.datasets import get_loader
    from surf_amd.evaluation import dtu_eval
    from surf_amd.surf import SuRF
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    H, W = 96, 128
    root = tmp_path / "dtu"
    _write_scene(root, H, W)
    dconf = conf.from_dict({"dataset_name": "DTUDataset", "data_dir": str(root), "scene": ["scan24"], "ref_view": [1], "light_idx": [3],
                            "num_src_view": 2, "val_res_level": 2, "factor": 1.0, "interval_scale": 1, "num_interval": 192,
                            "img_hw": [H, W], "total_views": 4})
    loader, _, dataset = get_loader(dconf, "val", False, num_workers=0)
    assert len(dataset) == 1
    np.random.seed(0)
    item = next(iter(loader))
    inputs = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in item.items()}             # runner.py's tocuda
    inputs["mesh_resolution"] = 128

    torch.manual_seed(0)
    mcfg = surf_conf(base_dim=16)
    model = SuRF(conf.from_dict(mcfg)).to(dev).eval()
    model.logit_override = synthetic.sphere_logit          # untrained U-Nets: a surface-concentrated pyramid like a trained one's
    with torch.no_grad():
        out = model("val", inputs, cos_anneal_ratio=1.0, step=0)                                  # runner.py:216
    h, w = H // 2, W // 2
    assert out["img_fine"].shape == (h, w, 3) and out["normal_img"].shape == (h, w, 3) and out["render_depth"].shape == (h, w)
    assert np.isfinite(out["img_fine"]).all() and out["depth_stage3"].shape == (H, W)
    v, t = out["vertices"], out["triangles"]
    assert len(v) > 1000 and len(t) > 2000
    # the geometric initialisation (sdf_network.py:62-86 zeroes the feature columns): a closed, star-shaped, roughly spherical
    # surface around the origin whatever the feature volumes hold
    r_norm = np.linalg.norm(v, axis=1)
    assert 0.4 < float(r_norm.mean()) < 0.9 and float(r_norm.std()) < 0.15 * float(r_norm.mean()), (r_norm.mean(), r_norm.std())
    edges = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), axis=1)
    _, counts = np.unique(edges, axis=0, return_counts=True)
    assert (counts == 2).all() and len(v) - len(counts) + len(t) == 2            # closed 2-manifold of genus 0
    # the central ray of the reference view hits it: rendered depth ~ |camera centre| - (surface radius along that ray)
    c = item["c2ws"][0, :3, 3].norm().item()
    centre_depth = float(out["render_depth"][h // 2, w // 2])
    assert c - float(r_norm.max()) - 0.05 < centre_depth < c - float(r_norm.min()) + 0.05, (centre_depth, c)

    # ---- runner.py:231-240: world-frame PLY ----
    mesh_path = tmp_path / "exp" / "meshes" / "final" / "scan24.ply"
    vw = mesh_io.export_mesh(str(mesh_path), v, t, item["scale_mat"])
    S = item["scale_mat"].double().numpy()
    centre_w, radius_scale = S[:3, 3], float(np.linalg.norm(S[:3, 0]))
    r_world = float(np.linalg.norm(vw - centre_w[None], axis=1).mean())
    assert abs(r_world - radius_scale * float(r_norm.mean())) < 1e-3 * r_world

    # ---- DTU evaluation files for "scan 24": the scan is the mesh surface pushed outwards by delta ----
    delta = 6.0
    density = r_world / 60.0
    ev = tmp_path / "dtu_eval"
    os.makedirs(ev / "ObsMask")
    os.makedirs(ev / "Points" / "stl")
    surf = dtu_eval.sample_mesh_points(vw, t, density)
    # outward normals of the near-spherical surface ~ radial directions (cos of the angle between them >= 0.9 here)
    radial = (surf - centre_w[None]) / np.linalg.norm(surf - centre_w[None], axis=1, keepdims=True)
    stl = surf + delta * radial
    with open(ev / "Points" / "stl" / "stl024_total.ply", "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {len(stl)}\nproperty float x\nproperty float y\n"
                 "property float z\nend_header\n").encode())
        f.write(np.ascontiguousarray(stl, dtype="<f4").tobytes())
    lo, hi = centre_w - 2 * r_world, centre_w + 2 * r_world
    res = 4.0 * r_world / 63
    savemat(ev / "ObsMask" / "ObsMask24_10.mat", {"ObsMask": np.ones((64, 64, 64), np.uint8), "BB": np.stack([lo, hi]).astype(np.float32),
                                                  "Res": np.float32(res)})
    savemat(ev / "ObsMask" / "Plane24.mat", {"P": np.array([[0.0, 0.0, 1.0, -(lo[2] - 1.0)]])})       # everything is above it
    d2s, s2d, overall = dtu_eval.evaluate_scan(str(mesh_path), str(ev), 24, downsample_density=density, patch_size=60,
                                               max_dist=20, rng=np.random.default_rng(0))
    # both directions measure the offset: delta x cos(angle between radial direction and surface normal), plus at most the
    # spacing of the thinned clouds
    assert 0.85 * delta < d2s < delta + 0.1, d2s
    assert 0.85 * delta < s2d < delta + density, s2d
    assert abs(overall - 0.5 * (d2s + s2d)) < 1e-9

    # ---- the same chain as ONE command: scripts/dtu_chamfer.py (conf + data_dir + eval_dir + scan -> one JSON record) ----
    import json
    import sys
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root_dir, "scripts"))
    import dtu_chamfer
    conf_path = tmp_path / "surf_synth.conf"
    conf_path.write_text(json.dumps({"model": mcfg, "val_dataset": {k: dconf[k] for k in dconf}}, indent=1))    # (JSON is a HOCON subset)
    rec = dtu_chamfer.run(dtu_chamfer.parse_args([
        "--conf", str(conf_path), "--eval_dir", str(ev), "--scan", "24", "--ref_view", "1", "--out_dir", str(tmp_path / "exp2"),
        "--mesh_resolution", "128", "--downsample_density", str(density), "--logit_override", "sphere", "--reference_chamfer", str(overall)]))
    assert rec["scan"] == 24 and rec["down_rule"] == "pad0" and rec["views"] == 3 and rec["triangles"] == len(t)
    # same seeded weights, same files, same evaluator seed: the command reproduces the chain above
    assert abs(rec["chamfer"] - overall) < 1e-6 and abs(rec["delta"]) < 1e-6, (rec["chamfer"], overall)
    assert os.path.exists(rec["mesh"]) and json.load(open(tmp_path / "exp2" / "chamfer_scan24.json"))["chamfer"] == rec["chamfer"]
    # ... and with another stride-2 site rule it is a different network (row a5: three candidates behind one switch)
    rec0 = dtu_chamfer.run(dtu_chamfer.parse_args([
        "--conf", str(conf_path), "--eval_dir", str(ev), "--scan", "24", "--ref_view", "1", "--out_dir", str(tmp_path / "exp3"),
        "--mesh_resolution", "64", "--downsample_density", str(density), "--logit_override", "sphere", "--down_rule", "dilate"]))
    assert rec0["down_rule"] == "dilate" and np.isfinite(rec0["chamfer"])
    # ---- --sweep: the 3 x 2 x 2 grid of torchsparse conventions on ONE loaded model, the discriminator of VERDICT r5 item 3 ----
    # A fresh SDF network ignores its feature columns (geometric initialisation zeroes them: every candidate would give the same
    # mesh), so the sweep runs on a CHECKPOINT whose lin1..lin5 read the sparse features.  The "reference" number is the
    # pad0 / zfast / mirrored network's own Chamfer: the sweep must single that combination out of the twelve.
    torch.manual_seed(0)
    ck_model = SuRF(conf.from_dict(mcfg))
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for l in range(1, 6):
            wv = getattr(ck_model.implicit_surface.sdf_network, f"lin{l}").weight_v
            wv[:, -28:] += 0.05 * torch.randn(wv.shape[0], 28, generator=g)
    ckpt_path = tmp_path / "ckpt_features.pth"
    torch.save({"model": ck_model.state_dict()}, ckpt_path)
    base = ["--conf", str(conf_path), "--ckpt", str(ckpt_path), "--eval_dir", str(ev), "--scan", "24", "--ref_view", "1",
            "--mesh_resolution", "64", "--downsample_density", str(density), "--logit_override", "sphere"]
    target = dtu_chamfer.run(dtu_chamfer.parse_args(base + ["--out_dir", str(tmp_path / "exp4"), "--kernel_order", "zfast",
                                                            "--transposed_pairing", "mirrored"]))
    assert (target["down_rule"], target["kernel_order"], target["transposed_pairing"]) == ("pad0", "zfast", "mirrored")
    assert target["missing_keys"] == [] and target["unexpected_keys"] == []
    sw = dtu_chamfer.run(dtu_chamfer.parse_args(base + ["--out_dir", str(tmp_path / "exp5"), "--sweep",
                                                        "--reference_chamfer", repr(target["chamfer"])]))
    assert len(sw["sweep"]) == 12 and len({(r["down_rule"], r["kernel_order"], r["transposed_pairing"]) for r in sw["sweep"]}) == 12
    chamfers = [r["chamfer"] for r in sw["sweep"] if r["chamfer"] is not None]
    assert len({round(c, 7) for c in chamfers}) >= 8, chamfers                       # the candidates are different networks
    best = sw["best"]
    assert (best["down_rule"], best["kernel_order"], best["transposed_pairing"]) == ("pad0", "zfast", "mirrored"), sw["sweep"]
    assert abs(best["chamfer"] - target["chamfer"]) < 1e-9
    assert [(r["down_rule"], r["kernel_order"], r["transposed_pairing"]) for r in sw["within_0.01_of_reference"]][:1] != []
    assert os.path.exists(tmp_path / "exp5" / "chamfer_sweep_scan24.json")
