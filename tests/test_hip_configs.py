"""GPU parity of the render path in the other BASELINE.json configurations (parity-test cases, not bench lines):
config 1 (3 views, 64 samples/ray), config 5 (7 views, 192 samples/ray, non-square images) and the reference's
default 136-sample split, each against the CPU oracle on a small synthetic sphere pyramid."""
import pytest
import torch

from oracle import surf_oracle as O

pytestmark = pytest.mark.gpu


def _run(nv, H, W, n_samples, step, base=8, bands=(float("inf"), 0.92, 0.3, 0.1)):
    from bench import model_conf
    from surf_amd import synthetic
    from surf_amd.implicit_surface import ImplicitSurface
    dev = torch.device("cuda:0")
    torch.manual_seed(nv)
    model = ImplicitSurface(model_conf(n_samples)).to(dev)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        model.deviation_network.variance.fill_(0.45)
        for l in range(1, 7):                       # let the sparse-volume channels matter (zero at geometric init)
            lin = getattr(model.sdf_network, f"lin{l}")
            lin.weight_v[:, -28:] += (0.05 * torch.randn(lin.weight_v.shape[0], 28, generator=g)).to(dev)
    intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
    imgs = synthetic.procedural_images(nv, H, W, 0, dev)
    feats = synthetic.feature_pyramid(nv, H, W, 0, dev)
    vols, tabs, mvol = synthetic.sphere_pyramid(base, dev, bands=bands)
    scene = model.scene(mvol, vols[::-1], tabs[::-1], None, feats, imgs, intrs.to(dev), c2ws.to(dev))
    rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], H, W, step, dev)
    R = rays_o.shape[0]
    near = near_fars[0, 0].reshape(1, 1).repeat(R, 1)
    far = near_fars[0, 1].reshape(1, 1).repeat(R, 1)
    out = model.render_scene(rays_o, rays_d, near.to(dev), far.to(dev), scene, 1.0)
    torch.cuda.synchronize()
    sd = {"implicit_surface." + k: v.detach().cpu() for k, v in model.state_dict().items()}
    tabs_c = [t.cpu().long() for t in tabs[::-1]]
    ref = O.render(sd, rays_o.cpu(), rays_d.cpu(), near, far, mvol.cpu(), [v[:, :7].cpu() for v in vols[::-1]], tabs_c,
                   [(t >= 0).float() for t in tabs_c], [f.cpu() for f in feats], imgs.cpu(), intrs, c2ws, n_samples,
                   [1.0, 0.4, 0.1, 0.01], 256, 1.0)
    return out, ref


def _check(out, ref):
    def close(a, b, rtol, atol):
        a, b = a.detach().float().cpu().reshape(-1), b.detach().float().reshape(-1)
        err = (a - b).abs()
        assert bool((err <= atol + rtol * b.abs()).all()), float(err.max())
    close(out["sdf"], ref["sdf"], 0, 1e-4)
    close(out["mid_z_vals"], ref["mid_z_vals"], 0, 3e-6)
    close(out["weights"], ref["weights"], 1e-3, 3e-5)
    close(out["color_fine"], ref["color_fine"], 1e-3, 3e-5)
    close(out["render_depth"], ref["render_depth"], 1e-3, 3e-5)
    close(out["sdf_depth"], ref["sdf_depth"], 1e-3, 3e-5)
    assert torch.equal(out["valid_mask"].cpu(), ref["valid_mask"])
    assert float(ref["weights"].sum(1).max()) > 0.5


def test_config1_three_views_64_samples():
    _check(*_run(3, 48, 64, [32, 16, 8, 8], 4))


def test_config5_seven_views_192_samples():
    _check(*_run(7, 72, 128, [96, 48, 32, 16], 8))


def test_reference_default_136_samples_five_views():
    _check(*_run(5, 48, 64, [64, 32, 24, 16], 4))


def test_bench_train_workload_line_on_one_gpu():
    """BASELINE configs[3] at reduced size: `python bench.py --workload train` (the data-parallel training step as the bench runs
    it; N = 1 here, the same code path that DistributedDataParallel wraps for N > 1) prints ONE well-formed line: training rays/s,
    steps/s, per-kernel rooflines of the backward kernels from HIP events inside the timed region, a finite loss."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "train", "--steps", "2", "--warmup", "1", "--height", "96",
                          "--width", "128", "--base-dim", "16", "--rays", "128"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["scaling"] == "weak" and r["data"] == "synthetic" and r["unit"] == "rays/s"
    assert r["config"]["rays_per_rank_step"] == 128 and r["config"]["parallelism"] == "ddp1" and r["config"]["train_precision"] == "fp32"      # the forced world-1 RCCL group
    assert r["collective_backend"].startswith("nccl") and r["gradient_allreduce_ms"] > 0
    assert abs(r["value"] - 128 * r["steps_per_s"]) < 1e-6 * r["value"] and r["loss"] == r["loss"]
    names = {e["kernel"] for e in r["roofline_kernels"]}
    assert {"costvol_bwd", "matching_depth_bwd", "sdf_bwd", "blend_bwd", "colgram"} <= names
    assert any(n.startswith("spconv_wgrad<") for n in names)
    top = r["roofline"]
    assert top is not None and top["bound"] in ("hbm", "valu") and 0 < top["frac"]


def _bench(args, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


SMALL = ["--steps", "2", "--warmup", "1", "--height", "48", "--width", "64", "--base-dim", "16", "--cpu-seconds", "0", "--build", "0",
         "--train-step", "0", "--mesh-grid", "0", "--also", ""]


def test_bench_scene_dealing_on_one_gpu():
    """BASELINE configs[2] (15 scans scene-parallel) at reduced size: `bench.py --gpus 1 --scenes 3` deals three different synthetic
    scenes (seeds 0..2) to the one rank, renders each with the real kernels inside the timed region and reports one record per scene."""
    r = _bench(["--gpus", "1", "--scenes", "3"] + SMALL)
    assert r["n_gpus"] == 1 and r["config"]["scenes"] == 3 and r["config"]["rays_per_step"] == 3 * 48 * 64
    recs = r["scenes"]
    assert [x["scene"] for x in recs] == [0, 1, 2] and all(x["rank"] == 0 and x["ms_per_render"] > 0 for x in recs)
    assert abs(r["value"] - 3 * 48 * 64 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    assert sum(x["ms_per_render"] for x in recs) <= r["ms_per_step"] * 1.05


def test_bench_two_ranks_share_the_gpu_render():
    """The N > 1 path of the inference bench with its REAL kernels: two fresh ranks started by bench.py's own launcher, both on
    cuda:0, process group gloo (RCCL wants one device per rank; the collectives here are only the barrier, the MAX of the elapsed
    time on a device tensor and the record gather).  Five scenes dealt 3 + 2; rank 0 relays one line."""
    r = _bench(["--gpus", "2", "--scenes", "5", "--backend", "gloo", "--one-gpu"] + SMALL)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["collective_backend"].startswith("gloo")
    recs = r["scenes"]
    assert [x["scene"] for x in recs] == [0, 1, 2, 3, 4] and [x["rank"] for x in recs] == [0, 1, 0, 1, 0]
    assert abs(r["value"] - 5 * 48 * 64 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]


def test_bench_two_ranks_share_the_gpu_train():
    """BASELINE configs[3]'s N > 1 leg with its real kernels: two ranks on cuda:0, DistributedDataParallel over gloo (bucket hooks,
    the broadcast at wrap time, the gradient all-reduce timing) - what RCCL serves on an 8-GPU node."""
    r = _bench(["--workload", "train", "--gpus", "2", "--backend", "gloo", "--one-gpu", "--steps", "2", "--warmup", "1", "--height", "96",
                "--width", "128", "--base-dim", "16", "--rays", "128"])
    assert r["n_gpus"] == 2 and r["config"]["parallelism"] == "ddp2" and r["config"]["rays_per_rank_step"] == 128
    assert r["gradient_allreduce_ms"] > 0 and r["loss"] == r["loss"] and r["collective_backend"].startswith("gloo")
    assert abs(r["value"] - 2 * 128 * r["steps_per_s"]) < 1e-6 * r["value"]


def test_bench_single_scene_split_over_two_ranks_is_bit_equal():
    """SURVEY 8e's single-scene split: `bench.py --split rays --gpus 2` - ONE image on two ranks (both on cuda:0,
    gloo), rank r renders its half of the pixel rays and its x-range of the mesh lattice, rank 0 stitches.  Rays and lattice points
    are independent, so the stitched image and the stitched `u` lattice must be BIT-equal to rank 0's own single-rank results."""
    r = _bench(["--split", "rays", "--check-split", "--gpus", "2", "--backend", "gloo", "--one-gpu", "--steps", "2", "--warmup", "1",
                "--height", "48", "--width", "64", "--base-dim", "16", "--mesh-grid", "48"])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["config"]["split"] == "rays"
    assert r["config"]["rays_per_rank"] == [1536, 1536] and r["config"]["rays_per_step"] == 48 * 64
    assert abs(r["value"] - 48 * 64 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    assert r["split"]["check"] == {"image_bit_equal": True, "lattice_bit_equal": True}
    assert r["split"]["lattice"]["resolution"] == 48 and r["split"]["lattice"]["sdf_ms"] > 0
    # ... and N = 1 through the same code path
    r1 = _bench(["--split", "rays", "--check-split", "--gpus", "1", "--steps", "1", "--warmup", "1", "--height", "48", "--width", "64",
                 "--base-dim", "16", "--mesh-grid", "32"])
    assert r1["n_gpus"] == 1 and r1["split"]["check"]["image_bit_equal"] and r1["split"]["check"]["lattice_bit_equal"]
