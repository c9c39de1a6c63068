#!/usr/bin/env python3
"""Headline benchmark of the SuRF hot path on MI355X: full-image render throughput (rays/s).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the render hot path (ray set-up -> SDF MLP + gradient -> multi-view blending ->
NeuS compositing) over every pixel ray of the 576x800 reference view of a synthetic 5-view scene with
128 samples per ray (BASELINE.json configs[1]; synthetic sphere pyramid 88^3 -> 704^3).
Inputs are resident in HBM before the timed region.  With N > 1 every rank renders its own scene
(seed = rank): scenes are independent, there is no data-path collective (weak scaling).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (sdf_mlp, fp32 MFMA bound) timed
with HIP events on the launch stream inside the timed region; `cpu_baseline` is the CPU oracle
(oracle/surf_oracle.py, a port of the reference algorithm) timed on this host on a bounded ray subset.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE_SDF = 2 * (99240 + 99240)          # forward + reverse-mode gradient MACs x 2
FLOP_PER_SAMPLE_BLEND_PER_VIEW = 2 * 9928
CPU_THREADS = min(32, os.cpu_count() or 1)
# Dominant kernel per SDF precision: (kernel name as the profile summaries spell it, matrix pipe, dense peak of that
# pipe in TFLOP/s from MI355X_MICROARCH.md, MFMA products issued per fp32-equivalent product).  `roofline.peak` is the
# pipe's dense peak divided by the products per fp32 product: the fp32-equivalent rate the pipe could deliver at best.
SDF_KERNELS = {
    "f32": ("sdf_mlp_kernel2<true>", "v_mfma_f32_32x32x2_f32", 157.3, 1),
    "bf16x3": ("sdf_mlp_split_kernel<PolBf3, true>", "v_mfma_f32_32x32x16_bf16", 2500.0, 6),
    "f16x2": ("sdf_mlp_split_kernel<PolH2, true>", "v_mfma_f32_32x32x16_f16", 2500.0, 3),
}


def model_conf(n_samples, sdf_precision="bf16x3"):
    from surf_amd import conf
    return conf.from_dict({
        "sdf_network": {"d_out": 129, "d_in": 3, "d_hidden": 128, "n_layers": 6, "skip_in": [3], "multires": 4,
                        "bias": 0.5, "scale": 1.0, "geometric_init": True, "weight_norm": True, "feat_channels": 28,
                        "feat_multires": 0},
        "color_network": {"d_feature": 16},
        "variance_network": {"init_val": 0.3},
        "render": {"n_samples": n_samples, "sample_ranges": [1.0, 0.4, 0.1, 0.01], "n_depth": 256, "perturb": 0.0,
                   "sdf_precision": sdf_precision},
    })


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary of this same command
    (profiles/rNN_bench_pmc.csv: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE is
    doubled per the gfx950 correction of MI355X_MICROARCH.md section HBM).  None if no profile is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc.csv")))
    if not files:
        return None
    with open(files[-1]) as f:
        rows = list(csv.DictReader(l for l in f if not l.startswith("#")))
    for r in rows:
        if r["kernel"] == kernel and r["FETCH_SIZE"] and r["WRITE_SIZE"]:
            return (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0
    return None


def cpu_baseline(model, cpu_scene, rays_o, rays_d, near, far, n_samples, budget_s, gpu_out, ray_idx):
    """Time the CPU oracle on 256-ray chunks (implicit_surface.py:367) of a strided ray subset."""
    from oracle import surf_oracle as O
    torch.set_num_threads(CPU_THREADS)   # more threads than this only add OpenMP overhead on these small ops
    sd = {"implicit_surface." + k: v.detach().cpu() for k, v in model.state_dict().items()}
    done, t0, max_err = 0, time.perf_counter(), 0.0
    for s in range(0, rays_o.shape[0], 256):
        sl = slice(s, s + 256)
        out = O.render(sd, rays_o[sl], rays_d[sl], near[sl], far[sl], cpu_scene["mvol"], cpu_scene["vols"],
                       cpu_scene["tabs"], cpu_scene["masks"], cpu_scene["feats"], cpu_scene["imgs"], cpu_scene["intrs"],
                       cpu_scene["c2ws"], n_samples, [1.0, 0.4, 0.1, 0.01], 256, 1.0)
        done += out["color_fine"].shape[0]
        ref = out["color_fine"]
        got = gpu_out["color_fine"][ray_idx[sl]].cpu()
        max_err = max(max_err, float((got - ref).abs().max()))
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return done / dt, done, dt, max_err


def volume_build_timing(args, dev):
    """One full-size volume build through surf_amd.surf.SuRF (FPN -> cost volume -> sparsify -> sparse U-Net ->
    densify -> matching field, rows a1-a7) on the synthetic 5-view scene, reported beside the render metric.
    Weights are random-init, so the U-Net's matching logit is replaced by the analytic sphere logit
    (-20 | |x| - 0.5 |) to obtain the surface-concentrated pyramid a trained network would produce."""
    from surf_amd import conf, synthetic
    from surf_amd.surf import SuRF
    H, W, nv = args.height, args.width, args.views
    cfg = {
        "range_ratios": [1.0, 0.4, 0.1, 0.01],
        "feature_network": {"d_in": 3, "d_base": 8, "d_out": [4, 4, 4, 4]},
        "volume": {"base_volume_dim": [args.base_dim] * 3},
        "reg_network": {"d_in": [8, 16, 16, 16], "d_base": [8] * 4, "d_out": [8] * 4},
        "matching_field": {"n_samples_depths": [128, 64, 32, 16], "n_importance_depths": [128, 64, 32, 16],
                           "up_sample_steps": [4, 4, 4, 4], "depth_res_levels": [4, 2, 2, 1]},
        "implicit_surface": dict(model_conf([64, 32, 16, 16])),
    }
    torch.manual_seed(0)
    model = SuRF(conf.from_dict(cfg)).eval().to(dev)
    intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
    ipts = {"imgs": synthetic.procedural_images(nv, H, W, 0, dev), "intrs": intrs.to(dev), "c2ws": c2ws.to(dev),
            "near_fars": near_fars.to(dev), "near": near_fars[0, 0].reshape(1, 1).to(dev),
            "far": near_fars[0, 1].reshape(1, 1).to(dev)}

    def sphere_logit(coords, D):
        world = coords.float() * (2.0 / (D - 1)) - 1.0
        return -20.0 * (world.norm(dim=1) - 0.5).abs()

    res = {}
    for it in range(2):                      # first pass warms allocator and code objects
        timings = {}
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        feats = model.feature_network(ipts["imgs"])
        e1.record()
        model.build_volumes(ipts, feats, logit_override=sphere_logit, timings=timings)
        e2.record()
        torch.cuda.synchronize()
        res = {"fpn_ms": e0.elapsed_time(e1), "stages_ms": e1.elapsed_time(e2), "total_ms": e0.elapsed_time(e2), "stages": []}
        for s in sorted(timings):
            ev = timings[s]["events"]
            res["stages"].append({"n_voxels": timings[s]["n_voxels"],
                                  "filter_costvol_ms": ev[0].elapsed_time(ev[1]), "sparse_unet_ms": ev[1].elapsed_time(ev[2]),
                                  "densify_ms": ev[2].elapsed_time(ev[3]), "matching_field_ms": ev[3].elapsed_time(ev[4])})
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--base-dim", type=int, default=88)
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--n-samples", type=str, default="64,32,16,16")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="time budget of the CPU baseline (0 = skip)")
    ap.add_argument("--build", type=int, default=1, help="also time one full volume build (FPN + 4 stages), N=1 only")
    ap.add_argument("--sdf-precision", default="bf16x3", choices=sorted(SDF_KERNELS),
                    help="SDF kernel: f32 MFMA, bf16x3 (exact 3-way bf16 split, fp32-equivalent; default), "
                         "f16x2 (22-bit operands, fastest)")
    ap.add_argument("--also", default="f16x2", help="comma list of further precisions timed after the headline run "
                                                    "(reported under other_precisions; '' = none)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from surf_amd import dist as D
    D.init_from_env("nccl", dev)          # RCCL; only used for the barrier and the MAX of the elapsed time
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    from surf_amd import synthetic
    from surf_amd.implicit_surface import ImplicitSurface

    n_samples = [int(x) for x in args.n_samples.split(",")]
    S = sum(n_samples)
    H, W, nv = args.height, args.width, args.views
    torch.manual_seed(0)
    model = ImplicitSurface(model_conf(n_samples, args.sdf_precision)).to(dev)

    # ---- scene (seed = rank), resident in HBM before the timed region ------------------------------------
    seed = rank
    intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
    imgs = synthetic.procedural_images(nv, H, W, seed, dev)
    feats = synthetic.feature_pyramid(nv, H, W, seed, dev)                     # fine -> coarse
    vols, tabs, mvol = synthetic.sphere_pyramid(args.base_dim, dev, seed=seed)  # coarse -> fine
    scene = model.scene(mvol, vols[::-1], tabs[::-1], None, feats, imgs, intrs.to(dev), c2ws.to(dev))
    rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], H, W, 1, dev)
    R = rays_o.shape[0]
    near = near_fars[0, 0].reshape(1, 1).repeat(R, 1).to(dev)
    far = near_fars[0, 1].reshape(1, 1).repeat(R, 1).to(dev)

    def step():
        return model.render_scene(rays_o, rays_d, near, far, scene, 1.0, per_sample=False)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    model.kernel_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    events, model.kernel_events = model.kernel_events, None
    elapsed = D.max_over_ranks(elapsed, dev)

    # ---- per-kernel durations from the HIP events recorded inside the timed region -----------------------
    per_kernel = {}
    for name, a, b in events:
        per_kernel.setdefault(name, []).append(a.elapsed_time(b))
    kernel_ms = {k: sum(v) / len(v) for k, v in per_kernel.items()}
    active = int(model.last_active_samples)

    # ---- the other SDF precisions on the same scene (N = 1 only; after the timed region, reported separately) ----
    others = {}
    if world == 1:
        for prec in [p for p in args.also.split(",") if p and p != args.sdf_precision]:
            model.sdf_precision = prec
            out_o = step()
            torch.cuda.synchronize()
            model.kernel_events = []
            t1 = time.perf_counter()
            for _ in range(args.steps):
                out_o = step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            ev_o, model.kernel_events = model.kernel_events, None
            sdf_o = [a.elapsed_time(b) for name, a, b in ev_o if name == "sdf_mlp"]
            others[prec] = {"rays_per_s": R * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                            "sdf_mlp_ms": sum(sdf_o) / len(sdf_o),
                            "max_abs_rgb_diff_vs_headline": float((out_o["color_fine"] - out["color_fine"]).abs().max())}
        model.sdf_precision = args.sdf_precision

    if rank == 0:
        sdf_kernel, sdf_pipe, pipe_peak, n_prod = SDF_KERNELS[args.sdf_precision]
        sdf_ms = kernel_ms["sdf_mlp"]
        flops = active * FLOP_PER_SAMPLE_SDF
        achieved = flops / (sdf_ms * 1e-3) / 1e12
        peak = pipe_peak / n_prod
        result = {
            "metric": "rays/sec (576x800, 5-view, 128 samp/ray render, whole job)",
            "value": world * R * args.steps / elapsed,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 (operands split exactly into 3 bf16 pieces, fp32 accumulate)",
                      "f16x2": "f32 accumulate, 22-bit operands (2 fp16 pieces)"}[args.sdf_precision],
            "data": "synthetic",
            "config": {"workload": f"render {H}x{W} ref view, {nv} views, samples {n_samples} (={S}/ray), sphere pyramid "
                                   f"{args.base_dim}^3->{args.base_dim * 8}^3, one scene per GPU",
                       "rays_per_step": R, "samples_per_ray": S, "active_samples": active},
            "per_gpu_rays_per_s": R * args.steps / elapsed,
            "kernel_ms": kernel_ms,
            "roofline": {"kernel": sdf_kernel, "bound": "mfma", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": pmc_traffic(sdf_kernel), "flop_per_sample": FLOP_PER_SAMPLE_SDF, "samples_per_launch": active,
                         "avg_launch_ms": sdf_ms, "pipe": sdf_pipe, "pipe_dense_peak": pipe_peak,
                         "mfma_products_per_fp32_product": n_prod, "frac_of_fp32_mfma_peak": achieved / 157.3},
        }
        if others:
            result["other_precisions"] = others
        if world == 1 and args.cpu_seconds > 0:
            n_sub = 8192
            idx = torch.linspace(0, R - 1, n_sub).long()
            cpu_scene = {
                "mvol": mvol.cpu(), "vols": [v[:, :7].cpu() for v in vols[::-1]],
                "tabs": [t.cpu().long() for t in tabs[::-1]], "feats": [f.cpu() for f in feats], "imgs": imgs.cpu(),
                "intrs": intrs, "c2ws": c2ws,
            }
            cpu_scene["masks"] = [(t >= 0).float() for t in cpu_scene["tabs"]]
            rps, n_done, dt, err = cpu_baseline(model, cpu_scene, rays_o.cpu()[idx], rays_d.cpu()[idx], near.cpu()[idx],
                                                far.cpu()[idx], n_samples, args.cpu_seconds, out, idx)
            result["cpu_baseline"] = {"value": rps, "unit": "rays/s", "cores": CPU_THREADS, "kind": "port",
                                      "sample": f"{n_done} rays (every {R // n_sub}th pixel ray, 256-ray chunks) of the same "
                                                f"scene in {dt:.1f} s, torch CPU fp32",
                                      "max_abs_rgb_diff_vs_gpu": err}
        if world == 1 and args.build:
            result["volume_build"] = volume_build_timing(args, dev)
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
