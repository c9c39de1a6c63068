#!/usr/bin/env python3
"""Headline benchmark of the SuRF hot path on MI355X: full-image render throughput (rays/s).

    python bench.py --gpus N --steps K --warmup W [--workload dtu|tnt|train] [--scenes M]

One "step" = one pass of the render hot path (ray set-up -> SDF MLP + gradient -> multi-view blending ->
NeuS compositing) over every pixel ray of the reference view of a synthetic multi-view scene
(`--workload dtu`, default: BASELINE.json configs[1], 576x800, 5 views, 128 samples per ray, synthetic sphere pyramid
88^3 -> 704^3; `--workload tnt`: configs[4], 1080x1920, 7 views, 192 samples per ray).
Inputs are resident in HBM before the timed region.

Multi-GPU: scenes are independent, so they shard over ranks with NO data-path collective (weak scaling);
RCCL only carries the timing barrier, the MAX of the elapsed time and a gather of small per-scene records.
`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself (one child process per
GPU, env:// rendezvous on 127.0.0.1, before this process touches the GPU) and relays rank 0's line; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks are the launcher's.
`--scenes M` (configs[2]: 15 scans) deals M scenes round-robin over the ranks (scripts/run.sh:3,
datasets/__init__.py:37-38 of the reference); without it every rank renders one scene (seed = rank).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (sdf_mlp) timed with HIP events on the launch
stream inside the timed region; `roofline_kernels` lists the other contraction kernels; `cpu_baseline` is the CPU
oracle (oracle/surf_oracle.py, a port of the reference algorithm) timed on this host on a bounded ray subset.
`--workload train` (configs[3]) instead runs the data-parallel TRAINING step: every rank one scene + 512 rays per step
through the reference's own sequence (forward, Loss, loss.backward(), Adam) with the model wrapped in
DistributedDataParallel over RCCL; it reports training rays/s, steps/s, the all-reduce time of the gradient bucket alone
and `roofline_kernels` for the heaviest backward kernels.
`--dry` replaces the kernels by a sleep and RCCL by gloo: it exists so that tests/test_dist_gloo.py can drive this
file's own N > 1 control flow on a CPU-only host; a dry line says so ("data": "dry-run") and is not a measurement.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE_SDF = 2 * (99240 + 99240)          # forward + reverse-mode gradient MACs x 2
FLOP_PER_SAMPLE_SDF_FWD = 2 * 99240                 # forward only (the 512^3 lattice of extract_geometry)
FLOP_PER_SAMPLE_BLEND_PER_VIEW = 2 * 9928
HBM_PEAK = 8.0e12                                   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CPU_THREADS = min(32, os.cpu_count() or 1)
# Dominant kernel per SDF precision: (kernel name as the profile summaries spell it, matrix pipe, dense peak of that
# pipe in TFLOP/s from MI355X_MICROARCH.md, MFMA products issued per fp32-equivalent product).  `roofline.peak` is the
# pipe's dense peak divided by the products per fp32 product: the fp32-equivalent rate the pipe could deliver at best.
SDF_KERNELS = {
    "f32": ("sdf_mlp_kernel2<true>", "v_mfma_f32_32x32x2_f32", 157.3, 1),
    "bf16x3": ("sdf_mlp_split_kernel<PolBf3, true>", "v_mfma_f32_32x32x16_bf16", 2500.0, 6),
    "f16x2": ("sdf_mlp_split_kernel<PolH2, true>", "v_mfma_f32_32x32x16_f16", 2500.0, 3),
}
BLEND_KERNELS = {
    "f32": ("blend_kernel<{ns}>", "v_mfma_f32_32x32x2_f32", 157.3, 1),
    "bf16x3": ("blend_split_kernel<BPolBf3>", "v_mfma_f32_32x32x16_bf16", 2500.0, 6),
    "f16x2": ("blend_split_kernel<BPolH2>", "v_mfma_f32_32x32x16_f16", 2500.0, 3),
}
WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "dtu": {"height": 576, "width": 800, "views": 5, "n_samples": "64,32,16,16",
            "metric": "rays/sec (576x800, 5-view, 128 samp/ray render, whole job)"},
    # BASELINE.json configs[4]: Tanks&Temples, 7 views 1080p, 192 samples per ray
    "tnt": {"height": 1080, "width": 1920, "views": 7, "n_samples": "96,48,32,16",
            "metric": "rays/sec (1080x1920, 7-view, 192 samp/ray render, whole job)"},
    # BASELINE.json configs[3]: surf.conf unsupervised training, data-parallel over the ranks (runner.py:102: DDP): one scene
    # + 512 rays per rank and step; a step = forward (FPN, 4-stage volume build, render) + Loss + loss.backward() + Adam
    "train": {"height": 576, "width": 800, "views": 5, "n_samples": "64,32,16,16",
              "metric": "training rays/sec (surf.conf unsupervised training step, 5 views 576x800, 128 samp/ray, whole job)"},
}
RAY_CHUNK = 1 << 19      # rays per render call (bounds the per-sample buffers; 576x800 fits one call)


def model_conf(n_samples, sdf_precision="bf16x3", blend_precision=None):
    from surf_amd import conf
    render = {"n_samples": n_samples, "sample_ranges": [1.0, 0.4, 0.1, 0.01], "n_depth": 256, "perturb": 0.0,
              "sdf_precision": sdf_precision}
    if blend_precision is not None:
        render["blend_precision"] = blend_precision
    return conf.from_dict({
        "sdf_network": {"d_out": 129, "d_in": 3, "d_hidden": 128, "n_layers": 6, "skip_in": [3], "multires": 4,
                        "bias": 0.5, "scale": 1.0, "geometric_init": True, "weight_norm": True, "feat_channels": 28,
                        "feat_multires": 0},
        "color_network": {"d_feature": 16},
        "variance_network": {"init_val": 0.3},
        "render": render,
    })


def surf_conf(base_dim=88, n_samples=(64, 32, 16, 16)):
    """The model section of confs/surf.conf:66-124 as a dict (full SuRF: FPN, volume, sparse U-Nets, matching field,
    implicit surface)."""
    return {
        "range_ratios": [1.0, 0.4, 0.1, 0.01],
        "feature_network": {"d_in": 3, "d_base": 8, "d_out": [4, 4, 4, 4]},
        "volume": {"base_volume_dim": [base_dim] * 3},
        "reg_network": {"d_in": [8, 16, 16, 16], "d_base": [8] * 4, "d_out": [8] * 4},
        "matching_field": {"n_samples_depths": [128, 64, 32, 16], "n_importance_depths": [128, 64, 32, 16],
                           "up_sample_steps": [4, 4, 4, 4], "depth_res_levels": [4, 2, 2, 1]},
        "implicit_surface": dict(model_conf(list(n_samples))),
    }


def algorithmic_bytes_per_ray(S, nv, n_depth=256, index_bytes=8):
    """coarse matching taps + per sample [mask taps + sparse trilinear rows (index + 28 B of features per
    corner) + per source view 4 levels x 4 taps x 16 B + 4 x 12 B of RGB] + ray in / out.
    S = 128, nv = 5, 8-byte indices (the reference's int64 tables) -> 313,408 B."""
    return n_depth * 8 * 4 + S * (4 * 4 + 4 * 8 * (index_bytes + 28) + (nv - 1) * (4 * 4 * 16 + 4 * 12)) + 64


def csrc_digest():
    """sha256 over the kernel sources: ties a committed PMC summary to the code it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "surf_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary of this same command
    (profiles/rNN_bench_pmc.csv: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE is doubled per the
    gfx950 correction of MI355X_MICROARCH.md section HBM).  Returns (bytes or None, source note): None when no profile
    is committed or when the kernel sources have changed since it was taken (`# csrc_sha256:` header line)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc.csv")), key=os.path.basename)
    if not files:
        return None, "no profiles/r*_bench_pmc.csv committed"
    path = files[-1]
    sha = None
    with open(path) as f:
        lines = f.readlines()
    for l in lines:
        if l.startswith("# csrc_sha256:"):
            sha = l.split(":", 1)[1].strip()
    name = os.path.relpath(path, ROOT)
    if sha != csrc_digest():
        return None, f"{name} was taken on other kernel sources (csrc_sha256 {sha} != {csrc_digest()}): stale, not reported"
    rows = list(csv.DictReader(l for l in lines if not l.startswith("#")))
    for r in rows:
        if r["kernel"] == kernel and r["FETCH_SIZE"] and r["WRITE_SIZE"]:
            return (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0, f"{name} (csrc_sha256 {sha})"
    return None, f"{name} has no row for {kernel}"


def pmc_mfma_busy(kernel):
    """Matrix-pipe utilisation of `kernel` from the same committed PMC summary: SQ_VALU_MFMA_BUSY_CYCLES /
    (GRBM_GUI_ACTIVE per XCD x 256 CUs x 4 SIMDs) (GRBM_GUI_ACTIVE is summed over the 8 XCDs).  None when stale / absent."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc.csv")), key=os.path.basename)
    if not files:
        return None
    with open(files[-1]) as f:
        lines = f.readlines()
    sha = [l.split(":", 1)[1].strip() for l in lines if l.startswith("# csrc_sha256:")]
    if not sha or sha[-1] != csrc_digest():
        return None
    for r in csv.DictReader(l for l in lines if not l.startswith("#")):
        if r["kernel"] == kernel and r.get("SQ_VALU_MFMA_BUSY_CYCLES") and r.get("GRBM_GUI_ACTIVE"):
            gui = float(r["GRBM_GUI_ACTIVE"]) / 8.0
            return float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) / (gui * 256 * 4) if gui > 0 else None
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
    densify -> matching field, rows a1-a7) on the synthetic scene, reported beside the render metric.
    Weights are random-init, so the U-Net's matching logit is replaced by the analytic sphere logit
    (-20 | |x| - 0.5 |) to obtain the surface-concentrated pyramid a trained network would produce."""
    from surf_amd import conf, synthetic
    from surf_amd.surf import SuRF
    H, W, nv = args.height, args.width, args.views
    torch.manual_seed(0)
    model = SuRF(conf.from_dict(surf_conf(args.base_dim))).eval().to(dev)
    intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
    ipts = {"imgs": synthetic.procedural_images(nv, H, W, 0, dev), "intrs": intrs.to(dev), "c2ws": c2ws.to(dev),
            "near_fars": near_fars.to(dev), "near": near_fars[0, 0].reshape(1, 1).to(dev),
            "far": near_fars[0, 1].reshape(1, 1).to(dev)}

    res = {}
    for it in range(2):                      # first pass warms allocator and code objects
        timings = {}
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        feats = model.feature_network(ipts["imgs"])
        e1.record()
        model.build_volumes(ipts, feats, logit_override=synthetic.sphere_logit, timings=timings)
        e2.record()
        torch.cuda.synchronize()
        res = {"fpn_ms": e0.elapsed_time(e1), "stages_ms": e1.elapsed_time(e2), "total_ms": e0.elapsed_time(e2), "stages": []}
        for s in sorted(timings):
            ev = timings[s]["events"]
            res["stages"].append({"n_voxels": timings[s]["n_voxels"],
                                  "filter_costvol_ms": ev[0].elapsed_time(ev[1]), "sparse_unet_ms": ev[1].elapsed_time(ev[2]),
                                  "densify_ms": ev[2].elapsed_time(ev[3]), "matching_field_ms": ev[3].elapsed_time(ev[4])})
    return res


def scene_timing(args, dev, mesh_resolution=512):
    """BASELINE configs[1] as ONE number: a whole scene through surf_amd.surf.SuRF.forward("val") - FPN, 4-stage volume build,
    the full-resolution render of the reference view, the mesh_resolution^3 SDF lattice and marching cubes (surf.py:133-163,
    implicit_surface.py:337-402) - wall clock around the call, inputs resident in HBM, outputs delivered as the runner reads them
    (images / depth maps on the host, mesh arrays on the host).  Same synthetic scene and sphere logit as volume_build_timing."""
    from surf_amd import conf, synthetic
    from surf_amd.surf import SuRF
    H, W, nv = args.height, args.width, args.views
    torch.manual_seed(0)
    mc = surf_conf(args.base_dim)
    mc["implicit_surface"]["render"]["n_samples"] = [int(x) for x in args.n_samples.split(",")]
    model = SuRF(conf.from_dict(mc)).eval().to(dev)
    model.logit_override = synthetic.sphere_logit
    intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
    rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], H, W, 1, dev)
    ipts = {"imgs": synthetic.procedural_images(nv, H, W, 0, dev), "intrs": intrs.to(dev), "c2ws": c2ws.to(dev),
            "near_fars": near_fars.to(dev), "near": near_fars[0, 0].reshape(1, 1).to(dev), "far": near_fars[0, 1].reshape(1, 1).to(dev),
            "rays_o": rays_o, "rays_d": rays_d, "bound_min": torch.tensor([-1.0] * 3), "bound_max": torch.tensor([1.0] * 3),
            "hw": (H, W), "mesh_resolution": mesh_resolution}
    ms = []
    for it in range(5):                      # the first passes warm code objects and the caching allocator (it takes three calls to
        torch.cuda.synchronize()             # settle: 380 / 360 / 310 / 307 / 307 ms); reported: the median of the last three
        t0 = time.perf_counter()
        with torch.no_grad():
            out = model("val", ipts, 1.0)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    return {"scene_ms": sorted(ms[2:])[1], "first_call_ms": ms[0], "calls_ms": ms, "rays": int(rays_o.shape[0]), "mesh_resolution": mesh_resolution,
            "vertices": int(len(out["vertices"])), "triangles": int(len(out["triangles"])),
            "what": "SuRF.forward('val'): FPN + 4-stage volume build + full-resolution render of the reference view + "
                    f"{mesh_resolution}^3 SDF lattice + marching cubes, host wall clock incl. the device-to-host copies of the outputs"}


def training_step_setup(dev, H=576, W=800, nv=5, base_dim=88, rays=512, device_jitter=True, scene_seed=0, precision="fp32"):
    """A volume-building SuRF in train mode on the synthetic scene + the inputs / loss targets of one training step
    (runner.py:150-166).  Weights are random-init, so the analytic sphere logit replaces the U-Net's matching logit in the
    forward (as in volume_build_timing); the backward still runs through every kernel."""
    from surf_amd import conf, synthetic
    from surf_amd.losses import Loss
    from surf_amd.surf import SuRF
    torch.manual_seed(0)
    model = SuRF(conf.from_dict(dict(surf_conf(base_dim), train_precision=precision))).to(dev).train()
    model.logit_override = synthetic.sphere_logit
    model.matching_field.device_jitter = device_jitter      # False: the reference's CPU-generator draw (host-bound)
    intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
    imgs = synthetic.procedural_images(nv, H, W, scene_seed, dev)
    rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], H, W, 1, dev)
    sel = torch.randperm(rays_o.shape[0], generator=torch.Generator().manual_seed(1000 + scene_seed))[:rays].to(dev)
    ipts = {"imgs": imgs, "intrs": intrs.to(dev), "c2ws": c2ws.to(dev), "near_fars": near_fars.to(dev),
            "near": near_fars[0, 0].reshape(1, 1).to(dev), "far": near_fars[0, 1].reshape(1, 1).to(dev),
            "rays_o": rays_o[sel].contiguous(), "rays_d": rays_d[sel].contiguous(), "src_idx": 1}
    pp = torch.randn(4096, 3, device=dev)
    ipts["pseudo_pts"] = 0.5 * pp / pp.norm(dim=1, keepdim=True)        # the dataset's pseudo surface points (pseudo_sdf term)
    ones = torch.ones(H, W, device=dev)
    targets = {"color": torch.rand(rays, 3, device=dev), "imgs": imgs, "intrs": intrs, "c2ws": c2ws, "src_idx": 1,
               "mask_ref": ones, "mask_src": ones, "pseudo_depth_ref": ones * 2.0, "pseudo_depth_src": ones * 2.0,
               "depth_ref": ones * 2.0, "depth_src": ones * 2.0}
    loss_fn = Loss(conf.from_dict({"color_weight": 1.0, "sparse_scale_factor": 100, "sparse_weight": 0.02, "igr_weight": 0.1,
                                   "mfc_weight": 0.5, "smooth_weight": 0.0001, "depth_weight": 0.0, "ptloss_weight": 1.0,
                                   "pseudo_auxi_depth_weight": 1.0, "pseudo_sdf_weight": 1.0, "pseudo_depth_weight": 0.0,
                                   "stage_weights": [0.25, 0.5, 0.75, 1.0]}))
    opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3}))
    return model, ipts, targets, loss_fn, opt


def training_step_timing(args, dev, steps=5):
    """Wall time of one full training step (runner.py:152-165: forward -> Loss -> loss.backward() = the HIP backward of the
    render, the 4-stage volume build and the FPN -> Adam) on the bench scene, reported beside the render metric,
    for both training-precision policies."""
    from surf_amd import dist as D
    from surf_amd import ops, training
    res = {}
    for precision in ("fp32", "bf16"):
        model, ipts, targets, loss_fn, opt = training_step_setup(dev, args.height, args.width, args.views, args.base_dim,
                                                                 precision=precision)
        def timed(stepper, sync, opt=opt):
            for _ in range(2):
                out = training.train_step(stepper, ipts, targets, loss_fn, opt, 1.0, 3, sync=sync)
            torch.cuda.synchronize()
            per_step = []
            for _ in range(steps):        # a step ends with the host reading its loss values: timing steps one by one adds no sync
                t0 = time.perf_counter()
                out = training.train_step(stepper, ipts, targets, loss_fn, opt, 1.0, 3, sync=sync)
                torch.cuda.synchronize()
                per_step.append((time.perf_counter() - t0) * 1e3)
            return sorted(per_step)[len(per_step) // 2], per_step, out     # median: one slow step (allocator, clocks) is not the policy

        # the single-GPU step, as every earlier round measured it: no gradient averaging (sync=False beside the forced group)
        ms, per_step, out = timed(model, False)
        res[precision] = {"ms_per_step": ms, "loss": out["loss"], "steps_ms": [round(v, 2) for v in per_step]}
        if D._active():                   # the same step as runner.py:102 runs it on N GPUs: wrapped in DDP over the (world-1) RCCL group
            ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index])      # runner.py:102, argument for argument
            ms_d, per_d, _ = timed(ddp, True)
            res[precision]["ddp_ms_per_step"] = ms_d
            res[precision]["ddp_steps_ms"] = [round(v, 2) for v in per_d]
            del ddp
        if precision == "fp32":
            # context, not the headline: the reference constructs torch.optim.Adam(optim_param) (runner.py:94: the foreach form,
            # ~25 launches with the host in between at the very end of a step); one keyword there - fused=True - is one launch a group
            opt_f = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3}), fused=True)
            ms_f, per_f, _ = timed(model, False, opt_f)
            res[precision]["fused_adam_ms_per_step"] = ms_f
            res[precision]["fused_adam_steps_ms"] = [round(v, 2) for v in per_f]
            # the same step with every launch in order on one stream (SURF_SIDE_STREAM=0): what the side streams buy
            was, ops.side.enabled = ops.side.enabled, False
            ms_o, per_o, _ = timed(model, False)
            ops.side.enabled = was
            res[precision]["in_order_ms_per_step"] = ms_o
            del opt_f
        voxels = model.last_voxels_per_stage
        rays = int(ipts["rays_o"].shape[0])
        del model, opt
    ops.set_train_precision("fp32")
    return {"ms_per_step": res["fp32"]["ms_per_step"], "rays": rays, "samples_per_ray": 128, "voxels_per_stage": voxels,
            "loss": res["fp32"]["loss"], "steps_ms": res["fp32"]["steps_ms"], "train_precision_bf16": res["bf16"],
            "ddp_ms_per_step": res["fp32"].get("ddp_ms_per_step"),
            "ddp_steps_ms": res["fp32"].get("ddp_steps_ms"),
            "in_order_ms_per_step": res["fp32"].get("in_order_ms_per_step"),
            "fused_adam_ms_per_step": res["fp32"].get("fused_adam_ms_per_step"),
            "streams": "backward sweep on 4 HIP streams (surf_amd.ops.SideStream: U-Net kernel gradients, render branches, matching chain); "
                       "in_order_ms_per_step = SURF_SIDE_STREAM=0; fused_adam_ms_per_step = the same step with "
                       "torch.optim.Adam(..., fused=True) instead of runner.py:94's default (foreach) form",
            "data_parallel": ("ddp_ms_per_step: the same step with the model wrapped in DistributedDataParallel over " + D.backend_note()
                              + " (bucket hooks, buffer broadcast, the 5.6 MB all-reduce with one peer)") if D._active() else None,
            "what": "median of 5 steps of: forward (FPN, volume build, render) + loss + loss.backward() (HIP backward of all of it) + Adam; every term of "
                    "losses/loss.py; matching-field jitter on the device generator; train_precision_bf16: the same step with the "
                    "weight-gradient reductions on bf16 operands (model conf train_precision = bf16)"}


def mesh_grid_timing(model, scene, dev, resolution):
    """The lattice of extract_geometry (implicit_surface.py:337-351, row a16): resolution^3 forward-only SDF
    evaluations, timed end to end and per kernel launch (HIP events)."""
    bmin, bmax = torch.tensor([-1.0] * 3), torch.tensor([1.0] * 3)
    model.sdf_grid(scene, bmin, bmax, 64)                    # warms the forward-only code object
    torch.cuda.synchronize()
    model.kernel_events = []
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    u = model.sdf_grid(scene, bmin, bmax, resolution)
    b.record()
    torch.cuda.synchronize()
    ev, model.kernel_events = model.kernel_events, None
    k_ms = sum(x.elapsed_time(y) for name, x, y in ev if name == "sdf_grid")
    mesh_grid_timing.launches = sum(1 for name, x, y in ev if name == "sdf_grid")      # 1 in lattice mode (round 5)
    inside = int((u > 0).sum())
    # marching cubes on that lattice (row f1; mcubes.marching_cubes at implicit_surface.py:353), device side only
    from surf_amd import ops
    ops.marching_cubes(u[:64, :64, :64].contiguous(), 0.0)
    torch.cuda.synchronize()
    c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c.record()
    v, t = ops.marching_cubes(u, 0.0)
    d.record()
    torch.cuda.synchronize()
    mc = {"marching_cubes_ms": c.elapsed_time(d), "vertices": int(v.shape[0]), "triangles": int(t.shape[0])}
    del u, v, t
    return a.elapsed_time(b), k_ms, inside, mc


# ----------------------------------------------------------------------------------------------------------------------
# process group: RCCL for N > 1, and (default) a FORCED world-size-1 RCCL group at N = 1
# ----------------------------------------------------------------------------------------------------------------------

GROUP_NOTE = {"error": None}


def init_group(args, dev, world):
    """Bring up the process group of this rank.  N > 1: RCCL (or gloo with --backend gloo).  N = 1 with --force-group 1 (the
    default): a world-size-1 group over the same backend, so that the barrier, the MAX all-reduce of the elapsed time on a
    DEVICE tensor, the record gather, gather_rows' padded device all_gather, the gradient bucket's all-reduce and
    DistributedDataParallel's hooks all execute over RCCL on a one-GPU box - the calls an 8-GPU run makes, with one peer.
    A failure of the forced group is reported in the line (`collective_backend`) and the run continues without a group; a
    failure at N > 1 is fatal."""
    from surf_amd import dist as D
    force = bool(args.force_group) and world == 1 and not torch.distributed.is_initialized()
    try:
        D.init_from_env(args.backend, dev, force=force, timeout_s=args.group_timeout)
    except Exception as e:       # noqa: BLE001
        if world > 1:
            raise
        GROUP_NOTE["error"] = f"none (forced world-1 {args.backend} group failed: {type(e).__name__}: {str(e)[:200]})"


def collective_note(args):
    from surf_amd import dist as D
    if GROUP_NOTE["error"]:
        return GROUP_NOTE["error"]
    return D.backend_note(args.one_gpu)


def collective_probe(dev, n_floats=1_410_000, iters=10):
    """The collectives of the N-rank runs on this rank's group, timed: the all-reduce of ONE flat gradient bucket of SuRF's size
    (1.41 M floats = 5.6 MB: DDP's single bucket), the padded device all_gather of gather_rows on an image-sized
    (R, 3) tensor with its identity check, the barrier.  None without a group."""
    from surf_amd import dist as D
    if not D._active():
        return None
    flat = torch.arange(n_floats, dtype=torch.float32, device=dev)
    ref = flat.clone()
    for _ in range(3):
        torch.distributed.all_reduce(flat)
    torch.cuda.synchronize()
    world = torch.distributed.get_world_size()
    ok = bool(torch.equal(flat, ref)) if world == 1 else None
    t0 = time.perf_counter()
    for _ in range(iters):
        torch.distributed.all_reduce(flat)
    torch.cuda.synchronize()
    ar_ms = (time.perf_counter() - t0) / iters * 1e3
    rows = torch.rand(460800 // world, 3, device=dev)
    D.gather_rows(rows)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = D.gather_rows(rows)
    torch.cuda.synchronize()
    ag_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    D.barrier()
    torch.cuda.synchronize()
    b_ms = (time.perf_counter() - t0) * 1e3
    return {"world": world, "backend": torch.distributed.get_backend(),
            "gradient_bucket_allreduce_ms": D.max_over_ranks(ar_ms, dev), "bucket_bytes": n_floats * 4,
            "allreduce_identity_at_world_1": ok, "gather_rows_ms": ag_ms,
            "gather_rows_identity_at_world_1": (bool(torch.equal(got, rows)) if world == 1 else None), "barrier_ms": b_ms}


# ----------------------------------------------------------------------------------------------------------------------
# --workload train: BASELINE configs[3], the data-parallel training step (runner.py:102,152-165)
# ----------------------------------------------------------------------------------------------------------------------

VALU_FP32_PEAK = 157.3      # TFLOP/s, MI355X_MICROARCH.md (vector fp32 = the fp32 matrix rate on this part)


def train_kernel_rooflines(per_kernel):
    """`roofline_kernels` entries of the training step's heaviest backward kernels from the HIP events recorded inside the
    timed region (ops.kernel_events).  Algorithmic work per unit (DESIGN K12):
      spconv_wgrad<Ci,Co>  per EXISTING (output site, offset) pair one gathered C_i row (4 C_i B); per output site the 27 table
                           entries (108 B) and the dy row (4 C_o B).  The pairs are counted on the device (ops.spconv_backward)
      spconv_dgrad<Ci,Co>  the same sparse convolution on the swapped lattices + the output row written (4 C_o B per site)
      costvol_bwd          per (voxel, view, level) 64 B of taps gathered + 64 B of float atomics
      matching_depth_bwd   per (ray, sample) 32 B of corners read + 32 B of float atomics
      sdf_bwd / sdf_smooth_bwd   per sample 2x / 4x (forward + reverse) x 0.13 M MAC on the fp32 VALU
      blend_bwd            per (sample, view) forward recomputed + reverse ~ 20 K MAC each on the fp32 VALU."""
    out = []
    for name, rows in sorted(per_kernel.items(), key=lambda kv: -sum(r[0] for r in kv[1])):
        ms = sum(r[0] for r in rows)                       # per step (all launches of the kernel)
        e = {"kernel": name, "ms_per_step": ms, "launches_per_step": len(rows)}
        if name.startswith("spconv_"):
            ci, co = (int(v) for v in name[name.index("<") + 1:-1].split(","))
            pairs, sites = sum(r[1]["pairs"] for r in rows), sum(r[1]["sites"] for r in rows)
            b = pairs * 4.0 * ci + sites * (108.0 + 4.0 * co)
            if sites > 0 and not pairs > 0:       # pair counting was off (--warmup 0): no algorithmic byte count, no fraction
                e.update(bound="hbm", achieved=None, pairs_per_step=None, sites_per_step=sites,
                         note="pairs not counted (needs >= 1 warm-up step): achieved GB/s not reported")
                out.append(e)
                continue
            e.update(bound="hbm", achieved=b / (ms * 1e-3) / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", pairs_per_step=pairs,
                     sites_per_step=sites, algorithmic_bytes_per_step=b)
        else:
            units = sum(r[1] for r in rows)
            e["units_per_step"] = units
            if name == "costvol_bwd":
                e.update(bound="hbm", achieved=units * 128.0 / (ms * 1e-3) / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", bytes_per_unit=128)
            elif name == "matching_depth_bwd":
                e.update(bound="hbm", achieved=units * 64.0 / (ms * 1e-3) / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", bytes_per_unit=64)
            elif name in ("sdf_bwd", "sdf_smooth_bwd"):
                f = (2 if name == "sdf_bwd" else 4) * 2 * 2 * 0.13e6
                e.update(bound="valu", achieved=units * f / (ms * 1e-3) / 1e12, peak=VALU_FP32_PEAK, unit="TFLOP/s", flop_per_unit=f)
            elif name == "blend_bwd":
                f = 2 * 2 * 20e3
                e.update(bound="valu", achieved=units * f / (ms * 1e-3) / 1e12, peak=VALU_FP32_PEAK, unit="TFLOP/s", flop_per_unit=f)
        if e.get("achieved") is not None:
            e["frac"] = e["achieved"] / e["peak"]
        out.append(e)
    return out


def cpu_train_baseline(nv=5, H=144, W=200, base_dim=24, rays=256, n_samples=(64, 32, 16, 16)):
    """`cpu_baseline` of the `--workload train` line: ONE training step of the CPU oracle (oracle/surf_oracle.py: fpn_forward ->
    build_volumes with BatchNorm batch statistics and the matching-field jitter -> render with the patch warp -> colour /
    eikonal / per-stage photometric / patch-NCC terms -> torch autograd backward through all of it) on this host.  A BOUNDED
    SAMPLE of the workload: the bench's scene generator at nv views of H x W, a base_dim^3 -> (8 base_dim)^3 pyramid and `rays`
    rays instead of 5 x 576x800, 88^3 -> 704^3, 512 rays - at the bench shape the oracle's sparse U-Net alone (27-offset loops
    over 10 M voxels, recorded for autograd) is many minutes and tens of GB per step.  Kind "port"; context, not a target."""
    from oracle import surf_oracle as O
    from surf_amd import conf, synthetic
    from surf_amd.surf import SuRF
    torch.manual_seed(0)
    model = SuRF(conf.from_dict(surf_conf(base_dim, n_samples)))                    # CPU: only its randomly initialised state_dict
    sd = {k: v.detach().clone().float() for k, v in model.state_dict().items()}
    names = [k for k, _ in model.named_parameters()]
    for k in names:
        sd[k].requires_grad_(True)
    intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
    imgs = synthetic.procedural_images(nv, H, W, 0, "cpu")
    rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], H, W, 1, "cpu")
    sel = torch.randperm(rays_o.shape[0], generator=torch.Generator().manual_seed(1000))[:rays]
    ipts = {"imgs": imgs, "intrs": intrs, "c2ws": c2ws, "near_fars": near_fars, "near": near_fars[0, 0].reshape(1, 1),
            "far": near_fars[0, 1].reshape(1, 1)}
    cfg = {"range_ratios": [1.0, 0.4, 0.1, 0.01], "base_volume_dim": base_dim, "n_samples_depths": [128, 64, 32, 16],
           "depth_res_levels": [4, 2, 2, 1]}

    def reg_fn(f, c, d, stage):          # the bench's analytic matching logit on top of the real U-Net (as training_step_setup)
        out, mid = O.sparse_unet(sd, f, c.long(), d, stage, training=True)
        out = torch.cat([synthetic.sphere_logit(c, d).reshape(-1, 1), out[:, 1:]], dim=1)
        return out, mid

    t0 = time.perf_counter()
    feats = O.fpn_forward(sd, imgs)
    bv = O.build_volumes(sd, ipts, feats, cfg, reg_fn=reg_fn, perturb=True, src_idx=1, training=True)
    vols, tabs, masks = bv["volumes"][::-1], bv["tables"][::-1], bv["masks"][::-1]
    near = ipts["near"].repeat(rays, 1)
    far = ipts["far"].repeat(rays, 1)
    out = O.render(sd, rays_o[sel], rays_d[sel], near, far, bv["matching_volume"], vols, tabs, masks, feats[::-1], imgs, intrs, c2ws,
                   list(n_samples), [1.0, 0.4, 0.1, 0.01], 256, 1.0, patch_warp=True)
    target = torch.rand(rays, 3, generator=torch.Generator().manual_seed(5))
    loss = (out["color_fine"] - target).abs().mean()
    if torch.is_tensor(out.get("gradient_error")):
        loss = loss + 0.1 * out["gradient_error"].mean()
    if "ref_gray_val" in out:
        loss = loss + 0.5 * O.lncc(out["ref_gray_val"], out["sampled_gray_val"]).mean()
    ones = torch.ones(H, W)
    for i, w in enumerate((0.25, 0.5, 0.75, 1.0)):
        d_ref, d_src = bv["depths"][i][0], bv["depths"][i][1]
        loss = loss + w * (O.photometric_loss(d_ref, imgs, ones, intrs, c2ws, 0, 2)[0]
                           + O.photometric_loss(d_src, imgs, ones, intrs, c2ws, 1, 1)[0])
    t1 = time.perf_counter()
    loss.backward()
    dt = time.perf_counter() - t0
    n_grad = sum(1 for k in names if sd[k].grad is not None)
    return {"value": rays / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "threads": torch.get_num_threads(),
            "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"ONE oracle training step (forward {t1 - t0:.1f} s + autograd backward {dt - (t1 - t0):.1f} s) at a reduced scene: "
                      f"{nv} views {H}x{W}, {base_dim}^3 -> {base_dim * 8}^3 pyramid ({[int(c.shape[0]) for c in bv['coords']]} voxels), "
                      f"{rays} rays x {sum(n_samples)} samples; colour + eikonal + patch-NCC + per-stage photometric terms; "
                      f"{n_grad} of {len(names)} parameter tensors received a gradient",
            "loss": float(loss.detach())}


class _DryTrainModel(torch.nn.Module):
    """--dry: a CPU stand-in with SuRF's parameter count (1.41 M floats) and forward signature."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1410, 1000))

    def forward(self, mode, inputs, cos_anneal_ratio=1.0, step=None):
        time.sleep(0.002)
        return {"color_fine": (self.w.sum() * 0.0 + inputs["x"]).reshape(1, 1)}


def run_rank_train(args):
    """N ranks x (one synthetic scene + `--rays` rays) per step, gradients averaged by DistributedDataParallel over RCCL -
    the reference's own arrangement (scripts/run.sh:3, runner.py:102) on surf_amd's differentiable forward."""
    from torch.nn.parallel import DistributedDataParallel
    from surf_amd import dist as D
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry
    if dry:
        dev = torch.device("cpu")
        D.init_from_env("gloo")
    else:
        if args.one_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        init_group(args, dev, world)
    if args.fail_rank == rank:
        sys.exit(3)
    grouped = D._active()          # N > 1, or the forced world-1 group: wrap in DDP, time the bucket's all-reduce

    def sync():
        if not dry:
            torch.cuda.synchronize()

    H, W, nv, rays = args.height, args.width, args.views, args.rays
    if dry:
        model = _DryTrainModel()
        inputs = {"x": torch.tensor(float(rank))}
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        loss_of = lambda outputs: outputs["color_fine"].sum()          # noqa: E731
        ddp = DistributedDataParallel(model) if grouped else model
    else:
        from surf_amd import ops
        model, ipts, targets, loss_fn, opt = training_step_setup(dev, H, W, nv, args.base_dim, rays=rays, scene_seed=rank,
                                                                 precision=args.train_precision)
        inputs = {**targets, **ipts}                                    # the runner hands ONE dictionary to model and loss
        loss_of = lambda outputs: loss_fn(outputs, inputs, 3.0)["loss"]  # noqa: E731
        ddp = (DistributedDataParallel(model, device_ids=[local_rank])                                 # runner.py:102, argument for argument
               if grouped else model)

    def step():                                                        # runner.py:155-164
        outputs = ddp("train", inputs, cos_anneal_ratio=1.0, step=3.0)
        loss = loss_of(outputs)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    if not dry:
        ops.set_count_pairs(True)           # (site, offset) pair counts of the sparse-conv backward: warm-up only, cached
    for _ in range(args.warmup):
        loss = step()
    if not dry:
        ops.set_count_pairs(False)
    sync()
    D.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    D.barrier()
    sync()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    events = []
    if not dry and args.kernel_pass:
        # per-kernel rooflines: the same K steps again, OUTSIDE the timed region, with a HIP event pair around each instrumented
        # launch - event pairs want in-order launches, so this pass keeps the weight gradients off the side stream
        # (ops.SideStream.active()); the timed region above runs the step as a user runs it
        ops.kernel_events = []
        for _ in range(args.steps):
            step()
        sync()
        events, ops.kernel_events = ops.kernel_events, None
    # the gradient all-reduce alone: one flat bucket of the trainable parameters' size (DDP's single 5.6 MB bucket)
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    allreduce_ms = None
    if grouped:
        flat = torch.zeros(n_params, dtype=torch.float32, device=dev)
        for _ in range(3):
            torch.distributed.all_reduce(flat)
        sync()
        t1 = time.perf_counter()
        for _ in range(10):
            torch.distributed.all_reduce(flat)
        sync()
        allreduce_ms = D.max_over_ranks((time.perf_counter() - t1) / 10 * 1e3, dev)
    if rank == 0:
        per_kernel = {}
        for name, a, b, units in events:
            if isinstance(units, dict):
                units = {k: float(v) / args.steps for k, v in units.items()}
            else:
                units = units / args.steps
            per_kernel.setdefault(name, []).append((a.elapsed_time(b) / args.steps, units))
        rk = train_kernel_rooflines(per_kernel) if per_kernel else []
        top = next((e for e in rk if "frac" in e), None)
        result = {
            "metric": WORKLOADS["train"]["metric"], "value": world * rays * args.steps / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "none (dry run)" if dry else ("f32" if args.train_precision == "fp32" else
                                                   "f32 (weight-gradient reductions: bf16 operands, fp32 accumulate)"),
            "data": "dry-run" if dry else "synthetic",
            "config": {"workload": ("DRY RUN of the train control flow: no kernels executed" if dry else
                                    f"training step: {nv} views {H}x{W}, {rays} rays x 128 samples and one synthetic scene per rank, "
                                    f"{args.base_dim}^3 -> {args.base_dim * 8}^3 pyramid, every term of losses/loss.py, Adam"),
                       "rays_per_rank_step": rays, "parallelism": f"ddp{world}" if grouped else "single",
                       "train_precision": args.train_precision,
                       "trainable_parameters": n_params},
            "steps_per_s": args.steps / elapsed, "gradient_allreduce_ms": allreduce_ms,
            "loss": float(loss.detach()),
            "roofline": None if top is None else {k: top[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac")} | {"traffic": None},
            "roofline_kernels": rk,
            "cpu_baseline": None,
        }
        if not dry and world == 1 and args.cpu_seconds > 0:
            torch.set_num_threads(CPU_THREADS)
            result["cpu_baseline"] = cpu_train_baseline()
        if not dry:
            result["voxels_per_stage"] = getattr(model, "last_voxels_per_stage", None)
        result["collective_backend"] = collective_note(args)
        return result
    return None


# ----------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torch.distributed launcher
# ----------------------------------------------------------------------------------------------------------------------


def launch_ranks(n, argv, timeout_s=1700.0):
    """Start n ranks of this file as FRESH child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on
    127.0.0.1) and relay rank 0's stdout.  Runs before anything in this process has touched the GPU: the parent never
    initialises HIP and nothing is exec'd from a process that has.  Every rank's stdout / stderr goes to its own temporary
    file (no pipe can fill up); when a rank exits non-zero, or `timeout_s` passes, the others - who would otherwise sit in
    an RCCL collective until the caller's own limit - are killed (exactly the processes started here), and the failing
    ranks' stderr tails are shown."""
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, logs = [], []
    tmp = tempfile.mkdtemp(prefix="surf_bench_")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        fo, fe = open(os.path.join(tmp, f"rank{r}.out"), "w+"), open(os.path.join(tmp, f"rank{r}.err"), "w+")
        logs.append((fo, fe))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=fo, stderr=fe, text=True))
    t0, why = time.monotonic(), None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):
            why = "a rank failed"
        elif time.monotonic() - t0 > timeout_s:
            why = f"timeout after {timeout_s:.0f} s"
        if why:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            codes = [p.wait() for p in procs]
            break
        time.sleep(0.2)

    def tail(f, nbytes=3000):
        f.flush()
        f.seek(0)
        return f.read()[-nbytes:]
    sys.stdout.write(tail(logs[0][0], 1 << 24))
    sys.stdout.flush()
    bad = why is not None or any(codes)
    if bad:
        print(f"bench.py: rank exit codes {codes}" + (f" ({why}; the remaining ranks were killed)" if why else ""), file=sys.stderr)
        for r, (fo, fe) in enumerate(logs):
            if codes[r] != 0:
                print(f"---- rank {r} stderr (tail) ----\n{tail(fe)}", file=sys.stderr)
    for fo, fe in logs:
        fo.close()
        fe.close()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return 1 if bad else 0


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dtu", choices=sorted(WORKLOADS))
    ap.add_argument("--scenes", type=int, default=0, help="deal this many scenes round-robin over the ranks "
                                                          "(BASELINE configs[2]: 15); 0 = one scene per rank")
    ap.add_argument("--base-dim", type=int, default=88)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--views", type=int, default=None)
    ap.add_argument("--n-samples", type=str, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="time budget of the CPU baseline (0 = skip)")
    ap.add_argument("--build", type=int, default=1, help="also time one full volume build (FPN + 4 stages), N=1 only")
    ap.add_argument("--train-step", type=int, default=1, help="also time one full training step (N=1, dtu workload)")
    ap.add_argument("--mesh-grid", type=int, default=512, help="also time the resolution^3 SDF lattice of "
                                                               "extract_geometry (0 = skip), N=1 only")
    ap.add_argument("--sdf-precision", default="bf16x3", choices=sorted(SDF_KERNELS),
                    help="SDF kernel: f32 MFMA, bf16x3 (exact 3-way bf16 split, fp32-equivalent; default), "
                         "f16x2 (22-bit operands, fastest)")
    ap.add_argument("--blend-precision", default=None, choices=sorted(BLEND_KERNELS),
                    help="blending kernel (default: the library's default, see surf_amd.ops.BLEND_DEFAULT)")
    ap.add_argument("--also", default="f16x2", help="comma list of further precisions timed after the headline run "
                                                    "(reported under other_precisions; '' = none)")
    ap.add_argument("--dry", action="store_true", help="control-flow test: no GPU, no kernels, gloo instead of RCCL")
    ap.add_argument("--rank-timeout", type=float, default=1700.0, help="self-spawned ranks are killed after this many seconds")
    ap.add_argument("--kernel-pass", type=int, default=1, help="--workload train: after the timed steps, the same steps again "
                    "with a HIP event pair around every instrumented launch (in-order launches) for the per-kernel rooflines; "
                    "0 under rocprofv3, whose trace then holds exactly warmup + steps training steps")
    ap.add_argument("--rays", type=int, default=512, help="--workload train: rays per rank and step (confs/surf.conf: 512)")
    ap.add_argument("--train-precision", default="fp32", choices=["fp32", "bf16"],
                    help="training-backward policy (model conf key train_precision): bf16 = weight-gradient reductions on bf16 operands")
    ap.add_argument("--split", default="scenes", choices=["scenes", "rays"],
                    help="how N ranks share the work: scenes (default: one scene per rank, weak scaling) or rays (SURVEY 8e's "
                         "single-scene split: ONE scene on N GPUs, rank r renders rays [r R / N, (r + 1) R / N) and an x-range of the "
                         "mesh lattice, rank 0 stitches; strong scaling)")
    ap.add_argument("--check-split", action="store_true",
                    help="(--split rays) rank 0 also renders the whole image and the whole lattice alone and reports whether the "
                         "stitched results are bit-equal")
    ap.add_argument("--force-group", type=int, default=1,
                    help="N = 1: create a world-size-1 process group over --backend anyway (default 1), so that every collective of "
                         "the N-rank runs (barrier, device-tensor MAX, record gather, gather_rows, gradient bucket, DDP) executes over "
                         "RCCL on a one-GPU box; 0 = no group at N = 1")
    ap.add_argument("--group-timeout", type=float, default=600.0, help="process-group timeout in seconds")
    ap.add_argument("--other-configs", type=int, default=1,
                    help="default N = 1 dtu line: also measure configs[4] (T&T shape), configs[2] (15 scenes on this GPU) and the "
                         "single-scene ray split, reported under other_configs")
    ap.add_argument("--fail-rank", type=int, default=-1, help="(tests) this rank exits 3 before the first barrier")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of a multi-rank run: nccl (= RCCL, the measured configuration) or gloo (tests: "
                         "device tensors through gloo, so that N ranks can share ONE GPU with --one-gpu)")
    ap.add_argument("--one-gpu", action="store_true",
                    help="(tests) every rank uses cuda:0 - the real kernels, the record gather, the device-tensor MAX reduction and "
                         "DDP's bucket hooks of an N-rank run on a one-GPU box; needs --backend gloo (RCCL wants one device per rank)")
    args = ap.parse_args(argv)
    if args.one_gpu and args.backend != "gloo" and args.gpus > 1:
        ap.error("--one-gpu needs --backend gloo")
    wl = WORKLOADS[args.workload]
    for k in ("height", "width", "views", "n_samples"):
        if getattr(args, k) is None:
            setattr(args, k, wl[k])
    return args


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, argv, args.rank_timeout))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with matching values "
              f"(or leave WORLD_SIZE unset and let bench.py start the ranks)", file=sys.stderr)
        sys.exit(2)
    result = run_rank(args)
    if result is not None:
        print(json.dumps(result))
        sys.stdout.flush()
    if torch.distributed.is_initialized():
        from surf_amd import dist as D
        D.shutdown()


# ----------------------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------------------


class DryScene:
    """--dry: stands in for a scene and its render call (sleep ~ 2 ms per call)."""

    def __init__(self, seed, R):
        self.seed, self.R = seed, R


def run_rank_split(args):
    """--split rays: ONE scene on N GPUs (north_star's 1 / 2 / 4 / 8-GPU numbers for one image).  Every rank holds the
    same scene (the 19.5 ms volume build is replicated, not split: implicit_surface.py:367-370's rays and :338-351's lattice are
    what shard), renders its contiguous share of the R pixel rays inside the timed region and sends it to rank 0, which stitches
    the image: one all_gather of (R / N, 3) colours per step - the only data-path collective, RCCL over xGMI.  value = R x steps /
    max-over-ranks time: the throughput of ONE image, strong scaling.  After the timed region the mesh lattice is split the same
    way (x-ranges, lattice-mode kernel) and gathered; --check-split compares both with rank 0's own single-rank results."""
    from surf_amd import dist as D
    from surf_amd import ops, synthetic
    from surf_amd.implicit_surface import ImplicitSurface
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    init_group(args, dev, world)
    if args.fail_rank == rank:
        sys.exit(3)
    n_samples = [int(x) for x in args.n_samples.split(",")]
    H, W, nv = args.height, args.width, args.views
    R = H * W
    torch.manual_seed(0)
    model = ImplicitSurface(model_conf(n_samples, args.sdf_precision, args.blend_precision)).to(dev)
    intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
    imgs = synthetic.procedural_images(nv, H, W, 0, dev)
    feats = synthetic.feature_pyramid(nv, H, W, 0, dev)
    vols, tabs, mvol = synthetic.sphere_pyramid(args.base_dim, dev, seed=0)
    sc = model.scene(mvol, vols[::-1], tabs[::-1], None, feats, imgs, intrs.to(dev), c2ws.to(dev))
    rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], H, W, 1, dev)
    near = near_fars[0, 0].reshape(1, 1).repeat(R, 1).to(dev)
    far = near_fars[0, 1].reshape(1, 1).repeat(R, 1).to(dev)
    r0, r1 = rank * R // world, (rank + 1) * R // world

    def render(a, b):
        outs = [model.render_scene(rays_o[s:min(s + RAY_CHUNK, b)], rays_d[s:min(s + RAY_CHUNK, b)], near[s:min(s + RAY_CHUNK, b)],
                                   far[s:min(s + RAY_CHUNK, b)], sc, 1.0, per_sample=False)["color_fine"] for s in range(a, b, RAY_CHUNK)]
        return outs[0] if len(outs) == 1 else torch.cat(outs)

    def step():
        return D.gather_rows(render(r0, r1))           # rank 0: the stitched (R, 3) image

    for _ in range(args.warmup):
        image = step()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        image = step()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    # the rank's own share of the time: render alone (HIP events) vs render + gather
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    render(r0, r1)
    b.record()
    torch.cuda.synchronize()
    render_ms = D.max_over_ranks(a.elapsed_time(b), dev)

    # ---- the mesh lattice, split by x-range (after the timed region) ----
    lattice = None
    res = args.mesh_grid
    if res > 0:
        sdf_w, _ = model.packed_weights(dev)
        axes = [torch.linspace(-1.0, 1.0, res).to(dev) for _ in range(3)]
        x0, x1 = rank * res // world, (rank + 1) * res // world
        u = torch.empty(res, res, res, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        D.barrier()
        t1 = time.perf_counter()
        if x1 > x0:
            ops.sdf_lattice(axes, sc.sv, sdf_w, u, x0, x1 - x0, sign=-1.0)
        torch.cuda.synchronize()
        lat_ms = D.max_over_ranks((time.perf_counter() - t1) * 1e3, dev)
        t2 = time.perf_counter()
        u_all = D.gather_rows(u[x0:x1])
        torch.cuda.synchronize()
        gather_ms = D.max_over_ranks((time.perf_counter() - t2) * 1e3, dev)
        lattice = {"resolution": res, "sdf_ms": lat_ms, "gather_ms": gather_ms}
    check = None
    if args.check_split and rank == 0:
        whole = render(0, R)
        check = {"image_bit_equal": bool(torch.equal(whole, image))}
        if res > 0:
            u1 = torch.empty_like(u)
            ops.sdf_lattice(axes, sc.sv, sdf_w, u1, 0, res, sign=-1.0)
            check["lattice_bit_equal"] = bool(torch.equal(u1, u_all))
    if rank == 0:
        return ({
            "metric": WORKLOADS[args.workload]["metric"], "value": R * args.steps / elapsed, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 (operands split exactly into 3 bf16 pieces, fp32 accumulate)",
                      "f16x2": "f32 accumulate, 22-bit operands (2 fp16 pieces)"}[args.sdf_precision],
            "data": "synthetic",
            "config": {"workload": f"ONE {H}x{W} image ({nv} views, samples {n_samples}) split over {world} GPU(s) by ray range; "
                                   f"colours gathered on rank 0 inside the timed region", "rays_per_step": R, "split": "rays",
                       "rays_per_rank": [(k + 1) * R // world - k * R // world for k in range(world)]},
            "split": {"render_ms_max_over_ranks": render_ms, "step_ms": elapsed / args.steps * 1e3, "lattice": lattice, "check": check},
            "roofline": None, "cpu_baseline": None,
            "collective_backend": collective_note(args)})
    return None


def run_rank(args):
    """One rank of the job `args` names.  Returns the result line as a dict on rank 0 (None elsewhere); main() prints it and
    leaves the process group."""
    if args.workload == "train":
        return run_rank_train(args)
    if args.split == "rays":
        return run_rank_split(args)
    from surf_amd import dist as D
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry
    if dry:
        dev = torch.device("cpu")
        D.init_from_env("gloo")
    else:
        if args.one_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        init_group(args, dev, world)          # RCCL; only the barrier, the MAX of the elapsed time and the record gather
    if torch.distributed.is_initialized():
        assert torch.distributed.get_world_size() == world == args.gpus
    if args.fail_rank == rank:
        sys.exit(3)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    n_samples = [int(x) for x in args.n_samples.split(",")]
    S = sum(n_samples)
    H, W, nv = args.height, args.width, args.views
    R = H * W
    my_scenes = D.shard_scenes(args.scenes, rank, world) if args.scenes > 0 else [rank]

    model = None
    if not dry:
        from surf_amd import ops, synthetic
        from surf_amd.implicit_surface import ImplicitSurface
        torch.manual_seed(0)
        model = ImplicitSurface(model_conf(n_samples, args.sdf_precision, args.blend_precision)).to(dev)
        blend_precision = model.blend_precision

    # ---- scenes (seed = scene id), resident in HBM before the timed region -------------------------------
    scenes = []
    for sid in my_scenes:
        if dry:
            scenes.append(DryScene(sid, R))
            continue
        intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
        imgs = synthetic.procedural_images(nv, H, W, sid, dev)
        feats = synthetic.feature_pyramid(nv, H, W, sid, dev)                     # fine -> coarse
        vols, tabs, mvol = synthetic.sphere_pyramid(args.base_dim, dev, seed=sid)  # coarse -> fine
        sc = model.scene(mvol, vols[::-1], tabs[::-1], None, feats, imgs, intrs.to(dev), c2ws.to(dev))
        rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], H, W, 1, dev)
        near = near_fars[0, 0].reshape(1, 1).repeat(R, 1).to(dev)
        far = near_fars[0, 1].reshape(1, 1).repeat(R, 1).to(dev)
        scenes.append({"id": sid, "scene": sc, "rays_o": rays_o, "rays_d": rays_d, "near": near, "far": far,
                       "cpu": (mvol, vols, tabs, feats, imgs, intrs, c2ws)})
        if not (rank == 0 and sid == my_scenes[0] and world == 1):
            scenes[-1]["cpu"] = None
        del vols, tabs, mvol, feats, imgs

    scene_ms = {}        # scene id -> [ms per step] (HIP events on the launch stream / wall clock when dry)

    def render_scene_once(sc, record):
        if dry:
            t = time.perf_counter()
            time.sleep(0.002)
            if record:
                scene_ms.setdefault(sc.seed, []).append((time.perf_counter() - t) * 1e3)
            return None
        if record:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        outs = []
        for s0 in range(0, R, RAY_CHUNK):
            sl = slice(s0, s0 + RAY_CHUNK)
            outs.append(model.render_scene(sc["rays_o"][sl], sc["rays_d"][sl], sc["near"][sl], sc["far"][sl], sc["scene"],
                                           1.0, per_sample=False))
        if record:
            b.record()
            scene_ms.setdefault(sc["id"], []).append((a, b))
        return outs[0] if len(outs) == 1 else {"color_fine": torch.cat([o["color_fine"] for o in outs])}

    def step(record=False):
        out = None
        for sc in scenes:
            out = render_scene_once(sc, record)
        return out

    for _ in range(args.warmup):
        out = step()
    sync()
    D.barrier()
    sync()
    if model is not None:
        model.kernel_events = []
        model.active_samples_log = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(record=True)
    sync()
    D.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if model is not None:
        events, model.kernel_events = model.kernel_events, None
        active_log, model.active_samples_log = model.active_samples_log, None
    elapsed = D.max_over_ranks(elapsed, dev)

    # ---- per-scene records (gathered: one small object per rank, no tensor traffic) ------------------------
    my_rec = []
    for sid, lst in scene_ms.items():
        ms = [x if dry else x[0].elapsed_time(x[1]) for x in lst]
        my_rec.append({"scene": sid, "rank": rank, "ms_per_render": sum(ms) / len(ms), "rays_per_s": R / (sum(ms) / len(ms) * 1e-3)})
    records = D.gather_records(my_rec)
    n_scenes_total = args.scenes if args.scenes > 0 else world
    rays_per_step_job = n_scenes_total * R

    if dry:
        if rank == 0:
            recs = sorted((r for per_rank in records for r in per_rank), key=lambda r: r["scene"])
            return ({
                "metric": WORKLOADS[args.workload]["metric"], "value": rays_per_step_job * args.steps / elapsed, "unit": "rays/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none (dry run)",
                "data": "dry-run", "config": {"workload": f"DRY RUN of the {args.workload} control flow: no kernels executed",
                                              "scenes": n_scenes_total, "rays_per_step": rays_per_step_job},
                "scenes": recs})
        return None

    # ---- per-kernel durations from the HIP events recorded inside the timed region -----------------------
    per_kernel = {}
    for name, a, b in events:
        per_kernel.setdefault(name, []).append(a.elapsed_time(b))
    kernel_ms = {k: sum(v) / len(v) for k, v in per_kernel.items()}         # average per launch
    launches_per_step = {k: len(v) / args.steps for k, v in per_kernel.items()}
    # active samples per sdf / blend launch (device-side counts of the render calls, read after the timed region)
    active = sum(int(x) for x in active_log) / len(active_log)

    # ---- the other SDF precisions on the same scene (N = 1 only; after the timed region, reported separately) ----
    others = {}
    if world == 1 and args.scenes == 0:
        for prec in [p for p in args.also.split(",") if p and p != args.sdf_precision]:
            model.sdf_precision = prec
            if args.blend_precision is None and prec in ops.BLEND_PRECISIONS:
                model.blend_precision = prec
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
            bl_o = [a.elapsed_time(b) for name, a, b in ev_o if name == "blend"]
            others[prec] = {"rays_per_s": R * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                            "sdf_mlp_ms": sum(sdf_o) / len(sdf_o), "blend_ms": sum(bl_o) / len(bl_o),
                            "blend_precision": model.blend_precision,
                            "max_abs_rgb_diff_vs_headline": float((out_o["color_fine"] - out["color_fine"]).abs().max())}
        model.sdf_precision = args.sdf_precision
        model.blend_precision = blend_precision

    # the timed collectives of the group (EVERY rank takes part: all-reduce of the gradient bucket, gather_rows, barrier)
    probe = collective_probe(dev)
    if rank == 0:
        sdf_kernel, sdf_pipe, pipe_peak, n_prod = SDF_KERNELS[args.sdf_precision]
        sdf_ms = kernel_ms["sdf_mlp"]
        flops = active * FLOP_PER_SAMPLE_SDF
        achieved = flops / (sdf_ms * 1e-3) / 1e12
        peak = pipe_peak / n_prod
        per_gpu = R * len(scenes) * args.steps / elapsed
        bytes_per_ray = algorithmic_bytes_per_ray(S, nv)
        traffic, traffic_src = pmc_traffic(sdf_kernel)
        bl_kernel, bl_pipe, bl_pipe_peak, bl_prod = BLEND_KERNELS[blend_precision]
        bl_kernel = bl_kernel.format(ns=nv - 1)
        bl_ms = kernel_ms["blend"]
        bl_ach = active * (nv - 1) * FLOP_PER_SAMPLE_BLEND_PER_VIEW / (bl_ms * 1e-3) / 1e12
        bl_traffic, bl_src = pmc_traffic(bl_kernel)
        roofline_kernels = [{
            "kernel": bl_kernel, "bound": "mfma", "achieved": bl_ach, "peak": bl_pipe_peak / bl_prod, "unit": "TFLOP/s",
            "frac": bl_ach / (bl_pipe_peak / bl_prod), "traffic": bl_traffic, "traffic_source": bl_src,
            "mfma_busy": pmc_mfma_busy(bl_kernel),
            "avg_launch_ms": bl_ms, "flop_per_sample": (nv - 1) * FLOP_PER_SAMPLE_BLEND_PER_VIEW, "samples_per_launch": active,
            "pipe": bl_pipe, "mfma_products_per_fp32_product": bl_prod, "frac_of_fp32_mfma_peak": bl_ach / 157.3}]
        recs = sorted((r for per_rank in records for r in per_rank), key=lambda r: r["scene"])
        result = {
            "metric": WORKLOADS[args.workload]["metric"],
            "value": rays_per_step_job * args.steps / elapsed,
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
                                   f"{args.base_dim}^3->{args.base_dim * 8}^3, "
                                   + (f"{n_scenes_total} scenes round-robin over {world} GPU(s)" if args.scenes > 0 else "one scene per GPU"),
                       "rays_per_step": rays_per_step_job, "samples_per_ray": S, "active_samples_per_launch": active,
                       "scenes": n_scenes_total, "blend_precision": blend_precision},
            "per_gpu_rays_per_s": per_gpu,
            "kernel_ms": kernel_ms,
            "kernel_launches_per_step": launches_per_step,
            "roofline": {"kernel": sdf_kernel, "bound": "mfma", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": traffic_src,
                         # matrix-pipe busy fraction from the PMC pass of the same profile (None when stale)
                         "mfma_busy": pmc_mfma_busy(sdf_kernel),
                         "flop_per_sample": FLOP_PER_SAMPLE_SDF, "samples_per_launch": active,
                         "avg_launch_ms": sdf_ms, "pipe": sdf_pipe, "pipe_dense_peak": pipe_peak,
                         "mfma_products_per_fp32_product": n_prod, "frac_of_fp32_mfma_peak": achieved / 157.3,
                         "frac_of_raw_16bit_dense_peak": achieved * n_prod / pipe_peak if n_prod > 1 else None,
                         # the whole step against the HBM roofline (SURVEY 8d / north_star): algorithmic bytes per ray x
                         # rays/s per GPU / 8 TB/s
                         "hbm_frac": bytes_per_ray * per_gpu / HBM_PEAK, "hbm_bytes_per_ray": bytes_per_ray,
                         "hbm_peak": HBM_PEAK},
            "roofline_kernels": roofline_kernels,
        }
        if args.scenes > 0 or world > 1:
            result["scenes"] = recs
        if others:
            result["other_precisions"] = others
        sc0 = scenes[0]
        if world == 1 and args.mesh_grid > 0:
            total_ms, k_ms, inside, mc = mesh_grid_timing(model, sc0["scene"], dev, args.mesh_grid)
            n_lat = args.mesh_grid ** 3
            fk, fpipe, fpeak, fprod = SDF_KERNELS[args.sdf_precision]
            fk = fk.replace("true", "false")
            ach = n_lat * FLOP_PER_SAMPLE_SDF_FWD / (k_ms * 1e-3) / 1e12
            result["mesh_grid_ms"] = total_ms
            result["mesh_grid"] = {"resolution": args.mesh_grid, "points": n_lat, "total_ms": total_ms, "sdf_kernel_ms": k_ms,
                                   "lattice_points_inside": inside, **mc}
            roofline_kernels.append({"kernel": fk, "bound": "mfma", "achieved": ach, "peak": fpeak / fprod, "unit": "TFLOP/s",
                                     "frac": ach / (fpeak / fprod), "traffic": None,
                                     "avg_launch_ms": k_ms / max(1, getattr(mesh_grid_timing, "launches", 1)),
                                     "flop_per_sample": FLOP_PER_SAMPLE_SDF_FWD,
                                     "samples_per_launch": n_lat // max(1, getattr(mesh_grid_timing, "launches", 1)),
                                     "pipe": fpipe, "frac_of_fp32_mfma_peak": ach / 157.3,
                                     "note": f"{args.mesh_grid}^3 lattice of extract_geometry (row a16), forward only"})
        if world == 1 and args.cpu_seconds > 0 and sc0["cpu"] is not None:
            mvol, vols, tabs, feats, imgs, intrs, c2ws = sc0["cpu"]
            n_sub = 8192
            idx = torch.linspace(0, R - 1, n_sub).long()
            cpu_scene = {
                "mvol": mvol.cpu(), "vols": [v[:, :7].cpu() for v in vols[::-1]],
                "tabs": [t.cpu().long() for t in tabs[::-1]], "feats": [f.cpu() for f in feats], "imgs": imgs.cpu(),
                "intrs": intrs, "c2ws": c2ws,
            }
            cpu_scene["masks"] = [(t >= 0).float() for t in cpu_scene["tabs"]]
            rps, n_done, dt, err = cpu_baseline(model, cpu_scene, sc0["rays_o"].cpu()[idx], sc0["rays_d"].cpu()[idx],
                                                sc0["near"].cpu()[idx], sc0["far"].cpu()[idx], n_samples, args.cpu_seconds,
                                                out, idx)
            result["cpu_baseline"] = {"value": rps, "unit": "rays/s", "cores": CPU_THREADS, "threads": CPU_THREADS,
                                      "host_cores": os.cpu_count(), "kind": "port",
                                      "sample": f"{n_done} rays (every {R // n_sub}th pixel ray, 256-ray chunks) of the same "
                                                f"scene in {dt:.1f} s, torch CPU fp32",
                                      "max_abs_rgb_diff_vs_gpu": err}
        if world == 1 and args.build and args.workload == "dtu":
            sc0["cpu"] = None
            result["volume_build"] = volume_build_timing(args, dev)
            if args.mesh_grid > 0:
                result["scene"] = scene_timing(args, dev, args.mesh_grid)
        if world == 1 and args.train_step and args.workload == "dtu":
            result["training_step"] = training_step_timing(args, dev)
        result["collective_backend"] = collective_note(args)
        result["collectives"] = probe
        if world == 1 and args.other_configs and args.workload == "dtu" and args.scenes == 0:
            del scenes, sc0, out
            torch.cuda.empty_cache()
            result["other_configs"] = other_configs(args)
        return result
    return None


def other_configs(args):
    """The BASELINE configs the headline line does not quote, measured in this same process after it (N = 1, default run):
    configs[4] the Tanks&Temples shape, configs[2] the 15-scan split dealt onto this one GPU, and SURVEY 8e's single-scene ray
    split with its bit-equality check.  Each entry is that job's own line (own `config.workload`, `ms_per_step`, `roofline.frac`)
    without its CPU / build / training legs; the headline's timed region is over before any of them starts."""
    import copy
    jobs = [("tnt", ["--workload", "tnt", "--steps", "3", "--warmup", "1"]),
            ("scenes15", ["--scenes", "15", "--steps", "1", "--warmup", "1"]),
            ("split_rays", ["--split", "rays", "--check-split", "--steps", "3", "--warmup", "1", "--mesh-grid", "512"])]
    common = ["--gpus", "1", "--cpu-seconds", "0", "--build", "0", "--train-step", "0", "--also", "", "--other-configs", "0",
              "--sdf-precision", args.sdf_precision, "--base-dim", str(args.base_dim)]
    res = {}
    for name, argv in jobs:
        a = parse_args(argv + common + ([] if "--mesh-grid" in argv else ["--mesh-grid", "0"]))
        t0 = time.perf_counter()
        try:
            r = run_rank(a)
        except Exception as e:       # noqa: BLE001 - a failing extra must not take the headline line with it
            r = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        torch.cuda.synchronize()
        r["wall_s_incl_scene_setup"] = time.perf_counter() - t0
        if isinstance(r.get("scenes"), list) and len(r["scenes"]) > 4:
            ms = [x["ms_per_render"] for x in r["scenes"]]
            r["scenes"] = {"count": len(ms), "ms_per_render_min": min(ms), "ms_per_render_max": max(ms), "ms_per_render_mean": sum(ms) / len(ms)}
        for k in ("kernel_launches_per_step",):
            r.pop(k, None)
        res[name] = r
        torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
