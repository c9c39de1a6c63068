"""GPU: the training boundary is the reference's: a train-mode forward carries an autograd graph, so the literal
runner.py:152-164 sequence - forward, Loss, optimizer.zero_grad(), loss.backward(), optimizer.step() - fills every `.grad`
through the HIP backward kernels, and DistributedDataParallel(model) (runner.py:102) averages them.  Pinned by the
reference's own loss.backward() (tests/golden/train_grads.npz)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest
import torch

from oracle import surf_oracle as O
from tests.golden_cfg import CFG, pipeline_views

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def rel_close(a, b, rtol, atol, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {err.max().item():.3e}"


def _check_against_reference_backward(named_params, vol_grads_f2c, gg, param_tol=5e-4):
    n = 0
    for name, p_ in named_params:
        ref = gg["grad/" + name]
        assert p_.grad is not None, name
        if name == "color_network.s":          # see test_color_network_s_gradient_is_conditioned_like_this (test_hip_parity.py)
            assert abs(float(p_.grad) - float(ref)) <= 0.35 * abs(float(ref)) + 1e-6
            continue
        rel_close(p_.grad, ref, 5e-3, param_tol * float(ref.abs().max()) + 1e-7, name)
        n += 1
    assert n >= 7 * 3 + 1 + 20
    for lvl in range(4):
        ref = gg[f"grad_vol{lvl}"]
        rel_close(vol_grads_f2c[lvl], ref, 5e-3, 5e-4 * float(ref.abs().max()), f"vol{lvl}")


class TorchLoss(torch.nn.Module):
    """The reference's Loss.forward in mode "val" (losses/loss.py:27-111) with ITS OWN way of computing the patch term:
    compute_LNCC2 in torch ops on ref_gray_val / sampled_gray_val (the oracle's restatement, pinned by the reference's
    values), so autograd hands the patch stacks' gradients - not d/d ncc - back to the render node."""

    def __init__(self, c):
        super().__init__()
        self.c = c

    def forward(self, preds, targets, step=None, mode="train"):
        c = self.c
        vm = (preds["valid_mask"] * targets["mask"].reshape(-1, 1)).float()
        color = ((preds["color_fine"] - targets["color"]).abs() * vm).sum() / (vm.sum() + 1e-5)
        eik = preds["gradient_error"].mean()
        sparse = torch.exp(-preds["sparse_sdf"].abs() * c["sparse_scale_factor"]).mean() * min(1.0, step / 2)
        smooth = preds["smooth_error"].mean()
        ncc = O.lncc(preds["ref_gray_val"], preds["sampled_gray_val"])
        nm = vm * preds["mid_inside_sphere"]
        mfc = 0.5 * ((ncc * nm).sum(dim=0) / (nm.sum(dim=0) + 1e-8)).squeeze(-1)
        psdf = preds["pseudo_sdf"].abs().mean()

        def ml1(t):
            m = (t > 0).float()
            return ((preds["render_depth"] - t).abs() * m).sum() / (m.sum() + 1e-8)
        loss = (color * c["color_weight"] + eik * c["igr_weight"] + sparse * c["sparse_weight"] + mfc * c["mfc_weight"]
                + smooth * c["smooth_weight"] + ml1(targets["depth"]) * c["depth_weight"] + psdf * c["pseudo_sdf_weight"]
                + ml1(targets["pseudo_depth"]) * c["pseudo_depth_weight"])
        return {"loss": loss}


@pytest.mark.parametrize("loss_kind", ["hip_terms", "torch_terms"])
def test_module_swap_train_forward_equals_the_reference_loss_backward(scene, weights, golden_fpn, golden_pipe, golden_grads, loss_kind):
    """surf_amd's ImplicitSurface inside the reference's models/surf.py.  The very call of
    tests/golden/make_golden_grad.py - isurf("train", ipts, mvol, volumes(requires_grad), tables, masks, features, features,
    cos_anneal, step) -> Loss -> loss.backward() - on the HIP kernels: loss value, every parameter gradient and the gradients
    of the caller's own (N_s, 7) volume tensors equal the reference's.  loss_kind: surf_amd.losses.Loss (HIP NCC with its own
    backward) or a torch-ops Loss like the reference's (the patch stacks' gradients flow back instead)."""
    from bench import model_conf
    from surf_amd import conf
    from surf_amd.implicit_surface import ImplicitSurface
    from surf_amd.losses import Loss
    from tests.golden.make_golden_grad import COS_ANNEAL, SEED, STEP
    from tests.golden.make_golden_train import LOSS_CONF
    d = dev()
    gg = golden_grads
    isurf = ImplicitSurface(model_conf(CFG["n_samples"], "f32")).train()
    isurf.load_state_dict({k[len("implicit_surface."):]: v for k, v in weights.items() if k.startswith("implicit_surface.")})
    isurf = isurf.to(d)
    vols, tabs, masks, mvol = pipeline_views(golden_pipe)
    vols = [v.to(d).clone().requires_grad_(True) for v in vols]
    feats = [golden_fpn[f"out{i}"].to(d) for i in range(4)][::-1]
    ipts = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in scene.items()}
    ipts["pseudo_pts"] = gg["pseudo_pts"].to(d)
    targets = {k[len("target_"):]: v.to(d) for k, v in gg.items() if k.startswith("target_")}
    loss_fn = Loss(conf.from_dict(LOSS_CONF)) if loss_kind == "hip_terms" else TorchLoss(LOSS_CONF)
    optimizer = torch.optim.Adam(list(isurf.parameters()) + vols, lr=1e-4)
    torch.manual_seed(SEED)
    outs = isurf("train", ipts, mvol.to(d)[None, None], vols, [t.to(d) for t in tabs], [m.to(d) for m in masks], feats, feats,
                 COS_ANNEAL, STEP)
    assert outs["color_fine"].grad_fn is not None and outs["ref_gray_val"].grad_fn is not None
    lo = loss_fn(outs, targets, STEP, "val")
    optimizer.zero_grad()
    lo["loss"].backward()
    rel_close(lo["loss"].detach().reshape(1), gg["loss"], 1e-3, 1e-4, "loss")
    _check_against_reference_backward(isurf.named_parameters(), [v.grad for v in vols], gg)
    before = isurf.sdf_network.lin3.weight_v.detach().clone()
    optimizer.step()
    assert float((isurf.sdf_network.lin3.weight_v.detach() - before).abs().max()) > 0


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_runner_sequence_on_a_has_vol_model_equals_the_reference_loss_backward(scene, weights, golden_fpn, golden_pipe, golden_grads,
                                                                               precision):
    """runner.py:152-164 verbatim on surf_amd.surf.SuRF (finetune parameter set: implicit surface + per-scene volumes).
    precision = the model conf's `train_precision`: "bf16" (BASELINE configs[3]) rounds the operands of the weight-gradient
    reductions to bf16 (fp32 accumulate) - the reference's gradients are then reproduced to 2 % of each tensor's largest entry
    instead of 0.05 %; loss, forward and the feature rows' gradients are untouched by the policy."""
    from surf_amd import conf
    from surf_amd.losses import Loss
    from surf_amd.surf import SuRF
    from tests.golden.make_golden import MODEL_CONF
    from tests.golden.make_golden_grad import COS_ANNEAL, SEED, STEP
    from tests.golden.make_golden_train import LOSS_CONF
    d = dev()
    gg = golden_grads
    cfg = dict(MODEL_CONF, has_vol=True, train_precision=precision)
    cfg["implicit_surface"] = dict(cfg["implicit_surface"])
    cfg["implicit_surface"]["render"] = dict(cfg["implicit_surface"]["render"], sdf_precision="f32", blend_precision="f32", perturb=0.0)
    model = SuRF(conf.from_dict(cfg)).to(d)
    model.implicit_surface.load_state_dict({k[len("implicit_surface."):]: v for k, v in weights.items()
                                            if k.startswith("implicit_surface.")})
    vols, tabs, masks, mvol = pipeline_views(golden_pipe)                           # fine -> coarse
    model.volumes = torch.nn.ParameterList([torch.nn.Parameter(v.to(d).clone()) for v in vols[::-1]])
    model.sparse_idxes = torch.nn.ParameterList([torch.nn.Parameter(t.to(d).to(torch.int32), requires_grad=False) for t in tabs[::-1]])
    model.matching_volume = torch.nn.Parameter(mvol.to(d)[None, None].clone(), requires_grad=False)
    model.features = [golden_fpn[f"out{i}"].to(d) for i in range(4)]                # coarse -> fine
    model.train()
    inputs = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in scene.items()}
    inputs["pseudo_pts"] = gg["pseudo_pts"].to(d)
    inputs.update({k[len("target_"):]: v.to(d) for k, v in gg.items() if k.startswith("target_")})
    loss_fn = Loss(conf.from_dict(LOSS_CONF))
    optimizer = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "vol_lr": [1e-2] * 4}))
    torch.manual_seed(SEED)
    # ---- runner.py:155-164 ----
    outputs = model("train", inputs, cos_anneal_ratio=COS_ANNEAL, step=STEP)
    psnr = 20.0 * torch.log10(1.0 / (((outputs["color_fine"] - inputs["color"]) ** 2).mean()).sqrt())
    loss_res = loss_fn(outputs, inputs, STEP, "val")
    loss = loss_res["loss"]
    optimizer.zero_grad()
    loss.backward()
    # ----
    assert bool(torch.isfinite(psnr))
    rel_close(loss.detach().reshape(1), gg["loss"], 1e-3, 1e-4, "loss")
    named = [(n[len("implicit_surface."):], p) for n, p in model.named_parameters() if n.startswith("implicit_surface.")]
    try:
        _check_against_reference_backward(named, [p.grad for p in list(model.volumes)[::-1]], gg, param_tol=5e-4 if precision == "fp32" else 2e-2)
    finally:
        from surf_amd import ops
        ops.set_train_precision("fp32")
    optimizer.step()


def _full_model(seed=4):
    from surf_amd import conf
    from surf_amd.surf import SuRF
    from tests.golden.make_golden import MODEL_CONF
    cfg = dict(MODEL_CONF)
    cfg["reg_network"] = {"d_in": [8, 16, 16, 16], "d_base": [8] * 4, "d_out": [8] * 4}
    torch.manual_seed(seed)
    model = SuRF(conf.from_dict(cfg))
    with torch.no_grad():
        model.implicit_surface.deviation_network.variance.fill_(0.3)
        for net in model.reg_network.nets:
            net.out_lin.weight.mul_(4.0)
    return model


def _full_batch(scene, d):
    ipts = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in scene.items()}
    ipts["src_idx"] = 1
    R = scene["rays_o"].shape[0]
    H, W = scene["imgs"].shape[-2:]
    g = torch.Generator().manual_seed(5)
    ipts["pseudo_pts"] = ((torch.rand(300, 3, generator=g) * 2 - 1) * 0.7).to(d)
    ones = torch.ones(H, W, device=d)
    ipts.update({"color": torch.rand(R, 3, generator=g).to(d), "mask_ref": ones, "mask_src": ones, "pseudo_depth_ref": ones,
                 "pseudo_depth_src": ones, "depth_ref": ones, "depth_src": ones})
    return ipts


def test_runner_sequence_on_the_full_model_equals_the_explicit_backward(scene):
    """Generalisation training (configs[3]): runner.py:152-164 verbatim on a volume-building model; every parameter of
    surf.py:36-45's two groups receives through `loss.backward()` the gradient that the explicit chain (leaf copies of the
    outputs -> SuRF.backward -> SuRF.backward_volumes, each pinned elsewhere against the reference's / the oracle's autograd)
    produces for the same forward."""
    from surf_amd import conf, ops
    from surf_amd.losses import Loss
    from tests.golden.make_golden_train import LOSS_CONF
    d = dev()
    model = _full_model().to(d).train()
    inputs = _full_batch(scene, d)
    loss_fn = Loss(conf.from_dict(LOSS_CONF))
    optimizer = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3}))
    bn0 = model.reg_network.nets[0].conv0.net[1].running_mean.clone()
    torch.manual_seed(70)
    outputs = model("train", inputs, cos_anneal_ratio=1.0, step=3.0)
    loss_res = loss_fn(outputs, inputs, 3.0)
    loss = loss_res["loss"]
    optimizer.zero_grad()
    loss.backward()
    got = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    assert all(bool(torch.isfinite(g).all()) for g in got.values())
    assert float((model.reg_network.nets[0].conv0.net[1].running_mean - bn0).abs().max()) > 0       # train-mode BatchNorm ran

    # the explicit chain on an identical forward
    model.zero_grad(set_to_none=True)
    torch.manual_seed(70)
    preds = model("train", inputs, 1.0, 3.0, record=True)
    assert preds["color_fine"].grad_fn is None
    torch.testing.assert_close(preds["color_fine"], outputs["color_fine"].detach(), rtol=0, atol=1e-6)
    preds["ncc"] = ops.lncc(preds["ref_gray_val"].contiguous(), preds["sampled_gray_val"].contiguous())
    keys = ("color_fine", "render_depth", "gradient_error", "sparse_sdf", "ncc", "smooth_error", "pseudo_sdf")
    dkeys = [f"depth_stage{i}" for i in range(4)] + [f"depth_src_stage{i}" for i in range(4)]
    leaves = {k: preds[k].detach().clone().requires_grad_(True) for k in keys + tuple(dkeys)}
    lo = loss_fn({**preds, **leaves}, inputs, step=3.0)
    lo["loss"].backward()
    torch.testing.assert_close(lo["loss"].detach(), loss.detach(), rtol=1e-5, atol=1e-6)
    g = {k: v.grad for k, v in leaves.items()}
    rows = model.backward(g["color_fine"], g["render_depth"], float(g["gradient_error"]), g["sparse_sdf"], g["ncc"],
                          float(g["smooth_error"]), g["pseudo_sdf"])
    model.backward_volumes(rows, {i: (g[f"depth_stage{i}"], g[f"depth_src_stage{i}"]) for i in range(4)})
    worst = []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        if n == "volume.agg_mlp.2.bias":
            # the view softmax is shift invariant: this gradient is zero in exact arithmetic and atomics-order rounding noise
            # in both runs (tests/test_volume_backward.py::test_volume_build_backward_end_to_end)
            w2 = float(model.volume.agg_mlp[2].weight.grad.abs().max())
            assert float(got[n].abs().max()) < 1e-3 * w2 and float(ref.abs().max()) < 1e-3 * w2
            continue
        scale = max(float(ref.abs().max()), 1e-8)
        worst.append((float((got[n] - ref).abs().max()) / scale, n))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-3, worst[:6]          # float atomics: summation order differs between two launches
    optimizer.step()


DDP_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from surf_amd import conf
    from surf_amd.losses import Loss
    from tests.conftest import load_npz
    from tests.golden.make_golden_train import LOSS_CONF
    from tests.test_autograd_runner import _full_model, _full_batch

    mode = sys.argv[1]                      # "ddp": one of two ranks; "ref": single process, both batches
    dev = torch.device("cuda:0")
    scene = load_npz("scene.npz")
    R = scene["rays_o"].shape[0]
    loss_fn = Loss(conf.from_dict(LOSS_CONF))

    def batch(r):
        ipts = _full_batch(scene, dev)
        sel = torch.arange(r, R, 2)
        for k in ("rays_o", "rays_d", "color"):
            ipts[k] = ipts[k][sel.to(dev)].contiguous()
        return ipts

    def step(model, inputs):                # runner.py:155-163
        torch.manual_seed(70)
        outputs = model("train", inputs, cos_anneal_ratio=1.0, step=3.0)
        loss = loss_fn(outputs, inputs, 3.0)["loss"]
        loss.backward()

    if mode == "ddp":
        dist.init_process_group("gloo")     # one GPU on the box: both ranks on cuda:0, CUDA tensors through gloo
        rank = dist.get_rank()
        model = _full_model(seed=4 + 10 * rank).to(dev).train()          # DIFFERENT initial weights per rank
        ddp = DistributedDataParallel(model, device_ids=[0])              # runner.py:102
        opt = torch.optim.SGD(model.get_optim_params({"mlp_lr": 1e-2, "feat_lr": 1e-2}))
        opt.zero_grad()
        step(ddp, batch(rank))
        opt.step()
    else:
        model = _full_model(seed=4).to(dev).train()
        opt = torch.optim.SGD(model.get_optim_params({"mlp_lr": 1e-2, "feat_lr": 1e-2}))
        grads = []
        for r in range(2):
            bn = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
            model.zero_grad(set_to_none=True)
            step(model, batch(r))
            grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.requires_grad})
            if r == 0:
                model.load_state_dict(bn, strict=False)    # every rank starts the step from the same running statistics
        for n, p in model.named_parameters():
            if p.requires_grad:
                p.grad = 0.5 * (grads[0][n] + grads[1][n])
        opt.step()
    torch.cuda.synchronize()
    sd = {n: p.detach().double().cpu() for n, p in model.named_parameters() if p.requires_grad}
    print("RESULT " + json.dumps({"sum": {n: float(v.sum()) for n, v in sd.items()}, "abs": {n: float(v.abs().sum()) for n, v in sd.items()}}))
    if mode == "ddp":
        dist.barrier()
        dist.destroy_process_group()
""") % ROOT


def test_distributed_data_parallel_wrap_two_ranks(tmp_path):
    """runner.py:102: DistributedDataParallel(model) around surf_amd's SuRF, two fresh processes (one GPU on the test box, so
    both ranks share cuda:0 and the process group is gloo - the hooks and buckets are the ones RCCL serves on N GPUs).  The
    ranks start from DIFFERENT seeds: the wrap must broadcast rank 0's parameters; after one runner step on different ray
    batches both ranks hold identical parameters, equal to a single-process step on the mean of the two gradients."""
    script = tmp_path / "ddp_worker.py"
    script.write_text(DDP_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    procs = [subprocess.Popen([sys.executable, str(script), "ddp"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(base, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port))) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-3000:] for o in outs]

    def result(out):
        return json.loads([ln for ln in out.splitlines() if ln.startswith("RESULT ")][-1][7:])
    r0, r1 = result(outs[0][0]), result(outs[1][0])
    assert r0 == r1
    ref_p = subprocess.Popen([sys.executable, str(script), "ref"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=base)
    ref_out = ref_p.communicate(timeout=900)
    assert ref_p.returncode == 0, ref_out[1][-3000:]
    ref = result(ref_out[0])
    worst = max(abs(v - r0["sum"][n]) / max(ref["abs"][n], 1e-9) for n, v in ref["sum"].items())
    assert worst < 2e-5, worst


def test_lncc_backward_matches_autograd(golden_train):
    """surf_lncc_backward (the autograd of compute_LNCC2, losses/ncc.py:7-51) against torch autograd through the oracle's
    restatement (itself pinned by the reference's values): gradients of both patch stacks, the top-2 view selection (zero
    gradient for the other views), through surf_amd.autograd.lncc."""
    from surf_amd import autograd
    d = dev()
    gt = golden_train
    for rk, sk in (("unit_ref", "unit_src"), ("ref_gray_val", "sampled_gray_val")):
        ref, src = gt[rk].clone().requires_grad_(True), gt[sk].clone().requires_grad_(True)
        g = torch.Generator().manual_seed(3)
        up = torch.randn(ref.shape[1], 1, generator=g)
        (O.lncc(ref, src) * up).sum().backward()
        ref_d, src_d = gt[rk].to(d).requires_grad_(True), gt[sk].to(d).requires_grad_(True)
        out = autograd.lncc(ref_d, src_d)
        (out * up.to(d)).sum().backward()
        for a, b, nm in ((ref_d.grad, ref.grad, rk), (src_d.grad, src.grad, sk)):
            scale = float(b.abs().max())
            assert scale > 0
            rel_close(a, b, 1e-3, 2e-5 * scale, nm)
        assert int((src.grad.abs().sum(dim=(2, 3)) > 0).sum(dim=0).max()) <= 2


def test_a_batch_that_misses_the_volume_still_trains(scene, weights, golden_fpn, golden_pipe):
    """implicit_surface.py:88-89: when no sample of the batch is masked in, the reference sends the first ten points through the
    networks anyway (voxel_mask itself stays zero, so their compositing weights are zero): sparse_sdf holds ten real SDF values
    instead of the placeholder 100 and loss.backward() has a graph to walk.  Rays pointing away from the volume: the forward's
    ten values equal the oracle's SDF at those points, the colour is zero, the backward fills finite gradients."""
    from bench import model_conf
    from surf_amd import ops
    from surf_amd.implicit_surface import ImplicitSurface
    d = dev()
    isurf = ImplicitSurface(model_conf(CFG["n_samples"], "f32")).train()
    isurf.load_state_dict({k[len("implicit_surface."):]: v for k, v in weights.items() if k.startswith("implicit_surface.")})
    isurf = isurf.to(d)
    vols, tabs, masks, mvol = pipeline_views(golden_pipe)
    vols_d = [v.to(d).clone().requires_grad_(True) for v in vols]
    feats = [golden_fpn[f"out{i}"].to(d) for i in range(4)][::-1]
    ipts = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in scene.items()}
    ipts["rays_o"] = scene["rays_o"].to(d) + 50.0                       # far outside [-1, 1]^3 and looking away
    torch.manual_seed(0)
    outs = isurf("train", ipts, mvol.to(d)[None, None], vols_d, [t.to(d) for t in tabs], [m.to(d) for m in masks], feats, feats, 1.0, 3)
    R = scene["rays_o"].shape[0]
    S = sum(CFG["n_samples"])
    ss = outs["sparse_sdf"].detach().cpu().reshape(-1)[1024:]
    assert ss.shape[0] == R * S
    assert float((ss[10:] - 100.0).abs().max()) == 0.0 and float((ss[:10] - 100.0).abs().min()) > 1.0
    assert float(outs["color_fine"].detach().abs().max()) == 0.0 and float(outs["smooth_error"].detach()) == 0.0
    # the oracle's SDF at those ten points (all their feature lookups are empty: outside every table)
    st = ops.ray_setup(ipts["rays_o"].float().contiguous(), ipts["rays_d"].float().contiguous(), ipts["near"].repeat(R, 1).float(),
                       ipts["far"].repeat(R, 1).float(), mvol.to(d).contiguous(), ops.SparseVolumes([v.to(d) for v in vols], [t.to(d) for t in tabs]),
                       CFG["n_samples"], [1.0, 0.4, 0.1, 0.01], 256)
    pts10 = st["pts"][:10].cpu()
    sdf_ref = O.sdf_mlp(O.sdf_weights(weights), pts10, torch.zeros(10, 28))[0].reshape(-1)
    rel_close(ss[:10], sdf_ref, 1e-4, 1e-4)
    loss = torch.exp(-outs["sparse_sdf"].abs() * 0.01).mean() + outs["color_fine"].sum() + outs["gradient_error"]
    loss.backward()
    g = isurf.sdf_network.lin0.weight_v.grad
    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0


def test_module_swap_feature_maps_receive_the_colour_paths_gradient(scene, weights, golden_fpn, golden_pipe):
    """INTEGRATION.md 1 with a trainable FPN upstream: the NCHW feature maps the caller passes in (with autograd history) get the
    blending network's share of the gradient back - the texel4 maps of surf_blend_backward, re-laid out - equal to what the
    explicit chain (ImplicitSurface.backward_render with gfeats_t4) accumulates for the same upstream gradient."""
    from bench import model_conf
    from surf_amd.implicit_surface import ImplicitSurface
    d = dev()
    isurf = ImplicitSurface(model_conf(CFG["n_samples"], "f32")).train()
    isurf.load_state_dict({k[len("implicit_surface."):]: v for k, v in weights.items() if k.startswith("implicit_surface.")})
    isurf = isurf.to(d)
    vols, tabs, masks, mvol = pipeline_views(golden_pipe)
    vols_d = [v.to(d) for v in vols]
    feats = [golden_fpn[f"out{i}"].to(d).clone().requires_grad_(True) for i in range(4)][::-1]
    ipts = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in scene.items()}
    g = torch.Generator().manual_seed(9)
    gc = torch.randn(scene["rays_o"].shape[0], 3, generator=g).to(d)
    torch.manual_seed(0)
    outs = isurf("train", ipts, mvol.to(d)[None, None], vols_d, [t.to(d) for t in tabs], [m.to(d) for m in masks], feats, feats, 1.0, 3)
    (outs["color_fine"] * gc).sum().backward()
    got = [f.grad for f in feats]
    assert all(x is not None and float(x.abs().max()) > 0 for x in got)
    # the explicit chain on an identical forward
    with torch.no_grad():
        torch.manual_seed(0)
        sc = isurf.scene(mvol.to(d)[None, None], vols_d, [t.to(d) for t in tabs], None, [f.detach() for f in feats], ipts["imgs"],
                         ipts["intrs"], ipts["c2ws"])
        R = scene["rays_o"].shape[0]
        isurf.render_scene(ipts["rays_o"], ipts["rays_d"], ipts["near"].repeat(R, 1), ipts["far"].repeat(R, 1), sc, 1.0,
                           patch_warp=True, step=3)
        gf = [torch.zeros_like(f) for f in sc.feats_t4]
        for p_ in isurf.parameters():
            p_.grad = None
        isurf.backward_render(gc, gfeats_t4=gf)
    for a, b in zip(got, gf):
        ref = b.permute(0, 3, 1, 2)
        rel_close(a, ref, 1e-4, 1e-5 * float(ref.abs().max()), "feature map gradient")
