"""GPU parity of the volume-build backward kernels (row f2): each HIP backward against torch autograd through the CPU
oracle's restatement of the same reference function, on the golden scene.  Gradients are sums of many float atomics:
tolerances are relative to the largest reference entry of each tensor."""
import numpy as np
import os

import pytest
import torch

from oracle import surf_oracle as O
from tests.golden_cfg import CFG

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def grad_close(a, b, rtol=2e-3, floor=1e-6):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(float(b.abs().max()), floor)
    err = (a - b).abs()
    bad = err > rtol * b.abs() + 2e-4 * scale
    assert not bool(bad.any()), f"{int(bad.sum())}/{bad.numel()} off, max err {err.max().item():.3e} (scale {scale:.3e})"


def _cams(scene):
    from surf_amd import ops
    return ops._cams_ext(ops.Cameras(scene["intrs"], scene["c2ws"]), scene["intrs"], scene["c2ws"])


@pytest.mark.parametrize("stage,perturb", [(0, False), (1, True), (2, False)])
def test_matching_depth_backward(scene, golden_pipe, golden_train, stage, perturb):
    """d depth maps / d matching volume (matching_field.py:18-71 under autograd): views 0 and src_idx only, bands from the
    detached previous depths, with and without the train-mode jitter."""
    from surf_amd import conf, ops
    from surf_amd.matching_field import MatchingField
    d = dev()
    gp = golden_pipe
    H, W = scene["imgs"].shape[-2:]
    nv = scene["intrs"].shape[0]
    src_idx = int(golden_train["mf_perturb_src_idx"])
    mvol = gp[f"s{stage}_mvol"].clone().requires_grad_(True)
    pre = None if stage == 0 else gp[f"s{stage - 1}_depths"]
    g = torch.Generator().manual_seed(3 + stage)
    G = torch.randn(nv, H, W, generator=g)
    for v in range(nv):
        if v != 0 and v != src_idx:
            G[v] = 0
    torch.manual_seed(31)
    ref = O.matching_field((H, W), scene["intrs"], scene["c2ws"], scene["near_fars"], mvol, stage, CFG["range_ratios"],
                           CFG["n_samples_depths"], CFG["depth_res_levels"], None if pre is None else list(pre),
                           perturb=perturb, src_idx=src_idx)
    (torch.stack(ref) * G).sum().backward()

    mf = MatchingField(conf.from_dict({"n_samples_depths": CFG["n_samples_depths"], "n_importance_depths": [0] * 4,
                                       "up_sample_steps": [0] * 4, "depth_res_levels": CFG["depth_res_levels"]}))
    lvl = CFG["depth_res_levels"][stage]
    jitter = None
    if perturb:
        torch.manual_seed(31)
        jitter = mf.draw_jitter(nv, (H // lvl) * (W // lvl), 1 if pre is None else 2, src_idx).to(d).contiguous()
    dm = ops.matching_depth_backward(gp[f"s{stage}_mvol"].to(d).contiguous(), _cams(scene), scene["near_fars"], H, W, lvl,
                                     CFG["n_samples_depths"][stage], G.to(d).contiguous(),
                                     None if pre is None else pre.to(d).contiguous(), CFG["range_ratios"][stage],
                                     CFG["range_ratios"][stage - 1] if stage > 0 else 1.0, jitter=jitter, views=(0, src_idx))
    assert float(mvol.grad.abs().max()) > 0
    grad_close(dm, mvol.grad)
    # round 5: the backward started from the forward's per-ray softmax statistics (one walk over the samples instead of two)
    saved = {}
    ops.matching_depth(gp[f"s{stage}_mvol"].to(d).contiguous(), _cams(scene), scene["near_fars"], H, W, lvl,
                       CFG["n_samples_depths"][stage], None if pre is None else pre.to(d).contiguous(), CFG["range_ratios"][stage],
                       CFG["range_ratios"][stage - 1] if stage > 0 else 1.0, jitter=jitter, saved=saved)
    assert tuple(saved["stats"].shape) == (nv, H // lvl, W // lvl, 4)
    dm2 = ops.matching_depth_backward(gp[f"s{stage}_mvol"].to(d).contiguous(), _cams(scene), scene["near_fars"], H, W, lvl,
                                      CFG["n_samples_depths"][stage], G.to(d).contiguous(),
                                      None if pre is None else pre.to(d).contiguous(), CFG["range_ratios"][stage],
                                      CFG["range_ratios"][stage - 1] if stage > 0 else 1.0, jitter=jitter, views=(0, src_idx),
                                      stats=saved["stats"])
    grad_close(dm2, mvol.grad)
    grad_close(dm2, dm, rtol=1e-4)


@pytest.mark.parametrize("D", [4, 8, 10, 12, 32, 70])      # 70: two z-tiles of the gather kernel, ragged in x / y / z
def test_densify_backward(D):
    from surf_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(D)
    occ = torch.rand(D, D, D, generator=g) < 0.3
    coords = occ.nonzero().to(torch.int32)
    n = coords.shape[0]
    logit = torch.randn(n, generator=g).requires_grad_(True)
    prev = torch.randn(D // 2, D // 2, D // 2, generator=g).requires_grad_(True)
    Gd = torch.randn(D, D, D, generator=g)
    dense, _ = O.sparse2dense(logit, coords, D, prev)
    (dense * Gd).sum().backward()
    rows = torch.randn(n, 8, generator=g)
    _, table = ops.densify(coords.to(d).contiguous(), rows.to(d).contiguous(), D, prev.detach().to(d).contiguous())
    g_rows = torch.zeros(n, 8, device=d)
    g_prev = torch.zeros(D // 2, D // 2, D // 2, device=d)
    ops.densify_backward(coords.to(d).contiguous(), table, Gd.to(d).contiguous(), g_rows, g_prev)
    grad_close(g_rows[:, 0], logit.grad)
    assert float(g_rows[:, 1:].abs().max()) == 0.0
    grad_close(g_prev, prev.grad)
    # stage 0: no background
    g_rows2 = torch.zeros(n, 8, device=d)
    ops.densify_backward(coords.to(d).contiguous(), table, Gd.to(d).contiguous(), g_rows2, None)
    assert torch.equal(g_rows2, g_rows)
    # a mostly-zero dense gradient (what training produces: the gather kernel's workgroups leave when their footprint is zero),
    # accumulated INTO a non-zero g_prev
    Gs = torch.zeros(D, D, D)
    Gs[D // 2:, :2, 1] = Gd[D // 2:, :2, 1]
    Gs[0, 0, 0], Gs[D - 1, D - 1, D - 1] = 1.5, -2.5
    logit.grad, prev.grad = None, None
    dense2, _ = O.sparse2dense(logit, coords, D, prev)
    (dense2 * Gs).sum().backward()
    base = torch.randn(D // 2, D // 2, D // 2, generator=g)
    g_prev3 = base.to(d).clone()
    ops.densify_backward(coords.to(d).contiguous(), table, Gs.to(d).contiguous(), torch.zeros(n, 8, device=d), g_prev3)
    grad_close(g_prev3 - base.to(d), prev.grad)


def test_scatter_rows_add():
    from surf_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(1)
    n_src, n = 300, 1500
    idx = torch.randint(0, n_src * 8, (n,), generator=g).to(torch.int32)
    g_dst = torch.randn(n, 16, generator=g)
    ref = torch.zeros(n_src, 8).index_add_(0, (idx >> 3).long(), g_dst[:, 8:])
    out = ops.scatter_rows_add(g_dst.to(d).contiguous(), idx.to(d), torch.zeros(n_src, 8, device=d), shift=3, dst_off=8)
    grad_close(out, ref, 1e-5)


@pytest.mark.parametrize("stage", [0, 2, 3])
def test_costvol_backward(scene, weights, golden_fpn, golden_pipe, stage):
    """d [mean | var] rows -> the summed FPN levels' maps and agg_mlp (volume.py:54-97 under autograd)."""
    from surf_amd import ops
    d = dev()
    gp = golden_pipe
    D = CFG["base_volume_dim"] * 2 ** stage
    coords = gp[f"s{stage}_coords"].to(torch.int32)
    sd = {k: v.clone().requires_grad_(True) for k, v in weights.items() if k.startswith("volume.agg_mlp")}
    feats = [golden_fpn[f"out{i}"].clone().requires_grad_(True) for i in range(4)]
    cv, _ = O.back_proj_multiscale(sd, feats, coords.float(), D, scene["intrs"], scene["c2ws"], stage)
    g = torch.Generator().manual_seed(stage)
    G = torch.randn(cv.shape, generator=g)
    (cv * G).sum().backward()

    feats_t4 = [ops.pack_texel4(golden_fpn[f"out{i}"].to(d).contiguous()) for i in range(4)]
    gfeats = [torch.zeros_like(f) for f in feats_t4]
    g_agg = torch.zeros(49, device=d)
    ops.costvol_backward(feats_t4, gfeats, stage, D, _cams(scene), ops.agg_mlp_host(weights), coords.to(d).contiguous(),
                         G.to(d).contiguous(), g_agg)
    for l in range(4):
        if l < stage:
            assert feats[l].grad is None and float(gfeats[l].abs().max()) == 0.0
        else:
            grad_close(gfeats[l].permute(0, 3, 1, 2), feats[l].grad)
    ref_agg = torch.cat([sd["volume.agg_mlp.0.weight"].grad.reshape(-1), sd["volume.agg_mlp.0.bias"].grad.reshape(-1),
                         sd["volume.agg_mlp.2.weight"].grad.reshape(-1), sd["volume.agg_mlp.2.bias"].grad.reshape(-1)])
    grad_close(g_agg, ref_agg)
    # round 5: what ran above is the BINNED form (adds sorted by image tile, 64-bit fixed-point LDS images); the direct scatter
    # remains as the fallback of shapes with more than 16,384 (view, tile) buckets - SURF_CVB_DIRECT forces it: same gradients
    import os
    gfeats2 = [torch.zeros_like(f) for f in feats_t4]
    g_agg2 = torch.zeros(49, device=d)
    os.environ["SURF_CVB_DIRECT"] = "1"
    try:
        ops.costvol_backward(feats_t4, gfeats2, stage, D, _cams(scene), ops.agg_mlp_host(weights), coords.to(d).contiguous(),
                             G.to(d).contiguous(), g_agg2)
    finally:
        del os.environ["SURF_CVB_DIRECT"]
    for l in range(stage, 4):
        grad_close(gfeats2[l].permute(0, 3, 1, 2), feats[l].grad)
        scale = float(gfeats2[l].abs().max())
        assert scale > 0 and float((gfeats[l] - gfeats2[l]).abs().max()) <= 2e-5 * scale      # the two forms agree far inside the bar
    grad_close(g_agg2, ref_agg)


def test_fpn_backward_matches_autograd(scene):
    """FeatureNetwork.backward (input gradients on the forward kernels, surf_conv3x3_wgrad, InstanceNorm backward) against
    torch autograd through the oracle's fpn_forward: every convolution weight, for upstream gradients on all four maps."""
    from surf_amd import conf, ops
    from surf_amd.feature_network import FeatureNetwork
    d = dev()
    torch.manual_seed(2)
    net = FeatureNetwork(conf.from_dict({"d_in": 3, "d_base": 8, "d_out": [4, 4, 4, 4]}))
    sd = {"feature_network." + k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    imgs = scene["imgs"][:3, :, :64, :96].contiguous()
    outs_ref = O.fpn_forward(sd, imgs)
    g = torch.Generator().manual_seed(6)
    G = [torch.randn(o.shape, generator=g) for o in outs_ref]
    sum((o * gg).sum() for o, gg in zip(outs_ref, G)).backward()
    net = net.to(d)
    tape = []
    outs = net(imgs.to(d), tape=tape)
    for o, r in zip(outs, outs_ref):
        grad_close(o.permute(0, 3, 1, 2), r.detach(), 1e-3)
    net.backward(tape, [gg.permute(0, 2, 3, 1).contiguous().to(d) for gg in G])
    n = 0
    for name, p_ in net.named_parameters():
        ref = sd["feature_network." + name].grad
        assert ref is not None and p_.grad is not None, name
        grad_close(p_.grad, ref, 5e-3)
        n += 1
    assert n == 8 + 3 + 4


def test_photometric_loss_backward(scene, golden_pipe, golden_train):
    """d compute_ptloss / d depth (losses/photometric_loss.py:54-125 under autograd): smooth-L1, image-gradient and SSIM
    terms through the top-k source selection.  The loss is piecewise smooth (top-k switches, SSIM clamp, bilinear cell
    borders): a small share of pixels may sit on a different piece in fp32, so the check is on the bulk + the total."""
    from surf_amd import ops
    d = dev()
    gt, gp = golden_train, golden_pipe
    imgs_t4 = ops.pack_texel4(scene["imgs"].to(d).contiguous())
    cams = ops.Cameras(scene["intrs"], scene["c2ws"])
    for depth, mask, ref_idx, topk in ((gp["s3_depths"][0], gt["pt_mask_ref"], 0, 2), (gp["s3_depths"][2], gt["pt_mask_src"], 2, 1)):
        dep = depth.clone().requires_grad_(True)
        loss, _, _ = O.photometric_loss(dep, scene["imgs"], mask, scene["intrs"], scene["c2ws"], ref_idx, topk)
        (loss * 3.0).backward()
        g = ops.photometric_loss_backward(depth.to(d).contiguous(), imgs_t4, mask.to(d).contiguous(), cams, ref_idx, topk,
                                          upstream=3.0).cpu()
        ref = dep.grad
        scale = float(ref.abs().max())
        assert scale > 0
        err = (g - ref).abs()
        bad = err > 5e-3 * ref.abs() + 1e-3 * scale
        assert float(bad.float().mean()) < 5e-3, f"{int(bad.sum())}/{bad.numel()} off, max err {err.max().item():.3e} (scale {scale:.3e})"
        assert abs(float(g.sum() - ref.sum())) < 2e-2 * float(ref.abs().sum())


def test_volume_build_backward_end_to_end(scene):
    """SuRF.backward_volumes (matching field -> densify -> sparse U-Net -> cost volume / parent rows -> FPN, four stages)
    against torch autograd through the oracle's fpn_forward + build_volumes (train mode: BatchNorm batch statistics, the
    matching-field jitter on the CPU generator seeded alike), for random upstream gradients on every stage's feature rows
    and on the depth maps of views 0 / src_idx: every parameter of feature_network, volume.agg_mlp and reg_network."""
    from surf_amd import conf
    from surf_amd.surf import SuRF
    from tests.golden.make_golden import MODEL_CONF
    d = dev()
    cfg = {k: v for k, v in MODEL_CONF.items()}
    cfg["reg_network"] = {"d_in": [8, 16, 16, 16], "d_base": [8] * 4, "d_out": [8] * 4}
    # Weight seed: the gradient of a ReLU network is discontinuous where a pre-activation crosses zero, and the comparison is
    # between two fp32 evaluations whose forwards differ by rounding (1e-6).  With seed 1 (rounds 3 - 5) one BatchNorm
    # pre-activation of nets.3.conv9 (804 sites x 16 channels) lies within 2e-6 of zero: with the FPN on the matrix cores
    # (round 6: another summation order, features equal to 2e-6) that ONE mask bit flips and the element's whole upstream
    # gradient (7 % of the tensor's maximum) appears / disappears in everything below it - scripts/dbg_r06_fpn.py traces it op
    # by op.  Seeds 2, 3, 4 have no pre-activation that close to the kink and pass with both FPN paths at the tolerances below;
    # seed 5 fails with both (another such element).  SURF_TEST_VB_SEED overrides.
    torch.manual_seed(int(os.environ.get("SURF_TEST_VB_SEED", "2")))
    model = SuRF(conf.from_dict(cfg))
    with torch.no_grad():
        for net in model.reg_network.nets:
            net.out_lin.weight.mul_(4.0)
    names = [k for k, _ in model.named_parameters() if k.split(".")[0] in ("feature_network", "volume", "reg_network")]
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k in names:
        sd[k].requires_grad_(True)
    src_idx = 1
    H, W = scene["imgs"].shape[-2:]
    nv = scene["imgs"].shape[0]

    # oracle
    torch.manual_seed(77)
    feats_ref = O.fpn_forward(sd, scene["imgs"])
    ocfg = {"range_ratios": cfg["range_ratios"], "base_volume_dim": 8, "n_samples_depths": [128, 64, 32, 16],
            "depth_res_levels": [4, 2, 2, 1]}
    ref = O.build_volumes(sd, scene, feats_ref, ocfg, perturb=True, src_idx=src_idx, training=True)

    # ours
    model = model.to(d).train()
    ipts = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in scene.items()}
    ipts["src_idx"] = src_idx
    torch.manual_seed(77)
    fpn_tape, vol_tape = [], []
    feats = model.feature_network(ipts["imgs"], tape=fpn_tape)
    cams = _cams(scene)
    outputs, volumes, tables, mvol = model.build_volumes(ipts, feats, cams, perturb=True, tape=vol_tape)
    model._train_tape = dict(fpn=fpn_tape, vol=vol_tape, feats=feats, cams=cams, near_fars=ipts["near_fars"], hw=(H, W),
                             src_idx=src_idx)
    for s in range(4):
        assert torch.equal(vol_tape[s]["coords"].cpu().long(), ref["coords"][s].long()), f"stage {s}: voxel sets differ"
        grad_close(volumes[s][:, 1:], ref["volumes"][s].detach(), 2e-3)

    g = torch.Generator().manual_seed(12)
    G_rows = [torch.randn(ref["volumes"][s].shape, generator=g) for s in range(4)]
    G_dep = [(torch.randn(H, W, generator=g) * 3, torch.randn(H, W, generator=g) * 3) for s in range(4)]
    loss = sum((ref["volumes"][s] * G_rows[s]).sum() for s in range(4))
    loss = loss + sum((ref["depths"][s][0] * G_dep[s][0]).sum() + (ref["depths"][s][src_idx] * G_dep[s][1]).sum() for s in range(4))
    loss.backward()

    model.zero_grad(set_to_none=True)
    model.backward_volumes([G_rows[s].to(d) for s in (3, 2, 1, 0)],
                           {s: (G_dep[s][0].to(d), G_dep[s][1].to(d)) for s in range(4)})
    params = dict(model.named_parameters())
    worst = []
    for k in names:
        r = sd[k].grad
        assert r is not None, k
        if params[k].grad is None:          # blocks of an empty U-Net level (stage 0's 8^3 lattice can go empty three levels down)
            assert float(torch.nan_to_num(r).abs().max()) == 0.0, k
            continue
        a, b = params[k].grad.cpu().float(), torch.nan_to_num(r.float())
        scale = max(float(b.abs().max()), 1e-6)
        if k == "volume.agg_mlp.2.bias":
            # the view softmax is shift invariant: this gradient is exactly zero in exact arithmetic (views outside the
            # frustum carry weight exp(-1e9) = 0) and pure rounding noise in fp32 on both sides
            w2 = float(sd["volume.agg_mlp.2.weight"].grad.abs().max())
            assert float(a.abs().max()) < 1e-3 * w2 and float(b.abs().max()) < 1e-3 * w2
            continue
        worst.append((float((a - b).abs().max()) / scale, k))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-2, worst[:5]
    assert sum(1 for w_, _ in worst if w_ > 5e-3) <= len(worst) // 10, worst[:8]


def test_train_steps_move_every_parameter_group(scene):
    """surf_amd.training.train_step on a volume-building model in train mode (runner.py:150-166): a few Adam steps with the
    reference's loss weights lower the loss on a fixed batch, and every parameter group of surf.py:36-45 (implicit surface,
    FPN, sparse U-Net, agg_mlp) receives a finite gradient and moves."""
    from surf_amd import conf
    from surf_amd.losses import Loss
    from surf_amd.surf import SuRF
    from surf_amd.training import train_step
    from tests.golden.make_golden import MODEL_CONF
    from tests.golden.make_golden_train import LOSS_CONF
    d = dev()
    cfg = {k: v for k, v in MODEL_CONF.items()}
    cfg["reg_network"] = {"d_in": [8, 16, 16, 16], "d_base": [8] * 4, "d_out": [8] * 4}
    torch.manual_seed(4)
    model = SuRF(conf.from_dict(cfg))
    with torch.no_grad():
        model.implicit_surface.deviation_network.variance.fill_(0.3)
        for net in model.reg_network.nets:
            net.out_lin.weight.mul_(4.0)
    model = model.to(d).train()
    ipts = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in scene.items()}
    ipts["src_idx"] = 1
    R = scene["rays_o"].shape[0]
    H, W = scene["imgs"].shape[-2:]
    g = torch.Generator().manual_seed(5)
    ipts["pseudo_pts"] = ((torch.rand(300, 3, generator=g) * 2 - 1) * 0.7).to(d)      # -> preds["pseudo_sdf"], pseudo_sdf_loss
    targets = {"color": torch.rand(R, 3, generator=g).to(d), "imgs": ipts["imgs"], "intrs": scene["intrs"], "c2ws": scene["c2ws"],
               "src_idx": 1, "mask_ref": torch.ones(H, W, device=d), "mask_src": torch.ones(H, W, device=d),
               "pseudo_depth_ref": torch.full((H, W), 1.0, device=d), "pseudo_depth_src": torch.full((H, W), 1.0, device=d),
               "depth_ref": torch.full((H, W), 1.0, device=d), "depth_src": torch.full((H, W), 1.0, device=d)}
    opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3}))
    before = {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad}
    loss_fn = Loss(conf.from_dict(LOSS_CONF))
    hist = []
    for step in range(5):
        torch.manual_seed(70)
        out = train_step(model, ipts, targets, loss_fn, opt, 1.0, step + 2)
        hist.append(out["loss"])
        assert out["pseudo_sdf_loss"] > 0
        if step == 0:
            for k, v in model.named_parameters():
                if v.grad is not None:
                    assert bool(torch.isfinite(v.grad).all()), k
    assert all(np.isfinite(hist)), hist
    assert hist[-1] < hist[0], hist
    moved = {grp: 0 for grp in ("feature_network", "reg_network", "volume", "implicit_surface")}
    for k, v in model.named_parameters():
        if v.requires_grad and float((v.detach() - before[k]).abs().max()) > 0:
            moved[k.split(".")[0]] += 1
    assert all(c > 0 for c in moved.values()), moved
    assert moved["feature_network"] == 15 and moved["volume"] >= 3


def test_seven_view_backward_kernels(weights):
    """The Tanks&Temples view count (7 views = 6 sources, SURF_MAX_VIEWS - 1): the view loops of surf_costvol_backward,
    surf_blend_backward (parameters + the feature maps' gradient, two chunks of four source views), surf_ptloss_backward and
    surf_matching_depth_backward against autograd through the oracle on a synthetic ring of cameras."""
    from surf_amd import ops, synthetic
    d = dev()
    nv, H, W = 7, 48, 64
    intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
    g = torch.Generator().manual_seed(8)
    imgs = torch.rand(nv, 3, H, W, generator=g)
    feats_c2f = [torch.randn(nv, 4, H >> (3 - l), W >> (3 - l), generator=g) * 0.5 for l in range(4)]
    cams = ops._cams_ext(ops.Cameras(intrs, c2ws), intrs, c2ws)
    feats_t4 = [ops.pack_texel4(f.to(d).contiguous()) for f in feats_c2f]
    imgs_t4 = ops.pack_texel4(imgs.to(d).contiguous())

    # cost volume, stage 1 (levels 1..3), a random sparse set of voxels
    D, stage = 16, 1
    coords = (torch.rand(D, D, D, generator=g) < 0.2).nonzero().to(torch.int32)
    sd = {k: v.clone().requires_grad_(True) for k, v in weights.items() if k.startswith("volume.agg_mlp")}
    fr = [f.clone().requires_grad_(True) for f in feats_c2f]
    cv, _ = O.back_proj_multiscale(sd, fr, coords.float(), D, intrs, c2ws, stage)
    G = torch.randn(cv.shape, generator=g)
    (cv * G).sum().backward()
    gfeats = [torch.zeros_like(f) for f in feats_t4]
    g_agg = torch.zeros(49, device=d)
    ops.costvol_backward(feats_t4, gfeats, stage, D, cams, ops.agg_mlp_host(weights), coords.to(d).contiguous(), G.to(d).contiguous(), g_agg)
    for l in range(1, 4):
        grad_close(gfeats[l].permute(0, 3, 1, 2), fr[l].grad)
    grad_close(g_agg[:48], torch.cat([sd[f"volume.agg_mlp.{k}"].grad.reshape(-1) for k in ("0.weight", "0.bias", "2.weight")]))

    # blending network: parameters and the sampled feature maps (fine -> coarse)
    pts = (torch.rand(600, 3, generator=g) * 2 - 1) * 0.6
    gcolor = torch.randn(600, 3, generator=g)
    prefix = "implicit_surface.color_network."
    sdc = {k: v.clone().requires_grad_(True) for k, v in weights.items() if k.startswith(prefix)}
    f2c = [f.clone().requires_grad_(True) for f in feats_c2f[::-1]]
    rf, rdiff, mval = O.lookup_feature(pts, imgs, intrs, c2ws, f2c)
    assert int(mval.sum()) > 0
    (O.blending(sdc, rf, rdiff, mval) * gcolor).sum().backward()
    raw = torch.from_numpy(ops.blend_raw_weights(weights)).to(d)
    gf = [torch.zeros_like(f) for f in feats_t4[::-1]]
    res = ops.blend_backward(pts.to(d).contiguous(), None, gcolor.to(d).contiguous(), feats_t4[::-1], imgs_t4, cams, raw, gfeats_t4=gf)
    for l in range(4):
        grad_close(gf[l].permute(0, 3, 1, 2), f2c[l].grad, 5e-3)
    for k, v in sdc.items():
        name = k[len(prefix):]
        if name == "s":
            continue                                    # ill-conditioned in fp32 (test_blend_backward_matches_autograd)
        ref = v.grad if v.grad is not None else torch.zeros_like(v)
        grad_close(res[name].reshape(ref.shape), ref, 5e-3, 1e-2)

    # photometric term with six sources
    depth = torch.full((H, W), 2.4) + torch.rand(H, W, generator=g) * 0.2
    mask = (torch.rand(H, W, generator=g) > 0.1).float()
    dep = depth.clone().requires_grad_(True)
    loss, _, _ = O.photometric_loss(dep, imgs, mask, intrs, c2ws, 3, 2)
    loss.backward()
    gd = ops.photometric_loss_backward(depth.to(d).contiguous(), imgs_t4, mask.to(d).contiguous(), ops.Cameras(intrs, c2ws), 3, 2, 1.0).cpu()
    scale = float(dep.grad.abs().max())
    bad = (gd - dep.grad).abs() > 5e-3 * dep.grad.abs() + 1e-3 * scale
    assert float(bad.float().mean()) < 1e-2, int(bad.sum())

    # matching field, stage 0, views 0 and 5
    mvol = torch.randn(16, 16, 16, generator=g)
    mv = mvol.clone().requires_grad_(True)
    ratios, nsd, lv = [1.0, 0.4, 0.1, 0.01], [32, 16, 8, 8], [4, 2, 2, 1]
    ref = O.matching_field((H, W), intrs, c2ws, near_fars, mv, 0, ratios, nsd, lv, None)
    Gd = torch.zeros(nv, H, W)
    Gd[0], Gd[5] = torch.randn(H, W, generator=g), torch.randn(H, W, generator=g)
    (torch.stack(ref) * Gd).sum().backward()
    dm = ops.matching_depth_backward(mvol.to(d).contiguous(), cams, near_fars, H, W, lv[0], nsd[0], Gd.to(d).contiguous(), views=(0, 5))
    grad_close(dm, mv.grad)


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("C,n", [(8, 1000), (64, 777), (16, 1)])
def test_bn_relu_backward_both_modes(train, C, n):
    """surf_bn_relu_backward against torch autograd of BatchNorm1d + ReLU (+ skip): batch statistics (train) and running
    statistics (eval), every channel count of the U-Net, ragged and single-row inputs."""
    from surf_amd import ops
    d = dev()
    if train and n == 1:
        pytest.skip("batch statistics of one row: the variance is zero and torch refuses it")
    g = torch.Generator().manual_seed(C + n)
    x = torch.randn(n, C, generator=g).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g)
        bn.bias.uniform_(-0.3, 0.3, generator=g)
        bn.running_mean.uniform_(-0.2, 0.2, generator=g)
        bn.running_var.uniform_(0.5, 2.0, generator=g)
    bn.train(train)
    if train:
        mean, var = x.detach().mean(0), x.detach().var(0, unbiased=False)
    else:
        mean, var = bn.running_mean.clone(), bn.running_var.clone()
    skip = torch.randn(n, C, generator=g).requires_grad_(True)
    dy = torch.randn(n, C, generator=g)
    y = torch.relu(bn(x)) + skip
    (y * dy).sum().backward()
    invstd = 1.0 / torch.sqrt(var + bn.eps)
    scale = (bn.weight.detach() * invstd).contiguous()
    shift = (bn.bias.detach() - mean * scale).contiguous()
    stats = torch.cat([mean, invstd]).contiguous()
    dx, dgamma, dbeta = ops.bn_relu_backward(x.detach().to(d).contiguous(), dy.to(d).contiguous(), scale.to(d), shift.to(d), stats.to(d),
                                            train=train)
    grad_close(dx, x.grad, 2e-3)
    grad_close(dgamma, bn.weight.grad, 2e-3)
    grad_close(dbeta, bn.bias.grad, 2e-3)
    assert torch.equal(skip.grad, dy)                      # the skip's gradient is dy itself (the caller adds it)


def test_train_step_seven_views():
    """The whole training step at the Tanks&Temples view count (7 views) on a small synthetic scene: every stage's tapes,
    the jitter shapes, the view loops of every backward kernel; finite loss and gradients, parameters move."""
    from bench import training_step_setup
    from surf_amd import training
    d = dev()
    model, ipts, targets, loss_fn, opt = training_step_setup(d, H=96, W=128, nv=7, base_dim=16, rays=128)
    before = {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad}
    for step in range(2):
        out = training.train_step(model, ipts, targets, loss_fn, opt, 1.0, step + 3)
        assert np.isfinite(out["loss"]), out
        for k, v in model.named_parameters():
            if v.grad is not None:
                assert bool(torch.isfinite(v.grad).all()), k
    assert len(model.last_voxels_per_stage) == 4 and min(model.last_voxels_per_stage) > 0
    moved = sum(1 for k, v in model.named_parameters() if v.requires_grad and float((v.detach() - before[k]).abs().max()) > 0)
    assert moved > 100


# ---- against the REFERENCE's own autograd (tests/golden/volume_grads.npz, make_golden_grad.py) ---------------------------------
def test_fpn_backward_equals_the_reference_autograd(scene, weights, golden_vgrads):
    from surf_amd import conf
    from surf_amd.feature_network import FeatureNetwork
    d = dev()
    gv = golden_vgrads
    net = FeatureNetwork(conf.from_dict({"d_in": 3, "d_base": 8, "d_out": [4, 4, 4, 4]}))
    net.load_state_dict({k[len("feature_network."):]: v for k, v in weights.items() if k.startswith("feature_network.")})
    net = net.to(d)
    tape = []
    net(scene["imgs"].to(d), tape=tape)
    net.backward(tape, [gv[f"fpn_up{i}"].permute(0, 2, 3, 1).contiguous().to(d) for i in range(4)])
    n = 0
    for name, p_ in net.named_parameters():
        grad_close(p_.grad, gv["fpn_grad/" + name], 5e-3)
        n += 1
    assert n == 15


@pytest.mark.parametrize("stage", [0, 2])
def test_costvol_backward_equals_the_reference_autograd(scene, weights, golden_fpn, golden_pipe, golden_vgrads, stage):
    from surf_amd import ops
    d = dev()
    gv = golden_vgrads
    D = CFG["base_volume_dim"] * 2 ** stage
    coords = golden_pipe[f"s{stage}_coords"].to(torch.int32).to(d).contiguous()
    feats_t4 = [ops.pack_texel4(golden_fpn[f"out{i}"].to(d).contiguous()) for i in range(4)]
    gfeats = [torch.zeros_like(f) for f in feats_t4]
    g_agg = torch.zeros(49, device=d)
    ops.costvol_backward(feats_t4, gfeats, stage, D, _cams(scene), ops.agg_mlp_host(weights), coords, gv[f"cv{stage}_up"].to(d).contiguous(), g_agg)
    for l in range(4):
        ref = gv[f"cv{stage}_gfeat{l}"]
        if l < stage:
            assert float(ref.abs().max()) == 0.0 and float(gfeats[l].abs().max()) == 0.0
        else:
            grad_close(gfeats[l].permute(0, 3, 1, 2), ref)
    ref_agg = torch.cat([gv[f"cv{stage}_grad/agg_mlp.{k}"].reshape(-1) for k in ("0.weight", "0.bias", "2.weight")])
    grad_close(g_agg[:48], ref_agg)


def test_densify_matching_photometric_backward_equal_the_reference_autograd(scene, golden_pipe, golden_vgrads):
    from surf_amd import conf, ops
    from surf_amd.matching_field import MatchingField
    d = dev()
    gv, gp = golden_vgrads, golden_pipe
    # sparse2dense, stage 1
    D = CFG["base_volume_dim"] * 2
    coords = gp["s1_coords"].to(torch.int32).to(d).contiguous()
    rows = gp["s1_reg_out"].to(d).contiguous()
    _, table = ops.densify(coords, rows, D, gp["s0_mvol"].to(d).contiguous())
    g_rows = torch.zeros_like(rows)
    g_prev = torch.zeros(D // 2, D // 2, D // 2, device=d)
    ops.densify_backward(coords, table, gv["s2d_up"].to(d).contiguous(), g_rows, g_prev)
    grad_close(g_rows[:, 0], gv["s2d_glogit"])
    grad_close(g_prev, gv["s2d_gprev"])
    # matching field, stage 1, jitter as the reference drew it (CPU generator, seed 31), views 0 and 2
    H, W = scene["imgs"].shape[-2:]
    nv = scene["intrs"].shape[0]
    mf = MatchingField(conf.from_dict({"n_samples_depths": CFG["n_samples_depths"], "n_importance_depths": [0] * 4,
                                       "up_sample_steps": [0] * 4, "depth_res_levels": CFG["depth_res_levels"]}))
    lvl = CFG["depth_res_levels"][1]
    torch.manual_seed(31)
    jitter = mf.draw_jitter(nv, (H // lvl) * (W // lvl), 2, 2).to(d).contiguous()
    G = torch.zeros(nv, H, W)
    G[0], G[2] = gv["mf_up0"], gv["mf_up2"]
    dm = ops.matching_depth_backward(gp["s1_mvol"].to(d).contiguous(), _cams(scene), scene["near_fars"], H, W, lvl,
                                     CFG["n_samples_depths"][1], G.to(d).contiguous(), gp["s0_depths"].to(d).contiguous(),
                                     CFG["range_ratios"][1], CFG["range_ratios"][0], jitter=jitter, views=(0, 2))
    grad_close(dm, gv["mf_gmvol"])
    # photometric term
    imgs_t4 = ops.pack_texel4(scene["imgs"].to(d).contiguous())
    gd = ops.photometric_loss_backward(gp["s3_depths"][0].to(d).contiguous(), imgs_t4, gv["pt_mask"].to(d).contiguous(),
                                       ops.Cameras(scene["intrs"], scene["c2ws"]), 0, 2, 1.0).cpu()
    ref = gv["pt_gdepth"]
    scale = float(ref.abs().max())
    bad = (gd - ref).abs() > 5e-3 * ref.abs() + 1e-3 * scale
    assert float(bad.float().mean()) < 5e-3, int(bad.sum())
