"""GPU: the backward kernels of the volume build / FPN at the bench scene's full sizes (5 views 576x800, 88^3 -> 704^3, up to
5 M voxels per stage), through size-independent identities (the oracle's autograd does not finish at these sizes):

* a convolution is linear in its input and in its kernel:  <dy, conv(x; W)> = <dx, x> = <dW, W>  (sparse, all three modes; FPN
  3x3 stride 1 / 2 / transposed);
* BatchNorm(batch statistics) backward:  sum dx = 0 and sum dx xhat = 0 per channel;
* densify backward is a partition of the dense gradient:  sum g_rows + sum g_prev = sum g_dense  (upsample weights sum to 1);
* the matching-field depth is invariant to a constant shift of the logits along a ray:  sum dmvol = 0 when no tap leaves the
  volume; the cost volume's view softmax likewise:  d / d agg_mlp.2.bias = 0;
* the photometric loss does not depend on depths outside the reference mask.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _close(a, b, rel=2e-3):
    assert abs(a - b) <= rel * max(abs(a), abs(b), 1e-12), (a, b)


@pytest.fixture(scope="module")
def full_train():
    """One recorded train-mode forward of the volume-building model on the bench scene."""
    from bench import training_step_setup
    dev = torch.device("cuda:0")
    model, ipts, targets, loss_fn, opt = training_step_setup(dev)
    torch.manual_seed(5)
    model("train", ipts, 1.0, 3, record=True)
    torch.cuda.synchronize()
    return dict(model=model, ipts=ipts, targets=targets, tape=model._train_tape, dev=dev)


@pytest.mark.parametrize("stage", [0, 2])
def test_sparse_conv_backward_identities_full_size(full_train, stage):
    """Every block of the stage's U-Net (SUBM / DOWN / UP, 8..64 channels, up to 5 M rows): <dy, y> = <dx, x> = <dW, W>."""
    from surf_amd import ops
    tape = full_train["tape"]["vol"][stage]["reg_tape"]
    g = torch.Generator(device="cuda").manual_seed(stage)
    seen = set()
    for e in tape[:-1]:
        n_out = e["raw"].shape[0]
        key = (e["mode"], e["w"].shape[1], e["w"].shape[2])
        if key in seen or n_out == 0:
            continue
        seen.add(key)
        dy = torch.randn(e["raw"].shape, device="cuda", generator=g)
        dx, dW = ops.spconv_backward(e["x"], e["in_site"][0], e["in_site"][1], e["out_site"][0], e["out_site"][1], e["mode"], e["w"], dy)
        ref = _dot(dy, e["raw"])                       # raw = conv(x; W) of the forward
        # <dx, x> is a sum of millions of signed terms that cancels to ~1e-5 of their total magnitude; the wide layers' dx comes
        # from the matrix-core kernel (fp32 accumulation of up to 27 x 64 products in another order: 5e-6 absolute per element
        # against the per-voxel kernel, scripts/dbg_spconv_dgrad.py), so the identity is held to 2e-8 of the UN-cancelled sum
        slack = 2e-8 * float((dx.double().abs() * e["x"].double().abs()).sum())
        got = _dot(dx, e["x"])
        assert abs(got - ref) <= 2e-3 * max(abs(got), abs(ref), 1e-12) + slack, (got, ref, slack)
        _close(_dot(dW, e["w"]), ref)
    assert len(seen) >= 8


def test_batchnorm_backward_identities_full_size(full_train):
    from surf_amd import ops
    tape = full_train["tape"]["vol"][2]["reg_tape"]
    e = tape[0]                                        # conv0 of stage 2: ~5 M rows x 8 channels
    assert e["raw"].shape[0] > 4_000_000
    g = torch.Generator(device="cuda").manual_seed(1)
    dy = torch.randn(e["raw"].shape, device="cuda", generator=g)
    dx, dgamma, dbeta = ops.bn_relu_backward(e["raw"], dy, e["scale"], e["shift"], e["stats"], train=True)
    C = e["raw"].shape[1]
    mean, invstd = e["stats"][:C], e["stats"][C:]
    xhat = (e["raw"] - mean) * invstd
    scale = float(dx.abs().double().sum())
    assert float(dx.double().sum(0).abs().max()) < 1e-5 * scale
    assert float((dx.double() * xhat.double()).sum(0).abs().max()) < 1e-5 * scale
    z = e["raw"] * e["scale"] + e["shift"]
    zb = dy * (z > 0)
    assert torch.allclose(dbeta.double(), zb.double().sum(0), rtol=1e-4, atol=1e-3)
    assert torch.allclose(dgamma.double(), (zb.double() * xhat.double()).sum(0), rtol=1e-4, atol=1e-3)


def test_densify_backward_partition_full_size(full_train):
    from surf_amd import ops
    r = full_train["tape"]["vol"][3]                   # 704^3
    D = r["D"]
    assert D == 704
    g = torch.Generator(device="cuda").manual_seed(2)
    g_dense = torch.randn(D, D, D, device="cuda", generator=g)
    g_rows = torch.zeros(r["coords"].shape[0], 8, device="cuda")
    g_prev = torch.zeros(D // 2, D // 2, D // 2, device="cuda")
    ops.densify_backward(r["coords"], r["table"], g_dense, g_rows, g_prev)
    c = r["coords"].long()
    assert torch.equal(g_rows[:, 0], g_dense[c[:, 0], c[:, 1], c[:, 2]])
    free = r["table"] < 0
    total = float(g_dense[free].double().sum())
    got = float(g_prev.double().sum())
    assert abs(total - got) < 1e-6 * float(g_dense[free].abs().double().sum())


@pytest.mark.parametrize("stage", [1, 3])
def test_matching_field_backward_shift_invariance_full_size(full_train, stage):
    t = full_train["tape"]
    r = t["vol"][stage]
    model = full_train["model"]
    H, W = t["hw"]
    nv = t["feats"][0].shape[0]
    g = torch.Generator(device="cuda").manual_seed(3)
    g_full = torch.zeros(nv, H, W, device="cuda")
    g_full[0] = torch.randn(H, W, device="cuda", generator=g)
    g_full[t["src_idx"]] = torch.randn(H, W, device="cuda", generator=g)
    dm = model.matching_field.backward(t["cams"], t["near_fars"], (H, W), r["mvol"], stage, model.range_ratios, g_full,
                                       r["pre_depths"], r.get("jitter"), src_idx=t["src_idx"])
    assert float(dm.abs().max()) > 0
    # softmax over a ray's samples: shifting every logit by a constant leaves the depth unchanged -> the taps' gradients sum to
    # zero ray by ray, hence in total (up to the rays whose samples leave the [-1,1]^3 lattice: zero padding)
    assert abs(float(dm.double().sum())) < 2e-2 * float(dm.abs().double().sum())
    assert bool(torch.isfinite(dm).all())


def test_cost_volume_backward_full_size(full_train):
    from surf_amd import ops
    t = full_train["tape"]
    model = full_train["model"]
    r = t["vol"][2]
    n = r["coords"].shape[0]
    g = torch.Generator(device="cuda").manual_seed(4)
    G = torch.randn(n, 8, device="cuda", generator=g)
    gfeats = [torch.zeros_like(f) for f in t["feats"]]
    g_agg = torch.zeros(49, device="cuda")
    ops.costvol_backward(t["feats"], gfeats, 2, r["D"], t["cams"], model.volume.agg_host(), r["coords"], G, g_agg)
    assert float(gfeats[0].abs().max()) == 0.0 and float(gfeats[1].abs().max()) == 0.0     # stage 2 sums levels 2, 3 only
    assert float(gfeats[2].abs().max()) > 0 and float(gfeats[3].abs().max()) > 0
    assert bool(torch.isfinite(g_agg).all()) and all(bool(torch.isfinite(f).all()) for f in gfeats)
    assert abs(float(g_agg[48])) < 1e-3 * float(g_agg[40:48].abs().max())                  # shift invariance of the view softmax
    # linearity in the upstream gradient
    gfeats2 = [torch.zeros_like(f) for f in t["feats"]]
    g_agg2 = torch.zeros(49, device="cuda")
    ops.costvol_backward(t["feats"], gfeats2, 2, r["D"], t["cams"], model.volume.agg_host(), r["coords"], (G * 2.0).contiguous(), g_agg2)
    assert torch.allclose(gfeats2[3], gfeats[3] * 2.0, rtol=1e-3, atol=1e-3 * float(gfeats[3].abs().max()))
    assert torch.allclose(g_agg2[:48], g_agg[:48] * 2.0, rtol=1e-3, atol=1e-3 * float(g_agg[:48].abs().max()))


def test_fpn_conv_backward_identities_full_size(full_train):
    """FPN layers at 5 x 576 x 800: stride-1, stride-2 and transposed 3x3 convolutions, <dy, y> = <dx, x> = <dW, W>."""
    from surf_amd import ops
    t = full_train["tape"]["fpn"][-1]
    g = torch.Generator(device="cuda").manual_seed(6)
    for k in (1, 2, 7):                                  # 8->8 s1 (finest), 8->16 s2, 64->64 s1 (coarsest)
        r = t["enc"][k]
        w = r["blk"].conv.weight.detach().float()
        dy = torch.randn(r["raw"].shape, device="cuda", generator=g)
        ref = _dot(dy, r["raw"])
        dW = ops.conv3x3_wgrad(r["x_in"], dy, r["blk"].stride)                              # [ky][kx][ci][co]
        _close(_dot(dW.permute(3, 2, 0, 1), w), ref)
        if r["blk"].stride == 1:
            dx = ops.conv3x3(dy, w.flip(2, 3).permute(2, 3, 0, 1).contiguous(), w.shape[1], 1)
        else:
            dx = ops.deconv3x3_s2(dy, w.permute(2, 3, 0, 1).contiguous(), w.shape[1])
        _close(_dot(dx, r["x_in"]), ref)
    r = t["dec"][0]                                       # transposed 16 -> 8 onto the finest level
    w = r["blk"].conv.weight.detach().float()             # (Cin, Cout, 3, 3)
    dy = torch.randn(r["raw"].shape, device="cuda", generator=g)
    ref = _dot(dy, r["raw"])
    dW = ops.conv3x3_wgrad(dy, r["x_in"], 2)              # [ky][kx][co][ci]
    _close(_dot(dW.permute(3, 2, 0, 1), w), ref)
    dx = ops.conv3x3(dy, w.permute(2, 3, 1, 0).contiguous(), w.shape[0], 2)
    _close(_dot(dx, r["x_in"]), ref)


def test_photometric_backward_respects_the_mask_full_size(full_train):
    from surf_amd import ops
    t = full_train["tape"]
    ipts, tg = full_train["ipts"], full_train["targets"]
    H, W = t["hw"]
    imgs_t4 = ops.pack_texel4(ipts["imgs"].float().contiguous())
    cams = ops.Cameras(tg["intrs"], tg["c2ws"])
    depth = torch.full((H, W), 2.0, device="cuda")
    mask = torch.zeros(H, W, device="cuda")
    mask[100:400, 200:600] = 1.0
    gd = ops.photometric_loss_backward(depth, imgs_t4, mask, cams, 0, 2, 1.0)
    assert bool(torch.isfinite(gd).all()) and float(gd.abs().max()) > 0
    outside = torch.ones(H, W, dtype=torch.bool, device="cuda")
    outside[98:402, 198:602] = False                      # the SSIM / gradient windows reach one pixel past the mask
    assert float(gd[outside].abs().max()) == 0.0
