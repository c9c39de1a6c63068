"""GPU: the multi-stream backward sweep (surf_amd.ops.SideStream: U-Net kernel gradients, the colour / smooth / SDF branches of the
render backward and the matching chain on their own HIP streams, the matching chain launched from the graph's depth tap) computes
what the in-order sweep computes.  Two optimiser steps of the reference's runner sequence (runner.py:152-165) in fresh
processes with SURF_SIDE_STREAM = 0 | 1 | all; the step-1 gradients of every parameter are compared tensor by tensor, bounded by
the spread of two in-order runs (the step has float atomics; see test_rccl_world1._grad_gap)."""
import os

import pytest
import torch

from tests.test_rccl_world1 import TRAIN, _env, _grad_gap, _worker   # the same worker script, mode "nogroup"

pytestmark = pytest.mark.gpu


def _run(tmp_path, monkeypatch, side):
    monkeypatch.setenv("SURF_SIDE_STREAM", side)
    return _worker(tmp_path, "train.py", TRAIN, "nogroup")


def test_multi_stream_backward_equals_the_in_order_sweep(tmp_path, monkeypatch):
    a = _run(tmp_path, monkeypatch, "0")
    b = _run(tmp_path, monkeypatch, "0")
    spread = _grad_gap(a, b)
    tol = max(10.0 * spread[0], 1e-4)          # fp32 atomics in another order: 1e-6 .. 3e-5 observed; a wrong gradient is >> 1e-3
    assert tol < 1e-2, ("the in-order step itself is not reproducible", spread)
    for side in ("1", "all"):
        s = _run(tmp_path, monkeypatch, side)
        gap = _grad_gap(s, a)
        print(f"SURF_SIDE_STREAM={side}: step-1 gradient gap {gap} (in-order spread {spread})")
        assert gap[0] <= tol, (side, gap, spread)
        assert s["losses"] == pytest.approx(a["losses"], rel=1e-4), (side, s["losses"], a["losses"])


def test_side_stream_switch_and_lanes():
    """The switch's users and the lane bookkeeping (fork / keep / join) on the device."""
    from surf_amd import ops
    s = ops.SideStream()
    assert (s.enabled and {"unet", "render", "match"} <= s.users) or "SURF_SIDE_STREAM" in os.environ
    dev = torch.device("cuda", 0)
    x = torch.ones(1 << 20, device=dev)
    out = s.run(lambda: x * 2.0, lane=1, keep=(x,))
    y = s.run(lambda: x + 1.0, lane=0)
    assert {k[1] for k in s._open} == {0, 1} and s._keep
    s.join(lanes=(1,))
    assert {k[1] for k in s._open} == {0}
    s.join()
    assert not s._open and not s._keep
    assert float(out.sum()) == 2.0 * (1 << 20) and float(y.sum()) == 2.0 * (1 << 20)
    with s.fork(lane=2):
        z = x * 3.0
        ev = s.mark()
    s.wait_for(ev)
    assert float(z.sum()) == 3.0 * (1 << 20)
    s.join()


def test_photometric_terms_as_one_node_equal_the_single_nodes():
    """autograd.photometric_losses (one graph node, the 2 n launches dealt out over side streams both ways) against
    autograd.photometric_loss per map: values bit-equal, gradients equal up to the order of the backward's float atomics."""
    from surf_amd import autograd, ops
    from tests.conftest import load_npz
    scene = load_npz("scene.npz")
    d = torch.device("cuda", 0)
    imgs_t4 = ops.pack_texel4(scene["imgs"].to(d).contiguous())
    cams = ops.Cameras(scene["intrs"], scene["c2ws"])
    H, W = imgs_t4.shape[1:3]
    g = torch.Generator().manual_seed(3)
    masks = [(torch.rand(H, W, generator=g) < 0.9).float().to(d) for _ in range(2)]
    specs, depths_a, depths_b = [], [], []
    for i in range(6):
        base = (2.0 + 0.3 * torch.rand(H, W, generator=g)).to(d)
        depths_a.append(base.clone().requires_grad_(True))
        depths_b.append(base.clone().requires_grad_(True))
        specs.append((masks[i % 2], 0, 2) if i < 3 else (masks[i % 2], 1, 1))
    w = [0.3 + 0.1 * i for i in range(6)]
    multi = autograd.photometric_losses(depths_a, imgs_t4, cams, specs)
    single = [autograd.photometric_loss(x, imgs_t4, m, cams, ref_idx=r, topk=k) for x, (m, r, k) in zip(depths_b, specs)]
    for a, b in zip(multi, single):
        assert float(a.detach()) == float(b.detach())
    sum(wi * a for wi, a in zip(w, multi[:5])).backward()          # the sixth map gets no gradient: None, not zeros
    sum(wi * b for wi, b in zip(w, single[:5])).backward()
    assert depths_a[5].grad is None and depths_b[5].grad is None
    for xa, xb in zip(depths_a[:5], depths_b[:5]):
        assert float(xb.grad.abs().max()) > 0
        assert float((xa.grad - xb.grad).abs().max()) <= 1e-5 * float(xb.grad.abs().max())


def test_matching_depth_sample_per_lane_form_equals_the_corner_per_lane_forms(monkeypatch):
    """Round 6: matching_depth_spl_kernel (one lane = one sample, segmented wave reduction of the softmax triples) against the
    corner-per-lane kernels it replaced (SURF_MD_FORM=0), on the golden pipeline's four stage shapes (128 / 2x64 / 2x32 / 2x16
    samples per ray: both lane counts, one and two passes per lane), with and without the train-mode jitter; the per-ray
    statistics handed to the backward agree too.  Same sums in another order: 1e-5 of the depth range."""
    from surf_amd import ops
    from tests.conftest import load_npz
    from tests.golden.make_golden import MODEL_CONF as CFG
    scene = load_npz("scene.npz")
    gp = load_npz("pipeline.npz")
    d = torch.device("cuda", 0)
    cams = ops._cams_ext(ops.Cameras(scene["intrs"], scene["c2ws"]), scene["intrs"], scene["c2ws"])
    H, W = scene["imgs"].shape[-2:]
    mf = CFG["matching_field"]
    nv = scene["imgs"].shape[0]
    g = torch.Generator().manual_seed(11)
    for s in range(4):
        mvol = gp[f"s{s}_mvol"].to(d).contiguous()
        pre = gp[f"s{s - 1}_depths"].to(d).contiguous() if s > 0 else None
        lvl, n = mf["depth_res_levels"][s], mf["n_samples_depths"][s]
        h, w = H // lvl, W // lvl
        for jit in (None, (torch.rand(nv, h * w, 2, generator=g) - 0.5).to(d).contiguous()):
            res = {}
            for form in ("0", "1"):
                monkeypatch.setenv("SURF_MD_FORM", form)
                saved = {}
                full, lr = ops.matching_depth(mvol, cams, scene["near_fars"], H, W, lvl, n, pre, CFG["range_ratios"][s],
                                              CFG["range_ratios"][s - 1] if s > 0 else 1.0, return_lr=True, jitter=jit, saved=saved)
                res[form] = (full, lr, saved["stats"])
            span = float(res["0"][1].max() - res["0"][1].min()) + 1e-6
            for a, b in zip(res["0"][:2], res["1"][:2]):
                assert torch.isfinite(b).all()
                assert float((a - b).abs().max()) <= 1e-5 * max(span, 1.0), (s, jit is not None)
            sa, sb = res["0"][2], res["1"][2]
            assert float((sa[..., 2] - sb[..., 2]).abs().max()) <= 1e-5 * max(span, 1.0)        # expected z
            # max logit (the corner sums of a sample are formed in another order) and softmax denominator: relative
            assert float(((sa[..., 0] - sb[..., 0]).abs() / sb[..., 0].abs().clamp_min(1.0)).max()) <= 1e-5
            assert float(((sa[..., 1] - sb[..., 1]).abs() / sb[..., 1]).max()) <= 1e-4


def _small_training_setup():
    from surf_amd import conf
    from surf_amd.surf import SuRF
    from tests.conftest import load_npz
    from tests.golden.make_golden import MODEL_CONF
    dev = torch.device("cuda", 0)
    scene = load_npz("scene.npz")
    cfg = dict(MODEL_CONF)
    cfg["reg_network"] = {"d_in": [8, 16, 16, 16], "d_base": [8] * 4, "d_out": [8] * 4}
    torch.manual_seed(4)
    model = SuRF(conf.from_dict(cfg))
    with torch.no_grad():
        model.implicit_surface.deviation_network.variance.fill_(0.3)
        for net in model.reg_network.nets:
            net.out_lin.weight.mul_(4.0)
    model = model.to(dev).train()
    ipts = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
    ipts["src_idx"] = 1
    return model, ipts


@pytest.mark.parametrize("which", ["colour", "depth"])
def test_partial_losses_on_the_multi_stream_sweep(which, monkeypatch):
    """A loss that reaches only ONE of the two graph nodes: colour alone (the depth tap sees no gradient: the matching chain has
    nothing to do, the render backward's three branches still fork and join) or the depth maps alone (the tap launches the
    matching chain, the render node never runs, the volume backward gets no row gradients).  Multi-stream == in-order, and the
    parameters the loss cannot reach get no gradient either way."""
    from surf_amd import ops
    grads = {}
    for side in (False, True):
        monkeypatch.setattr(ops.side, "enabled", side)
        model, ipts = _small_training_setup()
        torch.manual_seed(70)
        out = model("train", ipts, cos_anneal_ratio=1.0, step=3)
        if which == "colour":
            loss = (out["color_fine"] * torch.linspace(0.5, 1.5, 3, device=out["color_fine"].device)).sum()
        else:
            loss = sum((0.3 + 0.1 * s) * out[f"depth_stage{s}"].mean() + 0.2 * out[f"depth_src_stage{s}"].mean() for s in range(4))
        loss.backward()
        torch.cuda.synchronize()
        grads[side] = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in model.named_parameters()}
    a, b = grads[False], grads[True]
    top = max(float(g.abs().max()) for g in a.values() if g is not None)
    assert top > 0
    reached = 0
    for n in a:
        assert (a[n] is None) == (b[n] is None), n
        if a[n] is None:
            continue
        assert torch.isfinite(b[n]).all(), n
        reached += int(float(a[n].abs().max()) > 0)
        # (volume.agg_mlp.2.bias shifts every softmax logit alike: its gradient is rounding noise, 1e-9 of the largest one)
        assert float((a[n] - b[n]).abs().max()) <= 1e-4 * float(a[n].abs().max()) + 1e-6 * top, n
    assert reached > 10
    if which == "depth":        # the implicit surface is not on the path of the depth maps
        assert all(g is None or float(g.abs().max()) == 0.0 for n, g in b.items() if n.startswith("implicit_surface."))
