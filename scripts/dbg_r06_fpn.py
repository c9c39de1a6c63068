"""Debug: the e2e volume backward with the FPN on the VALU kernels vs on the matrix cores - where do they part?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import conf, ops
from surf_amd.surf import SuRF
from tests.golden.make_golden import MODEL_CONF
from tests.conftest import load_npz
from tests.test_volume_backward import _cams
d = torch.device("cuda:0")
scene = load_npz("scene.npz")
cfg = {k: v for k, v in MODEL_CONF.items()}
cfg["reg_network"] = {"d_in": [8, 16, 16, 16], "d_base": [8] * 4, "d_out": [8] * 4}
H, W = scene["imgs"].shape[-2:]
print("image", H, W, "views", scene["imgs"].shape[0])

def run(valu):
    if valu: os.environ["SURF_FPN_VALU"] = "1"
    else: os.environ.pop("SURF_FPN_VALU", None)
    torch.manual_seed(1)
    model = SuRF(conf.from_dict(cfg))
    with torch.no_grad():
        for net in model.reg_network.nets:
            net.out_lin.weight.mul_(4.0)
    model = model.to(d).train()
    ipts = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in scene.items()}
    ipts["src_idx"] = 1
    torch.manual_seed(77)
    fpn_tape, vol_tape = [], []
    feats = model.feature_network(ipts["imgs"], tape=fpn_tape)
    cams = _cams(scene)
    outputs, volumes, tables, mvol = model.build_volumes(ipts, feats, cams, perturb=True, tape=vol_tape)
    model._train_tape = dict(fpn=fpn_tape, vol=vol_tape, feats=feats, cams=cams, near_fars=ipts["near_fars"], hw=(H, W), src_idx=1)
    reg3 = vol_tape[3]["reg_tape"]
    blocks = [(e["raw"].clone(), e["y"].clone(), int(e["raw"].shape[0])) for e in reg3[:-1]]
    g = torch.Generator().manual_seed(12)
    G_rows = [torch.randn(volumes[s].shape[0], 7, generator=g) for s in range(4)]
    G_dep = [(torch.randn(H, W, generator=g) * 3, torch.randn(H, W, generator=g) * 3) for s in range(4)]
    rec = {}
    net3 = model.reg_network.nets[3]
    orig = net3.backward
    def spy(tape, g_out, d_mid=None, sink=None):
        rec["g_out"] = g_out.clone()
        rec["d_mid"] = None if d_mid is None else d_mid.clone()
        seq = rec["seq"] = []
        o_bn, o_sp = ops.bn_relu_backward, ops.spconv_backward
        def bn(x, dy, scale, shift, stats, train=True):
            r = o_bn(x, dy, scale, shift, stats, train)
            seq.append(("bn_in_dy", dy.clone())); seq.append(("bn_dx", r[0].clone())); seq.append(("bn_dgamma", r[1].clone()))
            return r
        def sp(*a, **k):
            r = o_sp(*a, **k)
            seq.append(("sp_dx", r[0].clone())); seq.append(("sp_dW", r[1].clone()))
            return r
        ops.bn_relu_backward, ops.spconv_backward = bn, sp
        try:
            res = orig(tape, g_out, d_mid, sink=sink)
        finally:
            ops.bn_relu_backward, ops.spconv_backward = o_bn, o_sp
        rec["d_in"] = res.clone()
        return res
    net3.backward = spy
    rec["mvol"] = vol_tape[3]["mvol"].clone()
    rec["depths"] = [outputs[f"depth_stage{s}"].clone() for s in range(4)] if "depth_stage0" in outputs else []
    model.zero_grad(set_to_none=True)
    model.backward_volumes([G_rows[s].to(d) for s in (3, 2, 1, 0)], {s: (G_dep[s][0].to(d), G_dep[s][1].to(d)) for s in range(4)})
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    enc = [r["raw"].clone() for r in fpn_tape[-1]["enc"]] + [r["raw"].clone() for r in fpn_tape[-1]["dec"] if r is not None]
    blocks.append(rec)
    return [f.clone() for f in feats], [v.clone() for v in volumes], grads, enc, blocks

A, B = [x == "valu" for x in (sys.argv[1:3] if len(sys.argv) > 2 else ["valu", "mfma"])]
print("first run valu =", A, " second run valu =", B)
fa, va, ga, ea, ba = run(A)
fb, vb, gb, eb, bb = run(B)
ra, rb = ba.pop(), bb.pop()
for k in ("mvol", "g_out", "d_in"):
    x, y = ra[k], rb[k]
    print(k, tuple(x.shape), "max diff", float((x - y).abs().max()), "of", float(x.abs().max()), "n differing > 1e-3 of max:",
          int(((x - y).abs() > 1e-3 * x.abs().max()).sum()))
for (na, x), (nb, y) in zip(ra["seq"], rb["seq"]):
    print("  ", na, tuple(x.shape), "diff", float((x - y).abs().max()), "of", float(x.abs().max()))
gd = (ra["g_out"] - rb["g_out"]).abs()
print("g_out col 0 (logit) diff", float(gd[:, 0].max()), "cols 1..7 diff", float(gd[:, 1:].max()))
for i, (x, y) in enumerate(zip(ra["depths"], rb["depths"])):
    print("depth stage", i, float((x - y).abs().max()), float(x.abs().max()))
for i, (x, y) in enumerate(zip(ba, bb)):
    print("nets.3 block", i, "sites", x[2], "raw diff", float((x[0] - y[0]).abs().max()), "of", float(x[0].abs().max()),
          "y diff", float((x[1] - y[1]).abs().max()), "of", float(x[1].abs().max()))
for i, (x, y) in enumerate(zip(ea, eb)):
    print("raw conv out", i, tuple(x.shape), float((x - y).abs().max()), float(x.abs().max()))
for i, (x, y) in enumerate(zip(fa, fb)):
    print("feat level", i, tuple(x.shape), float((x - y).abs().max()), float(x.abs().max()))
for i, (x, y) in enumerate(zip(va, vb)):
    print("volume", i, tuple(x.shape), float((x - y).abs().max()) if x.shape == y.shape else "shape differs", float(x.abs().max()))
w = sorted(((float((ga[k] - gb[k]).abs().max()) / max(float(ga[k].abs().max()), 1e-9), k) for k in ga), reverse=True)[:12]
for e in w: print(e)
