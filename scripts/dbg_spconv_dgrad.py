"""Per-layer check of the sparse-conv input gradient at the bench scene's sizes: the matrix-core path (surf_spconv_mfma on the
transposed kernel) against the per-voxel fp32 kernel, and both against the identity <dy, conv(x; W)> = <dx, x>."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import training_step_setup
from surf_amd import ops
dev = torch.device("cuda:0")
model, ipts, targets, loss_fn, opt = training_step_setup(dev)
torch.manual_seed(5)
model("train", ipts, 1.0, 3, record=True)
torch.cuda.synchronize()
for stage in (0, 2):
    tape = model._train_tape["vol"][stage]["reg_tape"]
    g = torch.Generator(device="cuda").manual_seed(stage)
    for li, e in enumerate(tape[:-1]):
        if e["raw"].shape[0] == 0:
            continue
        dy = torch.randn(e["raw"].shape, device="cuda", generator=g)
        w, mode = e["w"], e["mode"]
        wt = w.transpose(1, 2).contiguous()
        if mode == ops.SUBM:
            wt = wt.flip(0).contiguous()
        m2 = {ops.SUBM: ops.SUBM, ops.DOWN: ops.UP, ops.UP: ops.DOWN}[mode]
        a = ops.spconv(dy, e["out_site"][0], e["in_site"][1], m2, wt)
        pk = ops.spconv_pack_weights(wt)
        ref = float((dy.double() * e["raw"].double()).sum())
        da = float((a.double() * e["x"].double()).sum())
        line = f"stage {stage} layer {li} mode {mode} {tuple(w.shape[1:])} rows in {e['x'].shape[0]} out {e['raw'].shape[0]}: <dy,y> {ref:.6f} valu <dx,x> {da:.6f}"
        if pk is not None:
            b = ops.spconv(dy, e["out_site"][0], e["in_site"][1], m2, wt, packed=pk)
            db = float((b.double() * e["x"].double()).sum())
            bad = (a - b).abs().amax(dim=1) > 1e-3 * float(a.abs().max())
            line += f"  mfma <dx,x> {db:.6f}  max |a-b| {float((a - b).abs().max()):.3e} (|a| max {float(a.abs().max()):.3f}) bad rows {int(bad.sum())}"
            if int(bad.sum()):
                idx = bad.nonzero().view(-1)
                line += f" first bad rows {idx[:6].tolist()} last {idx[-3:].tolist()}"
        print(line)
