"""Race screen for the split SDF kernels: many launches of a multi-round, masked, compacted problem, every launch
compared with the fp32-MFMA kernel's result on the same points (scripts/stress_sdf.py [iters] [n_points])."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import synthetic, ops
from bench import model_conf
from surf_amd.implicit_surface import ImplicitSurface

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600_000
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ImplicitSurface(model_conf([64, 32, 16, 16])).to(dev)
vols, tabs, mvol = synthetic.sphere_pyramid(44, dev)
sv = ops.SparseVolumes(vols[::-1], tabs[::-1])
g = torch.Generator().manual_seed(3)
pts = ((torch.rand(n, 3, generator=g) * 2 - 1) * 0.6).to(dev).contiguous()
mask = (torch.arange(n) % 5 != 0).to(torch.uint8).to(dev)
sd = {k: v for k, v in model.state_dict().items()}
w32 = ops.sdf_pack_weights(sd, dev, "sdf_network.")
s_ref, g_ref = ops.sdf_mlp(pts, sv, w32, mask=mask)
for prec in ("bf16x3", "f16x2") if len(sys.argv) < 4 else sys.argv[3].split(","):
    w = ops.sdf_pack_weights_split(sd, dev, "sdf_network.", prec)
    bad_runs, worst = 0, 0.0
    for it in range(iters):
        s, gr = ops.sdf_mlp(pts, sv, w, mask=mask)
        es = float((s - s_ref).abs().max())
        eg = float((gr - g_ref).abs().max())
        worst = max(worst, es, eg)
        if es > 1e-4 or eg > 1e-3:
            bad_runs += 1
    print(f"{prec}: {bad_runs}/{iters} launches off (max |diff| vs fp32 kernel over all launches {worst:.3g})")
