"""Time surf_sdf_smooth (training-only H.1 kernel) on n random points inside the synthetic sphere pyramid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import synthetic, ops
from bench import model_conf
from surf_amd.implicit_surface import ImplicitSurface

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d = torch.device("cuda:0")
torch.manual_seed(0)
model = ImplicitSurface(model_conf([64, 32, 16, 16])).to(d)
vols, tabs, mvol = synthetic.sphere_pyramid(88, d)
sv = ops.SparseVolumes(vols[::-1], tabs[::-1])
w = ops.sdf_smooth_pack_weights(model.state_dict(), d, prefix="sdf_network.")
pts = ((torch.rand(n, 3, device=d) * 2 - 1) * 0.6).contiguous()
for _ in range(2):
    ops.sdf_smooth(pts, sv, w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.sdf_smooth(pts, sv, w, want_grad=True)
e1.record()
torch.cuda.synchronize()
print(f"sdf_smooth {n} points: {e0.elapsed_time(e1) / 5:.3f} ms")
