import os, sys, torch
sys.path.insert(0, os.getcwd())
from surf_amd import synthetic, ops
from bench import model_conf
from surf_amd.implicit_surface import ImplicitSurface
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = ImplicitSurface(model_conf([64, 32, 16, 16])).to(dev)
vols, tabs, mvol = synthetic.sphere_pyramid(16, dev)
sv = ops.SparseVolumes(vols[::-1], tabs[::-1])
sdf_w, _ = model.packed_weights(dev)
res = 48
axes = [torch.linspace(-1.0, 1.0, res).to(dev) for _ in range(3)]
u = torch.empty(res, res, res, dtype=torch.float32, device=dev)
print("lattice launch", flush=True)
ops.sdf_lattice(axes, sv, sdf_w, u, 0, res, sign=-1.0)
torch.cuda.synchronize()
print("lattice ok", float(u.abs().max()), flush=True)
xx, yy, zz = torch.meshgrid(axes[0], axes[1], axes[2], indexing="ij")
pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1).contiguous()
sdf, _ = ops.sdf_mlp(pts, sv, sdf_w, want_grad=False)
torch.cuda.synchronize()
print("points ok; bit-equal:", bool(torch.equal(u.reshape(-1), -sdf)), flush=True)
s2, g2 = ops.sdf_mlp(pts, sv, sdf_w, want_grad=True)
torch.cuda.synchronize()
print("grad ok", float(g2.abs().max()), bool(torch.equal(s2, sdf)), flush=True)
