"""Time one finetune-mode optimisation step (forward + loss + HIP backward + Adam) on the bench scene's volumes for a batch of
512 rays x 128 samples (the reference's training batch, confs/surf.conf), kernel by kernel with HIP events."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import synthetic, ops, conf
from surf_amd.losses import Loss
from bench import model_conf
from surf_amd.implicit_surface import ImplicitSurface

dev = torch.device("cuda:0")
nv, H, W, R = 5, 576, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_samples = [64, 32, 16, 16]
torch.manual_seed(0)
model = ImplicitSurface(model_conf(n_samples)).to(dev)
intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
imgs = synthetic.procedural_images(nv, H, W, 0, dev)
feats = synthetic.feature_pyramid(nv, H, W, 0, dev)
vols, tabs, mvol = synthetic.sphere_pyramid(88, dev)
scene = model.scene(mvol, vols[::-1], tabs[::-1], None, feats, imgs, intrs.to(dev), c2ws.to(dev))
rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], H, W, 1, dev)
sel = torch.randperm(rays_o.shape[0], device=dev)[:R]
rays_o, rays_d = rays_o[sel].contiguous(), rays_d[sel].contiguous()
near = near_fars[0, 0].reshape(1, 1).repeat(R, 1).to(dev); far = near_fars[0, 1].reshape(1, 1).repeat(R, 1).to(dev)
target = torch.rand(R, 3, device=dev)
opt = torch.optim.Adam(model.parameters(), lr=5e-4)


def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e


def step():
    t = [ev()]
    out = model.render_scene(rays_o, rays_d, near, far, scene, 1.0, patch_warp=True, step=3)
    t.append(ev())
    leaves = {k: out[k].detach().clone().requires_grad_(True) for k in ("color_fine", "gradient_error", "sparse_sdf")}
    vm = out["valid_mask"].float()
    loss = (((leaves["color_fine"] - target).abs() * vm).sum() / (vm.sum() + 1e-5) + 0.1 * leaves["gradient_error"]
            + 0.02 * torch.exp(-leaves["sparse_sdf"].abs() * 100).mean())
    loss.backward()
    ncc = ops.lncc(out["ref_gray_val"].contiguous(), out["sampled_gray_val"].contiguous())
    t.append(ev())
    opt.zero_grad(set_to_none=True)
    model.backward_render(leaves["color_fine"].grad, None, float(leaves["gradient_error"].grad), leaves["sparse_sdf"].grad)
    t.append(ev())
    opt.step()
    t.append(ev())
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in zip(t[:-1], t[1:])], float(loss.detach()), model.last_active if hasattr(model, "last_active") else None


for _ in range(3):
    step()
acc = [0.0] * 4
N = 10
t0 = time.perf_counter()
for _ in range(N):
    ts, loss, _ = step()
    acc = [a + b for a, b in zip(acc, ts)]
wall = (time.perf_counter() - t0) / N * 1e3
names = ["forward (render + patch warp + H.1 + sparse sdf)", "loss terms (torch autograd on per-ray outputs) + LNCC", "backward (composite, sdf, blend kernels + GEMMs)", "Adam"]
print(f"{R} rays x {sum(n_samples)} samples, {int(model._ctx['act'].shape[0])} active samples; wall {wall:.2f} ms per step, loss {loss:.4f}")
for n, a in zip(names, acc):
    print(f"  {n}: {a / N:.2f} ms")
