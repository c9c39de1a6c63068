import sys, time, os, torch, cProfile, pstats
sys.path.insert(0, '.')
from surf_amd import synthetic
from bench import model_conf
from surf_amd.implicit_surface import ImplicitSurface
from oracle import surf_oracle as O
dev = torch.device('cuda:0')
n_samples = [64, 32, 16, 16]; H, W, nv = 576, 800, 5
torch.manual_seed(0)
model = ImplicitSurface(model_conf(n_samples))
intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
imgs = synthetic.procedural_images(nv, H, W, 0, 'cpu')
feats = synthetic.feature_pyramid(nv, H, W, 0, 'cpu')
vols, tabs, mvol = synthetic.sphere_pyramid(88, dev)
vols = [v[:, :7].cpu() for v in vols[::-1]]; tabs = [t.cpu().long() for t in tabs[::-1]]; mvol = mvol.cpu()
masks = [(t >= 0).float() for t in tabs]
rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], H, W, 1, 'cpu')
idx = torch.linspace(0, rays_o.shape[0] - 1, 256).long()
ro, rd = rays_o[idx], rays_d[idx]
near = near_fars[0, 0].reshape(1, 1).repeat(256, 1); far = near_fars[0, 1].reshape(1, 1).repeat(256, 1)
sd = {"implicit_surface." + k: v.detach() for k, v in model.state_dict().items()}
for nt in (8, 32, 128):
    torch.set_num_threads(nt)
    t0 = time.perf_counter()
    out = O.render(sd, ro, rd, near, far, mvol, vols, tabs, masks, feats, imgs, intrs, c2ws, n_samples, [1.0, 0.4, 0.1, 0.01], 256, 1.0)
    print('threads', nt, 'time', time.perf_counter() - t0, flush=True)
torch.set_num_threads(32)
pr = cProfile.Profile(); pr.enable()
out = O.render(sd, ro, rd, near, far, mvol, vols, tabs, masks, feats, imgs, intrs, c2ws, n_samples, [1.0, 0.4, 0.1, 0.01], 256, 1.0)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
