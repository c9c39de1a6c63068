"""Time the SDF kernel variants (forward-only vs forward+gradient) on the bench scene's sample points."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import synthetic, ops
from bench import model_conf
from surf_amd.implicit_surface import ImplicitSurface
dev = torch.device('cuda:0')
H, W, nv = (int(sys.argv[1]) if len(sys.argv) > 1 else 288), 800, 5
n_samples = [64, 32, 16, 16]
torch.manual_seed(0)
model = ImplicitSurface(model_conf(n_samples)).to(dev)
intrs, c2ws, near_fars = synthetic.ring_cameras(nv, 576, W)
vols, tabs, mvol = synthetic.sphere_pyramid(88, dev)
sv = ops.SparseVolumes(vols[::-1], tabs[::-1])
rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], 576, W, 1, dev)
rays_o, rays_d = rays_o[:H * W].contiguous(), rays_d[:H * W].contiguous()
R = rays_o.shape[0]
near = near_fars[0, 0].reshape(1, 1).repeat(R, 1).to(dev); far = near_fars[0, 1].reshape(1, 1).repeat(R, 1).to(dev)
st = ops.ray_setup(rays_o, rays_d, near, far, mvol, sv, n_samples, [1.0, 0.4, 0.1, 0.01], 256)
sdf_w, blend_w = model.packed_weights(dev)
if os.environ.get('SURF_BF16'):
    os.environ.setdefault('SURF_PREC', 'bf16x3')
if os.environ.get('SURF_PREC', 'f32') != 'f32':
    sd = {k: v for k, v in model.state_dict().items()}
    sdf_w = ops.sdf_pack_weights_split(sd, dev, 'sdf_network.', os.environ['SURF_PREC'])
n_act = int(st["vmask"].sum())
for grad in (False, True):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.sdf_mlp(st["pts"], sv, sdf_w, mask=st["vmask"], want_grad=grad)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    flop = n_act * (396960 if grad else 198480)
    print(f"grad={grad}: {dt*1e3:.1f} ms, {flop/dt/1e12:.1f} TFLOP/s algorithmic ({flop/dt/1e12/157.3:.3f} of peak), tiles/s {st['pts'].shape[0]/32/dt:.3e}")
    from surf_amd import _lib
    import ctypes
    L = _lib.lib()
    fn = 'surf_debug_phases_f16' if os.environ.get('SURF_PREC') == 'f16x2' else 'surf_debug_phases'
    if hasattr(L, fn):
        buf = (ctypes.c_ulonglong * 8)()
        getattr(L, fn)(buf, 1)
        v = list(buf); tot = sum(v) or 1
        names = ['gather', 'forward', 'tail', 'backward', 'epilogue', 'stage_wait', 'lds_commit', 'barrier']
        print('  phases (3 runs, wave 0 clocks): ' + ', '.join(f'{n} {x/tot:.3f}' for n, x in zip(names, v)) + f'  total/run/block {tot/3/256:.0f} clk')
