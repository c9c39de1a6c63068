"""Time the blending kernels alone on the bench scene's sample points (scripts/time_blend.py [H] ; SURF_BLEND=f32|bf16x3|f16x2,
comma list allowed; SURF_VIEWS=5)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import synthetic, ops
from bench import model_conf
from surf_amd.implicit_surface import ImplicitSurface
dev = torch.device('cuda:0')
nv = int(os.environ.get('SURF_VIEWS', '5'))
H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 288), 800
n_samples = [64, 32, 16, 16]
torch.manual_seed(0)
model = ImplicitSurface(model_conf(n_samples)).to(dev)
intrs, c2ws, near_fars = synthetic.ring_cameras(nv, 576, W)
imgs = synthetic.procedural_images(nv, 576, W, 0, dev)
feats = synthetic.feature_pyramid(nv, 576, W, 0, dev)
vols, tabs, mvol = synthetic.sphere_pyramid(88, dev)
scene = model.scene(mvol, vols[::-1], tabs[::-1], None, feats, imgs, intrs.to(dev), c2ws.to(dev))
rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], 576, W, 1, dev)
rays_o, rays_d = rays_o[:H * W].contiguous(), rays_d[:H * W].contiguous()
R = rays_o.shape[0]
near = near_fars[0, 0].reshape(1, 1).repeat(R, 1).to(dev); far = near_fars[0, 1].reshape(1, 1).repeat(R, 1).to(dev)
st = ops.ray_setup(rays_o, rays_d, near, far, mvol, scene.sv, n_samples, [1.0, 0.4, 0.1, 0.01], 256)
act = ops.compact(st["vmask"])
sd = {k: v for k, v in model.state_dict().items()}
ref = None
for prec in os.environ.get('SURF_BLEND', 'f32,f32lds,bf16x3,f16x2').split(','):
    w = ops.blend_pack_weights(sd, dev, "color_network.", prec)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        col, nval = ops.blend(st["pts"], scene.feats_t4, scene.imgs_t4, scene.cams, w, mask=st["vmask"], active_idx=act)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n_act = int(act.shape[0])
    flop = n_act * (nv - 1) * 2 * 9928
    if ref is None:
        ref = col
    if os.environ.get('SURF_PHASES'):         # a -DSURF_BLEND_TIMING build: shader clocks of wavefront 0 of every workgroup, by phase
        import ctypes
        from surf_amd import _lib
        raw = ctypes.CDLL(_lib.LIB_PATH)
        buf = (ctypes.c_ulonglong * 10)()
        ph = raw.surf_debug_blend_phases
        ph(buf, 1)
        col, nval = ops.blend(st["pts"], scene.feats_t4, scene.imgs_t4, scene.cams, w, mask=st["vmask"], active_idx=act)
        torch.cuda.synchronize()
        ph(buf, 0)
        tiles_per_wave = (n_act / 32) / (256 * 8)
        names = ['tile setup', 'pass 1 (4 views)', 'pooling sweeps', 'shared base_fc.0', 'p2 base_fc', 'p2 vis_fc', 'p2 vis_fc2', 'p2 rgb_fc+softmax']
        tot = sum(buf[:8])
        for k in range(8):
            print(f"   phase {k} {names[k]:20s} {buf[k] / 256 / tiles_per_wave:9.0f} clocks per tile  ({100.0 * buf[k] / tot:4.1f} %)")
        print(f"   total {tot / 256 / tiles_per_wave:9.0f} clocks per tile per wave;  workgroup 0: {buf[8]} s_memtime ticks in {buf[9] / 100.0:.1f} us "
              f"(s_memrealtime, 100 MHz) = {buf[8] / max(buf[9], 1) * 0.1:.3f} GHz")
        wg = (ctypes.c_ulonglong * 512)()
        raw.surf_debug_blend_wg(wg)
        t0 = min(wg[2 * i] for i in range(256))
        dur = [(wg[2 * i + 1] - wg[2 * i]) / 100.0 for i in range(256)]
        start = [(wg[2 * i] - t0) / 100.0 for i in range(256)]
        print("   workgroup durations (us): min %.0f  median %.0f  max %.0f;  latest start %.0f us" % (min(dur), sorted(dur)[128], max(dur), max(start)))
        for x in range(8):
            d = [dur[i] for i in range(x, 256, 8)]
            print(f"   XCD {x}: min {min(d):.0f}  mean {sum(d) / len(d):.0f}  max {max(d):.0f} us")
    if os.environ.get('SURF_SAVE'):          # A/B of two library builds: SURF_SAVE=a.pt, then SURF_SAVE=b.pt SURF_CMP=a.pt
        torch.save({'col': col.cpu(), 'nval': nval.cpu()}, os.environ['SURF_SAVE'] + '.' + prec)
    if os.environ.get('SURF_CMP'):
        o = torch.load(os.environ['SURF_CMP'] + '.' + prec)
        print(f"   vs {os.environ['SURF_CMP']}: max |rgb diff| {float((o['col'] - col.cpu()).abs().max()):.3e}, "
              f"bitwise equal {bool(torch.equal(o['col'], col.cpu()))}, n_valid equal {bool(torch.equal(o['nval'], nval.cpu()))}")
    print(f"{prec}: {dt*1e3:.2f} ms for {n_act} samples x {nv-1} views, {flop/dt/1e12:.1f} TFLOP/s algorithmic, "
          f"{dt/ (n_act/32) * 1e9 * 256 * 4 / (nv-1):.0f} ns per (tile, view) per SIMD, max |rgb - first| {float((col-ref).abs().max()):.2e}")
