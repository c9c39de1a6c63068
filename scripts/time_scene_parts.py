import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from bench import surf_conf
from surf_amd import conf, synthetic
from surf_amd.surf import SuRF
dev = torch.device("cuda:0")
H, W, nv = 576, 800, 5
torch.manual_seed(0)
mc = surf_conf(88); mc["implicit_surface"]["render"]["n_samples"] = [64, 32, 16, 16]
model = SuRF(conf.from_dict(mc)).eval().to(dev)
model.logit_override = synthetic.sphere_logit
intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], H, W, 1, dev)
ipts = {"imgs": synthetic.procedural_images(nv, H, W, 0, dev), "intrs": intrs.to(dev), "c2ws": c2ws.to(dev),
        "near_fars": near_fars.to(dev), "near": near_fars[0, 0].reshape(1, 1).to(dev), "far": near_fars[0, 1].reshape(1, 1).to(dev),
        "rays_o": rays_o, "rays_d": rays_d, "bound_min": torch.tensor([-1.0] * 3), "bound_max": torch.tensor([1.0] * 3),
        "hw": (H, W), "mesh_resolution": 512}
def T(fn, n=6):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[1:])                       # the first call warms allocator / code objects
    return ts[len(ts) // 2], r, ts[0]
with torch.no_grad():
    r = T(lambda: model("val", ipts, 1.0)); print("full val forward: median %.1f ms, min %.1f" % (r[0], r[2]))
    dt, built, _ = T(lambda: model.run_build("val", ipts)); print("run_build %.1f ms" % dt)
    outputs, volumes, tables, mvol, features, cams, tape = built
    scene = model.build_scene("val", ipts, volumes, tables, mvol, features, cams, None)
    isf = model.implicit_surface
    R = rays_o.shape[0]
    near = ipts["near"].repeat(R, 1); far = ipts["far"].repeat(R, 1)
    for chunk in (65536, 1 << 17, 1 << 19, 65536, 1 << 19):
        r = T(lambda: isf.validate(rays_o, rays_d, near, far, scene, ipts["bound_min"], ipts["bound_max"], (H, W), 1.0, None, extract_geometry=False, chunk=chunk))
        print("validate(no mesh) chunk %d: median %.1f ms, min %.1f" % (chunk, r[0], r[2]))
    print("render_scene one call: %.1f ms" % T(lambda: isf.render_scene(rays_o, rays_d, near, far, scene, 1.0, per_sample=False))[0])
    print("sdf_grid 512: %.1f ms" % T(lambda: isf.sdf_grid(scene, ipts["bound_min"], ipts["bound_max"], 512))[0])
    print("extract_geometry 512: %.1f ms" % T(lambda: isf.extract_geometry(None, None, ipts["bound_min"], ipts["bound_max"], 512, 0.0, scene=scene))[0])
