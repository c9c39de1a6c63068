#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the reference implementation.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box).
It imports the reference's own modules (with the import shims of SURVEY.md Appendix B), feeds
them small seeded synthetic inputs and stores inputs + weights + outputs as .npz DATA.
No reference source is copied.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Fixtures
  scene.npz      cameras / images / rays of the tiny synthetic scene (3 ring cameras, 32x48)
  weights.npz    seeded state_dicts (reference key names) of FPN, agg_mlp, SDF MLP, blending, variance
  fpn.npz        FeatureNetwork.forward outputs                                   (row a1)
  pipeline.npz   per-stage up_sample / depth_filtering / back_proj_multiscale / sparse2dense /
                 get_index / MatchingField outputs with a stub regulariser          (rows a2-a4,a6,a7)
  render.npz     lookup_sparse_volume, SDFNetworkSparse.{sdf,gradient}, lookup_feature,
                 BlendingNetwork, ImplicitSurface.render (perturb=0) outputs        (rows a8-a14)
  sdf_grid.npz   extract_geometry lattice values at resolution 24                   (row a16)
  validate.npz   ImplicitSurface.validate's image outputs (img_fine, normal_img, sdf_depth, render_depth) on the 7 x 8
                 ray lattice, 256-ray chunks as the reference renders them            (row a16)
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


_captured = {}


def _fake_mc(u, thr):
    _captured["u"] = np.array(u)
    return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64)


def import_reference():
    sys.path.insert(0, REF)
    _stub("mcubes", marching_cubes=_fake_mc)
    _stub("models.modules.grid_sample_cuda")
    _stub("models.modules.grid_sample_cuda.cuda_gridsample",
          grid_sample_3d=lambda inp, grid, padding_mode="zeros", align_corners=True:
          F.grid_sample(inp, grid, mode="bilinear", padding_mode=padding_mode, align_corners=align_corners))
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models.modules.feature_network import FeatureNetwork
    from models.modules.volume import Volume
    from models.modules.matching_field import MatchingField
    from models.modules.implicit_surface import ImplicitSurface
    from models.modules import projector
    return FeatureNetwork, Volume, MatchingField, ImplicitSurface, projector


class Conf(dict):
    """dict-backed stand-in for the pyhocon ConfigTree getters used by the reference."""

    def _get(self, key, default=None):
        cur = self
        for part in key.split("."):
            if not isinstance(cur, dict) or not dict.__contains__(cur, part):
                return default
            cur = dict.__getitem__(cur, part)
        return Conf(cur) if isinstance(cur, dict) and not isinstance(cur, Conf) else cur

    def __getitem__(self, key):
        v = self._get(key)
        if v is None:
            raise KeyError(key)
        return v

    get_int = get_float = get_bool = get_list = get_string = get = lambda self, k, default=None: self._get(k, default)


MODEL_CONF = {
    "range_ratios": [1.0, 0.4, 0.1, 0.01],
    "feature_network": {"d_in": 3, "d_base": 8, "d_out": [4, 4, 4, 4]},
    "volume": {"base_volume_dim": [8, 8, 8]},
    "matching_field": {"n_samples_depths": [128, 64, 32, 16], "n_importance_depths": [128, 64, 32, 16],
                       "up_sample_steps": [4, 4, 4, 4], "depth_res_levels": [4, 2, 2, 1]},
    "implicit_surface": {
        "sdf_network": {"d_out": 129, "d_in": 3, "d_hidden": 128, "n_layers": 6, "skip_in": [3], "multires": 4,
                        "bias": 0.5, "scale": 1.0, "geometric_init": True, "weight_norm": True,
                        "feat_channels": 28, "feat_multires": 0},
        "color_network": {"d_feature": 16},
        "variance_network": {"init_val": 0.3},
        "render": {"n_samples": [64, 32, 16, 16], "sample_ranges": [1.0, 0.4, 0.1, 0.01], "n_depth": 256,
                   "perturb": 0.0},
    },
}


def make_scene(nv=3, H=32, W=48, radius=2.5, seed=0):
    """Ring cameras looking at the origin, smooth procedural images."""
    g = torch.Generator().manual_seed(seed)
    az = [0.0, 0.25, -0.25, 0.5, -0.5][:nv]
    c2ws, intrs = [], []
    for a in az:
        o = torch.tensor([radius * np.sin(a), 0.3 * np.sin(2 * a), -radius * np.cos(a)], dtype=torch.float32)
        z = -o / o.norm()
        x = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0]), z)
        x = x / x.norm()
        y = torch.linalg.cross(z, x)
        c2w = torch.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = x, y, z, o
        K = torch.eye(4)
        K[0, 0] = K[1, 1] = 1.6 * W
        K[0, 2], K[1, 2] = (W - 1) / 2, (H - 1) / 2
        c2ws.append(c2w)
        intrs.append(K)
    c2ws, intrs = torch.stack(c2ws), torch.stack(intrs)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    imgs = []
    for v in range(nv):
        chans = [0.5 + 0.45 * torch.sin(0.31 * (c + 1) * xx + 0.17 * (v + 1) * yy + c) for c in range(3)]
        imgs.append(torch.stack(chans))
    imgs = (torch.stack(imgs) + 0.05 * torch.rand(nv, 3, H, W, generator=g)).clamp(0, 0.999)
    dist = c2ws[:, :3, 3].norm(dim=1)
    near_fars = torch.stack([0.95 * (dist - 1), 1.05 * (dist + 1)], dim=1)
    # rays of the reference view on a strided pixel lattice
    ys, xs = torch.meshgrid(torch.arange(1, H, 5, dtype=torch.float32), torch.arange(2, W, 6, dtype=torch.float32), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(xs.numel())], dim=-1)
    d = pix @ torch.inverse(intrs[0])[:3, :3].t()
    d = d / d.norm(dim=-1, keepdim=True)
    rays_d = d @ c2ws[0, :3, :3].t()
    rays_o = c2ws[0, :3, 3][None].expand_as(rays_d).contiguous()
    return {"imgs": imgs, "intrs": intrs, "c2ws": c2ws, "near_fars": near_fars, "rays_o": rays_o, "rays_d": rays_d,
            "near": near_fars[0, 0].reshape(1, 1), "far": near_fars[0, 1].reshape(1, 1)}


def stub_regnet(feats, coords, D, stage):
    """Deterministic stand-in for the (un-importable) torchsparse U-Net used to pin the pipeline."""
    g = torch.Generator().manual_seed(100 + stage)
    A = torch.randn(feats.shape[1], 8, generator=g) * 0.5
    B = torch.randn(feats.shape[1], 8, generator=g) * 0.5
    world = coords * (2.0 / (D - 1)) - 1.0
    out = torch.tanh(feats @ A)
    out[:, 0] = -20.0 * (world.norm(dim=1) - 0.5).abs() + 0.5 * out[:, 0]
    return out, torch.tanh(feats @ B)


ONLY = set(a for a in sys.argv[1:] if a.endswith(".npz"))   # e.g. `make_golden.py render_perturb.npz`: rewrite just that file


def npz(name, **arrs):
    if ONLY and name not in ONLY:
        return
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(f"wrote {name}: {sum(a.nbytes for a in out.values()) / 1e6:.2f} MB raw, {len(out)} arrays")


def main():
    FeatureNetwork, Volume, MatchingField, ImplicitSurface, projector = import_reference()
    conf = Conf(MODEL_CONF)
    scene = make_scene()
    npz("scene.npz", **scene)

    # ---------------- weights ----------------
    torch.manual_seed(0)
    fpn = FeatureNetwork(conf["feature_network"]).eval()
    vol = Volume(conf["volume"]).eval()
    mf = MatchingField(conf["matching_field"]).eval()
    isurf = ImplicitSurface(conf["implicit_surface"]).eval()
    # perturb the SDF MLP away from the pure geometric init so that every weight block matters
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for l in range(7):
            lin = getattr(isurf.sdf_network, f"lin{l}")
            lin.weight_v.add_(torch.randn(lin.weight_v.shape, generator=g) * (0.02 if l < 6 else 0.01))
            lin.weight_g.mul_(1.0 + 0.1 * torch.randn(lin.weight_g.shape, generator=g))
            lin.bias.add_(torch.randn(lin.bias.shape, generator=g) * 0.02)
    # ... and recentre the zero level set on the r = 0.5 sphere so that the test rays hit a surface
    sphere = F.normalize(torch.randn(256, 3, generator=g), dim=1) * 0.5
    empty = [torch.zeros(1, 7)] * 4, [torch.full((2, 2, 2), -1, dtype=torch.int64)] * 4
    with torch.no_grad():
        lvl = isurf.sdf_network.sdf(sphere, *empty).mean()
        isurf.sdf_network.lin6.bias[0] -= lvl
        isurf.deviation_network.variance.fill_(0.45)      # a trained-like sharpness (inv_s = e^4.5)
    sd = {}
    sd.update({"feature_network." + k: v for k, v in fpn.state_dict().items()})
    sd.update({"volume." + k: v for k, v in vol.state_dict().items()})
    sd.update({"implicit_surface." + k: v for k, v in isurf.state_dict().items()})
    npz("weights.npz", **sd)

    # ---------------- a1 FPN ----------------
    with torch.no_grad():
        feats = fpn(scene["imgs"])
    npz("fpn.npz", **{f"out{i}": f for i, f in enumerate(feats)})

    # ---------------- a2-a4, a6, a7 pipeline with stub regulariser ----------------
    pipe = {}
    ratios = MODEL_CONF["range_ratios"]
    intrs, c2ws = scene["intrs"], scene["c2ws"]
    base_range = (scene["far"] - scene["near"]).squeeze()
    depths, mvol = None, None
    volumes, tables, masks = [], [], []
    with torch.no_grad():
        for s in range(4):
            if s == 0:
                coords = vol.init_coords().type_as(intrs)
            else:
                coords, up_feats = vol.up_sample(coords, mid)
                pipe[f"s{s}_up_coords"] = coords.to(torch.int16)
                coords, up_feats = vol.depth_filtering(depths, coords, up_feats, intrs, c2ws, base_range * ratios[s])
                pipe[f"s{s}_filt_coords"] = coords.to(torch.int16)
            D = int(vol.volume_dim[0])
            cv, keep = vol.back_proj_multiscale(feats, coords, intrs, c2ws, s)
            pipe[f"s{s}_costvol"] = cv
            pipe[f"s{s}_keep"] = keep
            cv, coords = cv[keep], coords[keep]
            if s > 0:
                up_feats = up_feats[keep]
                cv = torch.cat([cv, up_feats], dim=1)
            out, mid = stub_regnet(cv, coords, D, s)
            pipe[f"s{s}_reg_in"] = cv
            pipe[f"s{s}_reg_out"] = out
            pipe[f"s{s}_reg_mid"] = mid
            mvol, mask = vol.sparse2dense(out[:, :1], coords, mvol)
            table = vol.get_index(coords)
            pipe[f"s{s}_coords"] = coords.to(torch.int16)
            pipe[f"s{s}_mvol"] = mvol[0, 0]
            pipe[f"s{s}_mask_sum"] = mask.sum()
            pipe[f"s{s}_table"] = table.to(torch.int32)
            ipts = dict(scene)
            depths, _ = mf(ipts, mvol, s, ratios, depths, perturb=False)
            pipe[f"s{s}_depths"] = torch.stack(depths)
            volumes.append(out[:, 1:].contiguous())
            tables.append(table)
            masks.append(mask)
    npz("pipeline.npz", **pipe)

    # ---------------- a8-a14 render ----------------
    rend = {}
    vols_r, tabs_r, masks_r, feats_r = volumes[::-1], tables[::-1], masks[::-1], feats[::-1]
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand(400, 3, generator=g) * 2 - 1) * 0.8
    pts[:20] = pts[:20] * 1.6            # some outside the cube (extrapolating weights, clamped indices)
    with torch.no_grad():
        rend["pts"] = pts
        rend["phi"] = projector.lookup_sparse_volume(pts.clone(), vols_r, tabs_r)
        rend["mask_nearest"] = projector.lookup_volume(pts, masks_r, sample_mode="nearest")
        rend["mvol_trilinear"] = projector.lookup_volume(pts, mvol, sample_mode="bilinear")
        rend["sdf_out"] = isurf.sdf_network(pts.clone(), vols_r, tabs_r)
    gr, sm = isurf.sdf_network.gradient(pts.clone(), vols_r, tabs_r)
    rend["sdf_grad"], rend["sdf_smooth"] = gr.detach(), sm.detach()
    with torch.no_grad():
        rf, rdiff, mval = projector.lookup_feature(pts, scene["imgs"], intrs, c2ws, feats_r)
        rend["rgb_feat"], rend["ray_diff"], rend["mask_valid"] = rf, rdiff, mval
        rend["blend_rgb"] = isurf.color_network(rf, rdiff, mval)
    R = scene["rays_o"].shape[0]
    near = scene["near"].repeat(R, 1)
    far = scene["far"].repeat(R, 1)
    for ratio in (1.0, 0.3):
        outs = isurf.render(scene["rays_o"], scene["rays_d"], near, far, mvol, vols_r, tabs_r, masks_r, scene["imgs"],
                            feats_r, feats_r, intrs, c2ws, ratio, None)
        tag = "r%02d_" % int(ratio * 10)
        for k in ("color_fine", "render_depth", "sdf_depth", "normal", "weights", "gradients", "mid_z_vals",
                  "inside_sphere", "valid_mask", "mid_inside_sphere", "gradient_error", "weight_sum"):
            rend[tag + k] = outs[k].detach()
        rend[tag + "sdf"] = outs["sparse_sdf"][1024:].detach().reshape(R, -1)
    npz("render.npz", **rend)

    # ---------------- a8 with render.perturb = 1 (every shipped conf sets it; implicit_surface.py:274-277,304-306) ------
    # The jitters are the first four torch.rand([R, 1]) draws of ImplicitSurface.render on the CPU generator.
    isurf.perturb = 1.0
    torch.manual_seed(4321)
    outs = isurf.render(scene["rays_o"], scene["rays_d"], near, far, mvol, vols_r, tabs_r, masks_r, scene["imgs"],
                        feats_r, feats_r, intrs, c2ws, 1.0, None)
    torch.manual_seed(4321)
    t_rand = torch.cat([torch.rand([R, 1]) - 0.5 for _ in range(4)], dim=1)
    npz("render_perturb.npz", t_rand=t_rand, mid_z_vals=outs["mid_z_vals"].detach(), color_fine=outs["color_fine"].detach(),
        render_depth=outs["render_depth"].detach(), weights=outs["weights"].detach())
    isurf.perturb = 0.0

    # ---------------- a16 validate(): the image outputs the runner saves (implicit_surface.py:359-402, runner.py:222-229) ----
    # mesh extraction off (PyMCubes is absent; the lattice is pinned by sdf_grid.npz): what is pinned here is the assembly -
    # x 256 clip, normals rotated into the reference camera x 128 + 128, depth maps - on top of the chunked render
    val = isurf.validate(scene["rays_o"], scene["rays_d"], near, far, mvol, vols_r, tabs_r, masks_r, scene["imgs"], feats_r,
                         feats_r, intrs, c2ws, torch.tensor([-0.8] * 3), torch.tensor([0.8] * 3), (7, 8), 1.0, None,
                         extract_geometry=False)
    npz("validate.npz", color_fine=val["color_fine"], img_fine=val["img_fine"], normal_img=val["normal_img"],
        sdf_depth=val["sdf_depth"], render_depth=val["render_depth"])

    # ---------------- a16 SDF lattice ----------------
    bmin, bmax = torch.tensor([-0.7, -0.6, -0.65]), torch.tensor([0.7, 0.75, 0.6])
    with torch.no_grad():
        isurf.extract_geometry(vols_r, tabs_r, bmin, bmax, 24, 0.0)
    npz("sdf_grid.npz", u=_captured["u"], bound_min=bmin, bound_max=bmax)


if __name__ == "__main__":
    main()
