"""GPU parity: the HIP kernels (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerances (BASELINE.json north_star): 1e-4 abs on SDF values, 1e-3 relative on rendered RGB / depth.
Index / mask outputs must match exactly.
"""
import numpy as np
import pytest
import torch

from oracle import surf_oracle as O
from tests.golden_cfg import CFG, pipeline_views

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def rel_close(a, b, rtol, atol):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), f"{int(bad.sum())}/{bad.numel()} off, max err {err.max().item():.3e}"


@pytest.fixture(scope="module")
def gpu_scene(scene, weights, golden_fpn, golden_pipe):
    from surf_amd import ops
    d = dev()
    vols, tabs, masks, mvol = pipeline_views(golden_pipe)
    feats = [golden_fpn[f"out{i}"] for i in range(4)][::-1]         # fine -> coarse
    sv = ops.SparseVolumes([v.to(d) for v in vols], [t.to(d) for t in tabs])
    return {
        "cpu": dict(vols=vols, tabs=tabs, masks=masks, mvol=mvol, feats=feats),
        "sv": sv,
        "mvol": mvol.to(d).contiguous(),
        "feats_t4": [ops.pack_texel4(f.to(d).contiguous()) for f in feats],
        "imgs_t4": ops.pack_texel4(scene["imgs"].to(d).contiguous()),
        "cams": ops.Cameras(scene["intrs"], scene["c2ws"]),
        "sdf_w": ops.sdf_pack_weights(weights, d),
        "blend_w": ops.blend_pack_weights(weights, d),
    }


def test_pack_texel4(scene):
    from surf_amd import ops
    x = scene["imgs"].to(dev()).contiguous()
    t4 = ops.pack_texel4(x).cpu()
    assert torch.equal(t4[..., :3], scene["imgs"].permute(0, 2, 3, 1))
    assert float(t4[..., 3].abs().max()) == 0.0


def test_ray_setup_matches_oracle(scene, gpu_scene):
    from surf_amd import ops
    d = dev()
    R = scene["rays_o"].shape[0]
    near, far = scene["near"].repeat(R, 1), scene["far"].repeat(R, 1)
    c = gpu_scene["cpu"]
    z_ref = O.ray_zsample(scene["rays_o"], scene["rays_d"], near, far, c["mvol"], CFG["n_samples"],
                          CFG["sample_ranges"], CFG["n_depth"])
    out = ops.ray_setup(scene["rays_o"].to(d), scene["rays_d"].to(d), near.to(d), far.to(d), gpu_scene["mvol"],
                        gpu_scene["sv"], CFG["n_samples"], CFG["sample_ranges"], CFG["n_depth"], want_z=True)
    rel_close(out["z_vals"], z_ref, 0, 3e-6)
    S = z_ref.shape[1]
    dists = torch.cat([z_ref[:, 1:] - z_ref[:, :-1], torch.full((R, 1), 2.0 / CFG["n_samples"][0])], -1)
    mid = z_ref + dists * 0.5
    rel_close(out["mid_z"], mid, 0, 3e-6)
    # the mask is a discontinuous function of the position: compare on the kernel's own points
    pts = out["pts"].cpu()
    vm = torch.stack([O.lookup_volume_nearest(pts, m) for m in c["masks"]], -1).any(-1)
    assert torch.equal(out["vmask"].cpu().bool(), vm)
    assert 0.2 < float(vm.float().mean()) < 0.95
    p_ref = (scene["rays_o"][:, None] + scene["rays_d"][:, None] * out["mid_z"].cpu()[..., None]).reshape(-1, 3)
    rel_close(pts, p_ref, 0, 1e-6)
    assert S == sum(CFG["n_samples"])


def test_sdf_mlp_matches_oracle_and_golden(weights, gpu_scene, golden_render):
    from surf_amd import ops
    d = dev()
    c = gpu_scene["cpu"]
    pts = golden_render["pts"]
    sdf, grad = ops.sdf_mlp(pts.to(d).contiguous(), gpu_scene["sv"], gpu_scene["sdf_w"])
    torch.cuda.synchronize()
    # reference's own outputs
    rel_close(sdf, golden_render["sdf_out"][:, 0], 0, 1e-4)
    rel_close(grad, golden_render["sdf_grad"], 1e-3, 2e-4)
    # forward-only variant, odd point count, mask
    n = 333
    mask = (torch.arange(n) % 3 != 0).to(torch.uint8)
    sdf2, g2 = ops.sdf_mlp(pts[:n].to(d).contiguous(), gpu_scene["sv"], gpu_scene["sdf_w"], mask=mask.to(d),
                           want_grad=False)
    assert g2 is None
    exp = torch.where(mask.bool(), golden_render["sdf_out"][:n, 0], torch.full((n,), 100.0))
    rel_close(sdf2, exp, 0, 1e-4)


def test_sdf_mlp_many_tiles_consistent(weights, gpu_scene):
    """More tiles than resident waves: every wave loops; results must not depend on the tile -> wave map."""
    from surf_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(11)
    base = (torch.rand(4096, 3, generator=g) * 2 - 1) * 0.9
    pts = base.repeat(40, 1).contiguous()                      # 163,840 points = 5,120 tiles
    sdf, grad = ops.sdf_mlp(pts.to(d), gpu_scene["sv"], gpu_scene["sdf_w"])
    sdf, grad = sdf.cpu().view(40, -1), grad.cpu().view(40, -1, 3)
    assert torch.equal(sdf, sdf[:1].expand_as(sdf))
    assert torch.equal(grad, grad[:1].expand_as(grad))
    c = gpu_scene["cpu"]
    layers = O.sdf_weights(weights)
    phi, jphi = O.lookup_sparse_volume(base, c["vols"], c["tabs"], with_jac=True)
    s_ref, g_ref, _ = O.sdf_mlp(layers, base, phi, jphi)
    rel_close(sdf[0], s_ref, 0, 1e-4)
    rel_close(grad[0], g_ref, 1e-3, 2e-4)


def test_blend_matches_oracle_and_golden(scene, weights, gpu_scene, golden_render):
    from surf_amd import ops
    d = dev()
    pts = golden_render["pts"]
    color, nvalid = ops.blend(pts.to(d).contiguous(), gpu_scene["feats_t4"], gpu_scene["imgs_t4"], gpu_scene["cams"],
                              gpu_scene["blend_w"])
    rel_close(color, golden_render["blend_rgb"], 1e-3, 1e-5)
    assert torch.equal(nvalid.cpu().long(), golden_render["mask_valid"].long().sum(1))
    assert 0 < int((nvalid.cpu() == 0).sum()) < pts.shape[0]   # both all-masked and visible points present


def test_render_matches_golden(scene, weights, gpu_scene, golden_render):
    from surf_amd import ops
    d = dev()
    R = scene["rays_o"].shape[0]
    near, far = scene["near"].repeat(R, 1), scene["far"].repeat(R, 1)
    rays_o, rays_d = scene["rays_o"].to(d), scene["rays_d"].to(d)
    inv_s = float(torch.exp(weights["implicit_surface.deviation_network.variance"] * 10.0).clamp(1e-6, 1e6))
    for ratio in (1.0, 0.3):
        tag = "r%02d_" % int(ratio * 10)
        g = {k[len(tag):]: v for k, v in golden_render.items() if k.startswith(tag)}
        st = ops.ray_setup(rays_o, rays_d, near.to(d), far.to(d), gpu_scene["mvol"], gpu_scene["sv"], CFG["n_samples"],
                           CFG["sample_ranges"], CFG["n_depth"])
        sdf, grad = ops.sdf_mlp(st["pts"], gpu_scene["sv"], gpu_scene["sdf_w"], mask=st["vmask"])
        col, nvalid = ops.blend(st["pts"], gpu_scene["feats_t4"], gpu_scene["imgs_t4"], gpu_scene["cams"],
                                gpu_scene["blend_w"], mask=st["vmask"])
        out = ops.composite(sdf, grad, col, nvalid, st, rays_d, inv_s, ratio, gpu_scene["cams"])
        torch.cuda.synchronize()
        S = st["mid_z"].shape[1]
        rel_close(st["mid_z"], g["mid_z_vals"], 0, 3e-6)
        rel_close(sdf.view(R, S), g["sdf"], 0, 1e-4)
        rel_close(grad.view(R, S, 3), g["gradients"], 1e-3, 3e-4)
        rel_close(out["weights"], g["weights"], 1e-3, 2e-5)
        rel_close(out["color_fine"], g["color_fine"], 1e-3, 2e-5)
        rel_close(out["render_depth"], g["render_depth"], 1e-3, 2e-5)
        rel_close(out["sdf_depth"], g["sdf_depth"], 1e-3, 2e-5)
        rel_close(out["normal"], g["normal"], 1e-3, 1e-4)
        assert torch.equal(out["inside_sphere"].cpu(), g["inside_sphere"])
        assert torch.equal(out["valid_mask"].cpu().bool(), g["valid_mask"])
        assert torch.equal(out["mid_inside_sphere"].cpu().float(), g["mid_inside_sphere"])
        eik = out["eik"].cpu().sum(0)
        rel_close(eik[0] / (eik[1] + 1e-5), g["gradient_error"], 1e-3, 1e-5)
