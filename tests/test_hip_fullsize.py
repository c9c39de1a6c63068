"""GPU property tests at BASELINE.json's full sizes (576x800 rays, 5 views, 128 samples, 88^3 -> 704^3 pyramid),
where the CPU oracle is too slow to run everything: size-independent invariants of the hot path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_scene():
    from bench import model_conf
    from surf_amd import synthetic
    from surf_amd.implicit_surface import ImplicitSurface
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    n_samples = [64, 32, 16, 16]
    H, W, nv = 576, 800, 5
    torch.manual_seed(0)
    model = ImplicitSurface(model_conf(n_samples)).to(dev)
    with torch.no_grad():
        model.deviation_network.variance.fill_(0.5)
    intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
    imgs = synthetic.procedural_images(nv, H, W, 0, dev)
    feats = synthetic.feature_pyramid(nv, H, W, 0, dev)
    vols, tabs, mvol = synthetic.sphere_pyramid(88, dev)
    scene = model.scene(mvol, vols[::-1], tabs[::-1], None, feats, imgs, intrs.to(dev), c2ws.to(dev))
    rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], H, W, 1, dev)
    near = near_fars[0, 0].reshape(1, 1).repeat(rays_o.shape[0], 1).to(dev)
    far = near_fars[0, 1].reshape(1, 1).repeat(rays_o.shape[0], 1).to(dev)
    return dict(model=model, scene=scene, rays_o=rays_o, rays_d=rays_d, near=near, far=far, dev=dev, c2ws=c2ws, H=H, W=W)


def test_full_image_render_invariants(full_scene):
    fs = full_scene
    m = fs["model"]
    out = m.render_scene(fs["rays_o"], fs["rays_d"], fs["near"], fs["far"], fs["scene"], 1.0, per_sample=True)
    torch.cuda.synchronize()
    R = fs["rays_o"].shape[0]
    assert R == 576 * 800
    w = out["weights"]
    assert torch.isfinite(w).all() and float(w.min()) >= 0.0
    assert float(w.sum(dim=1).max()) <= 1.0 + 1e-3
    c = out["color_fine"]
    assert torch.isfinite(c).all() and float(c.min()) >= -1e-5 and float(c.max()) <= 1.0 + 1e-4
    # the geometric-init SDF is a sphere-like surface around the origin: centre rays must hit it
    centre = (576 // 2) * 800 + 400
    assert float(w[centre].sum()) > 0.9 and float(out["sdf_depth"][centre]) > 1.5
    # root property: the SDF vanishes at the reported zero crossing of every valid ray
    from surf_amd import ops
    valid = out["mid_inside_sphere"].view(-1) > 0
    assert int(valid.sum()) > 10000
    rot = torch.inverse(fs["c2ws"][0, :3, :3]).to(fs["dev"])
    cz = (fs["rays_d"] @ rot.t())[:, 2]
    z0 = out["sdf_depth"].view(-1) / cz
    pts = (fs["rays_o"] + fs["rays_d"] * z0[:, None])[valid].contiguous()
    sdf_w, _ = m.packed_weights(fs["dev"])
    sdf0, _ = ops.sdf_mlp(pts, fs["scene"].sv, sdf_w, want_grad=False)
    # linear interpolation between two samples: residual bounded by the curvature over one sample spacing
    assert float(sdf0.abs().quantile(0.99)) < 5e-3
    # chunk invariance: rays are independent, any batching gives bit-identical results
    sub = slice(100000, 104096)
    o2 = m.render_scene(fs["rays_o"][sub], fs["rays_d"][sub], fs["near"][sub], fs["far"][sub], fs["scene"], 1.0)
    assert torch.equal(o2["color_fine"], out["color_fine"][sub])
    assert torch.equal(o2["render_depth"], out["render_depth"][sub])
    assert torch.equal(o2["sdf"], out["sdf"][sub])


def test_full_size_gradient_is_the_derivative_of_the_sdf(full_scene):
    """Analytic gradient vs central differences of the kernel's own SDF on the 704^3 pyramid."""
    from surf_amd import ops
    fs = full_scene
    g = torch.Generator().manual_seed(9)
    p = torch.nn.functional.normalize(torch.randn(4096, 3, generator=g), dim=1) * (0.35 + 0.3 * torch.rand(4096, 1, generator=g))
    p = p.to(fs["dev"]).contiguous()
    sdf_w, _ = fs["model"].packed_weights(fs["dev"])
    sv = fs["scene"].sv
    _, grad = ops.sdf_mlp(p, sv, sdf_w)
    eps = 2e-4                      # well inside one 704^3 cell (2.8e-3), so the trilinear piece does not change often
    fd = torch.zeros_like(grad)
    for a in range(3):
        e = torch.zeros(1, 3, device=fs["dev"])
        e[0, a] = eps
        sp, _ = ops.sdf_mlp((p + e).contiguous(), sv, sdf_w, want_grad=False)
        sm, _ = ops.sdf_mlp((p - e).contiguous(), sv, sdf_w, want_grad=False)
        fd[:, a] = (sp - sm) / (2 * eps)
    err = (grad - fd).abs().max(dim=1).values
    # points whose +-eps stencil crosses a cell face see a kink; the bulk must agree to fp32 differencing accuracy
    assert float(err.median()) < 3e-3 and float((err < 2e-2).float().mean()) > 0.9
    assert float(grad.norm(dim=1).mean()) > 0.3


def test_compaction_and_masking_do_not_change_results(full_scene):
    from surf_amd import ops
    fs = full_scene
    st = ops.ray_setup(fs["rays_o"][:20000].contiguous(), fs["rays_d"][:20000].contiguous(), fs["near"][:20000],
                       fs["far"][:20000], fs["scene"].mvol, fs["scene"].sv, [64, 32, 16, 16], [1.0, 0.4, 0.1, 0.01], 256)
    sdf_w, blend_w = fs["model"].packed_weights(fs["dev"])
    a = ops.sdf_mlp(st["pts"], fs["scene"].sv, sdf_w, mask=st["vmask"], compact_active=True)
    b = ops.sdf_mlp(st["pts"], fs["scene"].sv, sdf_w, mask=st["vmask"], compact_active=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    ca = ops.blend(st["pts"], fs["scene"].feats_t4, fs["scene"].imgs_t4, fs["scene"].cams, blend_w, mask=st["vmask"])
    cb = ops.blend(st["pts"], fs["scene"].feats_t4, fs["scene"].imgs_t4, fs["scene"].cams, blend_w, mask=st["vmask"],
                   compact_active=False)
    assert torch.equal(ca[0], cb[0]) and torch.equal(ca[1], cb[1])
    assert 0.5 < float(st["vmask"].float().mean()) < 1.0


# ----------------------------------------------------------------------------------------------------------------------
# the volume build (rows a1-a7) at full size: 5 views 576x800, 88^3 -> 704^3
# ----------------------------------------------------------------------------------------------------------------------


@pytest.fixture(scope="module")
def full_build():
    from bench import surf_conf
    from surf_amd import conf, ops, synthetic
    from surf_amd.surf import SuRF
    dev = torch.device("cuda:0")
    H, W, nv = 576, 800, 5
    torch.manual_seed(0)
    model = SuRF(conf.from_dict(surf_conf(88))).eval().to(dev)
    intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
    ipts = {"imgs": synthetic.procedural_images(nv, H, W, 0, dev), "intrs": intrs.to(dev), "c2ws": c2ws.to(dev),
            "near_fars": near_fars.to(dev), "near": near_fars[0, 0].reshape(1, 1).to(dev),
            "far": near_fars[0, 1].reshape(1, 1).to(dev)}
    feats = model.feature_network(ipts["imgs"])                                  # texel4, coarse -> fine
    trace = {}
    outputs, volumes, tables, mvol = model.build_volumes(ipts, feats, logit_override=synthetic.sphere_logit, trace=trace)
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    feats_cpu = [f.permute(0, 3, 1, 2).contiguous().cpu() for f in feats]        # (nv,4,h,w), coarse -> fine
    return dict(trace=trace, sd=sd, feats_cpu=feats_cpu, intrs=intrs, c2ws=c2ws, near_fars=near_fars, H=H, W=W, nv=nv,
                outputs=outputs, dev=dev)


def test_full_size_build_tables_are_bijections_in_reference_order(full_build):
    """Every stage: the index table and the coordinate list are inverse to each other, the coordinates are in the
    reference's boolean-mask order (stage 0: lattice order; later: parent order x pos_list child order, volume.py:35-52,
    165-166, surf.py:104-108) and the dense volume holds the stage's logits at the occupied voxels and the x2 trilinear
    upsample of the previous stage elsewhere (volume.py:99-121)."""
    tr = full_build["trace"]
    dims = [88, 176, 352, 704]
    for s in range(4):
        t = tr[s]
        D, coords, table = t["D"], t["coords"].long(), t["table"]
        assert D == dims[s] and tuple(table.shape) == (D, D, D)
        N = coords.shape[0]
        assert N > 100000 and int((table >= 0).sum()) == N
        rows = table[coords[:, 0], coords[:, 1], coords[:, 2]].long()
        assert torch.equal(rows, torch.arange(N, device=rows.device))
        assert int(coords.min()) >= 0 and int(coords.max()) < D
        if s == 0:
            key = (coords[:, 0] * D + coords[:, 1]) * D + coords[:, 2]
            assert bool((key[1:] > key[:-1]).all())
        else:
            par = tr[s - 1]["table"][coords[:, 0] // 2, coords[:, 1] // 2, coords[:, 2] // 2].long()
            assert int(par.min()) >= 0                                            # every voxel descends from a kept parent
            assert bool((par[1:] >= par[:-1]).all())
            off = coords - 2 * (coords // 2)
            pos = torch.tensor([0, 3, 2, 6, 1, 5, 4, 7], device=off.device)       # pos_list rank of offset (x,y,z) as 4x+2y+z
            rank = pos[off[:, 0] * 4 + off[:, 1] * 2 + off[:, 2]]
            same = par[1:] == par[:-1]
            assert bool((rank[1:][same] > rank[:-1][same]).all())
        # dense matching volume
        mvol = t["mvol"]
        assert torch.equal(mvol[coords[:, 0], coords[:, 1], coords[:, 2]], t["out"][:, 0])
        empty = table < 0
        if s == 0:
            assert not bool(empty.any()) or float(mvol[empty].abs().max()) == 0.0
        else:
            up = torch.nn.functional.interpolate(tr[s - 1]["mvol"][None, None], scale_factor=2, mode="trilinear",
                                                 align_corners=False)[0, 0]
            err = (mvol - up)[empty].abs().max()
            assert float(err) < 1e-5, float(err)
            del up
    # the pyramid concentrates on the surface: the finest stage keeps well under 1 % of its lattice
    assert tr[3]["coords"].shape[0] < 0.01 * 704 ** 3


def test_full_size_build_keep_rule_and_cost_volume_vs_oracle(full_build):
    """Strided subsets of every stage's candidates through the oracle's depth_filtering + back_proj_multiscale: the kept
    set must be the kernel's (exactly, up to a handful of voxels whose band / frustum test sits at fp32 rounding), and the
    [mean | var] rows must agree to 1e-3."""
    from oracle import surf_oracle as O
    fb = full_build
    tr, sd = fb["trace"], fb["sd"]
    for s in range(4):
        t = tr[s]
        D, table = t["D"], t["table"]
        if s == 0:
            cand = O.init_coords(D)[::37]
            keep1 = torch.ones(cand.shape[0], dtype=torch.bool)
        else:
            parents = t["parents"]
            stride = max(1, parents.shape[0] // 3000)
            sub = parents[::stride].float().cpu()
            cand, _ = O.up_sample(sub, torch.zeros(sub.shape[0], 1))
            keep1 = O.depth_filtering([d for d in t["pre_depths"].cpu()], cand, D, fb["intrs"], fb["c2ws"],
                                      torch.tensor(t["depth_range"]))
        cv, keep2 = O.back_proj_multiscale(sd, fb["feats_cpu"], cand[keep1], D, fb["intrs"], fb["c2ws"], s)
        kept = cand[keep1][keep2].long()
        dropped = torch.cat([cand[~keep1], cand[keep1][~keep2]]).long()
        tab = table.cpu()
        rows_k = tab[kept[:, 0], kept[:, 1], kept[:, 2]]
        rows_d = tab[dropped[:, 0], dropped[:, 1], dropped[:, 2]]
        flips = int((rows_k < 0).sum()) + int((rows_d >= 0).sum())
        assert kept.shape[0] > 500 and (s == 0 or dropped.shape[0] > 500), (s, kept.shape, dropped.shape)
        assert flips <= max(2, cand.shape[0] // 2000), (s, flips, cand.shape[0])
        ok = rows_k >= 0
        got = t["reg_in"][rows_k[ok].long().to(fb["dev"])][:, :8].cpu()
        ref = cv[keep2][ok]
        err = (got - ref).abs()
        assert bool((err <= 2e-5 + 1e-3 * ref.abs()).all()), (s, float(err.max()))


def test_full_size_matching_field_vs_oracle(full_build):
    """Random low-resolution pixels of the first and last view of every stage against the oracle's depth_render on the
    same dense matching volume and previous depth maps; the bilinear upsample against F.interpolate."""
    from oracle import surf_oracle as O
    fb = full_build
    tr = fb["trace"]
    H, W = fb["H"], fb["W"]
    g = torch.Generator().manual_seed(2)
    ratios, n_dep, levels = [1.0, 0.4, 0.1, 0.01], [128, 64, 32, 16], [4, 2, 2, 1]
    for s in range(4):
        t = tr[s]
        L = levels[s]
        h, w = H // L, W // L
        lr, full = t["depths_lr"], t["depths"]
        assert tuple(lr.shape) == (fb["nv"], h, w) and tuple(full.shape) == (fb["nv"], H, W)
        up = torch.nn.functional.interpolate(lr[:, None], size=(H, W), mode="bilinear", align_corners=False)[:, 0]
        assert float((up - full).abs().max()) < 2e-5
        mv = t["mvol"].cpu()
        tx, ty = torch.linspace(0, W - 1, w), torch.linspace(0, H - 1, h)
        for v in (0, fb["nv"] - 1):
            pick = torch.randint(0, h * w, (1500,), generator=g)
            px, py = tx[pick % w], ty[pick // w]
            pre = None if s == 0 else t["pre_depths"][v].cpu()
            ref = O.matching_field_pixels(px, py, v, fb["intrs"], fb["c2ws"], fb["near_fars"], mv, s, ratios, n_dep, pre)
            got = lr[v].reshape(-1)[pick.to(fb["dev"])].cpu()
            err = (got - ref).abs()
            assert bool((err <= 5e-5 + 1e-3 * ref.abs()).all()), (s, v, float(err.max()))
        del mv
    # the final reference-view depth of rays through the sphere's centre region lies between its front and its back crossing
    d0 = fb["outputs"]["depth_stage3"]
    centre = float(d0[H // 2, W // 2])
    assert 1.9 < centre < 3.1, centre      # front (2.0) or back (3.0) crossing of the r = 0.5 sphere, or between


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] at full size: 7 views, 1080x1920, 192 samples per ray (blend kernel with six source views)
# ----------------------------------------------------------------------------------------------------------------------


def test_config5_full_size_seven_views_1080p(full_scene):
    """Every pixel ray of a 1080x1920 reference view, 7 views, samples [96,48,32,16], on the 704^3 pyramid: invariants on
    the whole image and a strided ray subset against the CPU oracle."""
    from bench import model_conf
    from oracle import surf_oracle as O
    from surf_amd import synthetic
    from surf_amd.implicit_surface import ImplicitSurface, SceneVolumes
    fs = full_scene
    dev = fs["dev"]
    n_samples = [96, 48, 32, 16]
    H, W, nv = 1080, 1920, 7
    torch.manual_seed(0)
    model = ImplicitSurface(model_conf(n_samples)).to(dev)
    with torch.no_grad():
        model.deviation_network.variance.fill_(0.5)
    intrs, c2ws, near_fars = synthetic.ring_cameras(nv, H, W)
    imgs = synthetic.procedural_images(nv, H, W, 0, dev)
    feats = synthetic.feature_pyramid(nv, H, W, 0, dev)
    base = fs["scene"]
    from surf_amd import ops
    scene = SceneVolumes.from_device_layouts(base.mvol, base.sv.vols, base.sv.tables,
                                             [ops.pack_texel4(f.contiguous()) for f in feats], ops.pack_texel4(imgs),
                                             ops.Cameras(intrs, c2ws))
    rays_o, rays_d = synthetic.pixel_rays(intrs[0], c2ws[0], H, W, 1, dev)
    R = rays_o.shape[0]
    assert R == 1080 * 1920
    near = near_fars[0, 0].reshape(1, 1).repeat(R, 1).to(dev)
    far = near_fars[0, 1].reshape(1, 1).repeat(R, 1).to(dev)
    cols, deps, wsum = [], [], []
    chunk = 1 << 19
    for s0 in range(0, R, chunk):
        sl = slice(s0, s0 + chunk)
        o = model.render_scene(rays_o[sl], rays_d[sl], near[sl], far[sl], scene, 1.0, per_sample=True)
        cols.append(o["color_fine"])
        deps.append(o["render_depth"])
        wsum.append(o["weights"].sum(dim=1))
        assert torch.isfinite(o["weights"]).all() and float(o["weights"].min()) >= 0.0
        del o
    torch.cuda.synchronize()
    color, depth, wsum = torch.cat(cols), torch.cat(deps), torch.cat(wsum)
    assert torch.isfinite(color).all() and float(color.min()) >= -1e-5 and float(color.max()) <= 1.0 + 1e-4
    assert float(wsum.max()) <= 1.0 + 1e-3
    centre = (H // 2) * W + W // 2
    assert float(wsum[centre]) > 0.9
    # a strided subset of the same rays through the CPU oracle
    idx = torch.linspace(0, R - 1, 384).long()
    sd = {"implicit_surface." + k: v.detach().cpu() for k, v in model.state_dict().items()}
    tabs_c = [t.cpu().long() for t in base.sv.tables]
    ref = O.render(sd, rays_o.cpu()[idx], rays_d.cpu()[idx], near.cpu()[idx], far.cpu()[idx], base.mvol.cpu(),
                   [v[:, :7].cpu() for v in base.sv.vols], tabs_c, [(t >= 0).float() for t in tabs_c],
                   [f.cpu() for f in feats], imgs.cpu(), intrs, c2ws, n_samples, [1.0, 0.4, 0.1, 0.01], 256, 1.0)
    err_c = (color[idx.to(dev)].cpu() - ref["color_fine"]).abs()
    assert bool((err_c <= 3e-5 + 1e-3 * ref["color_fine"].abs()).all()), float(err_c.max())
    err_d = (depth[idx.to(dev)].cpu() - ref["render_depth"]).abs()
    assert bool((err_d <= 3e-5 + 1e-3 * ref["render_depth"].abs()).all()), float(err_d.max())
    assert float(ref["weights"].sum(1).max()) > 0.5
