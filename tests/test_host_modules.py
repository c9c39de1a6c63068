"""CPU checks of the host-side mirror: checkpoint key contract, conf parser, argument validation."""
import pytest
import torch

from surf_amd import conf as C


def _model():
    from bench import model_conf
    from surf_amd.implicit_surface import ImplicitSurface
    return ImplicitSurface(model_conf([64, 32, 16, 16]))


def test_reference_state_dict_loads_strict(weights):
    m = _model()
    sd = {k[len("implicit_surface."):]: v for k, v in weights.items() if k.startswith("implicit_surface.")}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert m.sdf_network.lin2.weight_v.shape == (101, 156)
    assert m.color_network.base_fc[0].weight.shape == (64, 57)
    assert abs(m.deviation_network.inv_s() - float(torch.exp(sd["deviation_network.variance"] * 10))) < 1e-3


def test_geometric_init_properties():
    """sdf_network.py:62-86: negative near the origin, positive far out, and zero feature influence at init."""
    from oracle import surf_oracle as O
    torch.manual_seed(0)
    m = _model()
    sd = {"implicit_surface." + k: v.detach() for k, v in m.state_dict().items()}
    layers = O.sdf_weights(sd)
    g = torch.Generator().manual_seed(1)
    pts = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=1) * torch.linspace(0.05, 1.0, 64)[:, None]
    sdf_a, _, _ = O.sdf_mlp(layers, pts, torch.randn(64, 28, generator=g))
    sdf_b, _, _ = O.sdf_mlp(layers, pts, torch.zeros(64, 28))
    assert torch.equal(sdf_a, sdf_b)
    r = pts.norm(dim=1)
    assert float(sdf_a[r < 0.2].max()) < 0 < float(sdf_a[r > 0.9].min())


def test_conf_parser_subset():
    c = C.parse_string("""
    general { base_exp_dir = <your output save path> }
    model {
        range_ratios = [1.0, 0.4, 0.1, 0.01]   # comment
        volume { base_volume_dim = [88, 88, 88] }
        implicit_surface { render { n_samples = [64, 32, 24, 16], perturb = 1.0 }
                           color_network { d_feature = 16
                             # d_feature = 128
                           } }
        flag = true
        a.b.c = 3
    }""")
    assert c["general.base_exp_dir"] == "<your output save path>"
    assert c.get_list("model.range_ratios") == [1.0, 0.4, 0.1, 0.01]
    assert c["model"]["volume"].get_list("base_volume_dim") == [88, 88, 88]
    assert c.get_float("model.implicit_surface.render.perturb") == 1.0
    assert c.get_bool("model.flag") is True and c.get_bool("model.has_vol", default=False) is False
    assert c.get_int("model.a.b.c") == 3
    assert dict(c["model.implicit_surface.color_network"]) == {"d_feature": 16}
    with pytest.raises(KeyError):
        c["model.nope"]


def test_unsupported_architectures_fail_loudly():
    from surf_amd.implicit_surface import BlendingNetwork, SDFNetworkSparse
    with pytest.raises(NotImplementedError):
        SDFNetworkSparse(d_hidden=256)
    with pytest.raises(NotImplementedError):
        BlendingNetwork(d_feature=128)


def test_ops_reject_cpu_tensors():
    from surf_amd import ops
    with pytest.raises(TypeError):
        ops.pack_texel4(torch.zeros(1, 3, 4, 4))


def test_sdf_precision_conf_key():
    """render.sdf_precision selects the SDF kernel; absent -> bf16x3 (exact split); unknown values are rejected."""
    from bench import model_conf
    from surf_amd import ops
    from surf_amd.implicit_surface import ImplicitSurface
    assert ImplicitSurface(model_conf([64, 32, 16, 16])).sdf_precision == "bf16x3"
    for prec in ops.SDF_PRECISIONS:
        assert ImplicitSurface(model_conf([8, 8], prec)).sdf_precision == prec
    with pytest.raises(ValueError):
        ImplicitSurface(model_conf([8, 8], "fp8"))
    # the packers of the split kernels run on the host: stream sizes identify the kernel a packed tensor belongs to
    m = ImplicitSurface(model_conf([8, 8]))
    layers = ops.sdf_effective_weights({k: v for k, v in m.state_dict().items()}, "sdf_network.")
    sizes = {p: len(ops.sdf_pack_weights_split_host(layers, p)) for p in ("bf16x3", "f16x2")}
    assert sizes["bf16x3"] > sizes["f16x2"] > 0
    for p, n in sizes.items():
        assert ops.sdf_packed_precision(torch.zeros(n, dtype=torch.uint8)) == p
    assert ops.sdf_packed_precision(torch.zeros(4, dtype=torch.float32)) == "f32"
    with pytest.raises(ValueError):
        ops.sdf_packed_precision(torch.zeros(7, dtype=torch.uint8))


def test_perturb_draw_order_matches_reference():
    """ImplicitSurface.render_scene draws one torch.rand([R, 1]) - 0.5 per stage on the CPU generator, stage 0 first
    (implicit_surface.py:276, 305): under the fixture's seed that reproduces the reference's jitters exactly."""
    import numpy as np
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "render_perturb.npz"))
    t_ref = torch.from_numpy(g["t_rand"])
    R, n_stage = t_ref.shape
    torch.manual_seed(4321)
    t = torch.cat([torch.rand([R, 1]) - 0.5 for _ in range(n_stage)], dim=1)      # the expression in render_scene
    assert torch.equal(t, t_ref)


def test_matching_field_jitter_draw_order_matches_the_reference():
    """MatchingField.draw_jitter consumes the CPU generator like the reference's depth_render calls (matching_field.py:33-35,
    129-133): views in order, only view 0 and src_idx draw, one `torch.rand([n_rays, 1]) - 0.5` per band."""
    import torch
    from surf_amd import conf
    from surf_amd.matching_field import MatchingField
    mf = MatchingField(conf.from_dict({"n_samples_depths": [8, 4, 4, 2], "n_importance_depths": [0] * 4, "up_sample_steps": [0] * 4,
                                       "depth_res_levels": [4, 2, 2, 1]}))
    nv, n_rays, src_idx = 5, 37, 3
    for n_bands in (1, 2):
        torch.manual_seed(123)
        jit = mf.draw_jitter(nv, n_rays, n_bands, src_idx)
        torch.manual_seed(123)
        ref = torch.zeros(nv, n_rays, 2)
        for i in range(nv):                      # the reference's loop: depth_render(perturb) for i == 0 or i == src_idx only
            if i == 0 or i == src_idx:
                for b in range(n_bands):
                    ref[i, :, b] = (torch.rand([n_rays, 1]) - 0.5)[:, 0]
        assert torch.equal(jit, ref)
        assert float(jit[1].abs().max()) == 0.0 and float(jit[0].abs().max()) > 0
    torch.manual_seed(5)
    same = mf.draw_jitter(nv, n_rays, 2, 0)      # src_idx == 0: a single drawing view
    assert float(same[1:].abs().max()) == 0.0


def test_agg_mlp_gradient_split_and_step_helpers():
    """Volume.assign_agg_grad splits costvol_backward's 49 floats (w1 | b1 | w2 | b2) onto agg_mlp's parameters and accumulates;
    training._sync_gradients is a no-op without a process group."""
    import torch
    from surf_amd import conf, training
    from surf_amd.volume import Volume
    vol = Volume(conf.from_dict({"base_volume_dim": [8, 8, 8]}))
    g = torch.arange(49, dtype=torch.float32)
    vol.assign_agg_grad(g)
    assert torch.equal(vol.agg_mlp[0].weight.grad, g[:32].reshape(8, 4))
    assert torch.equal(vol.agg_mlp[0].bias.grad, g[32:40])
    assert torch.equal(vol.agg_mlp[2].weight.grad, g[40:48].reshape(1, 8))
    assert torch.equal(vol.agg_mlp[2].bias.grad, g[48:49])
    vol.assign_agg_grad(g)
    assert torch.equal(vol.agg_mlp[2].bias.grad, 2 * g[48:49])
    host = vol.agg_host()
    assert host.shape == (49,)
    opt = torch.optim.SGD(vol.parameters(), lr=0.1)
    before = vol.agg_mlp[0].bias.grad.clone()
    training._sync_gradients(vol, opt)
    assert torch.equal(vol.agg_mlp[0].bias.grad, before)


def test_backward_entry_points_fail_loudly_without_a_recorded_forward():
    import pytest as _pytest
    import torch
    from surf_amd import conf
    from surf_amd.surf import SuRF
    from tests.golden.make_golden import MODEL_CONF
    cfg = dict(MODEL_CONF)
    cfg["reg_network"] = {"d_in": [8, 16, 16, 16], "d_base": [8] * 4, "d_out": [8] * 4}
    model = SuRF(conf.from_dict(cfg))
    model._train_tape = None
    with _pytest.raises(RuntimeError, match="record=True"):
        model.backward_volumes([torch.zeros(1, 7)] * 4)


def test_packed_blend_image_carries_its_precision():
    """The f16x2 and f32lds LDS images of the split blend kernel have the same size: the layout travels with the bytes
    (ops.PackedBlend), a raw byte tensor of an ambiguous size is refused instead of guessed, a wrong-sized one is refused."""
    import pytest as _pytest
    import torch
    from surf_amd import _lib, ops
    L = _lib.lib()
    sizes = {p: L.surf_blend_split_packed_bytes(pid) for p, pid in ops._BLEND_ID.items()}
    assert sizes["f16x2"] == sizes["f32lds"] != sizes["bf16x3"]
    raw16 = torch.zeros(sizes["f16x2"], dtype=torch.uint8)
    for precision in ("f16x2", "f32lds"):
        pk = ops.PackedBlend(raw16, precision)
        assert ops.blend_packed_precision(pk) == precision and ops.blend_packed_precision(pk.to("cpu")) == precision
    with _pytest.raises(ValueError, match="fits the layouts"):
        ops.blend_packed_precision(raw16)
    assert ops.blend_packed_precision(torch.zeros(sizes["bf16x3"], dtype=torch.uint8)) == "bf16x3"
    with _pytest.raises(ValueError):
        ops.PackedBlend(raw16, "bf16x3")
    with _pytest.raises(ValueError):
        ops.blend_packed_precision(torch.zeros(17, dtype=torch.uint8))


def test_device_packing_is_byte_identical_to_the_host_packers():
    """surf_amd.packing builds the operand images of the split SDF / blend kernels and the fp32 image of the second-order
    kernel with torch ops from the live parameters (on the device in production; the same ops on the CPU here).  The C ABI's
    host packers are the definition: byte identity on random weights, for every split layout, incl. a negative `s`."""
    import torch
    from bench import model_conf
    from surf_amd import ops, packing
    from surf_amd.implicit_surface import ImplicitSurface
    torch.manual_seed(3)
    m = ImplicitSurface(model_conf([64, 32, 16, 16]))
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.05)
        m.color_network.s.fill_(-0.37)
    sd = dict(m.state_dict())
    for prec in ("bf16x3", "f16x2"):
        assert torch.equal(ops.sdf_pack_weights_split(sd, "cpu", "sdf_network.", prec), packing.sdf_pack_split_device(m.sdf_network, prec)), prec
    assert torch.equal(ops.sdf_smooth_pack_weights(sd, "cpu", "sdf_network."), packing.sdf_pack_smooth_device(m.sdf_network))
    for prec in ("bf16x3", "f16x2", "f32lds"):
        a, b = ops.blend_pack_weights(sd, "cpu", "color_network.", prec), packing.blend_pack_split_device(m.color_network, prec)
        assert b.precision == prec and torch.equal(a.tensor, b.tensor), prec
    assert torch.equal(packing.blend_raw_device(m.color_network), torch.from_numpy(ops.blend_raw_weights({"implicit_surface.color_network." + k: v
                       for k, v in m.color_network.state_dict().items()})))
    assert packing.supported("bf16x3", "bf16x3") and not packing.supported("f32", "bf16x3")


def test_agg_mlp_host_copy_is_cached_per_parameter_version():
    import torch
    from surf_amd import conf
    from surf_amd.volume import Volume
    vol = Volume(conf.from_dict({"base_volume_dim": [8, 8, 8]}))
    a = vol.agg_host()
    assert vol.agg_host() is a and a.shape == (49,)
    with torch.no_grad():
        vol.agg_mlp[2].bias.add_(1.0)
    b = vol.agg_host()
    assert b is not a and abs(float(b[48] - a[48]) - 1.0) < 1e-6


def test_colgram_workspace_covers_both_launch_plans():
    """surf_colgram_workspace_floats must cover the matrix-core plan of surf_colgram_p for every row count: with
    train_precision = bf16 that plan is taken for short inputs too (advisor finding, round 3: rows = 2050, M = 32 wrote 36
    partials into 33).  The plan is restated here from colgram.hip (cg_plan / slab_rows); no compute call, no GPU."""
    from surf_amd import _lib
    L = _lib.lib()

    def mfma_partials(rows, M):
        nsplit = 4 // ((M + 31) // 32)
        per = -(-rows // (256 * nsplit))
        per = max(64, -(-per // 16) * 16)
        return nsplit * -(-rows // (per * nsplit))

    for rows in list(range(1, 300)) + [2049, 2050, 4095, 4096, 4097, 60864, 243456, 1000003]:
        for M in (1, 8, 31, 32, 33, 64, 65, 96, 101, 128):
            for N in (1, 8, 33, 57, 156):
                need = int(L.surf_colgram_workspace_floats(rows, M, N))
                assert need >= mfma_partials(rows, M) * M * (N + 1), (rows, M, N)


def test_precision_scope_restores_the_process_wide_policy():
    """A backward runs under the training-precision policy its forward was recorded with (advisor finding, round 3: the
    policy was a process global read at backward time, so model B's forward between A's forward and A's loss.backward()
    changed A's weight-gradient arithmetic)."""
    from surf_amd import ops
    ops.set_train_precision("fp32")
    with ops.precision_scope(1):
        assert ops.colgram_precision == 1
        with ops.precision_scope(None):         # no recorded policy: leave whatever is set
            assert ops.colgram_precision == 1
        with ops.precision_scope(0):
            assert ops.colgram_precision == 0
        assert ops.colgram_precision == 1
    assert ops.colgram_precision == 0
    try:
        with ops.precision_scope(1):
            raise KeyError("x")
    except KeyError:
        pass
    assert ops.colgram_precision == 0


def test_small_zero_pool_hands_out_disjoint_zero_slices_once():
    """ops.small_zeros (round 5): the weight-gradient accumulators of a training step come from one pre-zeroed block instead of a
    fill launch each - every slice is zero, 256-byte aligned, disjoint from every other one, and never handed out twice; a request
    above the limit, or a block that has run out, gets fresh zeros."""
    from surf_amd import ops
    dev = torch.device("cpu")
    pool = ops._ZeroPool()
    a = pool.take((27, 16, 8), dev)
    b = pool.take((8,), dev)
    c = pool.take((0, 4), dev)
    assert a.shape == (27, 16, 8) and b.shape == (8,) and c.shape == (0, 4)
    assert float(a.abs().sum()) == 0.0 and float(b.abs().sum()) == 0.0
    assert a.data_ptr() % 256 == b.data_ptr() % 256                      # slices start on 64-float boundaries of one block
    a.fill_(1.0)
    assert float(b.abs().sum()) == 0.0                                    # disjoint
    big = pool.take((pool.LIMIT + 1,), dev)
    assert big.numel() == pool.LIMIT + 1 and float(big.abs().sum()) == 0.0
    blk = pool._blocks[(dev.type, dev.index)][0]
    n_before = pool._blocks[(dev.type, dev.index)][1]
    for _ in range(pool.BLOCK // pool.LIMIT + 2):                         # exhaust the block: a new one takes over
        t = pool.take((pool.LIMIT,), dev)
        assert float(t.abs().sum()) == 0.0
        t.fill_(2.0)
    assert pool._blocks[(dev.type, dev.index)][0] is not blk and n_before > 0
    assert float(b.abs().sum()) == 0.0 and float(a.min()) == 1.0          # the old block lives on through its views


def test_masked_l1_host_expression_for_the_three_mask_forms():
    """Loss._masked_l1 off the GPU (and for mismatched shapes on it) is the reference's expression
    sum(|pred - target| mask) / (sum(mask) + 1e-8), with the mask given as floats, booleans or "target>0" (loss.py:71-93)."""
    from surf_amd.losses import Loss
    g = torch.Generator().manual_seed(4)
    pred, target = torch.randn(7, 9, generator=g), torch.randn(7, 9, generator=g)
    fm = (torch.rand(7, 9, generator=g) < 0.6).float()
    want = ((pred - target).abs() * fm).sum() / (fm.sum() + 1e-8)
    assert torch.equal(Loss._masked_l1(pred, target, fm), want)
    assert torch.equal(Loss._masked_l1(pred, target, fm > 0), want)
    pos = (target > 0).float()
    assert torch.equal(Loss._masked_l1(pred, target, "target>0"), ((pred - target).abs() * pos).sum() / (pos.sum() + 1e-8))
    assert float(Loss._masked_l1(pred, target, torch.zeros(7, 9))) == 0.0
