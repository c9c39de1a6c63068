"""Self-consistency of the oracle's sparse-convolution restatement (row a5, parity unpinned): on a fully
occupied lattice a submanifold 3^3 conv equals a dense zero-padded conv3d with the same kernel, a k3/s2 down
conv equals the strided dense conv, and the transposed conv is the adjoint of the down conv."""
import torch
import torch.nn.functional as F

from oracle import surf_oracle as O


def _dense_weight(kernel):
    # kernel (27, Cin, Cout), offsets x fastest / z slowest -> conv3d weight (Cout, Cin, kx, ky, kz) on [x][y][z] volumes
    cin, cout = kernel.shape[1:]
    w = kernel.view(3, 3, 3, cin, cout)            # [z][y][x]
    return w.permute(4, 3, 2, 1, 0).contiguous()   # (cout, cin, x, y, z)


def test_subm_equals_dense_conv_on_full_grid():
    g = torch.Generator().manual_seed(0)
    D, cin, cout = 6, 8, 16
    coords = O.init_coords(D).long()
    feat = torch.randn(D ** 3, cin, generator=g)
    kernel = torch.randn(27, cin, cout, generator=g) * 0.1
    out = O.spconv_subm(feat, coords, D, kernel)
    dense = feat.view(D, D, D, cin).permute(3, 0, 1, 2)[None]
    ref = F.conv3d(dense, _dense_weight(kernel), padding=1)[0].permute(1, 2, 3, 0).reshape(-1, cout)
    assert torch.allclose(out, ref, atol=1e-5)


def test_down_equals_strided_dense_conv_and_up_is_its_adjoint():
    g = torch.Generator().manual_seed(1)
    D, cin, cout = 8, 8, 16
    coords = O.init_coords(D).long()
    feat = torch.randn(D ** 3, cin, generator=g)
    kernel = torch.randn(27, cin, cout, generator=g) * 0.1
    out, oc, D2 = O.spconv_down(feat, coords, D, kernel, "dilate")
    # (rule 'dilate') output sites are the even positions inside the bounding box: 0,2,4,6 per axis
    assert D2 == D // 2 + 1 and oc.shape[0] == 4 ** 3
    dense = feat.view(D, D, D, cin).permute(3, 0, 1, 2)[None]
    ref = F.conv3d(dense, _dense_weight(kernel), padding=1, stride=2)[0]        # (cout,4,4,4), centred on even sites
    assert torch.allclose(out, ref[:, oc[:, 0], oc[:, 1], oc[:, 2]].t(), atol=1e-5)
    # <down(x), y> == <x, up(y)> with the transposed kernel
    y = torch.randn(oc.shape[0], cout, generator=g)
    up = O.spconv_up(y, oc, coords, D, kernel.transpose(1, 2).contiguous(), "dilate")
    assert torch.allclose((out * y).sum(), (feat * up).sum(), rtol=1e-4)


def test_unet_runs_on_a_ragged_voxel_set():
    g = torch.Generator().manual_seed(2)
    D = 12
    coords = O.init_coords(D).long()
    coords = coords[torch.rand(coords.shape[0], generator=g) < 0.3]
    from surf_amd import conf
    from surf_amd.reg_network import SparseCostRegNetList
    torch.manual_seed(0)
    net = SparseCostRegNetList(conf.from_dict({"d_in": [8, 16], "d_out": [8, 8], "d_base": [8, 8]})).eval()
    sd = {"reg_network." + k: v.detach() for k, v in net.state_dict().items()}
    out, mid = O.sparse_unet(sd, torch.randn(coords.shape[0], 16, generator=g), coords, D, 1)
    assert out.shape == (coords.shape[0], 8) and mid.shape == (coords.shape[0], 8)
    assert torch.isfinite(out).all() and float(mid.abs().max()) > 0


def test_pad0_rule_equals_unpadded_strided_dense_conv_and_up_is_its_adjoint():
    """rule 'pad0' (torchsparse >= 2.1 builds strided maps spconv-style, out = (in + 2 pad - k) / stride, and
    the reference passes no padding): on a full lattice the down conv IS F.conv3d(stride 2, padding 0) - output lattice
    (D - 3) // 2 + 1, window 2q + {0,1,2}^3 - and the transposed conv inheriting the map is its adjoint; on a ragged set every
    output site has an input in its window and every input inside the covered range reaches an output."""
    g = torch.Generator().manual_seed(3)
    for D in (8, 9):
        cin, cout = 4, 8
        coords = O.init_coords(D).long()
        feat = torch.randn(D ** 3, cin, generator=g)
        kernel = torch.randn(27, cin, cout, generator=g) * 0.1
        out, oc, D2 = O.spconv_down(feat, coords, D, kernel, "pad0")
        assert D2 == (D - 3) // 2 + 1 and oc.shape[0] == D2 ** 3
        dense = feat.view(D, D, D, cin).permute(3, 0, 1, 2)[None]
        ref = F.conv3d(dense, _dense_weight(kernel), padding=0, stride=2)[0]
        assert ref.shape[1:] == (D2, D2, D2)
        assert torch.allclose(out, ref[:, oc[:, 0], oc[:, 1], oc[:, 2]].t(), atol=1e-5)
        y = torch.randn(oc.shape[0], cout, generator=g)
        up = O.spconv_up(y, oc, coords, D, kernel.transpose(1, 2).contiguous(), "pad0")
        assert torch.allclose((out * y).sum(), (feat * up).sum(), rtol=1e-4)
    D = 17
    coords = O.init_coords(D).long()
    coords = coords[torch.rand(coords.shape[0], generator=g) < 0.05]
    oc, D2 = O.down_coords(coords, D, "pad0")
    assert D2 == 8 and bool(((oc >= 0) & (oc < D2)).all())
    table = O._build_table(coords, D)
    hit = torch.zeros(oc.shape[0], dtype=torch.bool)
    for o in O.KOFFS:
        hit |= O._lookup(table, oc * 2 + torch.tensor(o) + 1, D) >= 0
    assert bool(hit.all())
    t2 = O._build_table(oc, D2)
    inside = ((coords <= 2 * (D2 - 1) + 2).all(dim=1))
    reached = torch.zeros(coords.shape[0], dtype=torch.bool)
    for o in O.KOFFS:
        c = coords - (torch.tensor(o) + 1)
        ok = ((c % 2) == 0).all(dim=1)
        reached |= ok & (O._lookup(t2, torch.div(c, 2, rounding_mode="floor"), D2) >= 0)
    assert bool(reached[inside].all())


def test_unet_runs_with_every_down_rule():
    g = torch.Generator().manual_seed(4)
    D = 20
    coords = O.init_coords(D).long()
    coords = coords[torch.rand(coords.shape[0], generator=g) < 0.3]
    from surf_amd import conf
    from surf_amd.reg_network import SparseCostRegNetList
    torch.manual_seed(0)
    net = SparseCostRegNetList(conf.from_dict({"d_in": [8, 16], "d_out": [8, 8], "d_base": [8, 8]})).eval()
    sd = {"reg_network." + k: v.detach() for k, v in net.state_dict().items()}
    x = torch.randn(coords.shape[0], 16, generator=g)
    outs = {rule: O.sparse_unet(sd, x, coords, D, 1, rule=rule)[0] for rule in ("dilate", "floor", "pad0")}
    assert all(o.shape == (coords.shape[0], 8) and torch.isfinite(o).all() for o in outs.values())
    assert float((outs["pad0"] - outs["dilate"]).abs().max()) > 1e-3        # the rules are different networks


def test_kernel_order_and_transposed_pairing_hedges():
    """the two other torchsparse recollections a loaded checkpoint depends on, hedged like `rule`.
    zfast == the xfast network on coordinates with x and z exchanged (a slice enumeration IS an axis naming); mirrored == the up
    layers' kernels flipped (slice 26 - k); the host module's permutation is the oracle's; all four combinations are different
    networks; both permutations are involutions (the weight gradient is mapped back with the same index)."""
    from surf_amd import conf
    from surf_amd.reg_network import SparseCostRegNetList, slice_permutation
    g = torch.Generator().manual_seed(7)
    D = 20
    coords = O.init_coords(D).long()
    coords = coords[torch.rand(coords.shape[0], generator=g) < 0.3]
    torch.manual_seed(1)
    net = SparseCostRegNetList(conf.from_dict({"d_in": [8, 16], "d_out": [8, 8], "d_base": [8, 8]})).eval()
    sd = {"reg_network." + k: v.detach() for k, v in net.state_dict().items()}
    x = torch.randn(coords.shape[0], 16, generator=g)
    outs = {}
    for order in ("xfast", "zfast"):
        for pairing in ("same", "mirrored"):
            outs[order, pairing] = O.sparse_unet(sd, x, coords, D, 1, kernel_order=order, transposed_pairing=pairing)[0]
            for mirrored in (False, True):
                perm = O.slice_permutation(order, mirrored)
                assert perm.tolist() == slice_permutation(order, mirrored)
                assert torch.equal(perm[perm], torch.arange(27))
    keys = list(outs)
    for i in range(4):
        for j in range(i + 1, 4):
            assert float((outs[keys[i]] - outs[keys[j]]).abs().max()) > 1e-3, (keys[i], keys[j])
    # zfast on (x, y, z) == xfast on (z, y, x): level-0 rows keep their order, so the outputs compare row by row
    swapped = O.sparse_unet(sd, x, coords.flip(1), D, 1, kernel_order="xfast")[0]
    assert torch.allclose(outs["zfast", "same"], swapped, atol=1e-5)
    # mirrored == flipped kernels in the three transposed layers
    sd_f = dict(sd)
    for i in (7, 9, 11):
        k = f"reg_network.nets.1.conv{i}.net.0.kernel"
        sd_f[k] = sd[k].flip(0)
    assert torch.allclose(outs["xfast", "mirrored"], O.sparse_unet(sd_f, x, coords, D, 1)[0], atol=1e-6)
    # the host module: conf keys, set_conventions, validation, the warning when a checkpoint arrives with unnamed conventions
    import warnings
    import pytest
    named = SparseCostRegNetList(conf.from_dict({"d_in": [8], "d_out": [8], "d_base": [8], "down_rule": "floor",
                                                 "kernel_order": "zfast", "transposed_pairing": "mirrored"}))
    assert named.conventions() == {"down_rule": "floor", "kernel_order": "zfast", "transposed_pairing": "mirrored"}
    assert named.nets[0].conv7._perm == slice_permutation("zfast", True) and named.nets[0].conv0._perm == slice_permutation("zfast")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        named.load_state_dict(named.state_dict())
    with pytest.warns(UserWarning, match="transposed_pairing"):
        net.load_state_dict(net.state_dict())
    net.set_conventions(kernel_order="zfast")
    assert net.conventions() == {"down_rule": "pad0", "kernel_order": "zfast", "transposed_pairing": "same"}
    assert net.nets[1].conv0._perm is not None and net.nets[1].conv0._wprep is None
    with pytest.raises(ValueError):
        SparseCostRegNetList(conf.from_dict({"d_in": [8], "d_out": [8], "d_base": [8], "kernel_order": "yfast"}))
    with pytest.raises(ValueError):
        SparseCostRegNetList(conf.from_dict({"d_in": [8], "d_out": [8], "d_base": [8], "transposed_pairing": "flipped"}))
