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
    out, oc, D2 = O.spconv_down(feat, coords, D, kernel)
    # output sites are the even positions inside the bounding box: 0,2,4,6 per axis
    assert D2 == D // 2 + 1 and oc.shape[0] == 4 ** 3
    dense = feat.view(D, D, D, cin).permute(3, 0, 1, 2)[None]
    ref = F.conv3d(dense, _dense_weight(kernel), padding=1, stride=2)[0]        # (cout,4,4,4), centred on even sites
    assert torch.allclose(out, ref[:, oc[:, 0], oc[:, 1], oc[:, 2]].t(), atol=1e-5)
    # <down(x), y> == <x, up(y)> with the transposed kernel
    y = torch.randn(oc.shape[0], cout, generator=g)
    up = O.spconv_up(y, oc, coords, D, kernel.transpose(1, 2).contiguous())
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
