"""Race screen of the split kernels as a GPU test (was scripts/stress_sdf.py): their LDS weight ring is retired by counted
`s_waitcnt vmcnt(N)`, so a miscounted or reordered vector-memory operation shows up as an occasional wrong tile, not as a
crash.  Many launches of a multi-round, masked, compacted problem, every launch compared with the fp32-MFMA kernel."""
import pytest
import torch

pytestmark = pytest.mark.gpu

LAUNCHES = 40


@pytest.fixture(scope="module")
def problem():
    from bench import model_conf
    from surf_amd import ops, synthetic
    from surf_amd.implicit_surface import ImplicitSurface
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = ImplicitSurface(model_conf([64, 32, 16, 16])).to(dev)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for l in range(1, 7):                       # let the sparse-volume channels matter (zero at geometric init)
            lin = getattr(model.sdf_network, f"lin{l}")
            lin.weight_v[:, -28:] += (0.05 * torch.randn(lin.weight_v.shape[0], 28, generator=g)).to(dev)
    vols, tabs, mvol = synthetic.sphere_pyramid(44, dev)
    sv = ops.SparseVolumes(vols[::-1], tabs[::-1])
    n = 600_000
    pts = ((torch.rand(n, 3, generator=g) * 2 - 1) * 0.6).to(dev).contiguous()
    mask = (torch.arange(n) % 5 != 0).to(torch.uint8).to(dev)
    sd = {k: v for k, v in model.state_dict().items()}
    w32 = ops.sdf_pack_weights(sd, dev, "sdf_network.")
    s_ref, g_ref = ops.sdf_mlp(pts, sv, w32, mask=mask)
    return dict(dev=dev, sv=sv, pts=pts, mask=mask, sd=sd, s_ref=s_ref, g_ref=g_ref)


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_split_sdf_kernels_race_screen(problem, precision):
    from surf_amd import ops
    p = problem
    w = ops.sdf_pack_weights_split(p["sd"], p["dev"], "sdf_network.", precision)
    bad, worst = 0, 0.0
    for it in range(LAUNCHES):
        s, gr = ops.sdf_mlp(p["pts"], p["sv"], w, mask=p["mask"])
        es = float((s - p["s_ref"]).abs().max())
        eg = float((gr - p["g_ref"]).abs().max())
        worst = max(worst, es, eg)
        bad += int(es > 1e-4 or eg > 1e-3)
    assert bad == 0, f"{precision}: {bad}/{LAUNCHES} launches off, worst |diff| vs the fp32 kernel {worst:.3g}"
    # forward-only variant (two workgroups per CU for f16x2): same screen on the SDF values
    for it in range(LAUNCHES // 2):
        s, _ = ops.sdf_mlp(p["pts"], p["sv"], w, mask=p["mask"], want_grad=False)
        assert float((s - p["s_ref"]).abs().max()) <= 1e-4
