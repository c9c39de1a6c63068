"""Time the SDF training kernels (surf_sdf_backward, surf_sdf_smooth_backward, surf_sdf_smooth) on n points of the bench pyramid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import synthetic, ops
from bench import model_conf
from surf_amd.implicit_surface import ImplicitSurface
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d = torch.device("cuda:0")
torch.manual_seed(0)
model = ImplicitSurface(model_conf([64, 32, 16, 16])).to(d)
vols, tabs, mvol = synthetic.sphere_pyramid(88, d)
sv = ops.SparseVolumes(vols[::-1], tabs[::-1])
w = model.smooth_weights(d)
pts = ((torch.rand(n, 3, device=d) * 2 - 1) * 0.6).contiguous()
ybar = torch.randn(n, device=d); gbar = torch.randn(n, 3, device=d); sbar = torch.randn(n, 3, device=d)
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
L = ops._lib.lib()
in_v = torch.empty(7, n, 160, device=d); in_d = torch.empty(7, n, 160, device=d); tb = torch.empty(6, n, 128, device=d); tdb = torch.empty(6, n, 128, device=d)
dv = [torch.zeros_like(v) for v in sv.vols]
def bwd():
    rc = L.surf_sdf_backward(ops._p(pts), ops._p(ybar), ops._p(gbar), n, sv._vp, sv._tp, sv._dp, sv.n, ops._ptr_array(dv), ops._p(w), ops._p(in_v), ops._p(in_d), ops._p(tb), ops._p(tdb), ops._stream())
    assert rc == 0
xin = torch.empty(7, 4, n, 160, device=d); ab = torch.empty(6, 4, n, 128, device=d)
def smb():
    rc = L.surf_sdf_smooth_backward(ops._p(pts), ops._p(sbar), n, sv._vp, sv._tp, sv._dp, sv.n, ops._ptr_array(dv), ops._p(w), ops._p(xin), ops._p(ab), ops._stream())
    assert rc == 0
print(f"sdf_smooth_backward kernels {t(smb):.3f} ms", end="   ")
print(f"{os.environ.get('SURF_HIP_LIB', 'default')}: n {n}  sdf_backward kernels {t(bwd):.3f} ms   sdf_smooth {t(lambda: ops.sdf_smooth(pts, sv, w, want_grad=True)):.3f} ms")
