"""Experiment (round 6): do the thin sparse convolutions get faster when the gathered rows are bf16 (half the bytes)?
Times spconv_pipe_kernel<Cin, Cout, SUBM> on the bench scene's stage lattices with fp32 rows and with a bf16 copy of the rows."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import _lib, ops, synthetic
d = torch.device("cuda:0")
L = ctypes.CDLL(_lib.LIB_PATH)
f = L.surf_x_spconv_rows16
f.restype = ctypes.c_int
vols, tabs, mvol = synthetic.sphere_pyramid(88, d)
for st in (1, 2, 3):
    tab = tabs[st]
    D = tab.shape[0]
    coords = (tab >= 0).nonzero().to(torch.int32).contiguous()
    n = coords.shape[0]
    order = tab[coords[:, 0].long(), coords[:, 1].long(), coords[:, 2].long()].long()
    c2 = torch.empty_like(coords); c2[order] = coords          # row i of the table order
    for cin, cout in ((16, 8), (8, 16), (16, 16)):
        x = torch.randn(n, cin, device=d)
        x16 = x.to(torch.bfloat16).contiguous()
        w = torch.randn(27, cin, cout, device=d) / (27 * cin) ** 0.5
        out = torch.empty(n, cout, device=d)
        def run32():
            return ops.spconv(x, tab, c2, ops.SUBM, w)
        def run16():
            rc = f(ctypes.c_void_p(x16.data_ptr()), cin, ctypes.c_void_p(tab.data_ptr()), D, ctypes.c_void_p(c2.data_ptr()), ctypes.c_int64(n),
                   0, ctypes.c_void_p(w.data_ptr()), cout, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc
            return out
        ref = ops.spconv(x16.float(), tab, c2, ops.SUBM, w)
        got = run16()
        torch.cuda.synchronize()
        err = float((got - ref).abs().max())
        res = []
        for fn in (run32, run16):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): fn()
            b.record(); torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / 10)
        print(f"stage {st} D={D} sites {n} <{cin},{cout}> fp32 rows {res[0]:.3f} ms  bf16 rows {res[1]:.3f} ms  (max |diff| vs fp32 kernel on the rounded rows {err:.1e})")
