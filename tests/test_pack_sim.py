"""CPU check of the MFMA weight re-layouts (host code of libsurf_hip.so) without a GPU.

A lane-level numpy model of v_mfma_f32_32x32x2_f32 (A: lane l holds A[l&31][l>>5]; B: B[l>>5][l&31];
D register r of lane l = D[(r&3) + 8(r>>2) + 4(l>>5)][l&31]) replays the data flow of sdf_mlp.hip and
blend.hip on the buffers produced by surf_sdf_pack_weights / surf_blend_pack_weights and compares with
the oracle.  It pins the k-permutation / row-map logic of the packers; the kernels themselves are
checked on the GPU (tests/test_hip_parity.py).
"""
import numpy as np
import pytest
import torch

from oracle import surf_oracle as O
from surf_amd import ops
from tests.golden_cfg import pipeline_views

LANES = np.arange(64)
J, H = LANES & 31, LANES >> 5
ROW = np.array([(r & 3) + 8 * (r >> 2) for r in range(16)])


def mma_seg(acc, b, packed, off, NQ, NT):
    """acc: list of NT arrays (64,16); b (64, >=NQ*4); packed [q][t][lane][4] at float offset off."""
    for q in range(NQ):
        for i in range(4):
            step = 4 * q + i
            B = np.zeros((2, 32), np.float64)
            B[H, J] = b[:, step]
            for t in range(NT):
                a = packed[off + ((q * NT + t) * 64 + LANES) * 4 + i]
                A = np.zeros((32, 2), np.float64)
                A[J, H] = a
                D = A @ B
                acc[t] += D[ROW[None, :] + 4 * H[:, None], J[:, None]]


def rows16(packed, off):
    """[h][16] table -> (64,16) per-lane view."""
    return packed[off + H[:, None] * 16 + np.arange(16)[None, :]].astype(np.float64)


def softplus(t):
    bt = t * 100.0
    hv = np.where(bt > 20, t, np.log1p(np.exp(np.minimum(bt, 20.0))) / 100.0)
    sv = np.where(bt > 20, 1.0, 1.0 / (1.0 + np.exp(-np.minimum(bt, 20.0))))
    return hv, sv


# offsets mirror sdf_mlp.hip
FWD_NQ = [4, 20, 20, 24, 20, 20]
BWD_NT = [1, 5, 5, 6, 5, 5]
FWD_OFF = np.concatenate([[0], np.cumsum([q * 4 * 256 for q in FWD_NQ])])
BWD_OFF = FWD_OFF[-1] + np.concatenate([[0], np.cumsum([16 * t * 256 for t in BWD_NT])])
BIAS_OFF = BWD_OFF[-1]
W6H_OFF = BIAS_OFF + 6 * 4 * 2 * 16
W6P_OFF = W6H_OFF + 128
B6_OFF = W6P_OFF + 32


def sim_sdf(packed, pts, phi_full, jphi_full, bias_step=False):
    """32 points -> sdf (32,), grad (32,3) through the packed buffers.  bias_step: accumulators start at zero and
    the bias comes in through k-step 14 of the last input segment with a constant-one operand (schedule v2)."""
    n = pts.shape[0]
    assert n == 32
    x = pts[J].astype(np.float64)
    e_all = np.zeros((64, 28))
    je_all = np.zeros((64, 28))
    e_all[:, :27] = O.posenc(torch.from_numpy(pts)).numpy()[J]
    je_all[:, :27] = O.posenc_jac_diag(torch.from_numpy(pts)).numpy()[J]
    sel = 14 * H[:, None] + np.arange(14)[None, :]
    e = np.zeros((64, 16)); e[:, :14] = np.take_along_axis(e_all, sel, 1)
    je = np.take_along_axis(je_all, sel, 1)
    phi = np.zeros((64, 16)); phi[:, :14] = np.take_along_axis(phi_full[J], sel, 1)
    Jl = np.stack([np.take_along_axis(jphi_full[J][:, :, a], sel, 1) for a in range(3)], -1)  # (64,14,3)

    if bias_step:
        e[:, 14] = 1.0
        phi[:, 14] = 1.0

    def bias(l):
        if bias_step:
            return [np.zeros((64, 16)) for t in range(4)]
        return [rows16(packed, BIAS_OFF + ((l * 4 + t) * 2) * 16) for t in range(4)]

    S = []
    acc = bias(0)
    mma_seg(acc, e, packed, FWD_OFF[0], 4, 4)
    for l in range(1, 6):
        hv, sv = softplus(np.concatenate(acc, 1))
        S.append(sv)
        acc = bias(l)
        mma_seg(acc, hv, packed, FWD_OFF[l], 16, 4)
        if l == 3:
            mma_seg(acc, e, packed, FWD_OFF[l] + 16 * 4 * 256, 4, 4)
            mma_seg(acc, phi, packed, FWD_OFF[l] + 20 * 4 * 256, 4, 4)
        else:
            mma_seg(acc, phi, packed, FWD_OFF[l] + 16 * 4 * 256, 4, 4)
    hv, sv = softplus(np.concatenate(acc, 1))
    w6h = packed[W6H_OFF + H[:, None] * 64 + np.arange(64)[None, :]]
    w6p = rows16(packed, W6P_OFF)
    y0 = (w6h * hv).sum(1) + (w6p * phi).sum(1)
    y0 = y0 + y0[LANES ^ 32] + packed[B6_OFF]
    delta = sv * w6h
    accP = w6p.copy()
    accE = np.zeros((64, 16))
    for l in (5, 4, 3, 2, 1):
        NT = BWD_NT[l]
        G = [np.zeros((64, 16)) for _ in range(4)]
        G += [accE, accP] if NT == 6 else [accP]
        mma_seg(G, delta, packed, BWD_OFF[l], 16, NT)
        delta = S[l - 1] * np.concatenate(G[:4], 1)
    G = [accE]
    mma_seg(G, delta, packed, BWD_OFF[0], 16, 1)
    g3 = np.zeros((64, 3))
    ch = 14 * H[:, None] + np.arange(14)[None, :]
    for a in range(3):
        g3[:, a] = (accE[:, :14] * je * ((ch % 3) == a)).sum(1) + (accP[:, :14] * Jl[:, :, a]).sum(1)
    g3 = g3 + g3[LANES ^ 32]
    return y0[:32], g3[:32]


def test_sdf_pack_matches_oracle(weights, golden_pipe, golden_render):
    vols, tabs, _, _ = pipeline_views(golden_pipe)
    pts = golden_render["pts"][100:132].contiguous()
    layers = O.sdf_weights(weights)
    packed = ops.sdf_pack_weights_host(ops.sdf_effective_weights(weights))
    phi, jphi = O.lookup_sparse_volume(pts, vols, tabs, with_jac=True)
    sdf_o, grad_o, _ = O.sdf_mlp(layers, pts, phi, jphi)
    sdf_s, grad_s = sim_sdf(packed.astype(np.float64), pts.numpy(), phi.numpy().astype(np.float64),
                            jphi.numpy().astype(np.float64))
    assert np.abs(sdf_s - sdf_o.numpy()).max() < 2e-5
    assert np.abs(grad_s - grad_o.numpy()).max() < 2e-4
    assert np.abs(grad_o.numpy()).max() > 0.1
    sdf_b, grad_b = sim_sdf(packed.astype(np.float64), pts.numpy(), phi.numpy().astype(np.float64),
                            jphi.numpy().astype(np.float64), bias_step=True)
    assert np.abs(sdf_b - sdf_o.numpy()).max() < 2e-5
    assert np.abs(grad_b - grad_o.numpy()).max() < 2e-4


# ---- blend ----------------------------------------------------------------------------------------
LNQ = [1, 2, 6, 3, 8, 4, 4, 4, 5, 2]
LNT = [1, 1, 2, 2, 1, 1, 1, 1, 1, 1]
MMA_OFF = np.concatenate([[0], np.cumsum([q * t * 256 for q, t in zip(LNQ, LNT)])])
(L_RD0, L_RD2, L_B0S, L_B0V, L_B2, L_V0, L_V2, L_W0, L_R0, L_R2) = range(10)
BB = MMA_OFF[-1]
(B_RD0, B_RD2, B_B0_T0, B_B0_T1, B_B2, B_V0, B_V2, B_W0, B_R0, B_R2) = range(10)
DOT = BB + 10 * 32
SCAL = DOT + 3 * 32


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def sigm(x):
    return 1.0 / (1.0 + np.exp(-x))


def sim_blend(P, rgb_feat, ray_diff, mask):
    """32 points, NS views: rgb_feat (32,NS,19), ray_diff (32,NS,4), mask (32,NS)."""
    NS = rgb_feat.shape[1]
    loc = np.where(H[:, None] == 0, np.arange(12)[None, :], 11 + np.arange(12)[None, :])
    valid = np.where(H[:, None] == 0, np.arange(12)[None, :] < 11, np.arange(12)[None, :] < 8)

    def local(v):  # (32,19) -> (64,12)
        full = np.concatenate([v, np.zeros((32, 12))], 1)[J]
        return np.take_along_axis(full, np.minimum(loc, 30), 1) * valid

    floc, rd, mk, ex = [], [], [], []
    for v in range(NS):
        r = ray_diff[:, v][J]
        bin_ = np.zeros((64, 4))
        bin_[:, 0] = np.where(H == 0, r[:, 0], r[:, 1])
        bin_[:, 1] = np.where(H == 0, r[:, 2], r[:, 3])
        a1 = [rows16(P, BB + B_RD0 * 32)]
        mma_seg(a1, bin_, P, MMA_OFF[L_RD0], 1, 1)
        h8 = elu(a1[0][:, :8])
        a2 = [rows16(P, BB + B_RD2 * 32)]
        mma_seg(a2, h8, P, MMA_OFF[L_RD2], 2, 1)
        f = local(rgb_feat[:, v])
        f[:, :11] += elu(a2[0][:, :11])
        floc.append(f); rd.append(r); mk.append(mask[:, v][J].astype(np.float64))
        ex.append(np.exp(P[SCAL] * (r[:, 3] - 1.0)))
    emin = np.min(ex, 0)
    wv = [(ex[v] - emin) * mk[v] for v in range(NS)]
    ws = sum(wv)
    wv = [w / (ws + 1e-8) for w in wv]
    mean = sum(floc[v] * wv[v][:, None] for v in range(NS))
    var = sum(wv[v][:, None] * (floc[v] - mean) ** 2 for v in range(NS))
    mvb = np.concatenate([mean, var], 1)
    G0 = [rows16(P, BB + B_B0_T0 * 32), rows16(P, BB + B_B0_T1 * 32)]
    mma_seg(G0, mvb, P, MMA_OFF[L_B0S], 6, 2)
    dvis, dvis2, drgb4 = rows16(P, DOT), rows16(P, DOT + 32), rows16(P, DOT + 64)
    rgb_in = rgb_feat[:, :, :3]
    logits = []
    for v in range(NS):
        a64 = [G0[0].copy(), G0[1].copy()]
        mma_seg(a64, floc[v], P, MMA_OFF[L_B0V], 3, 2)
        h32 = elu(np.concatenate(a64, 1))
        ax = [rows16(P, BB + B_B2 * 32)]
        mma_seg(ax, h32, P, MMA_OFF[L_B2], 8, 1)
        x = elu(ax[0])
        at = [rows16(P, BB + B_V0 * 32)]
        mma_seg(at, x * wv[v][:, None], P, MMA_OFF[L_V0], 4, 1)
        t16 = elu(at[0])
        ar = [rows16(P, BB + B_V2 * 32)]
        mma_seg(ar, t16, P, MMA_OFF[L_V2], 4, 1)
        vraw = (dvis * t16).sum(1)
        vraw = vraw + vraw[LANES ^ 32]
        vis = sigm(elu(vraw + P[SCAL + 1])) * mk[v]
        x = x + elu(ar[0])
        aw = [rows16(P, BB + B_W0 * 32)]
        mma_seg(aw, x * vis[:, None], P, MMA_OFF[L_W0], 4, 1)
        v2 = (dvis2 * elu(aw[0])).sum(1)
        v2 = v2 + v2[LANES ^ 32]
        vis2 = sigm(v2 + P[SCAL + 2]) * mk[v]
        rin = np.zeros((64, 20))
        rin[:, :16] = x
        rin[:, 16] = np.where(H == 0, vis2, rd[v][:, 0])
        rin[:, 17] = np.where(H == 0, rd[v][:, 1], rd[v][:, 2])
        rin[:, 18] = np.where(H == 0, rd[v][:, 3], 0.0)
        a16 = [rows16(P, BB + B_R0 * 32)]
        mma_seg(a16, rin, P, MMA_OFF[L_R0], 5, 1)
        r8 = elu(a16[0][:, :8])
        a8 = [rows16(P, BB + B_R2 * 32)]
        mma_seg(a8, r8, P, MMA_OFF[L_R2], 2, 1)
        rr = (drgb4[:, :4] * elu(a8[0][:, :4])).sum(1)
        rr = rr + rr[LANES ^ 32] + P[SCAL + 3]
        logits.append(np.where(mk[v] == 0, -1e9, rr)[:32])
    lg = np.stack(logits, 1)
    beta = np.exp(lg - lg.max(1, keepdims=True))
    beta /= beta.sum(1, keepdims=True)
    return (rgb_in * beta[:, :, None]).sum(1)


def test_blend_pack_matches_oracle(weights, golden_render):
    gr = golden_render
    P = ops.blend_pack_weights_host(ops.blend_raw_weights(weights)).astype(np.float64)
    sl = slice(40, 72)
    rf, rdf, mv = gr["rgb_feat"][sl], gr["ray_diff"][sl], gr["mask_valid"][sl]
    # make sure both masked and unmasked views occur
    assert 0 < int(mv.sum()) < mv.numel()
    out = sim_blend(P, rf.numpy().astype(np.float64), rdf.numpy().astype(np.float64), mv.numpy())
    ref = O.blending(weights, rf, rdf, mv).numpy()
    assert np.abs(out - ref).max() < 1e-5
    assert np.abs(out - gr["blend_rgb"][sl].numpy()).max() < 1e-5


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads on a CPU-only host and exports everything include/surf_hip.h declares."""
    import os
    import re
    from surf_amd import _lib
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "surf_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t)\s+(surf_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name)
    hdr_ver = int(re.search(r"#define\s+SURF_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert L.surf_abi_version() == hdr_ver == _lib.ABI_VERSION
    assert L.surf_sdf_scratch_bytes(1 << 20) > 0


# ---- split 16-bit streams (sdf_mlp_split.hip) --------------------------------------------------------------
def _bf16_to_f64(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


class SplitStream:
    """Decoder of surf_sdf_pack_weights_{bf16,f16}'s chunk stream: A operand of k-step ks = sum of its pieces."""
    BWD_NT = [1, 5, 5, 6, 5, 5]

    def __init__(self, packed, precision="bf16x3"):
        self.u16 = packed[: (len(packed) // 2) * 2].view(np.uint16)
        self.packed = packed
        self.precision = precision
        self.np = {"bf16x3": 3, "f16x2": 2}[precision]
        KS = self.np * 1024
        fwd_ks = [2, 10, 10, 11, 10, 10]
        self.ks, self.off = [], []
        o = 0
        for l in range(6):
            for t in range(4):
                self.off.append(o); self.ks.append(fwd_ks[l]); o += fwd_ks[l] * KS
        for l in range(5, -1, -1):
            for t in range(self.BWD_NT[l]):
                k = 7 if l == 2 else 8
                self.off.append(o); self.ks.append(k); o += k * KS
        self.stream_bytes = o
        self.tail = packed[o:o + 164 * 4].view(np.float32).astype(np.float64)

    def fwd_chunk(self, l, t):
        return l * 4 + t

    def bwd_chunk(self, l, t):
        return 24 + sum(self.BWD_NT[i] for i in range(5, l, -1)) + t

    def A(self, ci, ks):
        """(64 lanes, 8) float64 = exact sum of the pieces; also checks that piece sizes decay."""
        base = (self.off[ci] + ks * self.np * 1024) // 2
        raw = [self.u16[base + p * 512: base + (p + 1) * 512] for p in range(self.np)]
        if self.precision == "bf16x3":
            pieces = [_bf16_to_f64(r).reshape(64, 8) for r in raw]
            assert (np.abs(pieces[1]) <= np.abs(pieces[0]) * 2.0 ** -7 + 1e-45).all()
            return pieces[0] + pieces[1] + pieces[2]
        hi, lo = [r.view(np.float16).astype(np.float64).reshape(64, 8) for r in raw]
        assert (np.abs(lo) <= np.abs(hi) * 2.0 ** -11 + 2.0 ** -25).all()
        return (hi + lo) / 256.0                                        # weights are stored x 2^8


def mma16(acc, A, Bfrag):
    """one K=16 step: A (64,8): lane l -> A[row l&31][k = 8 (l>>5) + j]; Bfrag (64,8): B[k = 8 h + j][col l&31]."""
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for j in range(8):
        Am[J, 8 * H + j] = A[:, j]
        Bm[8 * H + j, J] = Bfrag[:, j]
    D = Am @ Bm
    acc += D[ROW[None, :] + 4 * H[:, None], J[:, None]]


def sim_sdf_split(packed, pts, phi_full, jphi_full, precision):
    S_ = SplitStream(packed, precision)
    e_all = np.zeros((64, 28)); je_all = np.zeros((64, 28))
    e_all[:, :27] = O.posenc(torch.from_numpy(pts)).numpy()[J]
    je_all[:, :27] = O.posenc_jac_diag(torch.from_numpy(pts)).numpy()[J]
    sel = 14 * H[:, None] + np.arange(14)[None, :]
    e = np.zeros((64, 16)); e[:, :14] = np.take_along_axis(e_all, sel, 1); e[:, 14] = 1.0
    je = np.take_along_axis(je_all, sel, 1)
    phi = np.zeros((64, 16)); phi[:, :14] = np.take_along_axis(phi_full[J], sel, 1); phi[:, 14] = 1.0
    Jl = np.stack([np.take_along_axis(jphi_full[J][:, :, a], sel, 1) for a in range(3)], -1)
    frags = lambda v: [v[:, 8 * s:8 * s + 8] for s in range(v.shape[1] // 8)]   # register order == fragment order

    def fwd_layer(l, hin):
        out = []
        for t in range(4):
            acc = np.zeros((64, 16))
            ci = S_.fwd_chunk(l, t)
            bs = (frags(e) if l in (0, 3) else []) + (frags(phi) if l >= 1 else []) + (frags(hin)[:7 if l == 3 else 8] if l else [])
            assert len(bs) == S_.ks[ci]
            for ks, b in enumerate(bs):
                mma16(acc, S_.A(ci, ks), b)
            out.append(acc)
        return np.concatenate(out, 1)

    # bf16x3 runs in units of the softplus exponent (PolBf3::PRESCALED): the accumulators are u = 100 log2(e) t, the
    # activations z = max(u, 0) + log2(1 + 2^-|u|) = y 100 / ln 2, and the packer has scaled biases / input columns / lin6
    def softplus_exp_units(u):
        return np.maximum(u, 0) + np.log2(1 + np.exp2(-np.abs(u))), 1.0 / (1.0 + np.exp2(-u))
    act = softplus_exp_units if precision == "bf16x3" else softplus
    Sg, h = [], None
    for l in range(6):
        t = fwd_layer(l, h)
        h, sv = act(t)
        if l < 5:
            Sg.append(sv)
    w6h = S_.tail[:128].reshape(2, 64)[H]
    w6p = S_.tail[128:160].reshape(2, 16)[H]
    y0 = (w6h * h).sum(1) + (w6p[:, :14] * phi[:, :14]).sum(1)
    y0 = y0 + y0[LANES ^ 32] + S_.tail[160]
    delta = sv * w6h
    accP = w6p.copy(); accE = np.zeros((64, 16))
    for l in (5, 4, 3, 2, 1):
        din = frags(delta)[:7 if l == 2 else 8]
        G = []
        for t in range(S_.BWD_NT[l]):
            acc = np.zeros((64, 16)) if t < 4 else (accE if (l == 3 and t == 4) else accP)
            ci = S_.bwd_chunk(l, t)
            for ks, b in enumerate(din):
                mma16(acc, S_.A(ci, ks), b)
            G.append(acc)
        delta = Sg[l - 1] * np.concatenate(G[:4], 1)
    for ks, b in enumerate(frags(delta)):
        mma16(accE, S_.A(S_.bwd_chunk(0, 0), ks), b)
    g3 = np.zeros((64, 3))
    ch = 14 * H[:, None] + np.arange(14)[None, :]
    for a in range(3):
        g3[:, a] = (accE[:, :14] * je * ((ch % 3) == a)).sum(1) + (accP[:, :14] * Jl[:, :, a]).sum(1)
    g3 = g3 + g3[LANES ^ 32]
    return y0[:32], g3[:32]


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_sdf_split_pack_matches_oracle(weights, golden_pipe, golden_render, precision):
    vols, tabs, _, _ = pipeline_views(golden_pipe)
    pts = golden_render["pts"][100:132].contiguous()
    layers = O.sdf_weights(weights)
    packed = ops.sdf_pack_weights_split_host(ops.sdf_effective_weights(weights), precision)
    phi, jphi = O.lookup_sparse_volume(pts, vols, tabs, with_jac=True)
    sdf_o, grad_o, _ = O.sdf_mlp(layers, pts, phi, jphi)
    sdf_s, grad_s = sim_sdf_split(packed, pts.numpy(), phi.numpy().astype(np.float64), jphi.numpy().astype(np.float64),
                                  precision)
    assert np.abs(sdf_s - sdf_o.numpy()).max() < 2e-5
    assert np.abs(grad_s - grad_o.numpy()).max() < 2e-4


def test_bf16_three_way_split_is_exact():
    g = np.random.default_rng(0)
    x = (g.standard_normal(4096) * np.exp(g.uniform(-20, 10, 4096))).astype(np.float32)

    def rne(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = u + 0x7FFF + ((u >> 16) & 1)
        return ((u >> 16).astype(np.uint32) << 16).view(np.float32)
    p1 = rne(x); r = x - p1; p2 = rne(r); p3 = rne(r - p2)
    assert np.array_equal((p1.astype(np.float64) + p2 + p3).astype(np.float32), x)
    assert np.abs(x - (p1 + p2 + p3)).max() == 0.0


def test_f16_two_way_split_error_bound():
    """f16x2 operands: hi = fp16(x), lo = fp16(x - hi); |x - (hi + lo)| <= max(2^-22 |x|, 2^-25) inside fp16's range
    (the absolute floor is the second piece going subnormal, for |x| < 0.125; weights and deltas are stored x 2^8)."""
    g = np.random.default_rng(1)
    x = (g.standard_normal(8192) * np.exp(g.uniform(-6, 6, 8192))).astype(np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    rec = hi.astype(np.float64) + lo.astype(np.float64)
    assert (np.abs(rec - x) <= np.maximum(np.abs(x) * 2.0 ** -22, 2.0 ** -25)).all()
