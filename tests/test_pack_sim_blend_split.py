"""CPU check of the split blend kernel's LDS image (surf_blend_pack_weights_split) without a GPU.

A lane-level numpy model of v_mfma_f32_32x32x16_{bf16,f16} (A: lane l holds A[l&31][8(l>>5)+i], B: lane l holds
B[8(l>>5)+i][l&31], D register r of lane l = D[(r&3) + 8(r>>2) + 4(l>>5)][l&31]) replays the data flow of
blend_split.hip on the image: 16-bit pieces summed back to fp32 weights (x log2 e), biases as accumulator
initial values, ELU on the log2(e)-scaled pre-activation.  It pins the block order, the k-slot / row maps and the
scaling of the packer; the kernel itself is checked on the GPU (tests/test_hip_parity.py)."""
import numpy as np
import pytest

from oracle import surf_oracle as O
from surf_amd import ops

LANES = np.arange(64)
J, H = LANES & 31, LANES >> 5
ROW = np.array([(r & 3) + 8 * (r >> 2) for r in range(16)])
NKS = [1, 1, 3, 2, 4, 2, 2, 2, 3, 1]
NT = [1, 1, 2, 2, 1, 1, 1, 1, 1, 1]
BLK_OFF = np.concatenate([[0], np.cumsum([k * t for k, t in zip(NKS, NT)])])
(L_RD0, L_RD2, L_B0S, L_B0V, L_B2, L_V0, L_V2, L_W0, L_R0, L_R2) = range(10)
(B_RD0, B_RD2, B_B0_T0, B_B0_T1, B_B2, B_V0, B_V2, B_W0, B_R0, B_R2) = range(10)
LOG2E, LN2 = 1.44269504088896341, 0.69314718055994531


class Image:
    def __init__(self, img, precision):
        self.NP = {"bf16x3": 3, "f16x2": 2, "f32lds": 1}[precision]
        n_blk = int(BLK_OFF[-1])
        assert n_blk == 26
        if precision == "f32lds":                                  # blocks of [lane][8 floats]
            a_bytes = n_blk * 2048
            self.A = img[:a_bytes].view(np.float32).reshape(n_blk, 64, 8).astype(np.float64)
        else:
            a_bytes = n_blk * self.NP * 1024
            u16 = img[:a_bytes].view(np.uint16).reshape(n_blk, self.NP, 64, 8)
            if precision == "bf16x3":
                vals = (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
            else:
                vals = u16.view(np.float16).astype(np.float64)
            self.A = vals.sum(axis=1)                              # (blk, lane, 8): pieces summed
        f = img[a_bytes:].view(np.float32).astype(np.float64)
        self.bias = f[: 10 * 32].reshape(10, 2, 16)
        self.dots = f[10 * 32: 13 * 32].reshape(3, 2, 16)
        self.scal = f[13 * 32: 13 * 32 + 4]
        assert img.size == a_bytes + 13 * 128 + 16

    def rows(self, table, row):
        return table[row][H]                                       # (64,16) per-lane view of an [h][16] row

    def layer(self, L, bvals, init):
        """bvals (64, 8 NKS[L]) k-slot values per lane; init: list of NT[L] (64,16) accumulators."""
        acc = [a.copy() for a in init]
        for ks in range(NKS[L]):
            B = np.zeros((16, 32))
            for i in range(8):
                B[8 * H + i, J] = bvals[:, 8 * ks + i]
            for t in range(NT[L]):
                blk = self.A[BLK_OFF[L] + ks * NT[L] + t]
                Af = np.zeros((32, 16))
                for i in range(8):
                    Af[J, 8 * H + i] = blk[:, i]
                D = Af @ B
                acc[t] += D[ROW[None, :] + 4 * H[:, None], J[:, None]]
        return acc


def elu_t(t):
    return LN2 * np.maximum(t, 0) + np.minimum(np.exp2(np.minimum(t, 0)), 1.0) - 1.0


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def sigm(x):
    return 1.0 / (1.0 + np.exp(-x))


def pad8(v, n):
    out = np.zeros((64, n))
    out[:, : v.shape[1]] = v
    return out


def sim_blend_split(I, rgb_feat, ray_diff, mask):
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
        bin_ = np.stack([np.where(H == 0, r[:, 0], r[:, 1]), np.where(H == 0, r[:, 2], r[:, 3])], 1)
        a1 = I.layer(L_RD0, pad8(bin_, 8), [I.rows(I.bias, B_RD0)])[0]
        h8 = elu_t(a1[:, :8])
        a2 = I.layer(L_RD2, h8, [I.rows(I.bias, B_RD2)])[0]
        f = local(rgb_feat[:, v])
        f[:, :11] += elu_t(a2[:, :11])
        floc.append(f); rd.append(r); mk.append(mask[:, v][J].astype(np.float64))
        ex.append(np.exp(I.scal[0] * (r[:, 3] - 1.0)))
    emin = np.min(ex, 0)
    wv = [(ex[v] - emin) * mk[v] for v in range(NS)]
    ws = sum(wv)
    wv = [w / (ws + 1e-8) for w in wv]
    mean = sum(floc[v] * wv[v][:, None] for v in range(NS))
    var = sum(wv[v][:, None] * (floc[v] - mean) ** 2 for v in range(NS))
    G0 = I.layer(L_B0S, np.concatenate([mean, var], 1), [I.rows(I.bias, B_B0_T0), I.rows(I.bias, B_B0_T1)])
    dvis, dvis2, drgb4 = I.rows(I.dots, 0), I.rows(I.dots, 1), I.rows(I.dots, 2)
    logits = []
    for v in range(NS):
        a64 = I.layer(L_B0V, pad8(floc[v], 16), G0)
        h32 = elu_t(np.concatenate(a64, 1))
        x = elu_t(I.layer(L_B2, h32, [I.rows(I.bias, B_B2)])[0])
        t16 = elu_t(I.layer(L_V0, x * wv[v][:, None], [I.rows(I.bias, B_V0)])[0])
        ar = I.layer(L_V2, t16, [I.rows(I.bias, B_V2)])[0]
        vraw = (dvis * t16).sum(1)
        vraw = vraw + vraw[LANES ^ 32]
        vis = sigm(elu(vraw + I.scal[1])) * mk[v]
        x = x + elu_t(ar)
        aw = I.layer(L_W0, x * vis[:, None], [I.rows(I.bias, B_W0)])[0]
        v2 = (dvis2 * elu_t(aw)).sum(1)
        v2 = v2 + v2[LANES ^ 32]
        vis2 = sigm(v2 + I.scal[2]) * mk[v]
        rin = np.zeros((64, 24))
        rin[:, :16] = x
        rin[:, 16] = np.where(H == 0, vis2, rd[v][:, 0])
        rin[:, 17] = np.where(H == 0, rd[v][:, 1], rd[v][:, 2])
        rin[:, 18] = np.where(H == 0, rd[v][:, 3], 0.0)
        r8 = elu_t(I.layer(L_R0, rin, [I.rows(I.bias, B_R0)])[0][:, :8])
        a8 = I.layer(L_R2, r8, [I.rows(I.bias, B_R2)])[0]
        rr = (drgb4[:, :4] * elu_t(a8[:, :4])).sum(1)
        rr = rr + rr[LANES ^ 32] + I.scal[3]
        logits.append(np.where(mk[v] == 0, -1e9, rr)[:32])
    lg = np.stack(logits, 1)
    beta = np.exp(lg - lg.max(1, keepdims=True))
    beta /= beta.sum(1, keepdims=True)
    return (rgb_feat[:, :, :3] * beta[:, :, None]).sum(1)


@pytest.mark.parametrize("precision,tol", [("bf16x3", 1e-5), ("f16x2", 1e-5), ("f32lds", 1e-5)])
def test_blend_split_image_matches_oracle(weights, golden_render, precision, tol):
    gr = golden_render
    img = ops.blend_pack_weights_split_host(ops.blend_raw_weights(weights), precision)
    I = Image(img, precision)
    sl = slice(40, 72)
    rf, rdf, mv = gr["rgb_feat"][sl], gr["ray_diff"][sl], gr["mask_valid"][sl]
    assert 0 < int(mv.sum()) < mv.numel()
    out = sim_blend_split(I, rf.numpy().astype(np.float64), rdf.numpy().astype(np.float64), mv.numpy())
    ref = O.blending(weights, rf, rdf, mv).numpy()
    assert np.abs(out - ref).max() < tol
    assert np.abs(out - gr["blend_rgb"][sl].numpy()).max() < tol


def test_blend_split_pieces_reassemble_the_weights(weights):
    """bf16x3 pieces sum back to the fp32 weight (x log2 e, rounded once to fp32) exactly; f16x2 to 22 bits."""
    raw = ops.blend_raw_weights(weights)
    W = raw[1 + 64 + 16 + 19 * 16 + 19 + 64 * 57 + 64:][: 32 * 64].reshape(32, 64)        # base_fc.2.weight
    want = (W.astype(np.float64) * np.float64(np.float32(LOG2E))).astype(np.float32).astype(np.float64)   # as the packer rounds
    for precision, rel in (("bf16x3", 0.0), ("f16x2", 2.0 ** -21)):
        I = Image(ops.blend_pack_weights_split_host(raw, precision), precision)
        got = np.zeros((32, 64))
        for ks in range(4):
            blk = I.A[BLK_OFF[L_B2] + ks]
            for lane in range(64):
                for i in range(8):
                    m = 8 * ks + i
                    col = 32 * (m // 16) + ((m % 16) & 3) + 8 * ((m % 16) >> 2) + 4 * (lane >> 5)
                    got[lane & 31, col] = blk[lane, i]
        err = np.abs(got - want)
        assert bool((err <= rel * np.abs(want) + (0 if rel == 0 else 2.0 ** -25)).all()), (precision, err.max())
