// K9: sparse trilinear feature gather + SDF MLP forward + analytic gradient, fp32 MFMA.
//
// Restates lookup_sparse_volume / grid_sample_3d_sparse   projector.py:217-390
//          SDFNetworkSparse.forward / sdf                   sdf_network.py:95-124
//          first-order part of SDFNetworkSparse.gradient    sdf_network.py:129-141 (closed form, no autograd)
//
// Design (CDNA4, wave64, v_mfma_f32_32x32x2_f32 = exact fp32):
//   * one wavefront owns a tile of 32 sample points for the whole network; lane l = (sample j = l & 31,
//     half h = l >> 5).  Every layer is computed TRANSPOSED:  T^T (features x samples) = W (A operand)
//     * X^T (B operand), so the 32x32 accumulator tile has the sample on the lane and 16 features in
//     registers -- which is exactly the B-operand shape of the next layer.  Activations therefore never
//     leave registers and never cross lanes: accumulator register r of input tile tt simply IS k-step
//     16*tt + r of the next layer, with the k-permutation  k = 32 tt + (r&3) + 8 (r>>2) + 4 h  baked into
//     the packed weights (surf_sdf_pack_weights).
//   * the 28 sparse-volume channels and the 27 positional-encoding channels are split between the two
//     lane halves (half h gathers stages 2h and 2h+1), so the gather work is not duplicated either.
//   * weights (0.89 MB packed) stream from L2 as one 16-byte load per lane per 4 k-steps.
//   * reverse sweep: G_in = W^T delta, same trick with the packed transpose; softplus'(t) of the five
//     inner layers and the 14x3 feature Jacobian round-trip through a per-wave scratch slot.
#include <math.h>
#include <stdlib.h>

#include "common.h"

// build-time knobs
#ifndef SURF_SDF_STAGGER
#define SURF_SDF_STAGGER 0   // start delay of waves 4..7, x 8,128 cycles (measured: no effect, kept as a knob)
#endif
#ifndef SURF_SDF_PRIO
#define SURF_SDF_PRIO 0
#endif
#ifndef SURF_SDF_NOWEIGHTS
#define SURF_SDF_NOWEIGHTS 0
#endif
#ifndef SURF_SDF_NOSOFTPLUS
#define SURF_SDF_NOSOFTPLUS 0
#endif
#ifndef SURF_SDF_NOSCRATCH
#define SURF_SDF_NOSCRATCH 0  // 1 = timing-only diagnostic build (wrong gradients): no softplus' round trip
#endif

namespace {

constexpr int HID = 128, NE = 27, H2 = 101;
constexpr int TILE = 32;  // samples per wavefront pass

// ---- packed weight buffer (floats) --------------------------------------------------------------
// forward layer l: [q][t][lane][4]; NQ groups of 4 k-steps, NT output tiles of 32 rows
constexpr int FWD_NQ[6] = {4, 20, 20, 24, 20, 20};
constexpr int BWD_NT[6] = {1, 5, 5, 6, 5, 5};  // backward of layer l: [H tiles..][E][P]
constexpr int fwd_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += FWD_NQ[i] * 4 * 256; return o; }
constexpr int FWD_TOTAL = fwd_off(6);
constexpr int bwd_off(int l) { int o = FWD_TOTAL; for (int i = 0; i < l; ++i) o += 16 * BWD_NT[i] * 256; return o; }
constexpr int BWD_TOTAL_END = bwd_off(6);
constexpr int BIAS_OFF = BWD_TOTAL_END;            // [l][t][h][16]
constexpr int W6H_OFF = BIAS_OFF + 6 * 4 * 2 * 16;  // [h][64]
constexpr int W6P_OFF = W6H_OFF + 2 * 64;           // [h][16]
constexpr int B6_OFF = W6P_OFF + 2 * 16;            // 1 (+3 pad)
constexpr int PACKED_FLOATS = B6_OFF + 4;

// ---- per-wave scratch slot (floats): softplus' of layers 0..4 + feature Jacobian ---------------------
constexpr int SCR_S = 5 * 16 * 64 * 4;
constexpr int SCR_J = 11 * 64 * 4;  // 42 floats per lane padded to 44
constexpr int SCR_SLOT = SCR_S + SCR_J;

#ifndef SURF_SDF_WPB
#define SURF_SDF_WPB 8
#endif
constexpr int WPB = SURF_SDF_WPB;          // wavefronts per workgroup; 8 = two per SIMD, partners are w and w + 4
constexpr int MAX_BLOCKS = 256 * 8 / WPB;  // two wavefronts per SIMD on every CU

struct SdfArgs {
  const float* pts;
  const uint8_t* mask;
  const int32_t* idx;  // optional list of point indices to evaluate (n entries); outputs are scattered back
  int64_t n;
  const float* vols[SURF_MAX_STAGES];
  const int32_t* tables[SURF_MAX_STAGES];
  int dims[SURF_MAX_STAGES];
  const float* packed;
  float* sdf;
  float* grad;
  float* scratch;
};
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// 16-byte buffer load / store: wave-uniform descriptor + per-lane byte offset + compile-time byte offset.
// (flat global loads made hipcc hoist one 64-bit address per unrolled load out of the tile loop and spill
// hundreds of them; the buffer form needs a single offset VGPR.)
__device__ __forceinline__ f32x4 bload(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// gfx950 hazard (observed, not handled by hipcc when soffset is an SGPR): a VALU write to the data VGPRs
// directly after a buffer_store_dwordx4 corrupts the stored data of some lanes.  Two wait states fix it; store and
// pad are one asm statement so that neither the scheduler nor a register-allocator copy can land between them.
__device__ __forceinline__ void bstore(rsrc_t r, int voff, int soff, f32x4 v) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(r), "s"(soff) : "memory");
}

__device__ __forceinline__ f32x4 wload(rsrc_t r, int voff, int soff) {
#if SURF_SDF_NOWEIGHTS  // timing-only diagnostic: no weight traffic at all
  const float v = __builtin_bit_cast(float, voff + soff);
  return f32x4{v, v, v, v};
#else
  return bload(r, voff, soff);
#endif
}

// acc[t] += W-tile(t) * b  over NQ groups of 4 k-steps; weights at byte offset `off`: [q][t][lane] x 16 B.
// The A operands are prefetched PF groups ahead through a ring of PF+1 register buffers (statically indexed after
// full unrolling); the L2 round trip of a group is longer than the 16 MFMAs (1,024 cycles) of one group.
#ifndef SURF_SDF_PF
#define SURF_SDF_PF 2
#endif
template <int NQ, int NT, int PF = SURF_SDF_PF>
__device__ __forceinline__ void mma_seg(f32x16 (&acc)[NT], const float* b, rsrc_t wr, int lane16, int off) {
  f32x4 a[PF + 1][NT];
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    if (p < NQ) {
#pragma unroll
      for (int t = 0; t < NT; ++t) a[p][t] = wload(wr, lane16, off + (p * NT + t) * 1024);
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (q + PF < NQ) {
#pragma unroll
      for (int t = 0; t < NT; ++t) a[(q + PF) % (PF + 1)][t] = wload(wr, lane16, off + ((q + PF) * NT + t) * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q % (PF + 1)][t][i], b[q * 4 + i], acc[t], 0, 0, 0);
    }
    // keep the hand-made prefetch distance in place
    __builtin_amdgcn_sched_barrier(0);
  }
}
// nn.Softplus(beta=100, threshold=20) and its derivative from the pre-activation.
// Raw v_exp_f32 / v_log_f32 / v_rcp_f32 (1 ulp): 1 + e >= 1 is never denormal, and an e that underflows to 0
// gives h = 0, s = 0, the fp32 limits.  Absolute error of h <= 2e-8, of s <= 3e-7.
__device__ __forceinline__ void softplus100(float t, float& hv, float& sv) {
#if SURF_SDF_NOSOFTPLUS  // timing-only diagnostic
  hv = fmaxf(t, 0.f); sv = t > 0.f ? 1.f : 0.f; return;
#endif
  const float bt = t * 100.0f;
  const float e = __builtin_amdgcn_exp2f(fminf(bt, 20.0f) * 1.44269504088896341f);
  const float d = 1.0f + e;
  const float hp = __builtin_amdgcn_logf(d) * (0.69314718055994531f * 0.01f);
  const float sp = e * __builtin_amdgcn_rcpf(d);
  const bool lin = bt > 20.0f;
  hv = lin ? t : hp;
  sv = lin ? 1.0f : sp;
}

// acc (4 tiles) -> h[64] (activations) and, if STORE, softplus' to the wave's scratch slot (byte offset off)
template <bool STORE>
__device__ __forceinline__ void activate(const f32x16 (&acc)[4], float (&h)[64], rsrc_t sr, int svoff, int off) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 s;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float hv, sv;
        softplus100(acc[t][4 * g + i], hv, sv);
        h[16 * t + 4 * g + i] = hv;
        s[i] = sv;
      }
      if (STORE && !SURF_SDF_NOSCRATCH) bstore(sr, svoff, off + (t * 4 + g) * 1024, s);
    }
  }
}

// sparse trilinear gather of the two stages owned by this lane half (projector.py:217-374):
// g = (p+1)/voxel_size, weights from the unclamped floor, indices clamped, row -1 -> zeros.
// Branch-free and batched: all 16 table lookups are issued before the first is consumed, then the 16 row
// fetches of a stage (empty corners read row 0 with weight 0), so a tile pays ~3 memory round trips, not 32.
template <bool GRAD>
__device__ __forceinline__ void gather_features(const SdfArgs& a, int h, float px, float py, float pz,
                                                float (&phi)[16], float (&J)[14][3]) {
#pragma unroll
  for (int c = 0; c < 16; ++c) phi[c] = 0.f;
  if (GRAD) {
#pragma unroll
    for (int c = 0; c < 14; ++c) J[c][0] = J[c][1] = J[c][2] = 0.f;
  }
  int rows[2][8];
  float tx[2], ty[2], tz[2], inv_vs[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int st = 2 * h + sl;
    const int D = a.dims[st];
    const int32_t* __restrict__ table = a.tables[st];
    const float vs = 2.0f / ((float)D - 1.0f);
    inv_vs[sl] = 1.0f / vs;
    const float gx = (px + 1.0f) / vs, gy = (py + 1.0f) / vs, gz = (pz + 1.0f) / vs;
    const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
    tx[sl] = gx - fx; ty[sl] = gy - fy; tz[sl] = gz - fz;
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int xi = min(max(x0 + (c >> 2), 0), D - 1);
      const int yi = min(max(y0 + ((c >> 1) & 1), 0), D - 1);
      const int zi = min(max(z0 + (c & 1), 0), D - 1);
      rows[sl][c] = D > 0 ? table[((int64_t)xi * D + yi) * D + zi] : -1;  // D == 0: stage absent
    }
  }
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const float* __restrict__ vol = a.vols[2 * h + sl];
    f32x4 f0[8], f1[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4* fr = reinterpret_cast<const f32x4*>(vol + (int64_t)max(rows[sl][c], 0) * 8);
      f0[c] = fr[0];
      f1[c] = fr[1];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {  // corner order x, y, z-fastest = the reference's summation order
      const int dx = c >> 2, dy = (c >> 1) & 1, dz = c & 1;
      const float ok = rows[sl][c] >= 0 ? 1.0f : 0.0f;
      const float wx = dx ? tx[sl] : 1.0f - tx[sl];
      const float wy = dy ? ty[sl] : 1.0f - ty[sl];
      const float wz = dz ? tz[sl] : 1.0f - tz[sl];
      const float w = wx * wy * wz * ok;
      const float f[7] = {f0[c][0], f0[c][1], f0[c][2], f0[c][3], f1[c][0], f1[c][1], f1[c][2]};
      float cx = 0.f, cy = 0.f, cz = 0.f;
      if (GRAD) {  // d w / d p: the product of the other two weights over the voxel size
        cx = ((dx ? 1.0f : -1.0f) * wy * wz) * (inv_vs[sl] * ok);
        cy = ((dy ? 1.0f : -1.0f) * wx * wz) * (inv_vs[sl] * ok);
        cz = ((dz ? 1.0f : -1.0f) * wx * wy) * (inv_vs[sl] * ok);
      }
#pragma unroll
      for (int ch = 0; ch < 7; ++ch) {
        phi[7 * sl + ch] += f[ch] * w;
        if (GRAD) {
          J[7 * sl + ch][0] += f[ch] * cx;
          J[7 * sl + ch][1] += f[ch] * cy;
          J[7 * sl + ch][2] += f[ch] * cz;
        }
      }
    }
  }
}

// positional encoding channels owned by this half (embedder.py:11-36): channel 14 h + s, s < 14;
// optionally the diagonal Jacobian d e_c / d x_{c % 3}
__device__ __forceinline__ void posenc_half(int h, float x, float y, float z, float (&e)[16], float (&je)[14], bool want_j) {
  float all[28], jall[28];
  all[0] = x; all[1] = y; all[2] = z;
  jall[0] = jall[1] = jall[2] = 1.0f;
  const float p[3] = {x, y, z};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // accurate sin/cos of the coordinate, then exact double-angle steps for 2x, 4x, 8x (error doubles per step:
    // <= 1e-6 at 8x, against the 1e-4 SDF tolerance)
    float s, co;
    sincosf(p[c], &s, &co);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float f = (float)(1 << k);
      all[3 + 6 * k + c] = s;
      all[3 + 6 * k + 3 + c] = co;
      jall[3 + 6 * k + c] = f * co;
      jall[3 + 6 * k + 3 + c] = -f * s;
      const float s2 = 2.0f * s * co;
      const float c2 = fmaf(-2.0f * s, s, 1.0f);
      s = s2;
      co = c2;
    }
  }
  all[27] = 0.f; jall[27] = 0.f;
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    e[s] = h ? all[14 + s] : all[s];
    if (want_j) je[s] = h ? jall[14 + s] : jall[s];
  }
  e[14] = e[15] = 0.f;
}

#ifndef SURF_SDF_OCC
#define SURF_SDF_OCC 2  // wavefronts per SIMD the register budget is sized for
#endif

// =================================================================================================================
// v2 schedule: output-tile-major layers.  Each 32-row output tile runs its whole k-loop alone (a dependent MFMA
// chain issues back to back at the 64-cycle rate), so the softplus VALU work of tile t is issued between the MFMAs
// of tile t+1 and hides under them instead of idling the matrix pipe; all-zero k-groups (features 104..127 of the
// 101-wide layer) are skipped; the weight stream is one compile-time sequence of 16-byte groups with a register ring
// that prefetches across tile, layer and forward/backward boundaries.
// =================================================================================================================
constexpr int fwd_ngt(int l) { return l == 0 ? 4 : (l == 3 ? 21 : 20); }  // k-groups per output tile
constexpr int fwd_gbase(int l, int t) {
  int g = 0;
  for (int i = 0; i < l; ++i) g += 4 * fwd_ngt(i);
  return g + t * fwd_ngt(l);
}
constexpr int FWD_GROUPS = fwd_gbase(6, 0);
constexpr int bwd_ngt(int l) { return l == 2 ? 13 : 16; }
constexpr int bwd_gbase(int l, int t) {  // layers run 5,4,3,2,1,0
  int g = 0;
  for (int i = 5; i > l; --i) g += BWD_NT[i] * bwd_ngt(i);
  return g + t * bwd_ngt(l);
}
constexpr int BWD_GROUPS = bwd_gbase(0, 0) + bwd_ngt(0);
constexpr int ALL_GROUPS = FWD_GROUPS + BWD_GROUPS;

struct GroupTable { int off[ALL_GROUPS + 8]; };
constexpr GroupTable make_groups() {
  GroupTable g{};
  int n = 0;
  for (int l = 0; l < 6; ++l)
    for (int t = 0; t < 4; ++t) {
      const int base = fwd_off(l) * 4 + t * 1024;
      if (l == 0) {
        for (int q = 0; q < 4; ++q) g.off[n++] = base + q * 4096;
      } else {
        const int nh = l == 3 ? 13 : 16;
        for (int q = 0; q < nh; ++q) g.off[n++] = base + q * 4096;
        for (int q = 16; q < FWD_NQ[l]; ++q) g.off[n++] = base + q * 4096;
      }
    }
  for (int l = 5; l >= 0; --l)
    for (int t = 0; t < BWD_NT[l]; ++t)
      for (int q = 0; q < bwd_ngt(l); ++q) g.off[n++] = bwd_off(l) * 4 + (q * BWD_NT[l] + t) * 1024;
  for (int k = 0; k < 8; ++k) g.off[n + k] = g.off[n - 1];
  return g;
}
constexpr GroupTable GROUPS = make_groups();

#ifndef SURF_SDF_RPF
#define SURF_SDF_RPF 4
#endif
constexpr int RPF = SURF_SDF_RPF;  // weight groups in flight ahead of the one being multiplied
struct WRing { f32x4 a[RPF + 1]; };

// NQ consecutive groups of the weight stream starting at global group GBASE: acc += W-groups x b[0 .. 4 NQ);
// fn(q) runs after the four MFMAs of group q (VALU work that hides under them).
template <int GBASE, int NQ, int NTOTAL, class F>
__device__ __forceinline__ void stream_mma(WRing& ring, f32x16& acc, const float* b, rsrc_t wr, int lane16, F fn) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (GBASE + q + RPF < NTOTAL) ring.a[(GBASE + q + RPF) % (RPF + 1)] = wload(wr, lane16, GROUPS.off[GBASE + q + RPF]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ring.a[(GBASE + q) % (RPF + 1)][i], b[q * 4 + i], acc, 0, 0, 0);
    fn(q);
    __builtin_amdgcn_sched_barrier(0);
  }
}
struct SdfCtx {
  rsrc_t wr, sr;
  int lane16, h64, h256, svoff;
};

// One forward output tile (layer L, tile T).  `raw` holds the pre-activations of the previously finished tile and is
// converted (softplus, softplus' -> scratch) under this tile's MFMAs; on return it holds this tile's pre-activations.
template <bool GRAD, int L, int T, int NTOTAL>
__device__ __forceinline__ void fwd_tile(const SdfCtx& c, WRing& ring, f32x16& raw, float* hin, float* hout, const float* e,
                                         const float* phi, float* delta, float& y0) {
  constexpr int NH = L == 0 ? 4 : (L == 3 ? 13 : 16);  // k-groups over the main input (e for layer 0)
  constexpr int NE = L == 3 ? 4 : 0;
  constexpr int GB = fwd_gbase(L, T);
  f32x16 acc;  // the bias arrives through k-step 14 of the last segment (B operand 1.0), no separate loads
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const f32x16 prev = raw;
  f32x4 w6[4];
  if (L == 5 && T > 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) w6[g] = bload(c.wr, c.h256, W6H_OFF * 4 + ((T - 1) * 4 + g) * 16);
  }
  f32x4 sbuf = {0.f, 0.f, 0.f, 0.f};
  auto cv = [&](int gi) __attribute__((always_inline)) {
    if (L == 0) {
      if (T > 0 && gi < 4) {  // 4 elements of layer-0 tile T-1 per group
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float hv, sv;
          softplus100(prev[4 * gi + u], hv, sv);
          hout[16 * (T - 1) + 4 * gi + u] = hv;
          sbuf[u] = sv;
        }
        if (GRAD && !SURF_SDF_NOSCRATCH) bstore(c.sr, c.svoff, ((T - 1) * 4 + gi) * 1024, sbuf);
      }
    } else if (T == 0) {
      if (gi >= 1 && gi <= 8) {  // tile 3 of the previous layer, needed from k-group 12 on
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int el = 2 * (gi - 1) + u;
          float hv, sv;
          softplus100(prev[el], hv, sv);
          hin[48 + el] = hv;
          sbuf[el & 3] = sv;
          if (GRAD && (el & 3) == 3 && !SURF_SDF_NOSCRATCH)
            bstore(c.sr, c.svoff, (L - 1) * 16384 + (12 + (el >> 2)) * 1024, sbuf);
        }
      }
    } else if (gi >= 1 && gi <= 16) {  // tile T-1 of this layer, one element per k-group
      const int el = gi - 1;
      float hv, sv;
      softplus100(prev[el], hv, sv);
      if (L < 5) {
        hout[16 * (T - 1) + el] = hv;
        sbuf[el & 3] = sv;
        if (GRAD && (el & 3) == 3 && !SURF_SDF_NOSCRATCH)
          bstore(c.sr, c.svoff, L * 16384 + ((T - 1) * 4 + (el >> 2)) * 1024, sbuf);
      } else {  // last hidden layer: fold straight into y0 = W6[0] . h5 and delta5 = softplus' * W6[0]
        const float w = w6[el >> 2][el & 3];
        y0 = fmaf(w, hv, y0);
        if (GRAD) delta[16 * (T - 1) + el] = sv * w;
      }
    }
  };
  stream_mma<GB, NH, NTOTAL>(ring, acc, L == 0 ? e : hin, c.wr, c.lane16, [&](int q) __attribute__((always_inline)) { cv(q); });
  if (NE) stream_mma<GB + NH, NE ? NE : 1, NTOTAL>(ring, acc, e, c.wr, c.lane16, [&](int q) __attribute__((always_inline)) { cv(NH + q); });
  if (L > 0)
    stream_mma<GB + NH + NE, 4, NTOTAL>(ring, acc, phi, c.wr, c.lane16, [&](int q) __attribute__((always_inline)) { cv(NH + NE + q); });
  raw = acc;
}

template <bool GRAD, int L, int NTOTAL>
__device__ __forceinline__ void fwd_layer(const SdfCtx& c, WRing& ring, f32x16& raw, float* hin, float* hout, const float* e,
                                          const float* phi, float* delta, float& y0) {
  fwd_tile<GRAD, L, 0, NTOTAL>(c, ring, raw, hin, hout, e, phi, delta, y0);
  fwd_tile<GRAD, L, 1, NTOTAL>(c, ring, raw, hin, hout, e, phi, delta, y0);
  fwd_tile<GRAD, L, 2, NTOTAL>(c, ring, raw, hin, hout, e, phi, delta, y0);
  fwd_tile<GRAD, L, 3, NTOTAL>(c, ring, raw, hin, hout, e, phi, delta, y0);
}

// One hidden output tile of the reverse sweep of layer L: dout[16T..] = softplus'(t_{L-1})[tile T] * (W_L^T din)[tile T]
template <int L, int T>
__device__ __forceinline__ void bwd_hidden_tile(const SdfCtx& c, WRing& ring, const float* din, float* dout) {
  f32x4 sS[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (SURF_SDF_NOSCRATCH) sS[g] = f32x4{0.5f, 0.5f, 0.5f, 0.5f};
    else sS[g] = bload(c.sr, c.svoff, (L - 1) * 16384 + (T * 4 + g) * 1024);
  }
  f32x16 G;
#pragma unroll
  for (int r = 0; r < 16; ++r) G[r] = 0.f;
  stream_mma<FWD_GROUPS + bwd_gbase(L, T), bwd_ngt(L), ALL_GROUPS>(ring, G, din, c.wr, c.lane16, [](int) {});
#pragma unroll
  for (int r = 0; r < 16; ++r) dout[16 * T + r] = sS[r >> 2][r & 3] * G[r];
}

template <int L, int T>
__device__ __forceinline__ void bwd_acc_tile(const SdfCtx& c, WRing& ring, const float* din, f32x16& acc) {
  stream_mma<FWD_GROUPS + bwd_gbase(L, T), bwd_ngt(L), ALL_GROUPS>(ring, acc, din, c.wr, c.lane16, [](int) {});
}

template <int L>
__device__ __forceinline__ void bwd_layer(const SdfCtx& c, WRing& ring, const float* din, float* dout, f32x16& accE, f32x16& accP) {
  bwd_hidden_tile<L, 0>(c, ring, din, dout);
  bwd_hidden_tile<L, 1>(c, ring, din, dout);
  bwd_hidden_tile<L, 2>(c, ring, din, dout);
  bwd_hidden_tile<L, 3>(c, ring, din, dout);
  if (L == 3) {
    bwd_acc_tile<L, 4>(c, ring, din, accE);
    bwd_acc_tile<L, BWD_NT[L] - 1>(c, ring, din, accP);
  } else {
    bwd_acc_tile<L, 4>(c, ring, din, accP);
  }
}

template <bool GRAD>
__global__ __launch_bounds__(WPB * 64, SURF_SDF_OCC) void sdf_mlp_kernel2(SdfArgs a) {
  constexpr int NTOTAL = GRAD ? ALL_GROUPS : FWD_GROUPS;
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5;
  const int64_t wave_id = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * WPB;
  const int64_t n_tiles = (a.n + TILE - 1) / TILE;
  SdfCtx c;
  c.wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.packed, 0, PACKED_FLOATS * 4, 0x00020000);
  c.sr = __builtin_amdgcn_make_buffer_rsrc((void*)a.scratch, 0, GRAD ? 0x7fffffff : 0, 0x00020000);
  c.lane16 = lane * 16;
  c.h64 = h * 64;
  c.h256 = h * 256;
  c.svoff = (int)(wave_id * (SCR_SLOT * 4)) + lane * 16;

  for (int64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
    const int64_t slot = tile * TILE + (lane & 31);
    const int64_t sc = slot < a.n ? slot : a.n - 1;
    const int64_t i = a.idx ? (int64_t)a.idx[sc] : sc;
    const bool active = (slot < a.n) && (!a.mask || a.mask[i] != 0);
    if (__ballot(active) == 0ull) continue;
    const float px = a.pts[i * 3 + 0], py = a.pts[i * 3 + 1], pz = a.pts[i * 3 + 2];

    WRing ring;
#pragma unroll
    for (int p = 0; p < RPF; ++p) ring.a[p] = wload(c.wr, c.lane16, GROUPS.off[p]);  // weights fly during the gather

    float phi[16], e[16];
    {
      float J[14][3];
      gather_features<GRAD>(a, h, px, py, pz, phi, J);
      if (GRAD) {
#pragma unroll
        for (int g = 0; g < 11; ++g) {
          f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int idx = 4 * g + q;
            v[q] = idx < 42 ? J[idx / 3][idx % 3] : 0.f;
          }
          bstore(c.sr, c.svoff, SCR_S * 4 + g * 1024, v);
        }
      }
      float je_unused[14];
      posenc_half(h, px, py, pz, e, je_unused, false);
      e[14] = 1.0f;    // bias steps (see surf_sdf_pack_weights)
      phi[14] = 1.0f;
    }

    // ------------------------------------------------ forward ------------------------------------------------
    float hA[64], hB[64], delta[64];
    f32x16 raw;
#pragma unroll
    for (int r = 0; r < 16; ++r) raw[r] = 0.f;
    float y0 = 0.f;
    fwd_layer<GRAD, 0, NTOTAL>(c, ring, raw, hA, hA, e, phi, delta, y0);   // h0 -> hA (tile 3 finished in layer 1)
    fwd_layer<GRAD, 1, NTOTAL>(c, ring, raw, hA, hB, e, phi, delta, y0);
    fwd_layer<GRAD, 2, NTOTAL>(c, ring, raw, hB, hA, e, phi, delta, y0);
    fwd_layer<GRAD, 3, NTOTAL>(c, ring, raw, hA, hB, e, phi, delta, y0);
    fwd_layer<GRAD, 4, NTOTAL>(c, ring, raw, hB, hA, e, phi, delta, y0);
    fwd_layer<GRAD, 5, NTOTAL>(c, ring, raw, hA, hB, e, phi, delta, y0);
    // tile 3 of layer 5 and the feature part of the last layer: y0 = W6[0] . [h5 | phi] + b6
    {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 w = bload(c.wr, c.h256, W6H_OFF * 4 + (12 + g) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float hv, sv;
          softplus100(raw[4 * g + q], hv, sv);
          y0 = fmaf(w[q], hv, y0);
          if (GRAD) delta[48 + 4 * g + q] = sv * w[q];
        }
      }
    }
    f32x16 accP;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 w = bload(c.wr, c.h64, W6P_OFF * 4 + g * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        y0 = fmaf(w[q], phi[4 * g + q], y0);
        accP[4 * g + q] = w[q];
      }
    }
    y0 += __shfl_xor(y0, 32);
    y0 += a.packed[B6_OFF];
    if (active && h == 0) a.sdf[i] = y0;
    if (!GRAD) continue;

    // ------------------------------------------------ reverse sweep ------------------------------------------
    f32x16 accE;
#pragma unroll
    for (int r = 0; r < 16; ++r) accE[r] = 0.f;
    bwd_layer<5>(c, ring, delta, hA, accE, accP);
    bwd_layer<4>(c, ring, hA, delta, accE, accP);
    bwd_layer<3>(c, ring, delta, hA, accE, accP);
    bwd_layer<2>(c, ring, hA, delta, accE, accP);
    bwd_layer<1>(c, ring, delta, hA, accE, accP);
    bwd_acc_tile<0, 0>(c, ring, hA, accE);

    float g3[3] = {0.f, 0.f, 0.f};
    {
      float e2[16], je[14];
      posenc_half(h, px, py, pz, e2, je, true);
#pragma unroll
      for (int s2 = 0; s2 < 14; ++s2) {
        const int c0 = s2 % 3, c1 = (14 + s2) % 3;
        const float v = accE[s2] * je[s2];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) g3[ax] += ((h ? c1 : c0) == ax) ? v : 0.f;
      }
      float Jf[44];
#pragma unroll
      for (int g = 0; g < 11; ++g) {
        f32x4 v = bload(c.sr, c.svoff, SCR_S * 4 + g * 1024);
        Jf[4 * g + 0] = v[0]; Jf[4 * g + 1] = v[1]; Jf[4 * g + 2] = v[2]; Jf[4 * g + 3] = v[3];
      }
#pragma unroll
      for (int ch = 0; ch < 14; ++ch) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) g3[ax] = fmaf(accP[ch], Jf[3 * ch + ax], g3[ax]);
      }
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) g3[ax] += __shfl_xor(g3[ax], 32);
    if (active && h == 0) {
      a.grad[i * 3 + 0] = g3[0];
      a.grad[i * 3 + 1] = g3[1];
      a.grad[i * 3 + 2] = g3[2];
    }
  }
}

int max_blocks() {
  static const int v = [] {
    const char* e = getenv("SURF_SDF_MAX_BLOCKS");  // tuning / diagnostics only
    int x = e ? atoi(e) : 0;
    return x > 0 ? x : MAX_BLOCKS;
  }();
  return v;
}

int grid_blocks(int64_t n) {
  int64_t tiles = (n + TILE - 1) / TILE;
  int64_t blocks = (tiles + WPB - 1) / WPB;
  return (int)(blocks < max_blocks() ? blocks : max_blocks());
}

}  // namespace

extern "C" int64_t surf_sdf_packed_floats(void) { return PACKED_FLOATS; }

extern "C" int64_t surf_sdf_scratch_bytes(int64_t n_points) {
  if (n_points <= 0) return 0;
  return (int64_t)grid_blocks(n_points) * WPB * SCR_SLOT * sizeof(float);
}

// Host-side packer: effective (weight-normed) matrices -> MFMA A-operand order.
extern "C" int surf_sdf_pack_weights(const float* const* h_W, const float* const* h_b, float* out) {
  if (!h_W || !h_b || !out) return SURF_E_ARG;
  for (int l = 0; l < 7; ++l)
    if (!h_W[l] || !h_b[l]) return SURF_E_ARG;
  const int in_dim[7] = {NE, 156, 156, 156, 156, 156, 156};
  const int out_dim[6] = {HID, HID, H2, HID, HID, HID};
  const float rsqrt2 = (float)(1.0 / sqrt(2.0));
  auto hk = [](int tt, int r, int h) { return 32 * tt + (r & 3) + 8 * (r >> 2) + 4 * h; };
  for (int64_t i = 0; i < PACKED_FLOATS; ++i) out[i] = 0.f;

  // -------- forward: value(l, step, h, row)
  for (int l = 0; l < 6; ++l) {
    const int NQ = FWD_NQ[l];
    float* dst = out + fwd_off(l);
    const int hid_in = (l == 3) ? H2 : HID;
    for (int q = 0; q < NQ; ++q)
      for (int t = 0; t < 4; ++t)
        for (int lane = 0; lane < 64; ++lane)
          for (int i = 0; i < 4; ++i) {
            const int step = 4 * q + i, h = lane >> 5, row = 32 * t + (lane & 31);
            int col = -1;
            float scale = 1.f;
            bool is_bias = false;  // k-step 14 of the last segment carries the bias (its B operand is the constant 1)
            if (l == 0) {  // 16 e-steps
              int ch = 14 * h + step;
              if (step < 14 && ch < NE) col = ch;
              is_bias = (step == 14 && h == 0);
            } else if (step < 64) {  // hidden steps
              int k = hk(step / 16, step % 16, h);
              if (k < hid_in) col = k;
              if (l == 3) scale = rsqrt2;
            } else if (l == 3 && step < 80) {  // e-steps of the skip layer
              int s = step - 64, ch = 14 * h + s;
              if (s < 14 && ch < NE) col = H2 + ch;
              scale = rsqrt2;
            } else {  // phi steps
              int s = step - (l == 3 ? 80 : 64);
              if (s < 14) col = 128 + 14 * h + s;
              is_bias = (s == 14 && h == 0);
            }
            float v = 0.f;
            if (col >= 0 && row < out_dim[l]) v = h_W[l][(int64_t)row * in_dim[l] + col] * scale;
            if (is_bias && row < out_dim[l]) v = h_b[l][row];
            dst[((q * 4 + t) * 64 + lane) * 4 + i] = v;
          }
  }
  // -------- backward: G_in = W_l^T delta_l; tiles [H0..H3][E][P] as listed in BWD_NT
  for (int l = 0; l < 6; ++l) {
    const int NT = BWD_NT[l];
    float* dst = out + bwd_off(l);
    const int hid_in = (l == 3) ? H2 : HID;
    for (int q = 0; q < 16; ++q)
      for (int t = 0; t < NT; ++t)
        for (int lane = 0; lane < 64; ++lane)
          for (int i = 0; i < 4; ++i) {
            const int step = 4 * q + i, h = lane >> 5, rho = lane & 31;
            const int krow = hk(step / 16, step % 16, h);  // feature of delta carried by this k-step
            const int h_row = (rho >> 2) & 1, r_row = (rho & 3) | ((rho >> 3) << 2);
            const int ch = 14 * h_row + r_row;
            int col = -1;
            float scale = 1.f;
            // tile kind
            int kind;  // 0 hidden, 1 E, 2 P
            if (l == 0) kind = 1;
            else if (t < 4) kind = 0;
            else if (l == 3 && t == 4) kind = 1;
            else kind = 2;
            if (kind == 0) {
              int c = 32 * t + rho;
              if (c < hid_in) col = c;
              if (l == 3) scale = rsqrt2;
            } else if (kind == 1) {
              if (r_row < 14 && ch < NE) col = (l == 3 ? H2 : 0) + ch;
              if (l == 3) scale = rsqrt2;
            } else {
              if (r_row < 14) col = 128 + ch;
            }
            float v = 0.f;
            if (col >= 0 && krow < out_dim[l]) v = h_W[l][(int64_t)krow * in_dim[l] + col] * scale;
            dst[((q * NT + t) * 64 + lane) * 4 + i] = v;
          }
  }
  // -------- biases [l][t][h][r]
  for (int l = 0; l < 6; ++l)
    for (int t = 0; t < 4; ++t)
      for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 16; ++r) {
          int k = hk(t, r, h);
          out[BIAS_OFF + ((l * 4 + t) * 2 + h) * 16 + r] = k < out_dim[l] ? h_b[l][k] : 0.f;
        }
  // -------- last layer row 0
  for (int h = 0; h < 2; ++h) {
    for (int s = 0; s < 64; ++s) out[W6H_OFF + h * 64 + s] = h_W[6][hk(s / 16, s % 16, h)];
    for (int s = 0; s < 14; ++s) out[W6P_OFF + h * 16 + s] = h_W[6][128 + 14 * h + s];
  }
  out[B6_OFF] = h_b[6][0];
  return 0;
}

extern "C" int surf_sdf_mlp(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n, const float* const* h_vols,
                            const int32_t* const* h_tables, const int* h_dims, int n_vol, const float* packed,
                            float* sdf, float* grad, void* scratch, void* stream) {
  if (!pts || !h_vols || !h_tables || !h_dims || !packed || !sdf) return SURF_E_ARG;
  if (n <= 0 || n_vol <= 0) return SURF_E_ARG;
  if (n_vol > SURF_MAX_STAGES) return SURF_E_LIMIT;
  if (grad && !scratch) return SURF_E_ARG;
  SdfArgs a;
  a.pts = pts; a.mask = mask; a.idx = idx; a.n = n; a.packed = packed; a.sdf = sdf; a.grad = grad; a.scratch = (float*)scratch;
  for (int s = 0; s < SURF_MAX_STAGES; ++s) {
    a.vols[s] = s < n_vol ? h_vols[s] : h_vols[0];  // absent stage: dims = 0, every corner misses, row 0 is read with weight 0
    a.tables[s] = s < n_vol ? h_tables[s] : nullptr;
    a.dims[s] = s < n_vol ? h_dims[s] : 0;
    if (s < n_vol && (!h_vols[s] || !h_tables[s] || h_dims[s] <= 1)) return SURF_E_ARG;
  }
  dim3 grid(grid_blocks(n)), block(WPB * 64);
  if (grad)
    hipLaunchKernelGGL(sdf_mlp_kernel2<true>, grid, block, 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(sdf_mlp_kernel2<false>, grid, block, 0, (hipStream_t)stream, a);
  return surf_check_launch();
}
