// K9b: SDF MLP forward + analytic gradient on the bf16 MFMA pipe with fp32-equivalent accuracy.
//
// Same network, same transposed / register-resident chaining as sdf_mlp.hip (see there for the reference citations),
// but every fp32 operand is split exactly into three bf16 pieces (8 + 8 + 8 significant bits):
//     a = a1 + a2 + a3,  b = b1 + b2 + b3,   a b ~= a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1     (error ~ 2^-24 |a b|)
// Six v_mfma_f32_32x32x16_bf16 (K = 16, 32 cycles) replace eight v_mfma_f32_32x32x2_f32 (K = 2, 64 cycles each):
// 2.67x less matrix-pipe time for the same contraction, fp32 accumulation throughout.
//
// The weight stream is 1.5x the fp32 one in bytes and is consumed ~4x faster, so it can no longer be streamed per
// wavefront from L2: the four wavefronts of a workgroup (one per SIMD, the whole register file each) run in lockstep
// over a double-buffered LDS image of the stream, one chunk = all k-steps x 3 pieces of one 32-row output tile;
// every wavefront copies a quarter of the next chunk (global -> registers -> LDS) under the MFMAs of the current one.
// Softplus, operand splitting and scratch stores of output tile t are issued between the MFMAs of tile t+1.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

constexpr int HID = 128, NE = 27, H2 = 101, TILE = 32;
constexpr int BWD_NT[6] = {1, 5, 5, 6, 5, 5};

// ---- chunk stream ------------------------------------------------------------------------------------------------
// forward chunk (l, t): k-steps = [hidden (tt, s) ...][e s=0,1 (l = 0, 3)][phi s=0,1 (l >= 1)]
constexpr int fwd_nh(int l) { return l == 0 ? 0 : (l == 3 ? 7 : 8); }
constexpr int fwd_ne(int l) { return (l == 0 || l == 3) ? 2 : 0; }
constexpr int fwd_np(int l) { return l == 0 ? 0 : 2; }
constexpr int fwd_ks(int l) { return fwd_nh(l) + fwd_ne(l) + fwd_np(l); }
constexpr int bwd_ks(int l) { return l == 2 ? 7 : 8; }
constexpr int N_FWD_CHUNKS = 24;
constexpr int n_bwd_chunks() { int n = 0; for (int l = 0; l < 6; ++l) n += BWD_NT[l]; return n; }
constexpr int N_BWD_CHUNKS = n_bwd_chunks();
constexpr int N_CHUNKS = N_FWD_CHUNKS + N_BWD_CHUNKS;
constexpr int KS_BYTES = 3 * 1024;  // one k-step: 3 pieces x 64 lanes x 16 B
constexpr int MAX_KS = 12;

struct ChunkTable {
  int off[N_CHUNKS + 1];  // byte offset into the packed stream
  int ks[N_CHUNKS + 1];
};
constexpr ChunkTable make_chunks() {
  ChunkTable c{};
  int n = 0, o = 0;
  for (int l = 0; l < 6; ++l)
    for (int t = 0; t < 4; ++t) { c.off[n] = o; c.ks[n] = fwd_ks(l); o += fwd_ks(l) * KS_BYTES; ++n; }
  for (int l = 5; l >= 0; --l)
    for (int t = 0; t < BWD_NT[l]; ++t) { c.off[n] = o; c.ks[n] = bwd_ks(l); o += bwd_ks(l) * KS_BYTES; ++n; }
  c.off[n] = o;
  c.ks[n] = 0;
  return c;
}
constexpr ChunkTable CHUNKS = make_chunks();
constexpr int STREAM_BYTES = CHUNKS.off[N_CHUNKS];
constexpr int fwd_chunk(int l, int t) { return l * 4 + t; }
constexpr int bwd_chunk(int l, int t) {
  int n = N_FWD_CHUNKS;
  for (int i = 5; i > l; --i) n += BWD_NT[i];
  return n + t;
}
// fp32 tail of the packed buffer (floats): W6[0] in lane order, b6
constexpr int TAIL_W6H = 0;             // [h][64]
constexpr int TAIL_W6P = TAIL_W6H + 128;  // [h][16]
constexpr int TAIL_B6 = TAIL_W6P + 32;
constexpr int TAIL_FLOATS = TAIL_B6 + 4;
constexpr int PACKED_BYTES = STREAM_BYTES + TAIL_FLOATS * 4;

constexpr int SLOT_BYTES = MAX_KS * KS_BYTES;  // 36 KB, two slots
constexpr int WPB = 4;

// per-wave scratch slot (floats): softplus' of layers 0..4 + feature Jacobian (same layout as sdf_mlp.hip)
constexpr int SCR_S = 5 * 16 * 64 * 4;
constexpr int SCR_J = 11 * 64 * 4;
constexpr int SCR_SLOT = SCR_S + SCR_J;
constexpr int MAX_BLOCKS = 256;

struct SdfArgs {
  const float* pts;
  const uint8_t* mask;
  const int32_t* idx;
  int64_t n;
  const float* vols[SURF_MAX_STAGES];
  const int32_t* tables[SURF_MAX_STAGES];
  int dims[SURF_MAX_STAGES];
  const unsigned char* packed;
  float* sdf;
  float* grad;
  float* scratch;
};

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 bload(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(rsrc_t r, int voff, int soff, f32x4 v) {  // see sdf_mlp.hip: store-data hazard
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
  asm volatile("s_nop 1");
  __builtin_amdgcn_sched_barrier(0);
}

// ---- exact 3-way bf16 split of a pair of floats: returns packed (lo, hi) pairs for the three pieces --------------------
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ void split3(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = pack2(a, b);
  const float ra = a - bf_lo(p1), rb = b - bf_hi(p1);
  p2 = pack2(ra, rb);
  p3 = pack2(ra - bf_lo(p2), rb - bf_hi(p2));
}

// B-operand fragments of a 16-wide k-step: three pieces x 4 dwords (8 bf16)
struct Frag3 { u32x4 p[3]; };

__device__ __forceinline__ void frag_set_pair(Frag3& f, int pair /*0..3*/, float a, float b) {
  uint32_t p1, p2, p3;
  split3(a, b, p1, p2, p3);
  f.p[0][pair] = p1;
  f.p[1][pair] = p2;
  f.p[2][pair] = p3;
}

__device__ __forceinline__ void softplus100(float t, float& hv, float& sv) {
  const float bt = t * 100.0f;
  const float e = __builtin_amdgcn_exp2f(fminf(bt, 20.0f) * 1.44269504088896341f);
  const float d = 1.0f + e;
  const float hp = __builtin_amdgcn_logf(d) * (0.69314718055994531f * 0.01f);
  const float sp = e * __builtin_amdgcn_rcpf(d);
  const bool lin = bt > 20.0f;
  hv = lin ? t : hp;
  sv = lin ? 1.0f : sp;
}

// six-term product of one k-step: acc += A(3 pieces from LDS) x B(3 pieces)
__device__ __forceinline__ void mfma6(f32x16& acc, const u32x4 (&a)[3], const Frag3& b) {
#define SURF_MF(x, y) \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[x]), __builtin_bit_cast(bf16x8, b.p[y]), acc, 0, 0, 0)
  SURF_MF(2, 0);  // smallest terms first
  SURF_MF(0, 2);
  SURF_MF(1, 1);
  SURF_MF(1, 0);
  SURF_MF(0, 1);
  SURF_MF(0, 0);
#undef SURF_MF
}

struct Ctx {
  rsrc_t wr, sr, tr;  // packed stream, scratch, fp32 tail
  int lane, lane16, h, svoff, wave;
  char* lds;  // two slots
};

// ---- staging: this wave's quarter of chunk CI, global -> registers (issue) and registers -> LDS (commit) ---------------
constexpr int stage_blocks(int ci) { return CHUNKS.ks[ci] * 3; }  // 1 KB blocks
struct Stage { f32x4 v[9]; };

template <int CI>
__device__ __forceinline__ void stage_issue(const Ctx& c, Stage& st) {
  if (CI >= N_CHUNKS) return;
  constexpr int NB = stage_blocks(CI < N_CHUNKS ? CI : 0);
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int blk = c.wave + 4 * k;  // wave-uniform
    if (4 * k < NB && blk < NB) st.v[k] = bload(c.wr, c.lane16, CHUNKS.off[CI < N_CHUNKS ? CI : 0] + blk * 1024);
  }
}
template <int CI>
__device__ __forceinline__ void stage_commit(const Ctx& c, const Stage& st) {
  if (CI >= N_CHUNKS) return;
  constexpr int NB = stage_blocks(CI < N_CHUNKS ? CI : 0);
  char* slot = c.lds + (CI & 1) * SLOT_BYTES;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int blk = c.wave + 4 * k;
    if (4 * k < NB && blk < NB) *reinterpret_cast<f32x4*>(slot + blk * 1024 + c.lane16) = st.v[k];
  }
}
__device__ __forceinline__ u32x4 lds_a(const Ctx& c, int slot, int ks, int piece) {
  return *reinterpret_cast<const u32x4*>(c.lds + slot * SLOT_BYTES + (ks * 3 + piece) * 1024 + c.lane16);
}

// One chunk: NKS k-steps read from LDS slot CI & 1; B fragments come from bsel(ks); fn(ks) = VALU work to interleave.
// While it runs, this wave's quarter of chunk CI+1 is in flight; it is committed to the other slot at the end, then the
// workgroup barrier hands the slots over.
template <int CI, int NCH, class BSel, class F>
__device__ __forceinline__ void run_chunk(const Ctx& c, f32x16& acc, BSel bsel, F fn) {
  constexpr int NKS = CHUNKS.ks[CI];
  constexpr int slot = CI & 1;
  constexpr int NEXT = CI + 1 < NCH ? CI + 1 : N_CHUNKS;  // nothing to stage after the last chunk of this variant
  Stage st;
  stage_issue<NEXT>(c, st);
  u32x4 a_cur[3], a_nxt[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) a_cur[p] = lds_a(c, slot, 0, p);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if (ks + 1 < NKS) {
#pragma unroll
      for (int p = 0; p < 3; ++p) a_nxt[p] = lds_a(c, slot, ks + 1, p);
    }
    mfma6(acc, a_cur, bsel(ks));
    fn(ks);
#pragma unroll
    for (int p = 0; p < 3; ++p) a_cur[p] = a_nxt[p];
    __builtin_amdgcn_sched_barrier(0);
  }
  stage_commit<NEXT>(c, st);
  __syncthreads();
}

// ---- gather / posenc (identical arithmetic to sdf_mlp.hip) -------------------------------------------------------------
template <bool GRAD>
__device__ __forceinline__ void gather_features(const SdfArgs& a, int h, float px, float py, float pz, float (&phi)[16],
                                                float (&J)[14][3]) {
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
      rows[sl][c] = D > 0 ? table[((int64_t)xi * D + yi) * D + zi] : -1;
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
    for (int c = 0; c < 8; ++c) {
      const int dx = c >> 2, dy = (c >> 1) & 1, dz = c & 1;
      const float ok = rows[sl][c] >= 0 ? 1.0f : 0.0f;
      const float wx = dx ? tx[sl] : 1.0f - tx[sl];
      const float wy = dy ? ty[sl] : 1.0f - ty[sl];
      const float wz = dz ? tz[sl] : 1.0f - tz[sl];
      const float w = wx * wy * wz * ok;
      const float f[7] = {f0[c][0], f0[c][1], f0[c][2], f0[c][3], f1[c][0], f1[c][1], f1[c][2]};
      float cx = 0.f, cy = 0.f, cz = 0.f;
      if (GRAD) {
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

__device__ __forceinline__ void posenc_half(int h, float x, float y, float z, float (&e)[16], float (&je)[14], bool want_j) {
  float all[28], jall[28];
  all[0] = x; all[1] = y; all[2] = z;
  jall[0] = jall[1] = jall[2] = 1.0f;
  const float p[3] = {x, y, z};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
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

// 16 local channels (14 data + the bias one + pad) -> two k-step fragments
__device__ __forceinline__ void local_frags(const float (&v)[16], Frag3 (&f)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) frag_set_pair(f[s], pr, v[8 * s + 2 * pr], v[8 * s + 2 * pr + 1]);
}

// ---- forward tile (layer L, tile T) -----------------------------------------------------------------------------------
// hin/hout: fragments of the 128 hidden activations: index 2*tile + s.  `raw` = pre-activations of the tile finished
// before this one; they are converted under this tile's MFMAs (two elements per k-step).
template <bool GRAD, int L, int T>
__device__ __forceinline__ void fwd_tile(const Ctx& c, f32x16& raw, Frag3* hin, Frag3* hout, const Frag3 (&ef)[2],
                                         const Frag3 (&pf)[2], Frag3* dfr, float& y0) {
  constexpr int CI = fwd_chunk(L, T);
  constexpr int NH = fwd_nh(L), NEk = fwd_ne(L);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const f32x16 prev = raw;
  f32x4 w6[4];
  if (L == 5 && T > 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) w6[g] = bload(c.tr, c.h * 256, (TAIL_W6H * 4) + ((T - 1) * 4 + g) * 16);
  }
  f32x4 sbuf = {0.f, 0.f, 0.f, 0.f};
  // element pair (2q, 2q+1) of the previous tile
  auto cvt_pair = [&](int q, Frag3* dst, int dst_tile, int s_layer) __attribute__((always_inline)) {
    float hv[2], sv[2];
    softplus100(prev[2 * q], hv[0], sv[0]);
    softplus100(prev[2 * q + 1], hv[1], sv[1]);
    const int el = 2 * q;
    if (L == 5 && T > 0) {
      const float w0 = w6[el >> 2][el & 3], w1 = w6[(el + 1) >> 2][(el + 1) & 3];
      y0 = fmaf(w0, hv[0], y0);
      y0 = fmaf(w1, hv[1], y0);
      if (GRAD) frag_set_pair(dfr[2 * dst_tile + (el >> 3)], (el & 7) >> 1, sv[0] * w0, sv[1] * w1);
    } else {
      frag_set_pair(dst[2 * dst_tile + (el >> 3)], (el & 7) >> 1, hv[0], hv[1]);
      sbuf[el & 3] = sv[0];
      sbuf[(el & 3) + 1] = sv[1];
      if (GRAD && (el & 3) == 2) bstore(c.sr, c.svoff, s_layer * 16384 + (dst_tile * 4 + (el >> 2)) * 1024, sbuf);
    }
  };
  auto fn = [&](int ks) __attribute__((always_inline)) {
    if (L == 0) {
      if (T > 0 && ks < 2) {  // only two k-steps per tile in layer 0: four pairs each
#pragma unroll
        for (int u = 0; u < 4; ++u) cvt_pair(4 * ks + u, hout, T - 1, 0);
      }
    } else if (T == 0) {
      if (ks < 4) {  // tile 3 of the previous layer (needed from hidden k-step 6 on), two pairs per k-step
        cvt_pair(2 * ks, hin, 3, L - 1);
        cvt_pair(2 * ks + 1, hin, 3, L - 1);
      }
    } else if (ks < 8) {
      cvt_pair(ks, hout, T - 1, L);
    }
  };
  auto bsel = [&](int ks) __attribute__((always_inline)) -> const Frag3& {
    if (L == 0) return ef[ks];
    if (ks < NH) return hin[ks];
    if (ks < NH + NEk) return ef[ks - NH];
    return pf[ks - NH - NEk];
  };
  run_chunk<CI, GRAD ? N_CHUNKS : N_FWD_CHUNKS>(c, acc, bsel, fn);
  raw = acc;
}

template <bool GRAD, int L>
__device__ __forceinline__ void fwd_layer(const Ctx& c, f32x16& raw, Frag3* hin, Frag3* hout, const Frag3 (&ef)[2],
                                          const Frag3 (&pf)[2], Frag3* dfr, float& y0) {
  fwd_tile<GRAD, L, 0>(c, raw, hin, hout, ef, pf, dfr, y0);
  fwd_tile<GRAD, L, 1>(c, raw, hin, hout, ef, pf, dfr, y0);
  fwd_tile<GRAD, L, 2>(c, raw, hin, hout, ef, pf, dfr, y0);
  fwd_tile<GRAD, L, 3>(c, raw, hin, hout, ef, pf, dfr, y0);
}

// ---- backward tiles ----------------------------------------------------------------------------------------------------
template <int L, int T>
__device__ __forceinline__ void bwd_hidden_tile(const Ctx& c, const Frag3* din, Frag3* dout) {
  constexpr int CI = bwd_chunk(L, T);
  f32x4 sS[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) sS[g] = bload(c.sr, c.svoff, (L - 1) * 16384 + (T * 4 + g) * 1024);
  f32x16 G;
#pragma unroll
  for (int r = 0; r < 16; ++r) G[r] = 0.f;
  run_chunk<CI, N_CHUNKS>(c, G, [&](int ks) __attribute__((always_inline)) -> const Frag3& { return din[ks]; }, [](int) {});
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int el = 2 * q;
    frag_set_pair(dout[2 * T + (el >> 3)], (el & 7) >> 1, sS[el >> 2][el & 3] * G[el], sS[(el + 1) >> 2][(el + 1) & 3] * G[el + 1]);
  }
}
template <int L, int T>
__device__ __forceinline__ void bwd_acc_tile(const Ctx& c, const Frag3* din, f32x16& acc) {
  run_chunk<bwd_chunk(L, T), N_CHUNKS>(c, acc, [&](int ks) __attribute__((always_inline)) -> const Frag3& { return din[ks]; }, [](int) {});
}
template <int L>
__device__ __forceinline__ void bwd_layer(const Ctx& c, const Frag3* din, Frag3* dout, f32x16& accE, f32x16& accP) {
  bwd_hidden_tile<L, 0>(c, din, dout);
  bwd_hidden_tile<L, 1>(c, din, dout);
  bwd_hidden_tile<L, 2>(c, din, dout);
  bwd_hidden_tile<L, 3>(c, din, dout);
  if (L == 3) {
    bwd_acc_tile<L, 4>(c, din, accE);
    bwd_acc_tile<L, BWD_NT[L] - 1>(c, din, accP);
  } else {
    bwd_acc_tile<L, 4>(c, din, accP);
  }
}

template <bool GRAD>
__global__ __launch_bounds__(WPB * 64, 1) void sdf_mlp_bf16_kernel(SdfArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[2 * SLOT_BYTES];
  Ctx c;
  c.lane = threadIdx.x & 63;
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.h = c.lane >> 5;
  c.lane16 = c.lane * 16;
  c.lds = lds;
  c.wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.packed, 0, STREAM_BYTES, 0x00020000);
  c.tr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.packed + STREAM_BYTES), 0, TAIL_FLOATS * 4, 0x00020000);
  c.sr = __builtin_amdgcn_make_buffer_rsrc((void*)a.scratch, 0, GRAD ? 0x7fffffff : 0, 0x00020000);
  const int64_t wave_id = (int64_t)blockIdx.x * WPB + c.wave;
  c.svoff = (int)(wave_id * (SCR_SLOT * 4)) + c.lane * 16;
  const int64_t n_tiles = (a.n + TILE - 1) / TILE;
  const int64_t n_rounds = (n_tiles + WPB - 1) / WPB;
  constexpr int NCH = GRAD ? N_CHUNKS : N_FWD_CHUNKS;
  (void)NCH;

  for (int64_t round = blockIdx.x; round < n_rounds; round += gridDim.x) {
    const int64_t tile = round * WPB + c.wave;
    const int64_t slot0 = tile * TILE + (c.lane & 31);
    const int64_t sc = slot0 < a.n ? slot0 : a.n - 1;
    const int64_t i = a.idx ? (int64_t)a.idx[sc] : sc;
    const bool active = (slot0 < a.n) && (!a.mask || a.mask[i] != 0);
    const float px = a.pts[i * 3 + 0], py = a.pts[i * 3 + 1], pz = a.pts[i * 3 + 2];

    // stage chunk 0 (the previous round has passed its last barrier, both slots are free)
    {
      Stage st;
      stage_issue<0>(c, st);
      stage_commit<0>(c, st);
    }

    float phi[16];
    Frag3 ef[2], pf[2];
    {
      float e[16];
      float J[14][3];
      gather_features<GRAD>(a, c.h, px, py, pz, phi, J);
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
      posenc_half(c.h, px, py, pz, e, je_unused, false);
      e[14] = 1.0f;  // bias k-element (weights carry the bias there, lane half 0 only)
      phi[14] = 1.0f;
      local_frags(e, ef);
      local_frags(phi, pf);
    }
    __syncthreads();  // chunk 0 visible

    // ------------------------------------------------ forward ----------------------------------------------------
    Frag3 hA[8], hB[8], dA[8];
    f32x16 raw;
#pragma unroll
    for (int r = 0; r < 16; ++r) raw[r] = 0.f;
    float y0 = 0.f;
    fwd_layer<GRAD, 0>(c, raw, hA, hA, ef, pf, dA, y0);
    fwd_layer<GRAD, 1>(c, raw, hA, hB, ef, pf, dA, y0);
    fwd_layer<GRAD, 2>(c, raw, hB, hA, ef, pf, dA, y0);
    fwd_layer<GRAD, 3>(c, raw, hA, hB, ef, pf, dA, y0);
    fwd_layer<GRAD, 4>(c, raw, hB, hA, ef, pf, dA, y0);
    fwd_layer<GRAD, 5>(c, raw, hA, hB, ef, pf, dA, y0);
    {  // tile 3 of layer 5 and the feature part of the last layer
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 w = bload(c.tr, c.h * 256, TAIL_W6H * 4 + (12 + g) * 16);
        float hv[4], sv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          softplus100(raw[4 * g + q], hv[q], sv[q]);
          y0 = fmaf(w[q], hv[q], y0);
        }
        if (GRAD) {
          frag_set_pair(dA[6 + (g >> 1)], 2 * (g & 1), sv[0] * w[0], sv[1] * w[1]);
          frag_set_pair(dA[6 + (g >> 1)], 2 * (g & 1) + 1, sv[2] * w[2], sv[3] * w[3]);
        }
      }
    }
    f32x16 accP;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 w = bload(c.tr, c.h * 64, TAIL_W6P * 4 + g * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (4 * g + q < 14) y0 = fmaf(w[q], phi[4 * g + q], y0);
        accP[4 * g + q] = w[q];
      }
    }
    y0 += __shfl_xor(y0, 32);
    {
      const f32x4 b6 = bload(c.tr, 0, TAIL_B6 * 4);
      y0 += b6[0];
    }
    if (active && c.h == 0) a.sdf[i] = y0;
    if (GRAD) {
      // ---------------------------------------------- reverse sweep ----------------------------------------------
      f32x16 accE;
#pragma unroll
      for (int r = 0; r < 16; ++r) accE[r] = 0.f;
      bwd_layer<5>(c, dA, hA, accE, accP);
      bwd_layer<4>(c, hA, dA, accE, accP);
      bwd_layer<3>(c, dA, hA, accE, accP);
      bwd_layer<2>(c, hA, dA, accE, accP);
      bwd_layer<1>(c, dA, hA, accE, accP);
      bwd_acc_tile<0, 0>(c, hA, accE);

      float g3[3] = {0.f, 0.f, 0.f};
      {
        float e2[16], je[14];
        posenc_half(c.h, px, py, pz, e2, je, true);
#pragma unroll
        for (int s2 = 0; s2 < 14; ++s2) {
          const int c0 = s2 % 3, c1 = (14 + s2) % 3;
          const float v = accE[s2] * je[s2];
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) g3[ax] += ((c.h ? c1 : c0) == ax) ? v : 0.f;
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
      if (active && c.h == 0) {
        a.grad[i * 3 + 0] = g3[0];
        a.grad[i * 3 + 1] = g3[1];
        a.grad[i * 3 + 2] = g3[2];
      }
    }
  }
}

int grid_blocks(int64_t n) {
  int64_t tiles = (n + TILE - 1) / TILE;
  int64_t rounds = (tiles + WPB - 1) / WPB;
  return (int)(rounds < MAX_BLOCKS ? rounds : MAX_BLOCKS);
}

// ---- host packer ------------------------------------------------------------------------------------------------------
inline uint16_t bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float bf16_to_f(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline void split3_host(float v, uint16_t (&p)[3]) {
  p[0] = bf16_rne(v);
  float r = v - bf16_to_f(p[0]);
  p[1] = bf16_rne(r);
  r = r - bf16_to_f(p[1]);
  p[2] = bf16_rne(r);
}
inline int frag_feat(int tt, int s, int j, int h) { return 32 * tt + 16 * s + (j & 3) + 8 * (j >> 2) + 4 * h; }

}  // namespace

extern "C" int64_t surf_sdf_bf16_packed_bytes(void) { return PACKED_BYTES; }

extern "C" int64_t surf_sdf_bf16_scratch_bytes(int64_t n_points) {
  if (n_points <= 0) return 0;
  return (int64_t)grid_blocks(n_points) * WPB * SCR_SLOT * sizeof(float);
}

// h_W / h_b: effective (weight-normed) fp32 matrices lin0..lin6, as for surf_sdf_pack_weights.
extern "C" int surf_sdf_pack_weights_bf16(const float* const* h_W, const float* const* h_b, unsigned char* out) {
  if (!h_W || !h_b || !out) return SURF_E_ARG;
  for (int l = 0; l < 7; ++l)
    if (!h_W[l] || !h_b[l]) return SURF_E_ARG;
  const int in_dim[7] = {NE, 156, 156, 156, 156, 156, 156};
  const int out_dim[6] = {HID, HID, H2, HID, HID, HID};
  const float rsqrt2 = (float)(1.0 / sqrt(2.0));
  memset(out, 0, PACKED_BYTES);
  auto put = [&](int chunk, int ks, int lane, int j, float v) {
    uint16_t p[3];
    split3_host(v, p);
    for (int pc = 0; pc < 3; ++pc) {
      uint16_t* dst = reinterpret_cast<uint16_t*>(out + CHUNKS.off[chunk] + (ks * 3 + pc) * 1024 + lane * 16);
      dst[j] = p[pc];
    }
  };
  // ---- forward
  for (int l = 0; l < 6; ++l)
    for (int t = 0; t < 4; ++t) {
      const int ci = fwd_chunk(l, t);
      const int hid_in = (l == 3) ? H2 : HID;
      for (int ks = 0; ks < fwd_ks(l); ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int h = lane >> 5, row = 32 * t + (lane & 31);
            int col = -1;
            float scale = 1.f;
            bool is_bias = false;
            if (ks < fwd_nh(l)) {
              const int f = frag_feat(ks >> 1, ks & 1, j, h);
              if (f < hid_in) col = f;
              if (l == 3) scale = rsqrt2;
            } else if (ks < fwd_nh(l) + fwd_ne(l)) {
              const int cidx = 8 * (ks - fwd_nh(l)) + j;
              if (cidx < 14 && 14 * h + cidx < NE) col = (l == 3 ? H2 : 0) + 14 * h + cidx;
              if (l == 3) scale = rsqrt2;
              if (l == 0) is_bias = (cidx == 14 && h == 0);
            } else {
              const int cidx = 8 * (ks - fwd_nh(l) - fwd_ne(l)) + j;
              if (cidx < 14) col = 128 + 14 * h + cidx;
              is_bias = (cidx == 14 && h == 0);
            }
            float v = 0.f;
            if (col >= 0 && row < out_dim[l]) v = h_W[l][(int64_t)row * in_dim[l] + col] * scale;
            if (is_bias && row < out_dim[l]) v = h_b[l][row];
            put(ci, ks, lane, j, v);
          }
    }
  // ---- backward: G_in = W_l^T delta_l
  for (int l = 0; l < 6; ++l)
    for (int t = 0; t < BWD_NT[l]; ++t) {
      const int ci = bwd_chunk(l, t);
      const int hid_in = (l == 3) ? H2 : HID;
      for (int ks = 0; ks < bwd_ks(l); ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int h = lane >> 5, rho = lane & 31;
            const int krow = frag_feat(ks >> 1, ks & 1, j, h);
            const int h_row = (rho >> 2) & 1, r_row = (rho & 3) | ((rho >> 3) << 2);
            const int ch = 14 * h_row + r_row;
            int kind;  // 0 hidden, 1 E, 2 P
            if (l == 0) kind = 1;
            else if (t < 4) kind = 0;
            else if (l == 3 && t == 4) kind = 1;
            else kind = 2;
            int col = -1;
            float scale = 1.f;
            if (kind == 0) {
              const int cc = 32 * t + rho;
              if (cc < hid_in) col = cc;
              if (l == 3) scale = rsqrt2;
            } else if (kind == 1) {
              if (r_row < 14 && ch < NE) col = (l == 3 ? H2 : 0) + ch;
              if (l == 3) scale = rsqrt2;
            } else {
              if (r_row < 14) col = 128 + ch;
            }
            float v = 0.f;
            if (col >= 0 && krow < out_dim[l]) v = h_W[l][(int64_t)krow * in_dim[l] + col] * scale;
            put(ci, ks, lane, j, v);
          }
    }
  // ---- fp32 tail: last layer row 0
  float* tail = reinterpret_cast<float*>(out + STREAM_BYTES);
  auto hk = [](int tt, int r, int h) { return 32 * tt + (r & 3) + 8 * (r >> 2) + 4 * h; };
  for (int h = 0; h < 2; ++h) {
    for (int s = 0; s < 64; ++s) tail[TAIL_W6H + h * 64 + s] = h_W[6][hk(s / 16, s % 16, h)];
    for (int s = 0; s < 14; ++s) tail[TAIL_W6P + h * 16 + s] = h_W[6][128 + 14 * h + s];
  }
  tail[TAIL_B6] = h_b[6][0];
  return 0;
}

extern "C" int surf_sdf_mlp_bf16x3(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n,
                                   const float* const* h_vols, const int32_t* const* h_tables, const int* h_dims, int n_vol,
                                   const void* packed, float* sdf, float* grad, void* scratch, void* stream) {
  if (!pts || !h_vols || !h_tables || !h_dims || !packed || !sdf) return SURF_E_ARG;
  if (n <= 0 || n_vol <= 0) return SURF_E_ARG;
  if (n_vol > SURF_MAX_STAGES) return SURF_E_LIMIT;
  if (grad && !scratch) return SURF_E_ARG;
  SdfArgs a;
  a.pts = pts; a.mask = mask; a.idx = idx; a.n = n; a.packed = (const unsigned char*)packed; a.sdf = sdf; a.grad = grad;
  a.scratch = (float*)scratch;
  for (int s = 0; s < SURF_MAX_STAGES; ++s) {
    a.vols[s] = s < n_vol ? h_vols[s] : h_vols[0];
    a.tables[s] = s < n_vol ? h_tables[s] : nullptr;
    a.dims[s] = s < n_vol ? h_dims[s] : 0;
    if (s < n_vol && (!h_vols[s] || !h_tables[s] || h_dims[s] <= 1)) return SURF_E_ARG;
  }
  dim3 grid(grid_blocks(n)), block(WPB * 64);
  if (grad)
    hipLaunchKernelGGL(sdf_mlp_bf16_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(sdf_mlp_bf16_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
  return surf_check_launch();
}
