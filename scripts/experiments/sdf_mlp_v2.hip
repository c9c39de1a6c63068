// K9/K10 v2: the split-operand SDF network with TWO wavefronts per SIMD.
//
// Same function, same packed weight buffer and same scratch layout as sdf_mlp_split.hip (SDFNetworkSparse.forward / .sdf /
// .gradient, sdf_network.py:95-141; sparse gather projector.py:217-390); what differs is the loop order and where the
// activations live:
//
//   sdf_mlp_split.hip   tile-outer: one 32-row output tile at a time, its 8-12 k-steps inside; the 128 hidden activations
//                       are kept as SPLIT 16-bit fragments in registers (3 x 96 registers for bf16x3) - the wavefront
//                       needs ~480 registers, one wavefront per SIMD, every memory latency is exposed to the matrix pipe.
//   this file           k-step-outer: the four output tiles of a layer accumulate side by side (4 x 16 accumulator
//                       registers); the activations of the layer below stay as the fp32 ACCUMULATORS they were produced in
//                       (the D layout of one MFMA tile is the B layout of two k-steps of the next layer, frag_feat) and are
//                       activated + split lazily, one k-step ahead of their use.  Live state: 2 x 64 accumulators + two
//                       B fragments + one A fragment: <= 256 registers, so a workgroup is EIGHT wavefronts, two per SIMD,
//                       and one wavefront's gather / LDS / scratch / barrier latency hides behind the other's MFMAs.
//
// Weight stream: one chunk = one k-step of a layer for all its output tiles (12-18 KB for bf16x3), read from the SAME
// packed buffer as the tile-outer kernel (a chunk is 4-6 strided 1 KB-block runs of it), staged by LDS-DMA into a ring of
// NS slots and retired by a counted vmcnt spanning NS-2 barrier intervals (check_isa.py checks the counts on the ISA).
#define SURF_GATHER_SEQUENTIAL 1
#include "sdf_split_common.h"

namespace {

#ifndef SURF_V2_WPB   // wavefronts per workgroup: 8 = one workgroup per CU, 4 = two independent workgroups per CU
#define SURF_V2_WPB 8
#endif
#ifndef SURF_V2_NS    // LDS ring length (slots of one k-step)
#define SURF_V2_NS 5
#endif
constexpr int WPB2 = SURF_V2_WPB;
constexpr int V2_OCC = 8 / WPB2;                // workgroups per CU: two wavefronts per SIMD either way

// ---- chunk stream: one k-step per chunk ------------------------------------------------------------------------------
constexpr int V2_NFWD = fwd_ks(0) + fwd_ks(1) + fwd_ks(2) + fwd_ks(3) + fwd_ks(4) + fwd_ks(5);   // 53
constexpr int V2_NBWD = bwd_ks(5) + bwd_ks(4) + bwd_ks(3) + bwd_ks(2) + bwd_ks(1) + bwd_ks(0);   // 47
constexpr int v2_fwd_chunk(int l, int ks) {
  int n = 0;
  for (int i = 0; i < l; ++i) n += fwd_ks(i);
  return n + ks;
}
constexpr int v2_bwd_chunk(int l, int ks) {  // backward runs l = 5 .. 0
  int n = V2_NFWD;
  for (int i = 5; i > l; --i) n += bwd_ks(i);
  return n + ks;
}
struct V2Chunk { int layer, ks, nt; bool fwd; };
constexpr V2Chunk v2_chunk(int ci, bool grad) {
  if (ci < V2_NFWD) {
    int l = 0;
    while (ci >= fwd_ks(l)) { ci -= fwd_ks(l); ++l; }
    return V2Chunk{l, ci, 4, true};
  }
  ci -= V2_NFWD;
  if (!grad || ci >= V2_NBWD) return V2Chunk{0, 0, 0, false};  // padding chunk
  int l = 5;
  while (ci >= bwd_ks(l)) { ci -= bwd_ks(l); --l; }
  return V2Chunk{l, ci, BWD_NT[l], false};
}
template <class P> constexpr int v2_ns(bool grad) { return SURF_V2_NS; }
template <class P> constexpr int v2_nch(bool grad) {
  const int n = grad ? V2_NFWD + V2_NBWD : V2_NFWD, ns = v2_ns<P>(grad);
  return (n + ns - 1) / ns * ns;
}
// DMA pieces of chunk ci EVERY wavefront issues (wavefronts below blocks % 8 issue one more: counting the minimum only
// makes the vmcnt waits stricter)
template <class P> constexpr int v2_ndma(int ci, bool grad) { return v2_chunk(ci, grad).nt * P::NP / WPB2; }
// slot = the largest chunk (6 tiles backward, 4 forward)
template <class P> constexpr int v2_slot(bool grad) { return (grad ? 6 : 4) * P::NP * 1024; }

#ifndef SURF_V2_X  // timing experiments only (wrong results): 1 no vmcnt wait, 2 no conversions, 4 no DMA, 8 no barrier
#define SURF_V2_X 0
#endif
#ifdef SURF_V2_TIMING  // debug builds: shader-clock totals of wavefront 0 of every workgroup: [work, barrier wait]
__device__ unsigned long long g_v2_phase[8];
#define V2_T(k)                                                   \
  do {                                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    c.tacc[k] += now_ - c.tprev;                                  \
    c.tprev = now_;                                               \
  } while (0)
#if SURF_V2_TIMING > 1   // per-step stamps (work / barrier wait) as well: +10 % run time
#define V2_TS(k) V2_T(k)
#else
#define V2_TS(k)
#endif
#else
#define V2_T(k)
#define V2_TS(k)
#endif

struct Ctx2 {
#ifdef SURF_V2_TIMING
  mutable unsigned long long tprev;
  mutable unsigned long long tacc[8];
#endif
  rsrc_t wr, sr, sl, tr;  // packed stream, scratch (stores / loads), fp32 tail
  int lane, lane16, h, svoff, wave;
  char* lds;
};

// DMA of this wavefront's share of chunk CI: blocks wave + 8 k below the chunk's block count.  Block b = tile b / NP,
// piece b % NP lives in the tile-outer stream at off[chunk(l, tile)] + (ks NP + piece) KB; the tiles of one layer are
// equally long and consecutive, so the source offset is affine in (tile, piece).
// An LDS-DMA instruction holds the issuing wavefront for 60-185 cycles (MI355X_MICROARCH.md, per-instruction constants)
// and the CU's address unit serves one at a time: issued by all eight wavefronts right behind the barrier they cost
// ~2.2 k cycles per k-step with every matrix pipe idle (measured: 90 ms vs 30 ms without any DMA).  So each wavefront
// issues its pieces in front of a DIFFERENT output tile of the k-step (dma_turn), the two wavefronts of a SIMD two tiles
// apart: the partner's MFMAs run while one is held.
template <class P, bool GRAD, int CI>
__device__ __forceinline__ void v2_dma(const Ctx2& c) {
  constexpr V2Chunk ch = v2_chunk(CI, GRAD);
  constexpr int NP = P::NP, NB = ch.nt * NP;
  constexpr int C0 = ch.nt == 0 ? 0 : (ch.fwd ? P::CH.off[fwd_chunk(ch.layer, 0)] : P::CH.off[bwd_chunk(ch.layer, 0)]) + ch.ks * NP * 1024;
  constexpr int C1 = (ch.fwd ? fwd_ks(ch.layer) : bwd_ks(ch.layer)) * NP * 1024;  // stream bytes of one tile
  constexpr int SLOT = (CI % v2_ns<P>(GRAD)) * v2_slot<P>(GRAD);
  if (SURF_V2_X & 4) return;
#pragma unroll
  for (int k = 0; k < (NB + WPB2 - 1) / WPB2; ++k) {
    const int b = c.wave + WPB2 * k;
    if (b < NB) {                                       // wave-uniform
      const int t = NP == 3 ? (b * 43) >> 7 : b >> 1;   // b / NP for b < 32
      const int p = b - t * NP;
      const int src = (SURF_V2_X & 256) ? b * 1024 : C0 + t * C1 + p * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(c.wr, (__attribute__((address_space(3))) void*)(c.lds + SLOT + b * 1024), 16,
                                               c.lane16, src, 0, 0);
    }
  }
}
// in front of which of the NT output tiles of a k-step this wavefront issues its DMA pieces
__device__ __forceinline__ int dma_turn(const Ctx2& c, int nt) {
  const int t = ((c.wave & 3) + 2 * (c.wave >> 2)) & 3;   // (wave >> 2 = 0 with four wavefronts per workgroup)
  return nt >= 4 ? t : 0;
}

// ---- the per-k-step MFMA block -----------------------------------------------------------------------------------------
template <class P>
__device__ __forceinline__ void mma_tile(f32x16& acc, const u32x4 (&a)[P::NP], const FragT<P::NP>& b) {
  typename P::Acc t;
#pragma unroll
  for (int q = 0; q < P::NA; ++q) t.v[q] = acc;
  static_assert(P::NA == 1, "one accumulator chain");
  P::mma(t, a, b);
  acc = t.v[0];
}

// Pair q (elements 2q, 2q+1 of the lane's eight) of hidden k-step hk, produced from the accumulators `src` of the layer
// below, one k-step ahead of its use and BETWEEN the MFMAs of output tile q of the running k-step:
//   MODE 0 (forward):   h = softplus(t);  fragment <- h;  (GRAD) softplus' -> scratch slice (S_LAYER, tile, groups 2s, 2s+1)
//   MODE 1 (backward 5): delta_5 = softplus'(t) w6 (x D);  y0 += w6 softplus(t)
//   MODE 2 (backward <5): delta_l = softplus'_l (from scratch, `sp`) x G_l
template <class P, bool GRAD, int MODE, int S_LAYER>
__device__ __forceinline__ void convert_pair(const Ctx2& c, const f32x16 (&src)[4], int hk, int q, FragT<P::NP>& dst, float& y0,
                                             const f32x4 (&w6)[2], const f32x4 (&sp)[2], f32x4& sbuf) {
  const int tt = hk >> 1, s = hk & 1;
  const int r = 8 * s + 2 * q;
  if (SURF_V2_X & 2) {
    dst.p[0][q] = __builtin_bit_cast(uint32_t, src[tt][r]);
    return;
  }
  const f32x2 t2 = {src[tt][r], src[tt][r + 1]};
  if (MODE == 2) {
    constexpr float inv_w = 1.0f / Scales<P>::W;
    float d0 = sp[q >> 1][2 * (q & 1)] * t2[0], d1 = sp[q >> 1][2 * (q & 1) + 1] * t2[1];
    if (inv_w != 1.0f) { d0 *= inv_w; d1 *= inv_w; }
    frag_set_pair<P>(dst, q, d0, d1);
  } else {
    f32x2 hv, sv;
    softplus_pair<(GRAD || MODE == 1)>(t2, 1.0f / Scales<P>::W, hv, sv);
    if (MODE == 1) {
      const float w0 = w6[q >> 1][2 * (q & 1)], w1 = w6[q >> 1][2 * (q & 1) + 1];
      y0 = fmaf(w0, hv[0], y0);
      y0 = fmaf(w1, hv[1], y0);
      frag_set_pair<P>(dst, q, sv[0] * (w0 * Scales<P>::D), sv[1] * (w1 * Scales<P>::D));
    } else {
      frag_set_pair<P>(dst, q, hv[0], hv[1]);
      sbuf[2 * (q & 1)] = sv[0];
      sbuf[2 * (q & 1) + 1] = sv[1];
      if (GRAD && (q & 1)) bstore(c.sr, c.svoff, S_LAYER * 16384 + (tt * 4 + 2 * s + (q >> 1)) * 1024, sbuf);
    }
  }
}
template <class P, bool GRAD, int MODE, int S_LAYER>
__device__ __forceinline__ void convert_kstep(const Ctx2& c, const f32x16 (&src)[4], int hk, FragT<P::NP>& dst, float& y0,
                                              const f32x4 (&w6)[2], const f32x4 (&sp)[2]) {
  f32x4 sbuf;
#pragma unroll
  for (int q = 0; q < 4; ++q) convert_pair<P, GRAD, MODE, S_LAYER>(c, src, hk, q, dst, y0, w6, sp, sbuf);
}

template <class P>
__device__ __forceinline__ void local_pair(const float (&v)[16], int s, int pr, FragT<P::NP>& f) {
  frag_set_pair<P>(f, pr, v[8 * s + 2 * pr], v[8 * s + 2 * pr + 1]);
}
template <class P>
__device__ __forceinline__ void local_frag(const float (&v)[16], int s, FragT<P::NP>& f) {
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) local_pair<P>(v, s, pr, f);
}

// MFMA : VALU interleave of one output tile's six products with the conversion work placed beside it
#ifndef SURF_V2_SGB
#define SURF_V2_SGB 5
#endif
__device__ __forceinline__ void tile_sched(int n_mfma) {
#if SURF_V2_SGB > 0
  for (int m = 0; m < 6; ++m) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, SURF_V2_SGB, 0);
  }
#endif
}

// ---- segments, barriers and the counted vmcnt ---------------------------------------------------------------------------
// Every k-step is two segments with a workgroup barrier after each:
//   M  the LDS reads of the A fragments and the MFMAs of all output tiles (no vector-memory operation),
//   V  the DMA pieces of chunk CI+NS-1, the loads feeding later conversions, the activation / split of the next B
//      fragment and its scratch stores.
// Wavefronts 4..7 (the SIMD partners of 0..3) run ONE SEGMENT BEHIND wavefronts 0..3 (one extra barrier at kernel start,
// one for the others at the end), so a SIMD always pairs one wavefront's M segment with the other's V segment: the matrix
// pipe is per-SIMD and in-order streams only overlap when the partner's work is complementary
// (MI355X_MICROARCH.md, Two waves per SIMD).  With both wavefronts in the same phase two per SIMD ran exactly twice as
// long as one (measured).
// The barrier that ends M of chunk CI retires the DMA of chunk CI+1, issued in V of chunk CI-(NS-2): the vector-memory
// operations of the NS-3 V segments since may stay in flight (2 (NS-3) barrier intervals for check_isa.py).
template <int N, int WINDOW>
__device__ __forceinline__ void v2_barrier_m() {
  static_assert(N >= 0 && N < 64, "vmcnt");
  if (SURF_V2_X & 8) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return; }
  asm volatile("; surf_ring_window %1\n\ts_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((SURF_V2_X & 1) ? 63 : N), "n"(WINDOW) : "memory");
}
__device__ __forceinline__ void v2_barrier_v() {
  if (SURF_V2_X & 8) return;
  asm volatile("s_barrier" ::: "memory");
}

// backward fragments: f = 0..46 <-> (layer 5..0, k-step): fragment f is consumed by backward step f, converted in the V
// segment of step f-1 from loads issued in the V segment of step f-2
constexpr int bfrag_layer(int f) {
  int l = 5;
  while (l > 0 && f >= bwd_ks(l)) { f -= bwd_ks(l); --l; }
  return l;
}
constexpr int bfrag_hk(int f) {
  int l = 5;
  while (l > 0 && f >= bwd_ks(l)) { f -= bwd_ks(l); --l; }
  return f;
}
// the two 16-byte groups feeding fragment F: layer 5: lin6 row 0 (w6); layers < 5: the stored softplus'
template <int F>
__device__ __forceinline__ void load_frag(const Ctx2& c, f32x4 (&ld)[2]) {
  if constexpr (F < V2_NBWD) {
    constexpr int L = bfrag_layer(F), HK = bfrag_hk(F), G0 = (HK >> 1) * 4 + 2 * (HK & 1);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (L == 5) ld[g] = bload(c.tr, c.h * 256, (TAIL_W6H * 4) + (G0 + g) * 16);
      else ld[g] = bload(c.sl, c.svoff, L * 16384 + (G0 + g) * 1024);
    }
  }
}

// vector-memory operations of the V segment of chunk ci (every wavefront issues at least these)
template <class P, bool GRAD>
constexpr int v2_ops(int ci) {
  constexpr int NCH = v2_nch<P>(GRAD), DIST = v2_ns<P>(GRAD) - 1;
  const V2Chunk ch = v2_chunk(ci, GRAD);
  int n = v2_ndma<P>((ci + DIST) % NCH, GRAD);
  if (ch.nt == 0) return n;
  if (ch.fwd) {
    const int nk = ch.ks + 1;  // stores of softplus' when the fragment produced here is a hidden one of this layer
    if (GRAD && nk < fwd_ks(ch.layer) && nk >= fwd_nl(ch.layer)) n += 2;
  } else {
    const int j = ci - V2_NFWD;
    if (j + 2 < V2_NBWD) n += 2;       // loads of fragment j + 2
  }
  return n;
}
template <class P, bool GRAD, int CI>
constexpr int v2_wait() {
  constexpr int NCH = v2_nch<P>(GRAD), W = v2_ns<P>(GRAD) - 3;
  static_assert(W >= 1, "ring of at least 4 slots");
  int n = 0;
  for (int j = 1; j <= W; ++j) n += v2_ops<P, GRAD>((CI - j + NCH) % NCH);
  return n;
}
template <class P, bool GRAD> constexpr int v2_window() { return 2 * (v2_ns<P>(GRAD) - 3); }

struct V2State {
  f32x16 accA[4], accB[4];
  f32x16 accE, accP;
  float e16[16], phi16[16];
  float y0;
};

// One forward k-step: B = bcur (prepared by the V segment before), 4 output tiles; V prepares the fragment of the next
// k-step of this layer (or the first of the next layer).
template <class P, bool GRAD, int L, int KS>
__device__ __forceinline__ void fwd_step(const Ctx2& c, V2State& st, const f32x16 (&prev)[4], f32x16 (&cur)[4],
                                         FragT<P::NP>& bcur, FragT<P::NP>& bnext) {
  constexpr int CI = v2_fwd_chunk(L, KS), NCH = v2_nch<P>(GRAD), NS = v2_ns<P>(GRAD), NP = P::NP;
  // ---- M
  const char* rd = c.lds + (CI % NS) * v2_slot<P>(GRAD) + c.lane16;
  if (KS == 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) cur[t][r] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {  // (no read-ahead: the SIMD partner's V segment covers the LDS latency, registers are scarce)
    u32x4 a[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p] = *reinterpret_cast<const u32x4*>(rd + (t * NP + p) * 1024);
    mma_tile<P>(cur[t], a, bcur);
  }
  __builtin_amdgcn_sched_barrier(0);
  V2_TS(1);
  v2_barrier_m<v2_wait<P, GRAD, CI>(), v2_window<P, GRAD>()>();
  V2_TS(2);
  // ---- V
  v2_dma<P, GRAD, (CI + NS - 1) % NCH>(c);
  constexpr int NK = fwd_ks(L);
  float y_unused = 0.f;
  const f32x4 none[2] = {};
  if constexpr (KS + 1 < NK) {
    constexpr int nk = KS + 1;
    if constexpr (nk < fwd_ne(L)) local_frag<P>(st.e16, nk, bnext);
    else if constexpr (nk < fwd_nl(L)) local_frag<P>(st.phi16, nk - fwd_ne(L), bnext);
    else convert_kstep<P, GRAD, 0, L - 1>(c, prev, nk - fwd_nl(L), bnext, y_unused, none, none);
  } else if constexpr (L < 5) {  // first k-step of layer L + 1: e (layer 3) or phi
    if constexpr (fwd_ne(L + 1) > 0) local_frag<P>(st.e16, 0, bnext);
    else local_frag<P>(st.phi16, 0, bnext);
  }
  __builtin_amdgcn_sched_barrier(0);
  v2_barrier_v();
}

template <class P, bool GRAD, int L, int KS>
__device__ __forceinline__ void fwd_steps(const Ctx2& c, V2State& st, const f32x16 (&prev)[4], f32x16 (&cur)[4],
                                          FragT<P::NP>& b0, FragT<P::NP>& b1) {
  if constexpr (KS < fwd_ks(L)) {
    fwd_step<P, GRAD, L, KS>(c, st, prev, cur, b0, b1);
    fwd_steps<P, GRAD, L, KS + 1>(c, st, prev, cur, b1, b0);
  }
}

// One backward k-step (layer L, k-step KS = backward step J): B = delta_L fragment J; output tiles: hidden 0..3 -> cur,
// then e (L = 3, 0) / phi.  V issues the loads of fragment J + 2 and converts fragment J + 1 (layer 5: from acc5 with
// w6; below: softplus' x G, where G = prev, or this layer's finished cur for the first fragment of the next layer down).
template <class P, int L, int KS>
__device__ __forceinline__ void bwd_step(const Ctx2& c, V2State& st, const f32x16 (&prev)[4], f32x16 (&cur)[4],
                                         FragT<P::NP>& bcur, FragT<P::NP>& bnext, f32x4 (&ld)[2][2], f32x4 (&Jq)[12]) {
  constexpr int CI = v2_bwd_chunk(L, KS), J = CI - V2_NFWD, NCH = v2_nch<P>(true), NS = v2_ns<P>(true), NP = P::NP;
  static_assert(bfrag_layer(J) == L && bfrag_hk(J) == KS, "fragment numbering");
  // ---- M
  const char* rd = c.lds + (CI % NS) * v2_slot<P>(true) + c.lane16;
  if (KS == 0 && L > 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) cur[t][r] = 0.f;
  }
  constexpr int NT = BWD_NT[L];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    u32x4 a[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p] = *reinterpret_cast<const u32x4*>(rd + (t * NP + p) * 1024);
    if (L == 0) mma_tile<P>(st.accE, a, bcur);
    else if (t < 4) mma_tile<P>(cur[t], a, bcur);
    else if (L == 3 && t == 4) mma_tile<P>(st.accE, a, bcur);
    else mma_tile<P>(st.accP, a, bcur);
  }
  __builtin_amdgcn_sched_barrier(0);
  V2_TS(3);
  v2_barrier_m<v2_wait<P, true, CI>(), v2_window<P, true>()>();
  V2_TS(4);
  // ---- V
  load_frag<J + 2>(c, ld[J & 1]);                   // (fragment J + 2 has the parity of J)
  v2_dma<P, true, (CI + NS - 1) % NCH>(c);
  if constexpr (J + 1 < V2_NBWD) {
    constexpr int NLY = bfrag_layer(J + 1), NHK = bfrag_hk(J + 1);
    const f32x4 (&lx)[2] = ld[(J + 1) & 1];
    if constexpr (NLY == 5) convert_kstep<P, true, 1, 0>(c, prev, NHK, bnext, st.y0, lx, lx);
    else if constexpr (NHK > 0) convert_kstep<P, true, 2, 0>(c, prev, NHK, bnext, st.y0, lx, lx);
    else convert_kstep<P, true, 2, 0>(c, cur, NHK, bnext, st.y0, lx, lx);
  }
  __builtin_amdgcn_sched_barrier(0);
  v2_barrier_v();
}
template <class P, int L, int KS>
__device__ __forceinline__ void bwd_steps(const Ctx2& c, V2State& st, const f32x16 (&prev)[4], f32x16 (&cur)[4],
                                          FragT<P::NP>& b0, FragT<P::NP>& b1, f32x4 (&ld)[2][2], f32x4 (&Jq)[12]) {
  if constexpr (KS < bwd_ks(L)) {
    bwd_step<P, L, KS>(c, st, prev, cur, b0, b1, ld, Jq);
    bwd_steps<P, L, KS + 1>(c, st, prev, cur, b1, b0, ld, Jq);
  }
}
template <class P, bool GRAD, int CI>
__device__ __forceinline__ void v2_pad(const Ctx2& c) {
  constexpr int NCH = v2_nch<P>(GRAD), NS = v2_ns<P>(GRAD);
  if constexpr (CI < NCH) {
    v2_barrier_m<v2_wait<P, GRAD, CI>(), v2_window<P, GRAD>()>();
    v2_dma<P, GRAD, (CI + NS - 1) % NCH>(c);
    v2_barrier_v();
    v2_pad<P, GRAD, CI + 1>(c);
  }
}
template <class P, bool GRAD, int CI>
__device__ __forceinline__ void v2_prologue(const Ctx2& c) {
  if constexpr (CI < v2_ns<P>(GRAD) - 1) {
    v2_dma<P, GRAD, CI>(c);
    v2_prologue<P, GRAD, CI + 1>(c);
  }
}

template <class P, bool GRAD>
__global__ __launch_bounds__(WPB2 * 64, V2_OCC) void sdf_mlp_v2_kernel(SdfArgs a) {
  typedef FragT<P::NP> Frag;
  constexpr int NS = v2_ns<P>(GRAD), NCH = v2_nch<P>(GRAD);
  __shared__ __attribute__((aligned(16))) char lds[NS * v2_slot<P>(GRAD)];
  static_assert(NCH % NS == 0, "slot of a chunk = index % ring length");
  Ctx2 c;
  c.lane = threadIdx.x & 63;
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.h = c.lane >> 5;
  c.lane16 = c.lane * 16;
  c.lds = lds;
  c.wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.packed, 0, stream_bytes<P>(), 0x00020000);
  c.tr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.packed + stream_bytes<P>()), 0, TAIL_FLOATS * 4, 0x00020000);
  c.sr = __builtin_amdgcn_make_buffer_rsrc((void*)a.scratch, 0, (GRAD && !(SURF_V2_X & 128)) ? 0x7fffffff : 0, 0x00020000);
  c.sl = c.sr;
  const int64_t wave_id = (int64_t)blockIdx.x * WPB2 + c.wave;
  c.svoff = (int)(wave_id * (SCR_SLOT * 4)) + c.lane * 16;
  const int64_t n_tiles = (a.n + TILE - 1) / TILE;
  const int64_t n_rounds = (n_tiles + WPB2 - 1) / WPB2;

  v2_prologue<P, GRAD, 0>(c);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (c.wave >= WPB2 / 2) v2_barrier_v();   // the second half of the workgroup runs one segment behind the first
#ifdef SURF_V2_TIMING
  c.tprev = __builtin_readcyclecounter();
  for (int k = 0; k < 8; ++k) c.tacc[k] = 0;
#endif
  for (int64_t round = blockIdx.x; round < n_rounds; round += gridDim.x) {
    const int64_t tile = round * WPB2 + c.wave;
    const int64_t slot0 = tile * TILE + (c.lane & 31);
    const int64_t sc = slot0 < a.n ? slot0 : a.n - 1;
    const int64_t i = a.idx ? (int64_t)a.idx[sc] : sc;
    const bool active = (slot0 < a.n) && (!a.mask || a.mask[i] != 0);
    const float px = a.pts[i * 3 + 0], py = a.pts[i * 3 + 1], pz = a.pts[i * 3 + 2];
    const SinCos3 base = sincos3(px, py, pz);
    V2State st;
    st.y0 = 0.f;
    {
      if (SURF_V2_X & 512) {  // (timing / register experiments: no gather)
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) st.phi16[ch] = px * (float)ch;
      } else {
        gather_features<GRAD>(a, c, px, py, pz, st.phi16);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // feature part of the last layer
        const f32x4 w = bload(c.tr, c.h * 64, TAIL_W6P * 4 + g * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * g + q < 14) st.y0 = fmaf(w[q], st.phi16[4 * g + q], st.y0);
      }
      float je_unused[14];
      posenc_half(c.h, px, py, pz, base, st.e16, je_unused, false);
      st.e16[14] = 1.0f;  // bias k-element (weights carry the bias there, lane half 0 only)
      st.phi16[14] = 1.0f;
    }
    Frag b0;   // ONE B fragment: the V segment of a k-step rewrites it after the M segment has issued its MFMAs
    local_frag<P>(st.e16, 0, b0);
    V2_T(0);
    // ------------------------------------------------ forward ----------------------------------------------------
    fwd_steps<P, GRAD, 0, 0>(c, st, st.accB, st.accA, b0, b0);   // 2 k-steps: the next fragment ends in b0
    fwd_steps<P, GRAD, 1, 0>(c, st, st.accA, st.accB, b0, b0);   // 10
    fwd_steps<P, GRAD, 2, 0>(c, st, st.accB, st.accA, b0, b0);   // 10
    fwd_steps<P, GRAD, 3, 0>(c, st, st.accA, st.accB, b0, b0);   // 11: ends in b0
    fwd_steps<P, GRAD, 4, 0>(c, st, st.accB, st.accA, b0, b0);   // 10
    fwd_steps<P, GRAD, 5, 0>(c, st, st.accA, st.accB, b0, b0);   // 10
    V2_T(2);
    if (!GRAD) {
      // lin6 row 0 on the activations of layer 5 (accB)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 w = bload(c.tr, c.h * 256, TAIL_W6H * 4 + (t * 4 + g) * 16);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x2 t2 = {st.accB[t][4 * g + 2 * q], st.accB[t][4 * g + 2 * q + 1]};
            f32x2 hv, sv;
            softplus_pair<false>(t2, 1.0f / Scales<P>::W, hv, sv);
            st.y0 = fmaf(w[2 * q], hv[0], st.y0);
            st.y0 = fmaf(w[2 * q + 1], hv[1], st.y0);
          }
        }
      v2_pad<P, false, V2_NFWD>(c);
    } else {
      // ---------------------------------------------- reverse sweep ----------------------------------------------
#pragma unroll
      for (int r = 0; r < 16; ++r) st.accE[r] = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 w = bload(c.tr, c.h * 64, TAIL_W6P * 4 + g * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) st.accP[4 * g + q] = w[q] * (Scales<P>::W * Scales<P>::D);
      }
      f32x4 ld[2][2];
      load_frag<0>(c, ld[0]);  // delta_5 fragments 0 (converted here: nothing to hide it behind) and 1
      load_frag<1>(c, ld[1]);
      convert_kstep<P, true, 1, 0>(c, st.accB, 0, b0, st.y0, ld[0], ld[0]);
      f32x4 Jq[12];
      bwd_steps<P, 5, 0>(c, st, st.accB, st.accA, b0, b0, ld, Jq);   // 8 k-steps: next fragment ends in b0
      bwd_steps<P, 4, 0>(c, st, st.accA, st.accB, b0, b0, ld, Jq);   // 8
      bwd_steps<P, 3, 0>(c, st, st.accB, st.accA, b0, b0, ld, Jq);   // 8
      bwd_steps<P, 2, 0>(c, st, st.accA, st.accB, b0, b0, ld, Jq);   // 7: ends in b0
      bwd_steps<P, 1, 0>(c, st, st.accB, st.accA, b0, b0, ld, Jq);   // 8
      bwd_steps<P, 0, 0>(c, st, st.accA, st.accB, b0, b0, ld, Jq);   // 8 (accE only)
      v2_pad<P, true, V2_NFWD + V2_NBWD>(c);
      V2_T(4);
#pragma unroll
      for (int g = 0; g < 12; ++g) Jq[g] = bload(c.sl, c.svoff, SCR_S * 4 + g * 1024);  // (the accumulators are dead now)
      float g3[3] = {0.f, 0.f, 0.f};
      {
        float e2[16], je[14];
        posenc_half(c.h, px, py, pz, base, e2, je, true);
#pragma unroll
        for (int s2 = 0; s2 < 14; ++s2) {
          const int c0 = s2 % 3, c1 = (14 + s2) % 3;
          const float v = st.accE[s2] * je[s2];
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) g3[ax] += ((c.h ? c1 : c0) == ax) ? v : 0.f;
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          float Jf[24];
#pragma unroll
          for (int g = 0; g < 6; ++g) {
            const f32x4 v = Jq[6 * sl + g];
            Jf[4 * g + 0] = v[0]; Jf[4 * g + 1] = v[1]; Jf[4 * g + 2] = v[2]; Jf[4 * g + 3] = v[3];
          }
#pragma unroll
          for (int ch = 0; ch < 7; ++ch) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) g3[ax] = fmaf(st.accP[7 * sl + ch], Jf[3 * ch + ax], g3[ax]);
          }
        }
      }
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) g3[ax] = (g3[ax] + __shfl_xor(g3[ax], 32)) * (1.0f / (Scales<P>::W * Scales<P>::D));
      if (active && c.h == 0) {
        a.grad[i * 3 + 0] = g3[0];
        a.grad[i * 3 + 1] = g3[1];
        a.grad[i * 3 + 2] = g3[2];
      }
    }
    float y0 = st.y0 + __shfl_xor(st.y0, 32);
    {
      const f32x4 b6 = bload(c.tr, 0, TAIL_B6 * 4);
      y0 += b6[0];
    }
    if (active && c.h == 0) a.sdf[i] = y0;
    V2_T(5);
  }
  if (c.wave < WPB2 / 2) v2_barrier_v();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last (unused) prefetches must land before the LDS is freed
#ifdef SURF_V2_TIMING
  if (threadIdx.x == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&g_v2_phase[k], c.tacc[k]);
#endif
}

template <class P>
int v2_grid(int64_t n) {
  const int64_t tiles = (n + TILE - 1) / TILE, rounds = (tiles + WPB2 - 1) / WPB2;
  return (int)(rounds < 256 * V2_OCC ? rounds : 256 * V2_OCC);
}

template <class P>
int v2_launch(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n, const float* const* h_vols,
              const int32_t* const* h_tables, const int* h_dims, int n_vol, const void* packed, float* sdf, float* grad,
              void* scratch, void* stream) {
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
    if (s < n_vol && h_dims[s] > 1024) return SURF_E_LIMIT;
  }
  dim3 grid(v2_grid<P>(n)), block(WPB2 * 64);
  if (grad)
    hipLaunchKernelGGL((sdf_mlp_v2_kernel<P, true>), grid, block, 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((sdf_mlp_v2_kernel<P, false>), grid, block, 0, (hipStream_t)stream, a);
  return surf_check_launch();
}

}  // namespace

#ifdef SURF_V2_TIMING
extern "C" int surf_debug_phases(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_v2_phase), sizeof(unsigned long long) * 8) != hipSuccess) return 100;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_v2_phase), z, sizeof(z)) != hipSuccess) return 100;
  }
  return 0;
}
#endif

extern "C" int64_t surf_sdf_bf16_v2_scratch_bytes(int64_t n_points) {
  if (n_points <= 0) return 0;
  return (int64_t)v2_grid<PolBf3>(n_points) * WPB2 * SCR_SLOT * sizeof(float);
}
extern "C" int surf_sdf_mlp_bf16x3_v2(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n,
                                      const float* const* h_vols, const int32_t* const* h_tables, const int* h_dims, int n_vol,
                                      const void* packed, float* sdf, float* grad, void* scratch, void* stream) {
  return v2_launch<PolBf3>(pts, mask, idx, n, h_vols, h_tables, h_dims, n_vol, packed, sdf, grad, scratch, stream);
}
