// K9b: SDF MLP forward + analytic gradient on the 16-bit MFMA pipes with fp32 accumulation ("split" kernels).
//
// Same network, same transposed / register-resident chaining as sdf_mlp.hip (see there for the reference citations),
// but every fp32 operand (weights on the host, activations on the fly) is split into 16-bit pieces and the partial
// products are accumulated in fp32 by v_mfma_f32_32x32x16_{bf16,f16} (K = 16, 32 cycles) instead of
// v_mfma_f32_32x32x2_f32 (K = 2, 64 cycles).  Two precisions:
//
//   bf16x3  exact three-way split a = a1 + a2 + a3 (8 + 8 + 8 significant bits); six products
//           a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1, dropped terms <= 2^-23 |a b|: fp32-equivalent.      2.67x fewer
//           matrix-pipe cycles than fp32.  One wavefront per SIMD (the three-piece activations need the register file).
//   f16x2   two-way split a ~= a1 + a2 (11 + 11 significant bits; weights and deltas pre-scaled by 2^8 so that the second
//           piece stays normal); three products a1b1 + a1b2 + a2b1; operand error <= 2^-22 (or 2^-25 absolute).
//           5.3x fewer matrix-pipe cycles than fp32.  |activations| must stay below 65504 and |weights| below 255.
//
// The weight stream is consumed several times faster than one wavefront could pull it from L2, so the four wavefronts of
// a workgroup (one per SIMD) run in lockstep over an LDS image of it: one chunk = all k-steps x pieces of one 32-row
// output tile, a ring of three slots filled by LDS-DMA (buffer_load ... lds) two chunks ahead, cyclically across rounds
// and retired by counted s_waitcnt vmcnt(N) (no scratch-memory spills allowed: see build.sh).
// Softplus, operand splitting and scratch stores of output tile t are issued between the MFMAs of tile t+1: cut into slots
// of <= ~24 issue cycles and dealt out over the MFMA gaps by a compile-time plan (sdf_split_common.h, "conversion slots").
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

#include "sdf_split_common.h"

namespace {

#ifdef SURF_SDF_TIMING  // debug builds only: per-phase shader-clock totals of wavefront 0 of every workgroup
__device__ unsigned long long g_phase[8];
#define SURF_T(k)                                                 \
  do {                                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    c.tacc[k] += now_ - c.tprev;                                  \
    c.tprev = now_;                                               \
  } while (0)
#else
#define SURF_T(k)
#endif

struct Ctx {
#ifdef SURF_SDF_TIMING
  mutable unsigned long long tprev;
  mutable unsigned long long tacc[8];
#endif
  rsrc_t wr, sr, sl, tr;  // packed stream, scratch (stores / loads), fp32 tail
  int lane, lane16, h, svoff, wave;
  int wave1024;       // wave * 1024 (scalar)
  int dma_voff;       // lane * 16 + wave * 1024: this wavefront's 1 KB block of every 4 KB DMA piece
  uint32_t dma_lds;   // LDS address of the ring (+ wave * 1024 in the piece-major assignment; made opaque once per round)
  char* lds;    // the ring
  char* lds_s;  // this wavefront's exponent slices (the softplus arguments the reverse sweep needs) + lane * 16
  u32x4 I0, I1;  // (P::MRES) A operands of the matrix-pipe residuals, k-steps 0 / 1 of a tile: -1 where lane row = the tile row of k-slot
};
// A[i][8 g + ip] of k-step s (lane: row i = lane & 31, k group g = lane >> 5) = -1 iff row i is the accumulator row of register
// 8 s + ip in lane half g: (ip & 3) + 8 (2 s + (ip >> 2)) + 4 g - the row <-> k-slot permutation every packed fragment uses.
__device__ __forceinline__ u32x4 minus_identity_frag(int lane, int s) {
  const int i = lane & 31, g = lane >> 5;
  u32x4 f = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int ip = 0; ip < 8; ++ip) {
    const int rho = (ip & 3) + 8 * (2 * s + (ip >> 2)) + 4 * g;
    if (rho == i) f[ip >> 1] |= (ip & 1) ? 0xBF800000u : 0x0000BF80u;  // bf16(-1.0)
  }
  return f;
}
// one residual MFMA: vt -= (the fragment's piece, as the B operand of its own k-step)
template <class P>
__device__ __forceinline__ void mres_step(const Ctx& c, f32x16& vt, const u32x4& piece, int s2) {
  if constexpr (P::MRES) {
    __builtin_amdgcn_sched_barrier(0);  // what precedes it in the gap runs in the product MFMA's shadow, what follows in its own
    const u32x4 I = s2 ? c.I1 : c.I0;
    vt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, I), __builtin_bit_cast(bf16x8, piece), vt, 0, 0, 0);
    asm volatile("" ::"v"(vt));  // tie the MFMA to this place (see run_chunk)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- staging ------------------------------------------------------------------------------------------------------------
// The chunk stream is cyclic over rounds and lives in a ring of NS = P::nslot LDS slots (chunk CI in slot CI % NS; the
// stream lengths are multiples of NS).  While chunk CI is consumed, chunks CI+2 .. CI+NS-1 are in flight by LDS-DMA
// (buffer_load ... lds, 1 KB per instruction, lane-linear image; chunk CI+NS-1 is issued during chunk CI) and chunk CI+1
// is retired at the end of chunk CI by a counted vmcnt.  vmcnt retires in issue order, so everything older than that DMA
// has to be complete too: NS-1 chunks of slack keep the exponent-slice stores of the chunks before (acknowledged late by L2)
// out of that wait.
template <class P> constexpr int n_dma(int ci) { return (P::CH.ks[ci] * P::NP + 3) / 4; }
#ifndef SURF_SDF_DMA_WAVEMAJOR
#define SURF_SDF_DMA_WAVEMAJOR 1
#endif
template <class P, int NS, int CI, int k>
__device__ __forceinline__ void stage_dma_piece(const Ctx& c) {
  constexpr int OFF = P::CH.off[CI];
  // blocks past the end of a chunk read into the next one / out of range (= 0) and land in the unused tail of the slot.
#if SURF_SDF_DMA_WAVEMAJOR
  // Round 4: wave-major assignment - of the 4 ND one-KB blocks of chunk CI (ND = n_dma pieces a wavefront issues) wavefront w
  // moves blocks w ND .. w ND + ND - 1 instead of 4 k + w: consecutive pieces of a wavefront are then 1 KB apart on BOTH sides
  // (stream and LDS image are unchanged: block b still lands at slot + 1024 b), so four of them share one M0 / one scalar
  // offset and differ in the instruction's 12-bit immediate (0, 1024, 2048, 3072).  Two SALU instructions per four DMAs
  // instead of eight (an SALU instruction costs ~7 issue cycles between MFMAs, microbench/mfma_issue_model.hip); the wave's
  // share w ND 1024 is a scalar product with a compile-time ND (a handful of distinct values per kernel).
  constexpr int ND = n_dma<P>(CI);
  const uint32_t wshare = (uint32_t)c.wave1024 * (uint32_t)ND;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(
      c.wr, (__attribute__((address_space(3))) void*)(uintptr_t)(c.dma_lds + wshare + ((CI % NS) * slot_bytes<P>() + (k >> 2) * 4096)), 16,
      c.lane16, (int)wshare + (OFF + (k >> 2) * 4096), (k & 3) * 1024, 0);
#else
  // The wave's share of the address sits in the offset VGPR / the per-round LDS base, so that the scalar offset is a
  // compile-time constant: with `wave` inside it the ~330 distinct offsets of a round were loop-invariant SGPR values that
  // the compiler hoisted out of the round loop and spilled to VGPR lanes (v_readlane + 5 wait states in front of every DMA).
  __builtin_amdgcn_raw_ptr_buffer_load_lds(
      c.wr, (__attribute__((address_space(3))) void*)(uintptr_t)(c.dma_lds + ((CI % NS) * slot_bytes<P>() + k * 4096)), 16, c.dma_voff,
      OFF + k * 4096, 0, 0);
#endif
}
template <class P, int NS, int CI, int K = 0>
__device__ __forceinline__ void stage_dma(const Ctx& c) {
  if constexpr (K < n_dma<P>(CI)) {
    stage_dma_piece<P, NS, CI, K>(c);
    stage_dma<P, NS, CI, K + 1>(c);
  }
}
// vector-memory operations a chunk issues by itself, in order: [pre: loads before its DMA] [DMA] [post: stores in fn]
// which exponent slice the backward chunk (L, T) loads (layer < 0: none): DEEP: that of the hidden tile computed by the
// NEXT chunk; otherwise that of its own tile
constexpr int sprime_layer(bool deep, int L, int T, bool r0 = false) {
  if (!deep) return (L >= 1 && T < 4) ? L - 1 : -1;
  if (L >= 1 && T < 3) return L - 1;                // (L, T + 1)
  if (r0 && L == 2 && T == BWD_NT[L] - 1) return -1;  // (R0: slice (0, 0) exists only after the recompute chunk that follows)
  if (L >= 2 && T == BWD_NT[L] - 1) return L - 2;   // last tile of the layer -> (L - 1, 0)
  return -1;
}
constexpr int sprime_tile(bool deep, int L, int T) { return !deep ? T : ((L >= 1 && T < 3) ? T + 1 : 0); }
// exponent slices kept in the workgroup's spare LDS instead of the scratch buffer (P::LDS_SLICES per wavefront, 4 KB each:
// layer 4 tiles first, then layer 3): the CU executes ~90 clocks per vector-memory wave-instruction in this kernel
// (SQ_INSTS_VMEM / time) and the slices are a quarter of them; an LDS slice costs eight ds_* instead.
template <class P> constexpr int lds_slice(int layer, int tile) {  // index in the wavefront's LDS area or -1
  const int k = (4 - layer) * 4 + tile;
  return (layer <= 4 && layer >= 0 && k < P::LDS_SLICES) ? k : -1;
}
// ... and the next P::REG_SLICES of them in registers (RegSlices below): no memory operation at all
template <class P> constexpr int reg_slice(int layer, int tile) {
  const int k = (4 - layer) * 4 + tile - P::LDS_SLICES;
  return (layer <= 4 && layer >= 0 && k >= 0 && k < P::REG_SLICES) ? k : -1;
}
// (R0: layer 0's slices are neither kept nor stored: the reverse sweep recomputes them into RegSlices::r0)
template <class P> constexpr bool on_chip(int layer, int tile) {
  return lds_slice<P>(layer, tile) >= 0 || reg_slice<P>(layer, tile) >= 0 || (P::R0 && layer == 0);
}
template <class P> struct RegSlices {
  f32x4 v[P::REG_SLICES > 0 ? P::REG_SLICES : 1][4];
  f32x16 r0[P::R0 ? 4 : 1];  // the recomputed accumulators of layer 0 (their registers are those of slices that are dead by then)
};
// 16-byte groups of an exponent slice that are read back: layer 2 has 101 rows, so the second half (k-step 7: rows
// 112..127) of its tile 3 feeds nothing.  Loading it anyway would leave the loads to dead-code elimination, i.e. leave
// the number of vector-memory operations of that chunk - which stage_barrier's vmcnt counts - to the optimiser.
constexpr int sprime_groups(int layer, int tile) { return (layer == 2 && tile == 3) ? 2 : 4; }
template <class P> constexpr int vm_pre(int ci) {
  if (ci < N_FWD_CHUNKS) return (ci / 4 == 5 && ci % 4 > 0) ? 4 : 0;  // W6 slices
  if (ci >= P::CH.n) return 0;                                         // padding chunk
  if (P::R0 && ci == r0_chunk()) return 0;                             // recompute chunk: MFMAs only
  int l = 5, t = ci - N_FWD_CHUNKS - ((P::R0 && ci > r0_chunk()) ? 1 : 0);
  while (t >= BWD_NT[l]) { t -= BWD_NT[l]; --l; }
  if (l == 0) return P::DEEPJ ? 12 : 0;  // feature Jacobian for the epilogue
  if (sprime_layer(P::DEEP, l, t, P::R0) < 0 || on_chip<P>(sprime_layer(P::DEEP, l, t, P::R0), sprime_tile(P::DEEP, l, t))) return 0;
  return sprime_groups(sprime_layer(P::DEEP, l, t, P::R0), sprime_tile(P::DEEP, l, t));
}
template <class P, bool GRAD> constexpr int vm_post(int ci) {
  if (!GRAD || ci >= N_FWD_CHUNKS) return 0;
  const int l = ci / 4, t = ci % 4;
  if ((l == 0 && t == 0) || (l == 5 && t > 0)) return 0;
  // the slice stored under chunk (l, t): tile 3 of the layer below for t = 0, else (l, t - 1)
  return on_chip<P>(t == 0 ? l - 1 : l, t == 0 ? 3 : t - 1) ? 0 : 4;
}
// Retire the DMA of chunk CI+1, issued during chunk CI-(NS-2): everything this wave issued in the chunks after that one
// may stay in flight (operations between rounds are not counted, which only makes the wait stricter; that DMA's pieces are
// spread over the k-steps of their chunk, so that chunk's own stores are not counted either).  Then the LDS-only
// workgroup barrier.  The `surf_ring_window` comment tells check_isa.py how many barrier intervals the count spans.
template <class P, bool GRAD, int CI, int NCH>
constexpr int barrier_vmcnt() {
  constexpr int DIST = P::nslot(GRAD) - 1;
  int n = 0;
  for (int j = 0; j + 1 < DIST; ++j) {
    const int cj = (CI - j + NCH) % NCH;
    n += vm_pre<P>(cj) + n_dma<P>((cj + DIST) % NCH) + vm_post<P, GRAD>(cj);
  }
  return n;
}
template <class P, bool GRAD, int CI, int NCH>
__device__ __forceinline__ void stage_barrier() {
  constexpr int N = barrier_vmcnt<P, GRAD, CI, NCH>();
  static_assert(N >= 0 && N < 64, "vmcnt");
  asm volatile("; surf_ring_window %1\n\ts_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N), "n"(P::nslot(GRAD) - 2) : "memory");
}

// One chunk: NKS k-steps of P::NM MFMAs read from its LDS slot; B fragments come from bsel(ks).  After MFMA m of k-step ks
// (gap g = ks * NM + m) come, pinned by a scheduling barrier: the LDS read of piece m of the next k-step's A fragment, the
// conversion slots the plan PL puts there (slot(s), in order) and the DMA pieces it puts there.
// NACC > 1 (the R0 chunk): k-steps [a NKS / NACC, (a + 1) NKS / NACC) accumulate into tile a.
template <class P, bool GRAD, int CI, int NCH, class PL, int NACC, class BSel, class F>
__device__ __forceinline__ void run_chunk_n(const Ctx& c, BSel bsel, F slot, f32x16 (&result)[NACC]) {
  constexpr int NKS = P::CH.ks[CI];
  constexpr int NP = P::NP, NM = P::NM;
  constexpr int NS = P::nslot(GRAD);
  constexpr int KPA = NKS > 0 ? NKS / NACC : 1;
  static_assert(NKS % NACC == 0, "k-steps per accumulator");
  const char* rd = c.lds + (CI % NS) * slot_bytes<P>() + c.lane16;
  typename P::Acc accs[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int q = 0; q < P::NA; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) accs[a].v[q][r] = 0.f;
  constexpr int PF = P::PF;  // A fragments are read PF k-steps ahead of their MFMAs
  static_assert(PF == 1, "the gap plan places the reads of k-step ks + 1 behind the MFMAs of k-step ks");
  u32x4 a_q[PF + 1][NP];
#pragma unroll
  for (int d = 0; d < PF; ++d)
#pragma unroll
    for (int p = 0; p < NP; ++p)
      if (d < NKS) a_q[d][p] = *reinterpret_cast<const u32x4*>(rd + (d * NP + p) * 1024);
  constexpr int NXT = (CI + NS - 1) % NCH, ND = n_dma<P>(NXT);
  static_assert(ND <= MAX_DMA && NKS * NM <= MAX_GAPS, "GapPlan sizes");
  if (NKS == 0 && !SURF_X_NODMA) stage_dma<P, NS, NXT>(c);  // padding chunk: nothing to hide the DMA issue behind
  __builtin_amdgcn_sched_barrier(0);
  static_for<0, NKS>([&](auto ksc) __attribute__((always_inline)) {
    constexpr int ks = decltype(ksc)::value;
    const FragT<NP>& b = bsel(ks);
    static_for<0, NM>([&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value, g = ks * NM + m;
      typename P::Acc& acc = accs[ks / KPA];
      if (!SURF_X_NOMMA) {
        P::mma_one(acc, a_q[SURF_X_NOLDS ? 0 : ks % (PF + 1)], b, m);
        // An MFMA is a pure value to the compiler: instruction selection is free to linearise it anywhere between its
        // operands and its use (it sank whole k-steps below their gaps' scheduling barriers, with the A pieces spilled to
        // scratch memory meanwhile).  Passing the accumulator through an empty volatile statement ties it to this gap.
        // (round 4: the statement only READS the accumulator.  As an in-out operand it made the hazard recogniser treat the
        // statement as a VALU write of the MFMA's srcC and put an s_nop in front of the next MFMA - 616 of the gradient kernel's
        // ~1,000 s_nop, 3.4 issue cycles each; as an input it still keeps the MFMA from sinking below this gap, and source-order
        // selection keeps the next one from rising above it: same interleave, no spills, -0.7 K s_nop.  SURF_SDF_PINS & 4: the
        // in-out form)
        if (SURF_SDF_PINS & 4) asm volatile("" : "+v"(acc.v[m & (P::NA - 1)]));
        else if (SURF_SDF_PINS & 2) asm volatile("" ::"v"(acc.v[m & (P::NA - 1)]));
      } else if (m == 0) {
        acc.v[0][ks % 16] += __builtin_bit_cast(float, b.p[0][0]) + __builtin_bit_cast(float, a_q[ks % (PF + 1)][0][0]);
      }
      if constexpr (m < NP && ks + PF < NKS)
        if (!SURF_X_NOLDS) a_q[(ks + PF) % (PF + 1)][m] = *reinterpret_cast<const u32x4*>(rd + ((ks + PF) * NP + m) * 1024);
      static_for<PL::v.first[g], PL::v.first[g + 1]>(slot);
      static_for<0, ND>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (PL::v.dma_gap[k] == g)
          if (!SURF_X_NODMA) stage_dma_piece<P, NS, NXT, k>(c);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  SURF_T(CI < N_FWD_CHUNKS ? 1 : 3);
  stage_barrier<P, GRAD, CI, NCH>();
  SURF_T(7);
#pragma unroll
  for (int a = 0; a < NACC; ++a) result[a] = P::finish(accs[a]);
}
template <class P, bool GRAD, int CI, int NCH, class PL, class BSel, class F>
__device__ __forceinline__ f32x16 run_chunk(const Ctx& c, BSel bsel, F slot) {
  f32x16 r[1];
  run_chunk_n<P, GRAD, CI, NCH, PL, 1>(c, bsel, slot, r);
  return r[0];
}
template <class P, bool GRAD, int CI, int NCH> constexpr int chunk_dma() { return n_dma<P>((CI + P::nslot(GRAD) - 1) % NCH); }


// ---- forward tile (layer L, tile T) -----------------------------------------------------------------------------------
// hin/hout: fragments of the 128 hidden activations: index 2*tile + s.  `raw` = pre-activations of the tile finished
// before this one; they are converted under this tile's MFMAs as two woven streams of mini-phases (sdf_split_common.h,
// "conversion slots").  Per pair of elements:
//   u | e = 2^-|u| | 1 + e | log2 | max(t, 0) | h | pack, value, subtract (x NP - 1) | pack: fragment dwords and, every second
//   pair, the 16-byte store of four exponent arguments u for the reverse sweep
// and for the tiles of layer 5 (LAST: h feeds lin6's row 0, h' w6 is the reverse sweep's first delta)
//   u | e | 1 + e | log2 | max | h | y0 += w6 h ( | 1 / (1 + e) | t >= 0 ? 1 : e | h' w6 | split as above ).
template <class P, bool GRAD, bool LAST>
constexpr MiniProg fwd_prog() {
  constexpr int scale = Scales<P>::W != 1.0f ? 8 : 0;
  MiniProg mp{};
  if (!P::PRESCALED) mini_add(mp, K_ARG, 0, 8);
  mini_add(mp, K_EXP, 0, 16);
  mini_add(mp, K_ADD1, 0, 8);
  mini_add(mp, K_LOG, 0, 16);
  mini_add(mp, K_MAX, 0, 8 + scale);
  mini_add(mp, K_FMA, 0, 8);
  if (LAST) {
    mini_add(mp, K_Y0, 0, 8);
    if (GRAD) {
      mini_add(mp, K_RCP, 0, 16);
      mini_add(mp, K_SEL, 0, 16);
      mini_add(mp, K_SIG, 0, 16);
      mini_add_split(mp, P::NP, P::FUSED_SUB, P::MRES);
    }
  } else {
    mini_add_split(mp, P::NP, P::FUSED_SUB, P::MRES);
  }
  return mp;
}
template <class P, bool GRAD, int L, int T>
struct FwdPlan {
  static constexpr int CI = fwd_chunk(L, T), NKS = P::CH.ks[CI], NG = NKS * P::NM;
  static constexpr bool LAST = (L == 5 && T > 0), CONV = !(L == 0 && T == 0);
  static constexpr MiniProg mini = fwd_prog<P, GRAD, LAST>();
  static constexpr bool SPLITS = !LAST || GRAD;  // (forward-only layer 5: h feeds lin6's row 0 and nothing is split)
  static constexpr SlotProg prog = CONV ? weave(mini, 8, (GRAD && !LAST) ? 8 : 0, (P::MRES && SPLITS) ? P::NP : 0) : SlotProg{};
  // T == 0 (L >= 1) converts tile 3 of the layer below into hin[6] (pairs 0..3) and hin[7], which this very chunk multiplies
  // in its last two k-steps (layer 3, 101 inputs: hin[6] in the last one, hin[7] not at all)
  static constexpr int DEADLINE = (T == 0 && L >= 1) ? (NKS - 1) * P::NM : NG;
  static constexpr GapPlan v = plan_gaps(NKS, P::NM, P::NP, chunk_dma<P, GRAD, CI, n_chunks<P>(GRAD)>(), DEADLINE, prog);
  static_assert(!(T == 0 && L >= 1) || pair_done_gap(v, prog, 3, NG) < (fwd_nl(L) + 6) * P::NM, "hin[6] is multiplied before it is complete");
  static_assert(plan_ok(v, prog, mini, NG, DEADLINE), "gap plan");
};
template <class P, bool GRAD, int L, int T>
__device__ __forceinline__ void fwd_tile(const Ctx& c, f32x16& raw, FragT<P::NP>* hin, FragT<P::NP>* hout,
                                         const FragT<P::NP> (&ef)[2], const FragT<P::NP> (&pf)[2], FragT<P::NP>* dfr,
                                         float& y0, RegSlices<P>& keep) {
  typedef FragT<P::NP> Frag;
  typedef FwdPlan<P, GRAD, L, T> PL;
  constexpr int CI = fwd_chunk(L, T);
  constexpr int NEk = fwd_ne(L), NL = fwd_nl(L);
  constexpr bool LAST = PL::LAST;  // previous tile belongs to layer 5: feeds lin6 row 0 directly
  constexpr bool CONV = PL::CONV;  // there is a previous tile to convert
  constexpr int STORES = (GRAD && CONV && !LAST) ? 4 : 0;
  constexpr float inv_w = 1.0f / Scales<P>::W;
  // destination of the converted tile: (L, T - 1) in hout, or for T == 0 tile 3 of the layer below in hin
  constexpr int dst_tile = T == 0 ? 3 : T - 1, s_layer = T == 0 ? L - 1 : L;
  Frag* const dst = LAST ? dfr : ((T == 0 && L > 0) ? hin : hout);
  const f32x16 prev = raw;
  f32x4 w6[4];
  if (LAST) {
#pragma unroll
    for (int g = 0; g < 4; ++g) w6[g] = bload(c.tr, c.h * 256, (TAIL_W6H * 4) + ((T - 1) * 4 + g) * 16);
  }
  f32x4 sbuf[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // exponent arguments of pairs (4k, 4k + 1) / (4k + 2, 4k + 3)
  MiniState<P::NP> st[2] = {};
  // (P::MRES) the tile the matrix pipe forms the residuals of: starts as the pre-activations, a pair's activations replace them in
  // its K_PACK slot (its last use of them is before that), so that the tile costs no registers beside the accumulators it was
  f32x16 vt = prev;
  auto slot = [&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    constexpr int q = PL::prog.q[s] < 0 ? 0 : PL::prog.q[s], kind = PL::prog.kind[s], i = PL::prog.arg[s], el = 2 * q;
    MiniState<P::NP>& S = st[q & 1];
    const f32x2 t2 = P::MRES ? f32x2{vt[el], vt[el + 1]} : f32x2{prev[el], prev[el + 1]};
    const f32x2 w2 = {LAST ? w6[el >> 2][el & 3] : 0.f, LAST ? w6[(el + 1) >> 2][(el + 1) & 3] : 0.f};
    if constexpr (kind == K_ARG) {
      S.u[0] = t2[0] * (144.269504088896341f * inv_w);  // 100 log2(e) t   (element by element: no packed fp32 beside MFMAs)
      S.u[1] = t2[1] * (144.269504088896341f * inv_w);
      slot_pin(S.u);
    } else if constexpr (kind == K_EXP) {
      if (P::PRESCALED) S.u = t2;  // the accumulators ARE the exponent arguments
      slot_pin(S.u);
      S.e[0] = SURF_X_NOSOFTPLUS ? S.u[0] : __builtin_amdgcn_exp2f(-__builtin_fabsf(S.u[0]));
      S.e[1] = SURF_X_NOSOFTPLUS ? S.u[1] : __builtin_amdgcn_exp2f(-__builtin_fabsf(S.u[1]));
      slot_pin(S.e);
    } else if constexpr (kind == K_ADD1) {
      slot_pin(S.e);
      S.d[0] = S.e[0] + 1.0f;
      S.d[1] = S.e[1] + 1.0f;
      slot_pin(S.d);
    } else if constexpr (kind == K_LOG) {
      slot_pin(S.d);
      S.l[0] = SURF_X_NOSOFTPLUS ? S.d[0] : __builtin_amdgcn_logf(S.d[0]);
      S.l[1] = SURF_X_NOSOFTPLUS ? S.d[1] : __builtin_amdgcn_logf(S.d[1]);
      slot_pin(S.l);
    } else if constexpr (kind == K_MAX) {
      S.m[0] = __builtin_amdgcn_fmed3f(t2[0], 0.0f, 3.0e38f);  // max(t, 0) without the canonicalising extra v_max
      S.m[1] = __builtin_amdgcn_fmed3f(t2[1], 0.0f, 3.0e38f);
      if (inv_w != 1.0f) {
        S.m[0] *= inv_w;
        S.m[1] *= inv_w;
      }
      slot_pin(S.m);
    } else if constexpr (kind == K_FMA) {  // h = max(t, 0) + ln(1 + e) / 100
      slot_pin(S.l, S.m);
      if (P::PRESCALED) {  // z = max(u, 0) + log2(1 + 2^-|u|)
        S.val[0] = S.m[0] + S.l[0];
        S.val[1] = S.m[1] + S.l[1];
      } else {
        S.val[0] = fmaf(S.l[0], 0.69314718055994531f * 0.01f, S.m[0]);
        S.val[1] = fmaf(S.l[1], 0.69314718055994531f * 0.01f, S.m[1]);
      }
      slot_pin(S.val);
      if (!LAST) {
        sbuf[q >> 1 & 1][el & 3] = S.u[0];
        sbuf[q >> 1 & 1][(el & 3) + 1] = S.u[1];
      }
    } else if constexpr (kind == K_Y0) {
      slot_pin(S.val);
      y0 = fmaf(w2[0], S.val[0], y0);
      y0 = fmaf(w2[1], S.val[1], y0);
      slot_pin(y0);
    } else if constexpr (kind == K_RCP) {
      slot_pin(S.d);
      S.r[0] = __builtin_amdgcn_rcpf(S.d[0]);
      S.r[1] = __builtin_amdgcn_rcpf(S.d[1]);
      slot_pin(S.r);
    } else if constexpr (kind == K_SEL) {  // h' = (t >= 0 ? 1 : e) / (1 + e)
      slot_pin(S.e);
      S.sel[0] = t2[0] >= 0.0f ? 1.0f : S.e[0];
      S.sel[1] = t2[1] >= 0.0f ? 1.0f : S.e[1];
      slot_pin(S.sel);
    } else if constexpr (kind == K_SIG) {
      slot_pin(S.r, S.sel);
      S.val[0] = (S.sel[0] * S.r[0]) * (w2[0] * Scales<P>::D);
      S.val[1] = (S.sel[1] * S.r[1]) * (w2[1] * Scales<P>::D);
      slot_pin(S.val);
    } else if constexpr (kind == K_PACK) {
      slot_pin(S.val);
      S.pc[i] = SURF_X_NOSPLIT ? __builtin_bit_cast(uint32_t, S.val[i & 1]) : P::pack(S.val);
      slot_pin(S.pc[i]);
      if (P::MRES) {  // piece 0 into its fragment, the value into the tile whose residuals the matrix pipe forms
        dst[2 * dst_tile + (el >> 3)].p[0][(el & 7) >> 1] = S.pc[0];
        vt[el] = S.val[0];
        vt[el + 1] = S.val[1];
      }
      if (i == P::NP - 1 || P::MRES) {
        if (!P::MRES) {
          Frag& f = dst[2 * dst_tile + (el >> 3)];
#pragma unroll
          for (int k = 0; k < P::NP; ++k) f.p[k][(el & 7) >> 1] = S.pc[k];
        }
        if (GRAD && !LAST && (el & 3) == 2) {
          constexpr int ls = lds_slice<P>(s_layer, dst_tile), rs = reg_slice<P>(s_layer, dst_tile);
          if constexpr (P::R0 && s_layer == 0) {}  // recomputed by the reverse sweep
          else if constexpr (rs >= 0) keep.v[rs][el >> 2] = sbuf[q >> 1 & 1];
          else if constexpr (ls >= 0) *reinterpret_cast<f32x4*>(c.lds_s + ls * 4096 + (el >> 2) * 1024) = sbuf[q >> 1 & 1];
          else bstore(c.sr, c.svoff, s_layer * 16384 + (dst_tile * 4 + (el >> 2)) * 1024, sbuf[q >> 1 & 1]);
        }
      }
    } else if constexpr (kind == K_MRES) {
      mres_step<P>(c, vt, dst[2 * dst_tile + (i & 1)].p[i >> 1], i & 1);
    } else if constexpr (kind == K_PACKT) {
      const f32x2 r2 = {vt[el], vt[el + 1]};
      dst[2 * dst_tile + (el >> 3)].p[i][(el & 7) >> 1] = P::pack(r2);
    } else if constexpr (kind == K_EXPAND) {
      slot_pin(S.pc[i]);
      S.x = SURF_X_NOSPLIT ? S.val : P::expand(S.pc[i]);
      slot_pin(S.x);
    } else if constexpr (kind == K_SUB) {
      slot_pin(S.val, S.x);
      S.val[0] -= S.x[0];
      S.val[1] -= S.x[1];
      slot_pin(S.val);
    } else if constexpr (kind == K_SUBP) {
      S.val = SURF_X_NOSPLIT ? S.val : P::sub_piece(S.val, S.pc[i]);
    }
  };
  auto bsel = [&](int ks) __attribute__((always_inline)) -> const Frag& {
    if (ks < NEk) return ef[ks];
    if (ks < NL) return pf[ks - NEk];
    return hin[ks - NL];
  };
  static_assert((on_chip<P>(T == 0 ? L - 1 : L, T == 0 ? 3 : T - 1) ? 0 : STORES) == vm_post<P, GRAD>(CI) && (LAST ? 4 : 0) == vm_pre<P>(CI), "vmcnt bookkeeping");
  raw = run_chunk<P, GRAD, CI, n_chunks<P>(GRAD), PL>(c, bsel, slot);
}

template <class P, bool GRAD, int L>
__device__ __forceinline__ void fwd_layer(const Ctx& c, f32x16& raw, FragT<P::NP>* hin, FragT<P::NP>* hout,
                                          const FragT<P::NP> (&ef)[2], const FragT<P::NP> (&pf)[2], FragT<P::NP>* dfr,
                                          float& y0, RegSlices<P>& keep) {
  fwd_tile<P, GRAD, L, 0>(c, raw, hin, hout, ef, pf, dfr, y0, keep);
  fwd_tile<P, GRAD, L, 1>(c, raw, hin, hout, ef, pf, dfr, y0, keep);
  fwd_tile<P, GRAD, L, 2>(c, raw, hin, hout, ef, pf, dfr, y0, keep);
  fwd_tile<P, GRAD, L, 3>(c, raw, hin, hout, ef, pf, dfr, y0, keep);
}

// ---- backward tiles ----------------------------------------------------------------------------------------------------
// G (= W^T delta) of hidden tile T is multiplied by softplus' (formed from the tile's exponent slice) and split under the
// MFMAs of the tile that follows it.  The exponent slice of a hidden tile is loaded from scratch during the chunk BEFORE the one that computes its G, two
// chunks before it is used (an HBM round trip is longer than one chunk).
struct BwdPend {
  f32x16 G;     // finished tile waiting for its conversion
  f32x4 s[4];   // its exponent slice
  f32x4 sn[4];  // slice of the tile whose G is being computed now
};
template <class P, int NG = 4>
__device__ __forceinline__ void load_sprime(const Ctx& c, int layer, int tile, f32x4 (&dst)[4], const RegSlices<P>& keep) {
  const int ls = lds_slice<P>(layer, tile), rs = reg_slice<P>(layer, tile);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (P::R0 && layer == 0) {
      const f32x16& t = keep.r0[P::R0 ? tile : 0];
      dst[g] = f32x4{t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
    } else if (rs >= 0) dst[g] = keep.v[rs][g];
    else if (ls >= 0) dst[g] = *reinterpret_cast<const f32x4*>(c.lds_s + ls * 4096 + g * 1024);
    else dst[g] = bload_scratch(c.sl, c.svoff, layer * 16384 + (tile * 4 + g) * 1024);
  }
}
// mini-phases of the reverse sweep, per pair of elements of the finished G tile:
//   2^-u | 1 + 2^-u | 1 / that (= h') | delta = h' G | pack, value, subtract (x NP - 1) | pack: fragment dwords
template <class P>
constexpr MiniProg bwd_prog() {
  MiniProg mp{};
  mini_add(mp, K_EXPN, 0, 16);
  mini_add(mp, K_ADD1, 0, 8);
  mini_add(mp, K_RCP, 0, 16);
  mini_add(mp, K_MULG, 0, 8 + (Scales<P>::W != 1.0f ? 8 : 0));
  mini_add_split(mp, P::NP, P::FUSED_SUB, P::MRES);
  return mp;
}
template <class P, int L, int T, bool CONVERT>
struct BwdPlan {
  static constexpr int CI = bwd_chunk(L, T, P::R0), NKS = P::CH.ks[CI];
  static constexpr MiniProg mini = bwd_prog<P>();
  static constexpr SlotProg prog = CONVERT ? weave(mini, 8, 0, P::MRES ? P::NP : 0) : SlotProg{};
  static constexpr GapPlan v = plan_gaps(NKS, P::NM, P::NP, chunk_dma<P, true, CI, n_chunks<P>(true)>(), NKS * P::NM, prog);
  static_assert(plan_ok(v, prog, mini, NKS * P::NM, NKS * P::NM), "gap plan");
};
template <class P, int L, int T, bool CONVERT>
__device__ __forceinline__ f32x16 bwd_tile(const Ctx& c, const FragT<P::NP>* din, FragT<P::NP>* dout, const BwdPend& prev,
                                           f32x4 (&s_load)[4], const RegSlices<P>& keep) {
  typedef FragT<P::NP> Frag;
  typedef BwdPlan<P, L, T, CONVERT> PL;
  constexpr int CI = bwd_chunk(L, T, P::R0);
  constexpr int SL = sprime_layer(P::DEEP, L, T, P::R0);
  constexpr int NG = SL >= 0 ? sprime_groups(SL, sprime_tile(P::DEEP, L, T)) : 0;
  static_assert((SL >= 0 && on_chip<P>(SL, sprime_tile(P::DEEP, L, T)) ? 0 : NG) + ((L == 0 && P::DEEPJ) ? 12 : 0) == vm_pre<P>(CI),
                "vmcnt bookkeeping");
  if (SL >= 0) load_sprime<P, NG>(c, SL, sprime_tile(P::DEEP, L, T), s_load, keep);
  MiniState<P::NP> st[2] = {};
  f32x16 vt = prev.G;  // (P::MRES) G, pair by pair replaced by the deltas (K_PACK), then their residuals
  auto slot = [&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    constexpr float inv_w = 1.0f / Scales<P>::W;  // G arrives x W_SCALE x D_SCALE, deltas are kept x D_SCALE
    constexpr int q = PL::prog.q[s] < 0 ? 0 : PL::prog.q[s], kind = PL::prog.kind[s], i = PL::prog.arg[s], el = 2 * q;
    MiniState<P::NP>& S = st[q & 1];
    if constexpr (kind == K_EXPN) {
      const f32x2 u = {prev.s[el >> 2][el & 3], prev.s[(el + 1) >> 2][(el + 1) & 3]};  // exponent arguments (forward K_ARG)
      S.e[0] = SURF_X_NOSOFTPLUS ? u[0] : __builtin_amdgcn_exp2f(-u[0]);
      S.e[1] = SURF_X_NOSOFTPLUS ? u[1] : __builtin_amdgcn_exp2f(-u[1]);
      slot_pin(S.e);
    } else if constexpr (kind == K_ADD1) {
      slot_pin(S.e);
      S.d[0] = S.e[0] + 1.0f;
      S.d[1] = S.e[1] + 1.0f;
      slot_pin(S.d);
    } else if constexpr (kind == K_RCP) {  // h' = 1 / (1 + 2^-u)
      slot_pin(S.d);
      S.r[0] = SURF_X_NOSOFTPLUS ? S.d[0] : __builtin_amdgcn_rcpf(S.d[0]);
      S.r[1] = SURF_X_NOSOFTPLUS ? S.d[1] : __builtin_amdgcn_rcpf(S.d[1]);
      slot_pin(S.r);
    } else if constexpr (kind == K_MULG) {
      slot_pin(S.r);
      S.val[0] = S.r[0] * (P::MRES ? vt[el] : prev.G[el]);
      S.val[1] = S.r[1] * (P::MRES ? vt[el + 1] : prev.G[el + 1]);
      if (inv_w != 1.0f) {
        S.val[0] *= inv_w;
        S.val[1] *= inv_w;
      }
      slot_pin(S.val);
    } else if constexpr (kind == K_PACK) {
      slot_pin(S.val);
      S.pc[i] = SURF_X_NOSPLIT ? __builtin_bit_cast(uint32_t, S.val[i & 1]) : P::pack(S.val);
      slot_pin(S.pc[i]);
      if (P::MRES) {
        dout[2 * (T - 1) + (el >> 3)].p[0][(el & 7) >> 1] = S.pc[0];
        vt[el] = S.val[0];
        vt[el + 1] = S.val[1];
      } else if (i == P::NP - 1) {
        Frag& f = dout[2 * (T - 1) + (el >> 3)];
#pragma unroll
        for (int k = 0; k < P::NP; ++k) f.p[k][(el & 7) >> 1] = S.pc[k];
      }
    } else if constexpr (kind == K_MRES) {
      mres_step<P>(c, vt, dout[2 * (T - 1) + (i & 1)].p[i >> 1], i & 1);
    } else if constexpr (kind == K_PACKT) {
      const f32x2 r2 = {vt[el], vt[el + 1]};
      dout[2 * (T - 1) + (el >> 3)].p[i][(el & 7) >> 1] = P::pack(r2);
    } else if constexpr (kind == K_EXPAND) {
      slot_pin(S.pc[i]);
      S.x = SURF_X_NOSPLIT ? S.val : P::expand(S.pc[i]);
      slot_pin(S.x);
    } else if constexpr (kind == K_SUB) {
      slot_pin(S.val, S.x);
      S.val[0] -= S.x[0];
      S.val[1] -= S.x[1];
      slot_pin(S.val);
    } else if constexpr (kind == K_SUBP) {
      S.val = SURF_X_NOSPLIT ? S.val : P::sub_piece(S.val, S.pc[i]);
    }
  };
  return run_chunk<P, true, CI, n_chunks<P>(true), PL>(
      c, [&](int ks) __attribute__((always_inline)) -> const Frag& { return din[ks]; }, slot);
}
template <class P, int L, int T>
__device__ __forceinline__ void bwd_hidden_tile(const Ctx& c, const FragT<P::NP>* din, FragT<P::NP>* dout, BwdPend& pend, const RegSlices<P>& keep) {
  const BwdPend prev = pend;
  f32x4 s_load[4] = {};
  const f32x16 G = bwd_tile<P, L, T, (T > 0)>(c, din, dout, prev, s_load, keep);
  pend.G = G;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (P::DEEP) {
      pend.s[g] = prev.sn[g];
      pend.sn[g] = s_load[g];  // (T == 3: nothing was loaded; the layer's last tile fills sn for the next layer)
    } else {
      pend.s[g] = s_load[g];
    }
  }
}
template <class P, int L>
__device__ __forceinline__ void bwd_layer(const Ctx& c, const FragT<P::NP>* din, FragT<P::NP>* dout, f32x16& accE,
                                          f32x16& accP, BwdPend& pend, const RegSlices<P>& keep) {
  bwd_hidden_tile<P, L, 0>(c, din, dout, pend, keep);
  bwd_hidden_tile<P, L, 1>(c, din, dout, pend, keep);
  bwd_hidden_tile<P, L, 2>(c, din, dout, pend, keep);
  bwd_hidden_tile<P, L, 3>(c, din, dout, pend, keep);
  const BwdPend prev = pend;
  if constexpr (L == 3) {
    f32x4 unused[4];
    accE += bwd_tile<P, L, 4, true>(c, din, dout, prev, unused, keep);
    accP += bwd_tile<P, L, 5, false>(c, din, dout, prev, pend.sn, keep);
  } else {
    accP += bwd_tile<P, L, 4, true>(c, din, dout, prev, pend.sn, keep);
  }
}

// R0: the recompute chunk (sdf_split_common.h, make_chunks): layer 0's four pre-activation tiles = W0 [e | 1] again, from the
// positional-encoding fragments the forward sweep used, in the same k-step and product order (the same bits).  MFMAs only:
// nothing is converted under it (the tile before it accumulates into accP), and the accumulators ARE the exponent arguments.
template <class P, int CI>
struct R0Plan {
  static constexpr GapPlan v = plan_gaps(R0_KS, P::NM, P::NP, chunk_dma<P, true, CI, n_chunks<P>(true)>(), R0_KS * P::NM, SlotProg{});
};
template <class P>
__device__ __forceinline__ void recompute_layer0(const Ctx& c, const FragT<P::NP> (&ef)[2], RegSlices<P>& keep) {
  if constexpr (P::R0) {
    constexpr int CI = r0_chunk();
    static_assert(P::CH.ks[CI] == R0_KS && P::CH.off[CI] == P::CH.off[fwd_chunk(0, 0)] && fwd_ne(0) == 2 && fwd_nl(0) == 2, "R0 chunk");
    run_chunk_n<P, true, CI, n_chunks<P>(true), R0Plan<P, CI>, 4>(
        c, [&](int ks) __attribute__((always_inline)) -> const FragT<P::NP>& { return ef[ks & 1]; }, [&](auto) __attribute__((always_inline)) {},
        keep.r0);
  }
}

// empty chunks at the end of the gradient stream (n_chunks): their only content is the DMA issue and the barrier
template <class P, int CI>
struct PadPlan {
  static constexpr GapPlan v = plan_gaps(0, P::NM, P::NP, chunk_dma<P, true, CI, n_chunks<P>(true)>(), 0, SlotProg{});
};
template <class P, int CI>
__device__ __forceinline__ void pad_chunks(const Ctx& c, const FragT<P::NP>* any) {
  if constexpr (CI < n_chunks<P>(true)) {
    run_chunk<P, true, CI, n_chunks<P>(true), PadPlan<P, CI>>(
        c, [&](int ks) __attribute__((always_inline)) -> const FragT<P::NP>& { return any[0]; }, [&](auto) __attribute__((always_inline)) {});
    pad_chunks<P, CI + 1>(c, any);
  }
}

template <class P, bool GRAD>
__global__ __launch_bounds__(WPB * 64, P::occ(GRAD)) void sdf_mlp_split_kernel(SdfArgs a) {
  typedef FragT<P::NP> Frag;
  constexpr int NS = P::nslot(GRAD), NCH = n_chunks<P>(GRAD);
  constexpr int NSL = GRAD ? P::LDS_SLICES : 0;
  // ONE LDS object (ring + exponent slices): with a second __shared__ array the compiler's LDS-DMA alias tracking falls
  // back to `s_waitcnt vmcnt(0)` in front of every ds_read (311 of them, kernel 64 -> 98 ms)
  __shared__ __attribute__((aligned(16))) char lds[NS * slot_bytes<P>() + WPB * NSL * 4096];
  static_assert(NCH % NS == 0 && (!GRAD || NCH - P::CH.n <= MAX_PAD) && NS >= 3, "slot of a chunk = index % ring length");
  Ctx c;
  c.lane = threadIdx.x & 63;
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.h = c.lane >> 5;
  c.lane16 = c.lane * 16;
  c.lds = lds;
  c.lds_s = lds + NS * slot_bytes<P>() + c.wave * (NSL * 4096) + c.lane16;
  if (P::MRES) {
    c.I0 = minus_identity_frag(c.lane, 0);
    c.I1 = minus_identity_frag(c.lane, 1);
  }
  c.wave1024 = c.wave * 1024;
  c.dma_voff = c.lane16 + c.wave * 1024;
  c.dma_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (SURF_SDF_DMA_WAVEMAJOR ? 0 : c.wave * 1024);
  c.wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.packed, 0, stream_bytes<P>(), 0x00020000);
  c.tr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.packed + stream_bytes<P>()), 0, TAIL_FLOATS * 4, 0x00020000);
  c.sr = __builtin_amdgcn_make_buffer_rsrc((void*)a.scratch, 0, (GRAD && !(SURF_X_NOSCRATCH & 1)) ? 0x7fffffff : 0, 0x00020000);
  c.sl = __builtin_amdgcn_make_buffer_rsrc((void*)a.scratch, 0, (GRAD && !(SURF_X_NOSCRATCH & 2)) ? 0x7fffffff : 0, 0x00020000);
  const int64_t wave_id = (int64_t)blockIdx.x * WPB + c.wave;
  c.svoff = (int)(wave_id * (SCR_SLOT * 4)) + c.lane * 16;
  const int64_t n_pts = a.n_dev ? (int64_t)*a.n_dev : a.n;
  const int64_t n_tiles = (n_pts + TILE - 1) / TILE;
  const int64_t n_rounds = (n_tiles + WPB - 1) / WPB;

  // stream prologue: chunks 0 .. NS-2 (later rounds inherit them from the last chunks of the round before)
  stage_dma<P, NS, 0>(c);
  stage_dma<P, NS, 1>(c);
  if (NS > 3) stage_dma<P, NS, 2>(c);
  if (NS > 4) stage_dma<P, NS, 3>(c);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef SURF_SDF_TIMING
  c.tprev = __builtin_readcyclecounter();
  for (int k = 0; k < 8; ++k) c.tacc[k] = 0;
#endif
  for (int64_t round = blockIdx.x; round < n_rounds; round += gridDim.x) {
    asm volatile("" : "+s"(c.dma_lds));  // not loop-invariant: the DMA addresses of a round are formed where they are used
    asm volatile("" : "+s"(c.wave1024));  // (likewise the wave's share of the scalar offsets)
    const int64_t tile = round * WPB + c.wave;
    const int64_t slot0 = tile * TILE + (c.lane & 31);
    const int64_t sc = slot0 < n_pts ? slot0 : n_pts - 1;
    const int64_t i = a.idx ? (int64_t)a.idx[sc] : sc;
    const bool active = (slot0 < n_pts) && (!a.mask || a.mask[i] != 0);
    float px, py, pz;
    if (a.pts) {
      px = a.pts[i * 3 + 0]; py = a.pts[i * 3 + 1]; pz = a.pts[i * 3 + 2];
    } else {  // lattice mode: the point from its linear index (n < 2^31, checked at launch)
      const uint32_t ii = (uint32_t)i, yz = ii / (uint32_t)a.lat_nz;
      px = a.lat_axes[0][yz / (uint32_t)a.lat_ny];
      py = a.lat_axes[1][yz % (uint32_t)a.lat_ny];
      pz = a.lat_axes[2][ii % (uint32_t)a.lat_nz];
    }

    Frag ef[2], pf[2];
    float y0 = 0.f;
    SinCos3 base;
    {
      float phi[16];
      auto posenc = [&]() __attribute__((always_inline)) {
        float e[16], je_unused[14];
        base = sincos3(px, py, pz);
        posenc_half(c.h, px, py, pz, base, e, je_unused, false);
        e[14] = 1.0f;  // bias k-element (weights carry the bias there, lane half 0 only)
        local_frags<P>(e, ef);
      };
      if (SURF_X_NOGATHER) {
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) phi[ch] = px * (float)ch;
        posenc();
      } else {
        gather_features<GRAD>(a, c, px, py, pz, phi, posenc);
      }
      f32x4 w6p[4];  // feature part of the last layer (all four loads first: the statements are emitted in source order)
#pragma unroll
      for (int g = 0; g < 4; ++g) w6p[g] = bload(c.tr, c.h * 64, TAIL_W6P * 4 + g * 16);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * g + q < 14) y0 = fmaf(w6p[g][q], phi[4 * g + q], y0);
      phi[14] = 1.0f;
      local_frags<P>(phi, pf);
    }
    SURF_T(0);

    // ------------------------------------------------ forward ----------------------------------------------------
    Frag hA[8], hB[8], dA[8];
    RegSlices<P> keep;
    f32x16 raw;
#pragma unroll
    for (int r = 0; r < 16; ++r) raw[r] = 0.f;
    fwd_layer<P, GRAD, 0>(c, raw, hA, hA, ef, pf, dA, y0, keep);
    fwd_layer<P, GRAD, 1>(c, raw, hA, hB, ef, pf, dA, y0, keep);
    fwd_layer<P, GRAD, 2>(c, raw, hB, hA, ef, pf, dA, y0, keep);
    fwd_layer<P, GRAD, 3>(c, raw, hA, hB, ef, pf, dA, y0, keep);
    fwd_layer<P, GRAD, 4>(c, raw, hB, hA, ef, pf, dA, y0, keep);
    fwd_layer<P, GRAD, 5>(c, raw, hA, hB, ef, pf, dA, y0, keep);
    SURF_T(1);
    {  // tile 3 of layer 5
      f32x4 w6t[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) w6t[g] = bload(c.tr, c.h * 256, TAIL_W6H * 4 + (12 + g) * 16);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 w = w6t[g];
        f32x2 hv[2], sv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x2 t2 = {raw[4 * g + 2 * q], raw[4 * g + 2 * q + 1]};
          softplus_pair<GRAD ? 1 : 0, P::PRESCALED>(t2, 1.0f / Scales<P>::W, hv[q], sv[q]);
          y0 = fmaf(w[2 * q], hv[q][0], y0);
          y0 = fmaf(w[2 * q + 1], hv[q][1], y0);
          if (GRAD)
            frag_set_pair<P>(dA[6 + (g >> 1)], 2 * (g & 1) + q, sv[q][0] * (w[2 * q] * Scales<P>::D),
                             sv[q][1] * (w[2 * q + 1] * Scales<P>::D));
        }
      }
    }
    {
      const f32x4 b6 = bload(c.tr, 0, TAIL_B6 * 4);
      y0 += __shfl_xor(y0, 32);
      y0 += b6[0];
    }
    if (active && c.h == 0) a.sdf[i] = GRAD ? y0 : y0 * a.out_sign;
    SURF_T(2);
    if (GRAD) {
      // ---------------------------------------------- reverse sweep ----------------------------------------------
      f32x16 accE, accP;
#pragma unroll
      for (int r = 0; r < 16; ++r) accE[r] = 0.f;
      {
        f32x4 w[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) w[g] = bload(c.tr, c.h * 64, TAIL_W6P * 4 + g * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int q = 0; q < 4; ++q) accP[4 * g + q] = w[g][q] * (Scales<P>::W * Scales<P>::D);
      }
      BwdPend pend = {};
      if (P::DEEP) load_sprime<P>(c, 4, 0, pend.sn, keep);  // slice of the first hidden tile (5, 0)
      bwd_layer<P, 5>(c, dA, hA, accE, accP, pend, keep);
      bwd_layer<P, 4>(c, hA, dA, accE, accP, pend, keep);
      bwd_layer<P, 3>(c, dA, hA, accE, accP, pend, keep);
      bwd_layer<P, 2>(c, hA, dA, accE, accP, pend, keep);
      if constexpr (P::R0) {
        recompute_layer0<P>(c, ef, keep);
        load_sprime<P>(c, 0, 0, pend.sn, keep);  // slice of hidden tile (1, 0): what reverse layer 2's last chunk fetches otherwise
      }
      bwd_layer<P, 1>(c, dA, hA, accE, accP, pend, keep);
      f32x4 Jq[12];  // feature Jacobian: DEEP fetches it under the last chunk
      if (P::DEEPJ) {
#pragma unroll
        for (int g = 0; g < 12; ++g) Jq[g] = bload_scratch(c.sl, c.svoff, SCR_S * 4 + g * 1024);
      }
      {
        f32x4 unused[4];
        accE += bwd_tile<P, 0, 0, false>(c, hA, dA, pend, unused, keep);
      }
      pad_chunks<P, P::CH.n>(c, hA);
      SURF_T(3);
      float g3[3] = {0.f, 0.f, 0.f};
      {
        float e2[16], je[14];
        posenc_half(c.h, px, py, pz, base, e2, je, true);
#pragma unroll
        for (int s2 = 0; s2 < 14; ++s2) {
          const int c0 = s2 % 3, c1 = (14 + s2) % 3;
          const float v = accE[s2] * je[s2];
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) g3[ax] += ((c.h ? c1 : c0) == ax) ? v : 0.f;
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          float Jf[24];
#pragma unroll
          for (int g = 0; g < 6; ++g) {
            const f32x4 v = P::DEEPJ ? Jq[6 * sl + g] : bload_scratch(c.sl, c.svoff, SCR_S * 4 + (6 * sl + g) * 1024);
            Jf[4 * g + 0] = v[0]; Jf[4 * g + 1] = v[1]; Jf[4 * g + 2] = v[2]; Jf[4 * g + 3] = v[3];
          }
#pragma unroll
          for (int ch = 0; ch < 7; ++ch) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) g3[ax] = fmaf(accP[7 * sl + ch], Jf[3 * ch + ax], g3[ax]);
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
      SURF_T(4);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last (unused) prefetches must land before the LDS is freed
#ifdef SURF_SDF_TIMING
  if (threadIdx.x == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&g_phase[k], c.tacc[k]);
#endif
}

template <class P>
int grid_blocks(int64_t n, bool grad) {
  int64_t tiles = (n + TILE - 1) / TILE;
  int64_t rounds = (tiles + WPB - 1) / WPB;
  return (int)(rounds < max_blocks<P>(grad) ? rounds : max_blocks<P>(grad));
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
inline uint16_t f16_bits(float v) {
  _Float16 h = (_Float16)v;  // round to nearest even
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}
inline float f16_to_f(uint16_t b) {
  _Float16 h;
  memcpy(&h, &b, 2);
  return (float)h;
}
template <class P> void split_host(float v, uint16_t* p);
template <> void split_host<PolBf3>(float v, uint16_t* p) {
  p[0] = bf16_rne(v);
  float r = v - bf16_to_f(p[0]);
  p[1] = bf16_rne(r);
  r = r - bf16_to_f(p[1]);
  p[2] = bf16_rne(r);
}
template <> void split_host<PolH2>(float v, uint16_t* p) {
  p[0] = f16_bits(v);
  p[1] = f16_bits(v - f16_to_f(p[0]));
}
inline int frag_feat(int tt, int s, int j, int h) { return 32 * tt + 16 * s + (j & 3) + 8 * (j >> 2) + 4 * h; }

// h_W / h_b: effective (weight-normed) fp32 matrices lin0..lin6, as for surf_sdf_pack_weights.
template <class P>
int pack_weights(const float* const* h_W, const float* const* h_b, unsigned char* out) {
  if (!h_W || !h_b || !out) return SURF_E_ARG;
  for (int l = 0; l < 7; ++l)
    if (!h_W[l] || !h_b[l]) return SURF_E_ARG;
  constexpr int NP = P::NP;
  const int in_dim[7] = {NE, 156, 156, 156, 156, 156, 156};
  const int out_dim[6] = {HID, HID, H2, HID, HID, HID};
  const float rsqrt2 = (float)(1.0 / sqrt(2.0));
  // P::PRESCALED (see the policy): input columns and biases x c = 100 log2 e, row 0 of lin6 x ln 2 / 100, hidden columns as they
  // are.  Every packed value stays ONE float product `weight * scale` (surf_amd/packing.py re-does exactly that on the device).
  const float c_in = P::PRESCALED ? (float)(100.0 * 1.4426950408889634) : 1.0f;
  const float c_in_rsqrt2 = P::PRESCALED ? (float)(100.0 * 1.4426950408889634 / sqrt(2.0)) : rsqrt2;
  const float k_out = P::PRESCALED ? (float)(0.6931471805599453 / 100.0) : 1.0f;
  memset(out, 0, stream_bytes<P>() + TAIL_FLOATS * 4);
  auto put = [&](int chunk, int ks, int lane, int j, float v) {
    uint16_t p[NP];
    split_host<P>(v * Scales<P>::W, p);
    for (int pc = 0; pc < NP; ++pc) {
      uint16_t* dst = reinterpret_cast<uint16_t*>(out + P::CH.off[chunk] + (ks * NP + pc) * 1024 + lane * 16);
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
            if (ks < fwd_ne(l)) {
              const int cidx = 8 * ks + j;
              if (cidx < 14 && 14 * h + cidx < NE) col = (l == 3 ? H2 : 0) + 14 * h + cidx;
              scale = l == 3 ? c_in_rsqrt2 : c_in;
              if (l == 0) is_bias = (cidx == 14 && h == 0);
            } else if (ks < fwd_nl(l)) {
              const int cidx = 8 * (ks - fwd_ne(l)) + j;
              if (cidx < 14) col = 128 + 14 * h + cidx;
              scale = c_in;
              is_bias = (cidx == 14 && h == 0);
            } else {
              const int hk = ks - fwd_nl(l);
              const int f = frag_feat(hk >> 1, hk & 1, j, h);
              if (f < hid_in) col = f;
              if (l == 3) scale = rsqrt2;
            }
            float v = 0.f;
            if (col >= 0 && row < out_dim[l]) v = h_W[l][(int64_t)row * in_dim[l] + col] * scale;
            if (is_bias && row < out_dim[l]) v = h_b[l][row] * c_in;
            put(ci, ks, lane, j, v);
          }
    }
  // ---- backward: G_in = W_l^T delta_l
  for (int l = 0; l < 6; ++l)
    for (int t = 0; t < BWD_NT[l]; ++t) {
      const int ci = bwd_chunk(l, t, P::R0);
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
              scale = l == 3 ? c_in_rsqrt2 : c_in;
            } else {
              if (r_row < 14) col = 128 + ch;
              scale = c_in;
            }
            float v = 0.f;
            if (col >= 0 && krow < out_dim[l]) v = h_W[l][(int64_t)krow * in_dim[l] + col] * scale;
            put(ci, ks, lane, j, v);
          }
    }
  // ---- fp32 tail: last layer row 0
  float* tail = reinterpret_cast<float*>(out + stream_bytes<P>());
  auto hk = [](int tt, int r, int h) { return 32 * tt + (r & 3) + 8 * (r >> 2) + 4 * h; };
  for (int h = 0; h < 2; ++h) {
    for (int s = 0; s < 64; ++s) tail[TAIL_W6H + h * 64 + s] = h_W[6][hk(s / 16, s % 16, h)] * k_out;
    for (int s = 0; s < 14; ++s) tail[TAIL_W6P + h * 16 + s] = h_W[6][128 + 14 * h + s];
  }
  tail[TAIL_B6] = h_b[6][0];
  return 0;
}

struct Lattice { const float* ax[3]; int ny, nz; float sign; };
template <class P>
int launch(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n, const float* const* h_vols,
           const int32_t* const* h_tables, const int* h_dims, int n_vol, const void* packed, float* sdf, float* grad,
           void* scratch, void* stream, const int32_t* d_n, const Lattice* lat = nullptr) {
  if ((!pts && !lat) || !h_vols || !h_tables || !h_dims || !packed || !sdf) return SURF_E_ARG;
  if (n <= 0 || n_vol <= 0) return SURF_E_ARG;
  if (n_vol > SURF_MAX_STAGES) return SURF_E_LIMIT;
  if (grad && !scratch) return SURF_E_ARG;
  SdfArgs a;
  a.pts = pts; a.mask = mask; a.idx = idx; a.n = n; a.n_dev = d_n; a.packed = (const unsigned char*)packed; a.sdf = sdf; a.grad = grad;
  a.scratch = (float*)scratch;
  a.lat_axes[0] = a.lat_axes[1] = a.lat_axes[2] = nullptr; a.lat_ny = a.lat_nz = 1; a.out_sign = 1.0f;
  if (lat) {
    if (grad || mask || idx || d_n || n >= (int64_t)1 << 31 || lat->ny < 1 || lat->nz < 1 || !lat->ax[0] || !lat->ax[1] || !lat->ax[2]) return SURF_E_ARG;
    for (int k = 0; k < 3; ++k) a.lat_axes[k] = lat->ax[k];
    a.lat_ny = lat->ny; a.lat_nz = lat->nz; a.out_sign = lat->sign;
  }
  for (int s = 0; s < SURF_MAX_STAGES; ++s) {
    a.vols[s] = s < n_vol ? h_vols[s] : h_vols[0];
    a.tables[s] = s < n_vol ? h_tables[s] : h_tables[0];  // absent levels: a valid address for gather_features' unconditional loads
    a.dims[s] = s < n_vol ? h_dims[s] : 0;
    if (s < n_vol && (!h_vols[s] || !h_tables[s] || h_dims[s] <= 1)) return SURF_E_ARG;
    if (s < n_vol && h_dims[s] > 1024) return SURF_E_LIMIT;  // 32-bit table indices (gather_features)
  }
  dim3 grid(grid_blocks<P>(n, grad != nullptr)), block(WPB * 64);
  if (grad)
    hipLaunchKernelGGL((sdf_mlp_split_kernel<P, true>), grid, block, 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((sdf_mlp_split_kernel<P, false>), grid, block, 0, (hipStream_t)stream, a);
  return surf_check_launch();
}

// extract_geometry's lattice (implicit_surface.py:337-351) without point tensors: nx x ny x nz values out[(ix ny + iy) nz + iz]
// = sign * sdf(ax[ix], ay[iy], az[iz]) by the forward-only kernel
template <class P>
int launch_lattice(const float* ax, const float* ay, const float* az, int nx, int ny, int nz, const float* const* h_vols,
                   const int32_t* const* h_tables, const int* h_dims, int n_vol, const void* packed, float* out, float sign, void* stream) {
  if (nx < 1 || ny < 1 || nz < 1) return SURF_E_ARG;
  const Lattice lat = {{ax, ay, az}, ny, nz, sign};
  return launch<P>(nullptr, nullptr, nullptr, (int64_t)nx * ny * nz, h_vols, h_tables, h_dims, n_vol, packed, out, nullptr, nullptr, stream,
                   nullptr, &lat);
}

template <class P>
int64_t scratch_bytes(int64_t n_points) {
  if (n_points <= 0) return 0;
  return (int64_t)grid_blocks<P>(n_points, true) * WPB * SCR_SLOT * sizeof(float);
}

}  // namespace

#ifdef SURF_SDF_TIMING
#ifdef SURF_SDF_TU_F16
#define SURF_DEBUG_PHASES surf_debug_phases_f16
#else
#define SURF_DEBUG_PHASES surf_debug_phases
#endif
extern "C" int SURF_DEBUG_PHASES(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 8) != hipSuccess) return 100;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return 100;
  }
  return 0;
}
#endif

#ifndef SURF_SDF_TU_F16  // (this translation unit: the bf16x3 kernels; sdf_mlp_split_f16.hip: the f16x2 ones)

extern "C" int64_t surf_sdf_bf16_packed_bytes(void) { return stream_bytes<PolBf3>() + TAIL_FLOATS * 4; }
extern "C" int64_t surf_sdf_bf16_scratch_bytes(int64_t n_points) { return scratch_bytes<PolBf3>(n_points); }
extern "C" int surf_sdf_pack_weights_bf16(const float* const* h_W, const float* const* h_b, unsigned char* out) {
  return pack_weights<PolBf3>(h_W, h_b, out);
}
extern "C" int surf_sdf_mlp_bf16x3(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n,
                                   const float* const* h_vols, const int32_t* const* h_tables, const int* h_dims, int n_vol,
                                   const void* packed, float* sdf, float* grad, void* scratch, void* stream) {
  return launch<PolBf3>(pts, mask, idx, n, h_vols, h_tables, h_dims, n_vol, packed, sdf, grad, scratch, stream, nullptr);
}
// The same with the number of idx entries read from DEVICE memory (d_n[0] <= n = capacity of idx): the launch that follows a
// compaction needs no host round trip for the count (device-side counters instead of host syncs).
extern "C" int surf_sdf_mlp_bf16x3_dn(const float* pts, const int32_t* idx, int64_t n_capacity, const int32_t* d_n,
                                     const float* const* h_vols, const int32_t* const* h_tables, const int* h_dims, int n_vol,
                                     const void* packed, float* sdf, float* grad, void* scratch, void* stream) {
  if (!idx || !d_n) return SURF_E_ARG;
  return launch<PolBf3>(pts, nullptr, idx, n_capacity, h_vols, h_tables, h_dims, n_vol, packed, sdf, grad, scratch, stream, d_n);
}

extern "C" int surf_sdf_lattice_bf16x3(const float* ax, const float* ay, const float* az, int nx, int ny, int nz, const float* const* h_vols,
                                       const int32_t* const* h_tables, const int* h_dims, int n_vol, const void* packed, float* out,
                                       float sign, void* stream) {
  return launch_lattice<PolBf3>(ax, ay, az, nx, ny, nz, h_vols, h_tables, h_dims, n_vol, packed, out, sign, stream);
}

#else
extern "C" int surf_sdf_lattice_f16x2(const float* ax, const float* ay, const float* az, int nx, int ny, int nz, const float* const* h_vols,
                                      const int32_t* const* h_tables, const int* h_dims, int n_vol, const void* packed, float* out,
                                      float sign, void* stream) {
  return launch_lattice<PolH2>(ax, ay, az, nx, ny, nz, h_vols, h_tables, h_dims, n_vol, packed, out, sign, stream);
}
extern "C" int64_t surf_sdf_f16_packed_bytes(void) { return stream_bytes<PolH2>() + TAIL_FLOATS * 4; }
extern "C" int64_t surf_sdf_f16_scratch_bytes(int64_t n_points) { return scratch_bytes<PolH2>(n_points); }
extern "C" int surf_sdf_pack_weights_f16(const float* const* h_W, const float* const* h_b, unsigned char* out) {
  return pack_weights<PolH2>(h_W, h_b, out);
}
extern "C" int surf_sdf_mlp_f16x2(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n,
                                  const float* const* h_vols, const int32_t* const* h_tables, const int* h_dims, int n_vol,
                                  const void* packed, float* sdf, float* grad, void* scratch, void* stream) {
  return launch<PolH2>(pts, mask, idx, n, h_vols, h_tables, h_dims, n_vol, packed, sdf, grad, scratch, stream, nullptr);
}
// The same with the number of idx entries read from DEVICE memory (d_n[0] <= n = capacity of idx): the launch that follows a
// compaction needs no host round trip for the count (device-side counters instead of host syncs).
extern "C" int surf_sdf_mlp_f16x2_dn(const float* pts, const int32_t* idx, int64_t n_capacity, const int32_t* d_n,
                                    const float* const* h_vols, const int32_t* const* h_tables, const int* h_dims, int n_vol,
                                    const void* packed, float* sdf, float* grad, void* scratch, void* stream) {
  if (!idx || !d_n) return SURF_E_ARG;
  return launch<PolH2>(pts, nullptr, idx, n_capacity, h_vols, h_tables, h_dims, n_vol, packed, sdf, grad, scratch, stream, d_n);
}
#endif
