// K10b: multi-view feature fetch + IBRNet-style blending MLP on the 16-bit MFMA pipes ("split" blend kernels).
//
// Restates lookup_feature / compute_angle   projector.py:485-556
//          BlendingNetwork.forward           blending_network.py:69-118
// (same arithmetic as blend.hip, which stays in the tree as the fp32-MFMA reference of this kernel.)
//
// Precisions, as in sdf_mlp_split.hip: every fp32 operand is split into 16-bit pieces, partial products accumulate in
// fp32 with v_mfma_f32_32x32x16_{bf16,f16}:
//   bf16x3  exact three-way split, six products per k-step: fp32-equivalent (default)
//   f16x2   two fp16 pieces (operand error <= 2^-22 relative or 2^-25 absolute), three products per k-step
//
// Design for CDNA4:
//   * one wavefront owns 32 sample points, lane l = (sample j = l & 31, half h = l >> 5); every Linear is
//     W (A operand) x activations^T (B operand), so the 32x32 accumulator tile of a layer IS the B operand of the next
//     one after ELU and splitting: accumulator registers 8s..8s+7 of half h are the eight k-values of k-step s.
//   * the whole weight set (26 k-step x tile blocks of NP KB = 78 KB bf16x3 / 52 KB f16x2, + bias and dot rows) is
//     copied ONCE per workgroup into LDS and stays there: A fragments are ds_read_b128 (1 KB per wave instruction,
//     lane-linear, conflict-free), there is no weight stream, no ring and no barrier inside the tile loop.
//   * workgroups of 8 wavefronts = two per SIMD at <= 256 registers: with the 16-bit pipes the kernel is bound by the
//     VALU work per activation (ELU + operand split), not by the matrix pipe, and two wavefronts per SIMD let one
//     wavefront's VALU work run beside the other's MFMAs without hand-scheduling.
//   * ELU on the pre-activation scaled by log2(e) (folded into the packed weights and biases):
//     elu(x) = ln2 max(t, 0) + (min(exp2 t, 1) - 1), t = x log2 e: four VALU operations per element;
//     biases are the accumulators' initial values (LDS rows), not VALU adds.
//   * the 19 per-view input channels are split between the lane halves as in blend.hip: no texel is fetched twice.
#include <math.h>
#include <string.h>

#include "blend_raw.h"
#include "common.h"

namespace {

using namespace blend_raw;

constexpr int TILE = 32;
#ifndef SURF_BLEND_WPB
#define SURF_BLEND_WPB 8
#endif
constexpr int WPB = SURF_BLEND_WPB;  // wavefronts per workgroup (8 = two per SIMD; one workgroup per CU owns the LDS image)

// ---- LDS image --------------------------------------------------------------------------------------------------------
enum { L_RD0, L_RD2, L_B0S, L_B0V, L_B2, L_V0, L_V2, L_W0, L_R0, L_R2, N_L };
constexpr int NKS[N_L] = {1, 1, 3, 2, 4, 2, 2, 2, 3, 1};  // k-steps of 16
constexpr int NT[N_L] = {1, 1, 2, 2, 1, 1, 1, 1, 1, 1};   // 32-row output tiles
constexpr int blk_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += NKS[i] * NT[i]; return o; }
constexpr int N_BLK = blk_off(N_L);  // 26 blocks of NP KB: [layer][k-step][tile][piece][lane] x 16 B
enum { B_RD0, B_RD2, B_B0_T0, B_B0_T1, B_B2, B_V0, B_V2, B_W0, B_R0, B_R2, N_BIAS };  // accumulator-init rows [h][16]
enum { D_VIS, D_VIS2, D_RGB4, N_DOT };                                                    // per-lane dot rows [h][16]
// BB = bytes of one block: NP x 1 KB for the 16-bit policies, 2 KB for the fp32 policy
template <int BB> constexpr int bias_off() { return N_BLK * BB; }
template <int BB> constexpr int dot_off() { return bias_off<BB>() + N_BIAS * 128; }
template <int BB> constexpr int scal_off() { return dot_off<BB>() + N_DOT * 128; }  // [|s|, b_vis, b_vis2, b_rgb4]
template <int BB> constexpr int image_bytes() { return scal_off<BB>() + 16; }

constexpr float LOG2E = 1.44269504088896341f, LN2 = 0.69314718055994531f;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NP>
struct FragT { u32x4 p[NP]; };  // B operand of one 16-wide k-step: NP pieces x 8 x 16 bit

// Matrix-pipe residuals of the exact bf16 split (round 4; see split_tile_mres below): 1 = on (default), 0 = the VALU form.
// Precondition of the matrix-pipe form: FINITE activations.  The residual multiplies a sample's whole B column by the identity
// operand, so one inf / NaN activation turns all 32 rows of that sample's column into NaN (the VALU form confines it to the
// one value).  Finite inputs and weights give finite activations (ELU / sigmoid / softmax layers); nothing upstream of this
// kernel produces a non-finite feature.
#ifndef SURF_BLEND_MRES
#define SURF_BLEND_MRES 1
#endif

// ... per call site of make_frags / elu_make_frags (bit k of the mask = site k in source order; default: every site).  Round 5's
// A/B of "residual MFMAs only at the wide layers' sites" (the ones with >= 16 values a tile): see DESIGN K10b.
#ifndef SURF_BLEND_MRES_MASK
#define SURF_BLEND_MRES_MASK 0x3ff
#endif
struct BPolBf3 {
  static constexpr int NP = 3, ID = 1, BB = NP * 1024, LANE_BYTES = 16;
  static constexpr bool MRES = SURF_BLEND_MRES != 0;
  static constexpr bool mres_at(int site) { return MRES && ((SURF_BLEND_MRES_MASK >> site) & 1); }
#ifndef SURF_BLEND_REG_VIEWS_BF3
#define SURF_BLEND_REG_VIEWS_BF3 2
#endif
  static constexpr int REG_VIEWS = SURF_BLEND_REG_VIEWS_BF3;  // staged views kept in registers (after the ones in LDS)
  typedef FragT<NP> Frag;
  static __device__ __forceinline__ void set_pair(Frag& f, int pr, float a, float b);
  static __device__ __forceinline__ void zero_pair(Frag& f, int pr);
  template <int NUSED> static __device__ __forceinline__ void mma_lds(f32x16& acc, const char* blk_lane, const Frag& b);
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    uint32_t u = __builtin_bit_cast(uint32_t, v);
    asm volatile("" : "+v"(u));  // keep the packed value: the residuals below come from its two halves
    return u;
  }
  static __device__ __forceinline__ void split(float a, float b, uint32_t (&p)[NP]) { surf_split3_bf16(a, b, p); }
  static __device__ __forceinline__ void mma(f32x16& acc, const u32x4 (&a)[NP], const FragT<NP>& b) {
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
};

struct BPolH2 {
  static constexpr int NP = 2, ID = 2, BB = NP * 1024, LANE_BYTES = 16;
  static constexpr bool MRES = false;  // one residual level, formed by v_fma_mix-style arithmetic: nothing to gain
  static constexpr bool mres_at(int) { return false; }
#ifndef SURF_BLEND_REG_VIEWS_H2
#define SURF_BLEND_REG_VIEWS_H2 3
#endif
  static constexpr int REG_VIEWS = SURF_BLEND_REG_VIEWS_H2;
  typedef FragT<NP> Frag;
  static __device__ __forceinline__ void set_pair(Frag& f, int pr, float a, float b);
  static __device__ __forceinline__ void zero_pair(Frag& f, int pr);
  template <int NUSED> static __device__ __forceinline__ void mma_lds(f32x16& acc, const char* blk_lane, const Frag& b);
  static __device__ __forceinline__ void split(float a, float b, uint32_t (&p)[NP]) {
    const f32x2 v = {a, b};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 r = v - __builtin_convertvector(h, f32x2);
    p[0] = __builtin_bit_cast(uint32_t, h);
    p[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
  }
  static __device__ __forceinline__ void mma(f32x16& acc, const u32x4 (&a)[NP], const FragT<NP>& b) {
#define SURF_MF(x, y) \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[x]), __builtin_bit_cast(f16x8, b.p[y]), acc, 0, 0, 0)
    SURF_MF(1, 0);
    SURF_MF(0, 1);
    SURF_MF(0, 0);
#undef SURF_MF
  }
};

// fragment helpers of the 16-bit policies
template <class P>
struct Frag16 {
  static __device__ __forceinline__ void set_pair(typename P::Frag& f, int pr, float a, float b) {
    uint32_t p[P::NP];
    P::split(a, b, p);
#pragma unroll
    for (int k = 0; k < P::NP; ++k) f.p[k][pr] = p[k];
  }
  static __device__ __forceinline__ void zero_pair(typename P::Frag& f, int pr) {
#pragma unroll
    for (int k = 0; k < P::NP; ++k) f.p[k][pr] = 0u;
  }
  // NUSED (number of k-slots that carry data) is irrelevant here: one MFMA covers all 16 k of the step
  template <int NUSED>
  static __device__ __forceinline__ void mma_lds(f32x16& acc, const char* blk_lane, const typename P::Frag& b) {
    u32x4 a[P::NP];
#pragma unroll
    for (int p = 0; p < P::NP; ++p) a[p] = *reinterpret_cast<const u32x4*>(blk_lane + p * 1024);
    P::mma(acc, a, b);
  }
};

__device__ __forceinline__ void BPolBf3::set_pair(Frag& f, int pr, float a, float b) { Frag16<BPolBf3>::set_pair(f, pr, a, b); }
__device__ __forceinline__ void BPolBf3::zero_pair(Frag& f, int pr) { Frag16<BPolBf3>::zero_pair(f, pr); }
template <int NUSED> __device__ __forceinline__ void BPolBf3::mma_lds(f32x16& acc, const char* l, const Frag& b) { Frag16<BPolBf3>::mma_lds<NUSED>(acc, l, b); }
__device__ __forceinline__ void BPolH2::set_pair(Frag& f, int pr, float a, float b) { Frag16<BPolH2>::set_pair(f, pr, a, b); }
__device__ __forceinline__ void BPolH2::zero_pair(Frag& f, int pr) { Frag16<BPolH2>::zero_pair(f, pr); }
template <int NUSED> __device__ __forceinline__ void BPolH2::mma_lds(f32x16& acc, const char* l, const Frag& b) { Frag16<BPolH2>::mma_lds<NUSED>(acc, l, b); }

// fp32 policy: no operand split, v_mfma_f32_32x32x2_f32 (exact fp32, 1/16 of the 16-bit rate).  A "k-step" is the same
// eight k-slots per lane half as in the 16-bit layouts (so the packer's row / column maps are shared); it takes one MFMA
// per slot that carries data.  Block = [lane][8 floats] = 2 KB.
struct BPolF32 {
  static constexpr int NP = 1, ID = 3, BB = 2048, LANE_BYTES = 32;
  static constexpr bool MRES = false;
  static constexpr bool mres_at(int) { return false; }
  static constexpr int REG_VIEWS = 0;
  struct Frag { float v[8]; };
  static __device__ __forceinline__ void set_pair(Frag& f, int pr, float a, float b) { f.v[2 * pr] = a; f.v[2 * pr + 1] = b; }
  static __device__ __forceinline__ void zero_pair(Frag& f, int pr) { f.v[2 * pr] = 0.f; f.v[2 * pr + 1] = 0.f; }
  template <int NUSED>
  static __device__ __forceinline__ void mma_lds(f32x16& acc, const char* blk_lane, const Frag& b) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(blk_lane);
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
    if (NUSED > 4) a1 = *reinterpret_cast<const f32x4*>(blk_lane + 16);
#pragma unroll
    for (int i = 0; i < NUSED; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(i < 4 ? a0[i] : a1[i - 4], b.v[i], acc, 0, 0, 0);
  }
};

#ifdef SURF_BLEND_TIMING  // debug builds only (scripts/build_variant.sh): per-phase shader-clock totals of wavefront 0 of every workgroup
__device__ unsigned long long g_bwg[256][2];  // s_memrealtime at the start / end of every workgroup's wavefront 0
__device__ unsigned long long g_bphase[10];  // [8], [9]: s_memtime / s_memrealtime (100 MHz) span of workgroup 0's wavefront 0
#define SURF_BT(k)                                                \
  do {                                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    tacc[k] += now_ - tprev;                                      \
    tprev = now_;                                                 \
  } while (0)
#else
#define SURF_BT(k)
#endif

struct BlendArgs {
  const float* pts;
  const uint8_t* mask;
  const int32_t* idx;  // optional list of point indices (n entries)
  int64_t n;
  const int32_t* n_dev;  // optional device-side count (<= n = capacity of idx)
  const float* feats[4];
  int hw[8];
  const float* imgs;
  float K[SURF_MAX_VIEWS][9];
  float w2c[SURF_MAX_VIEWS][12];
  float cpos[SURF_MAX_VIEWS][3];
  const unsigned char* w;  // LDS image (surf_blend_pack_weights_split)
  float* color;
  uint8_t* n_valid;
  float* scratch;  // per-wavefront staging slots (surf_blend_split_scratch_bytes)
  int nv;
};

// Phase boundary for the instruction scheduler: without it hipcc hoists the LDS reads (A fragments, bias rows) of later
// layers far ahead of their use and spills; latency is hidden by the second wavefront of the SIMD instead.
#ifndef SURF_BLEND_NOPHASE
#define SURF_PHASE() __builtin_amdgcn_sched_barrier(0)
#else
#define SURF_PHASE()
#endif

// ---- activations ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }
__device__ __forceinline__ float elu_x(float x) { return x > 0.f ? x : fexp(x) - 1.0f; }
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + fexp(-x)); }
// ELU of a pre-activation given as t = x log2(e):  ln2 max(t, 0) + (min(exp2(t), 1) - 1)
__device__ __forceinline__ float elu_t(float t) {
  const float e = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(t), 0.0f, 1.0f);  // the clamp modifier of v_exp_f32
  const float m = __builtin_amdgcn_fmed3f(t, 0.0f, 3.0e38f);                       // max(t, 0) without a canonicalising v_max
  return fmaf(m, LN2, e - 1.0f);
}
template <int N>
__device__ __forceinline__ void elu_rows(const f32x16& acc, float* out) {
#pragma unroll
  for (int r = 0; r < N; ++r) out[r] = elu_t(acc[r]);
}

// N values (N <= 8 * NF) -> NF k-step fragments, zero padded
template <class P, int N, int NF>
__device__ __forceinline__ void frags_from(const float* v, typename P::Frag* f) {
#pragma unroll
  for (int s = 0; s < NF; ++s)
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      const int i0 = 8 * s + 2 * pr;
      if (i0 < N) P::set_pair(f[s], pr, v[i0], i0 + 1 < N ? v[i0 + 1] : 0.0f);
      else P::zero_pair(f[s], pr);
    }
}
// ELU of the first N registers of an accumulator tile straight into k-step fragments (N = 8 or 16), pair by pair
template <class P, int N>
__device__ __forceinline__ void elu_frags(const f32x16& acc, typename P::Frag* f) {
#pragma unroll
  for (int r = 0; r < N; r += 2) P::set_pair(f[r >> 3], (r & 7) >> 1, elu_t(acc[r]), elu_t(acc[r + 1]));
}

// ---- the residuals of the exact bf16 split on the matrix pipe (bf16x3, round 4) ------------------------------------------
// The kernel is VALU-issue bound (SQ counters, profiles/r04_blend_sq_bf16x3_before.txt: VALU issue 77 % of SIMD time, matrix
// pipe 33 %), and a third of its VALU instructions were the operand split a = p0 + p1 + p2: per PAIR of values three
// v_cvt_pk_bf16_f32 plus, for each of the two residual levels, two expands (shift / and) and two subtracts = 11 instructions.
// A residual r = a - p0 of a whole activation tile is an accumulator update with a constant A operand,
//       R (32 rows x 32 samples) = A_tile - sum_s I_s * P0_s ,
// where P0_s is the packed first piece AS the B fragment of its own k-step s (which it is about to be anyway) and I_s has a
// single -1 per row: row rho of the tile = k-slot (g, i') of step s with rho = (i' & 3) + 8 (2 s + (i' >> 2)) + 4 g, the same
// register <-> feature map `hk` that makes an accumulator tile the next layer's B operand.  One v_mfma_f32_32x32x16_bf16 per
// k-step and level (12 issue cycles) replaces 16 VALU instructions (~70 issue cycles) of every lane; the products -1 x p0 and
// the sum a - p0 are exact (Sterbenz), and scripts/microbench/mfma_residual.hip shows the pipe returns them bit-identical to
// the VALU subtraction for 2 M values (zeros, bf16 ties, both exponent extremes; not inf: 0 x inf would poison a column -
// activations here are ELU outputs of O(1)).  So the kernel's results do not change by a single bit.
__device__ __forceinline__ u32x4 ident_frag(int lane, int s) {
  const int i = lane & 31, g = lane >> 5;
  u32x4 f = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int ip = 0; ip < 8; ++ip) {
    const int rho = (ip & 3) + 8 * (2 * s + (ip >> 2)) + 4 * g;
    if (rho == i) f[ip >> 1] |= (ip & 1) ? 0xBF800000u : 0x0000BF80u;  // bf16(-1.0)
  }
  return f;
}

struct Ctx {
  const char* lds;
  int lane16, h64;
  u32x4 I0, I1;  // (bf16x3) A operands of the matrix-pipe residual, k-steps 0 and 1 of a 32-row tile: see split_tile_mres
};
// A copy of the context whose LDS offsets the optimiser cannot see through.  The LDS image is read-only inside the tile
// loop, so without this every ds_read of a weight fragment / bias row is a loop invariant AND common to all (unrolled)
// views: hipcc hoists and merges them and then keeps hundreds of registers live (and spills).  One opaque copy per view
// makes each view read its own fragments when it needs them.
__device__ __forceinline__ Ctx opaque(const Ctx& c) {
  Ctx o = c;
  asm volatile("" : "+v"(o.lane16), "+v"(o.h64));
  return o;
}

// first N values of r (the rest of r must be finite: zeros) -> NF = ceil(N / 8) fragments, exact three-way split, residuals on
// the matrix pipe
template <class P, int N, int NF>
__device__ __forceinline__ void split_tile_mres(const Ctx& c, f32x16 r, typename P::Frag* f) {
  static_assert(NF == (N + 7) / 8 && NF <= 2 && P::NP == 3, "one 32-row tile");
#pragma unroll
  for (int lvl = 0; lvl < 3; ++lvl) {
#pragma unroll
    for (int s = 0; s < NF; ++s)
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const int i0 = 8 * s + 2 * pr;
        if (i0 < N) {
          bf16x2 v;
          v[0] = (__bf16)r[i0];
          v[1] = (__bf16)(i0 + 1 < N ? r[i0 + 1] : 0.0f);
          f[s].p[lvl][pr] = __builtin_bit_cast(uint32_t, v);
        } else {
          f[s].p[lvl][pr] = 0u;
        }
      }
    if (lvl == 2) break;
    r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c.I0), __builtin_bit_cast(bf16x8, f[0].p[lvl]), r, 0, 0, 0);
    if (NF > 1)
      r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c.I1), __builtin_bit_cast(bf16x8, f[1].p[lvl]), r, 0, 0, 0);
  }
}
// N values -> fragments (policy's way)
template <class P, int N, int NF, int SITE>
__device__ __forceinline__ void make_frags(const Ctx& c, const float* v, typename P::Frag* f) {
  if constexpr (P::mres_at(SITE) && N >= 3) {
    constexpr int N0 = N > 16 ? 16 : N;  // tile by tile (two k-steps each)
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = i < N0 ? v[i] : 0.0f;
    split_tile_mres<P, N0, (N0 + 7) / 8>(c, r, f);
    if constexpr (N > 16) {
      constexpr int N1 = N - 16;
      static_assert(N1 <= 16, "at most two tiles");
      f32x16 r1;
#pragma unroll
      for (int i = 0; i < 16; ++i) r1[i] = i < N1 ? v[16 + i] : 0.0f;
      split_tile_mres<P, N1, (N1 + 7) / 8>(c, r1, f + 2);
    }
  } else {
    frags_from<P, N, NF>(v, f);
  }
}
// ELU of the first N registers of an accumulator tile -> fragments
template <class P, int N, int SITE>
__device__ __forceinline__ void elu_make_frags(const Ctx& c, const f32x16& acc, typename P::Frag* f) {
  if constexpr (P::mres_at(SITE)) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = i < N ? elu_t(acc[i]) : 0.0f;
    split_tile_mres<P, N, (N + 7) / 8>(c, r, f);
  } else {
    elu_frags<P, N>(acc, f);
  }
}

template <class P>
__device__ __forceinline__ f32x16 lds_row16(const Ctx& c, int byte_off) {  // [h][16] floats, broadcast within a half
  f32x16 v;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(c.lds + byte_off + c.h64 + g * 16);
    v[4 * g + 0] = x[0]; v[4 * g + 1] = x[1]; v[4 * g + 2] = x[2]; v[4 * g + 3] = x[3];
  }
  return v;
}
template <class P> __device__ __forceinline__ f32x16 bias_row(const Ctx& c, int row) { return lds_row16<P>(c, bias_off<P::BB>() + row * 128); }
template <class P> __device__ __forceinline__ f32x16 dot_row(const Ctx& c, int row) { return lds_row16<P>(c, dot_off<P::BB>() + row * 128); }

// acc += W[layer L, k-step KS, tile T] x b   (A fragments from the LDS image); NUSED = k-slots of this step that carry data
template <class P, int L, int KS, int T, int NUSED = 8>
__device__ __forceinline__ void mma_blk(const Ctx& c, f32x16& acc, const typename P::Frag& b) {
  constexpr int OFF = (blk_off(L) + KS * NT[L] + T) * P::BB;
  P::template mma_lds<NUSED>(acc, c.lds + OFF + c.lane16 * (P::LANE_BYTES / 16), b);
}
// k-slots in use in the LAST k-step of every layer (max over the lane halves; all earlier k-steps are full)
constexpr int LAST_USED[N_L] = {2, 8, 7, 3, 8, 8, 8, 8, 3, 8};
// one 32-row tile of layer L over NK k-steps
template <class P, int L, int T, int NK>
__device__ __forceinline__ void mma_layer(const Ctx& c, f32x16& acc, const typename P::Frag* b) {
  static_assert(NK == NKS[L], "k-steps");
  if constexpr (NK == 1) mma_blk<P, L, 0, T, LAST_USED[L]>(c, acc, b[0]);
  if constexpr (NK >= 2) mma_blk<P, L, 0, T>(c, acc, b[0]);
  if constexpr (NK == 2) mma_blk<P, L, 1, T, LAST_USED[L]>(c, acc, b[1]);
  if constexpr (NK >= 3) mma_blk<P, L, 1, T>(c, acc, b[1]);
  if constexpr (NK == 3) mma_blk<P, L, 2, T, LAST_USED[L]>(c, acc, b[2]);
  if constexpr (NK >= 4) mma_blk<P, L, 2, T>(c, acc, b[2]);
  if constexpr (NK == 4) mma_blk<P, L, 3, T, LAST_USED[L]>(c, acc, b[3]);
  SURF_PHASE();
}

// Bilinear fetch of a texel4 map with zero padding (grid_sample, align_corners=False), branch-free: out-of-range taps read a
// clamped texel with weight 0, so all four 16-byte loads of a fetch (and of every fetch of a view) can be in flight together.
// Round 4: the kernel is VALU-issue bound and the fetch was ~75 instructions a map (per CORNER: range test, clamp, 64-bit
// address, weight select).  Now per AXIS two clamped indices and two weights that are zero outside (TapGeom, shared by the maps
// of one resolution: the image and the finest feature level), 32-bit texel indices and one v_lshl_add_u64 per load.
// a / c for a divisor with its correctly rounded reciprocal rc = RN(1 / c) at hand: q = RN(a rc), r = a - q c (exact, one FMA),
// RN(q + r rc) is the correctly rounded quotient (Markstein); checked against IEEE division for 1.3e8 random numerators over
// every divisor (W - 1) / 2 of the pyramids in use.  Three instructions instead of the ten of v_div_scale .. v_div_fixup.
__device__ __forceinline__ float div_const(float a, float c, float rc) {
  const float q = a * rc;
  const float r = fmaf(-q, c, a);
  return fmaf(r, rc, q);
}
struct TapGeom {
  int idx[4];   // texel index (y * W + x) of (y0,x0), (y0,x1), (y1,x0), (y1,x1) = grid_sample's nw, ne, sw, se
  float w[4];
};
__device__ __forceinline__ void axis_taps(float g, int n, int& i0, int& i1, float& w0, float& w1) {
  const float f = floorf(g);
  const float t = g - f;
  const int i = (int)f;
  i0 = min(max(i, 0), n - 1);
  i1 = min(max(i + 1, 0), n - 1);
  w0 = (unsigned)i < (unsigned)n ? 1.0f - t : 0.0f;
  w1 = (unsigned)(i + 1) < (unsigned)n ? t : 0.0f;
}
__device__ __forceinline__ TapGeom tap_geom(int H, int W, float x, float y) {
  int x0, x1, y0, y1;
  float wx0, wx1, wy0, wy1;
  axis_taps(x, W, x0, x1, wx0, wx1);
  axis_taps(y, H, y0, y1, wy0, wy1);
  TapGeom g;
  const int r0 = y0 * W, r1 = y1 * W;
  g.idx[0] = r0 + x0; g.idx[1] = r0 + x1; g.idx[2] = r1 + x0; g.idx[3] = r1 + x1;
  g.w[0] = wx0 * wy0; g.w[1] = wx1 * wy0; g.w[2] = wx0 * wy1; g.w[3] = wx1 * wy1;
  return g;
}
struct Tap4 {
  f32x4 v[4];
};
__device__ __forceinline__ void tap_issue(Tap4& t, const float* __restrict__ map, const TapGeom& g) {
#pragma unroll
  for (int k = 0; k < 4; ++k) t.v[k] = *reinterpret_cast<const f32x4*>(map + (int64_t)g.idx[k] * 4);
}
__device__ __forceinline__ f32x4 tap_finish(const Tap4& t, const TapGeom& g) {
#ifdef SURF_BX_NOFMA
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) acc += t.v[k] * g.w[k];
  return acc;
#else
  f32x4 acc = t.v[0] * g.w[0];
#pragma unroll
  for (int k = 1; k < 4; ++k)
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) acc[ch] = fmaf(t.v[k][ch], g.w[k], acc[ch]);
  return acc;
#endif
}

template <class P> constexpr int lds_bytes() { return image_bytes<P::BB>(); }

// Per-wavefront staging slot in global memory (L2 / Infinity-Cache resident: written in pass 1, read back within the
// same tile): per source view five 16-byte groups per lane = [floc 0..3][floc 4..7][floc 8..11][ray_diff][rgb, +-ex]
// (ex = exp(|s| (dot - 1)) > 0, stored negated where the view's mask is 0).  It makes the register state independent of
// the number of views, so the view loops are real loops: one code path for 1..7 source views, ~4x less code than the
// unrolled form and no spills at two wavefronts per SIMD.
constexpr int SLOT_GROUPS = 5;
constexpr int slot_floats(int ns) { return ns * SLOT_GROUPS * 64 * 4; }
// The first lds_views<P>() source views of a tile are staged in the workgroup's spare LDS instead (5 KB per wavefront and
// view behind the weight image: bf16x3 78 + 2 x 40 KB, f16x2 52 + 2 x 40, f32lds 104 + 40): each staged view is written once
// and read three times (mean sweep, variance sweep, pass 2), and what does not leave the CU costs neither vector-memory
// instructions nor - the SDF kernel's lesson of round 3 - the clock the chip sustains.
#ifndef SURF_BLEND_LDS_VIEWS
#define SURF_BLEND_LDS_VIEWS 2
#endif
constexpr int VIEW_BYTES = SLOT_GROUPS * 64 * 16;
template <class P> constexpr int lds_views() {
  const int n = (160 * 1024 - lds_bytes<P>()) / (WPB * VIEW_BYTES);
  return n < SURF_BLEND_LDS_VIEWS ? n : SURF_BLEND_LDS_VIEWS;
}
struct Stage {
  float* slot;   // this wavefront's staging slot in global memory
  char* lds_v;   // ... and in LDS (+ lane * 16)
  int n_lds;     // views [0, n_lds) are in LDS
};

struct ViewState {
  float floc[12];
  float rd[4];
  float rgb[3];
  float ex;  // > 0; mk = 1
  float mk;
};
// (v is wave-uniform: a scalar branch; the two address spaces get their own instructions - a pointer select would make
//  every access a FLAT one)
template <class PtrT>
__device__ __forceinline__ void slot_store_at(PtrT p, const ViewState& s, int stride = 64) {
  p[0 * stride] = f32x4{s.floc[0], s.floc[1], s.floc[2], s.floc[3]};
  p[1 * stride] = f32x4{s.floc[4], s.floc[5], s.floc[6], s.floc[7]};
  p[2 * stride] = f32x4{s.floc[8], s.floc[9], s.floc[10], s.floc[11]};
  p[3 * stride] = f32x4{s.rd[0], s.rd[1], s.rd[2], s.rd[3]};
  p[4 * stride] = f32x4{s.rgb[0], s.rgb[1], s.rgb[2], s.mk != 0.f ? s.ex : -s.ex};
}
template <class PtrT>
__device__ __forceinline__ void slot_load_at(PtrT p, ViewState& s, int stride = 64) {
  f32x4 x[5];
#pragma unroll
  for (int g = 0; g < 5; ++g) x[g] = p[g * stride];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    s.floc[4 * g + 0] = x[g][0]; s.floc[4 * g + 1] = x[g][1]; s.floc[4 * g + 2] = x[g][2]; s.floc[4 * g + 3] = x[g][3];
  }
  s.rd[0] = x[3][0]; s.rd[1] = x[3][1]; s.rd[2] = x[3][2]; s.rd[3] = x[3][3];
  s.rgb[0] = x[4][0]; s.rgb[1] = x[4][1]; s.rgb[2] = x[4][2];
  s.mk = x[4][3] > 0.f ? 1.f : 0.f;
  s.ex = fabsf(x[4][3]);
}
typedef __attribute__((address_space(3))) f32x4* lds_f32x4_ptr;
// ... and the next REG_VIEWS views in registers (20 per lane and view: what the kernel's 256-register budget has left)
struct ViewRegs { f32x4 x[5]; };
template <int NRV> struct RegViews { ViewRegs v0, v1, v2; };  // (named members: an indexed array ends up in private memory)
template <int NRV>
__device__ __forceinline__ void slot_store(const Stage& g, int v, int lane, const ViewState& s, RegViews<NRV>& keep) {
  if (v < g.n_lds) { slot_store_at((lds_f32x4_ptr)(g.lds_v + v * VIEW_BYTES), s); return; }
  if (NRV > 0 && v == g.n_lds) { slot_store_at(&keep.v0.x[0], s, 1); return; }
  if (NRV > 1 && v == g.n_lds + 1) { slot_store_at(&keep.v1.x[0], s, 1); return; }
  if (NRV > 2 && v == g.n_lds + 2) { slot_store_at(&keep.v2.x[0], s, 1); return; }
  slot_store_at(reinterpret_cast<f32x4*>(g.slot + (int64_t)(v * SLOT_GROUPS) * 256 + lane * 4), s);
}
template <int NRV>
__device__ __forceinline__ void slot_load(const Stage& g, int v, int lane, ViewState& s, const RegViews<NRV>& keep) {
  if (v < g.n_lds) { slot_load_at((lds_f32x4_ptr)(g.lds_v + v * VIEW_BYTES), s); return; }
  if (NRV > 0 && v == g.n_lds) { slot_load_at(&keep.v0.x[0], s, 1); return; }
  if (NRV > 1 && v == g.n_lds + 1) { slot_load_at(&keep.v1.x[0], s, 1); return; }
  if (NRV > 2 && v == g.n_lds + 2) { slot_load_at(&keep.v2.x[0], s, 1); return; }
  slot_load_at(reinterpret_cast<const f32x4*>(g.slot + (int64_t)(v * SLOT_GROUPS) * 256 + lane * 4), s);
}

// ---- pass 1 of one source view, in two halves so that a caller may put other work between the fetches and their use ------
// per-kernel constants of the two pyramid levels this lane half fetches (half 0: rgb + levels 0, 1; half 1: levels 2, 3)
struct Geo {
  const float* mapA;
  const float* mapB;
  const float* mapC;   // the source images (half 0); half 1 re-reads its level-A taps (same addresses: L1 hits), result unused
  int HA, WA, HB, WB;
  float scA, scB, cWA, cHA, cWB, cHB, rWA, rHA, rWB, rHB;
  int h;
};
__device__ __forceinline__ Geo make_geo(const BlendArgs& a, int h) {
  Geo g;
  g.h = h;
  g.mapA = h ? a.feats[2] : a.feats[0];
  g.mapB = h ? a.feats[3] : a.feats[1];
  g.mapC = h ? a.feats[2] : a.imgs;
  g.HA = h ? a.hw[4] : a.hw[0]; g.WA = h ? a.hw[5] : a.hw[1];
  g.HB = h ? a.hw[6] : a.hw[2]; g.WB = h ? a.hw[7] : a.hw[3];
  g.scA = h ? 0.25f : 1.0f; g.scB = h ? 0.125f : 0.5f;
  g.cWA = (float)(g.WA - 1) / 2.0f; g.cHA = (float)(g.HA - 1) / 2.0f; g.cWB = (float)(g.WB - 1) / 2.0f; g.cHB = (float)(g.HB - 1) / 2.0f;
  g.rWA = 1.0f / g.cWA; g.rHA = 1.0f / g.cHA; g.rWB = 1.0f / g.cWB; g.rHB = 1.0f / g.cHB;
  return g;
}
struct Fetch {   // the twelve 16-byte taps of a view in flight + what turns them into values
  Tap4 qA, qB, qC;
  TapGeom gA, gB;
  bool ok;
};
// projections + texel fetches of source view `cam` for the point (px, py, pz); (ax, ay, az) = unit direction to the reference camera
__device__ __forceinline__ void pass1_issue(const BlendArgs& a, const Geo& geo, int cam, float px, float py, float pz, float ax, float ay,
                                        float az, ViewState& st, Fetch& q) {
  const int h = geo.h;
  const float* __restrict__ mapA = geo.mapA;
  const float* __restrict__ mapB = geo.mapB;
  const int HA = geo.HA, WA = geo.WA, HB = geo.HB, WB = geo.WB;
  const float scA = geo.scA, scB = geo.scB, cWA = geo.cWA, cHA = geo.cHA, cWB = geo.cWB, cHB = geo.cHB;
  const float rWA = geo.rWA, rHA = geo.rHA, rWB = geo.rWB, rHB = geo.rHB;
  Tap4& qA = q.qA; Tap4& qB = q.qB; Tap4& qC = q.qC;
  TapGeom& gA = q.gA; TapGeom& gB = q.gB;
  (void)h;
  // ray_diff (projector.py:485-498).  The direction to the source camera keeps the reference's arithmetic (correctly rounded
  // sqrt and divisions): its dot product with the reference direction goes through exp(|s| (dot - 1)) - min over views, a
  // difference of nearly equal numbers wherever the views see the point under similar angles, and 1-ulp changes of the
  // direction moved single colours by 1e-4.  The DIFFERENCE direction below only feeds the direction MLP (continuous,
  // well conditioned): v_sqrt / v_rcp (1 ulp each) instead of a correctly rounded sqrt and three divisions (~45 instructions)
  float bx = a.cpos[cam][0] - px, by = a.cpos[cam][1] - py, bz = a.cpos[cam][2] - pz;
  {
    const float nn = sqrtf(bx * bx + by * by + bz * bz) + 1e-6f;
    bx /= nn; by /= nn; bz /= nn;
  }
  const float ddx = ax - bx, ddy = ay - by, ddz = az - bz;
#ifdef SURF_BX_EXACT_RD
  {
    const float dn = fmaxf(sqrtf(ddx * ddx + ddy * ddy + ddz * ddz), 1e-6f);
    st.rd[0] = ddx / dn; st.rd[1] = ddy / dn; st.rd[2] = ddz / dn;
  }
#else
  {
    const float inv = __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(ddx * ddx + ddy * ddy + ddz * ddz), 1e-6f));
    st.rd[0] = ddx * inv; st.rd[1] = ddy * inv; st.rd[2] = ddz * inv;
  }
#endif
  st.rd[3] = ax * bx + ay * by + az * bz;
  // projection (projector.py:527-539); level l uses intrinsics rows 0,1 x 0.5^l = exact scaling of u,v.  u0, v0 decide the
  // view mask (comparisons): separate multiplies / adds and correctly rounded divisions, as the reference
  const float* M = a.w2c[cam];
  const float X = M[0] * px + M[1] * py + M[2] * pz + M[3];
  const float Y = M[4] * px + M[5] * py + M[6] * pz + M[7];
  const float Z = M[8] * px + M[9] * py + M[10] * pz + M[11];
  const float* K = a.K[cam];
  const float qx = K[0] * X + K[1] * Y + K[2] * Z;
  const float qy = K[3] * X + K[4] * Y + K[5] * Z;
  const float qz = K[6] * X + K[7] * Y + K[8] * Z;
  const float u0 = qx / qz, v0 = qy / qz;
  bool ok = qz > 0.f;
  {
    // normalise / un-normalise (grid_sample's rule).  The quotient u / ((W - 1) / 2) has to be the reference's to the bit:
    // the detour through [-1, 1] magnifies one ulp of it to W / 2 ulps of the texel coordinate (2e-5 texels at W = 800,
    // 1e-4 in single colours on a noisy image) - div_const gives the correctly rounded quotient in three instructions
    const float u = u0 * scA, vv = v0 * scA;
    ok = ok && (u >= 0.f) && (u < (float)WA) && (vv >= 0.f) && (vv < (float)HA);
    const float nx = div_const(u, cWA, rWA) - 1.0f, ny = div_const(vv, cHA, rHA) - 1.0f;
    const float gx = ((nx + 1.0f) * (float)WA - 1.0f) * 0.5f, gy = ((ny + 1.0f) * (float)HA - 1.0f) * 0.5f;
    gA = tap_geom(HA, WA, gx, gy);
    tap_issue(qA, mapA + (int64_t)cam * HA * WA * 4, gA);
    // half 1 re-reads its level-A taps instead of the image (same addresses: L1 hits), result unused
    tap_issue(qC, geo.mapC + (int64_t)cam * HA * WA * 4, gA);
  }
  {
    const float u = u0 * scB, vv = v0 * scB;
    ok = ok && (u >= 0.f) && (u < (float)WB) && (vv >= 0.f) && (vv < (float)HB);
    const float nx = div_const(u, cWB, rWB) - 1.0f, ny = div_const(vv, cHB, rHB) - 1.0f;
    const float gx = ((nx + 1.0f) * (float)WB - 1.0f) * 0.5f, gy = ((ny + 1.0f) * (float)HB - 1.0f) * 0.5f;
    gB = tap_geom(HB, WB, gx, gy);
    tap_issue(qB, mapB + (int64_t)cam * HB * WB * 4, gB);
  }
  q.ok = ok;
}
// direction feature (under the fetches), bilinear sums, local features, mask; returns the view's validity
template <class P>
__device__ __forceinline__ bool pass1_finish(const Ctx& c, const Geo& geo, float s_abs, ViewState& st, const Fetch& q) {
  typedef typename P::Frag Frag;
  const int h = geo.h;
  const Tap4& qA = q.qA; const Tap4& qB = q.qB; const Tap4& qC = q.qC;
  const TapGeom& gA = q.gA; const TapGeom& gB = q.gB;
  bool ok = q.ok;
  // direction feature ELU(L(ELU(L(ray_diff))))  4 -> 16 -> 19  (blending_network.py:72-74), under the fetches
  float d12[12];
  {
    const float bin[2] = {h ? st.rd[1] : st.rd[0], h ? st.rd[3] : st.rd[2]};
    Frag fb, f8;
    frags_from<P, 2, 1>(bin, &fb);
    f32x16 acc1 = bias_row<P>(c, B_RD0);
    mma_layer<P, L_RD0, 0, 1>(c, acc1, &fb);
    elu_make_frags<P, 8, 0>(c, acc1, &f8);
    f32x16 acc2 = bias_row<P>(c, B_RD2);
    mma_layer<P, L_RD2, 0, 1>(c, acc2, &f8);
    // rows of half 1 beyond its 8 channels carry zero weights and zero bias: elu(0) = 0
    elu_rows<12>(acc2, d12);
  }
  const f32x4 tA = tap_finish(qA, gA), tB = tap_finish(qB, gB);
  f32x4 tC = tap_finish(qC, gA);
  if (h != 0) tC = f32x4{0.f, 0.f, 0.f, 0.f};
  ok = ok && (__shfl_xor((int)ok, 32) != 0);  // AND over all four levels
  st.mk = ok ? 1.f : 0.f;
  st.rgb[0] = tC[0]; st.rgb[1] = tC[1]; st.rgb[2] = tC[2];
  // local channel order: half 0 = [rgb, F0, F1], half 1 = [F2, F3, 0, 0, 0]
  float g[12];
  if (h == 0) {
    g[0] = tC[0]; g[1] = tC[1]; g[2] = tC[2];
    g[3] = tA[0]; g[4] = tA[1]; g[5] = tA[2]; g[6] = tA[3];
    g[7] = tB[0]; g[8] = tB[1]; g[9] = tB[2]; g[10] = tB[3];
  } else {
    g[0] = tA[0]; g[1] = tA[1]; g[2] = tA[2]; g[3] = tA[3];
    g[4] = tB[0]; g[5] = tB[1]; g[6] = tB[2]; g[7] = tB[3];
    g[8] = g[9] = g[10] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < 11; ++r) st.floc[r] = g[r] + d12[r];
  st.floc[11] = 0.f;
  st.ex = expf(s_abs * (st.rd[3] - 1.0f));
  return ok;
}

template <class P>
__global__ __launch_bounds__(WPB * 64, WPB / 4) void blend_split_kernel(BlendArgs a) {
  typedef typename P::Frag Frag;
  __shared__ __attribute__((aligned(16))) char lds[lds_bytes<P>() + WPB * lds_views<P>() * VIEW_BYTES];
  static_assert(lds_bytes<P>() % 16 == 0 && lds_bytes<P>() + WPB * lds_views<P>() * VIEW_BYTES <= 160 * 1024, "LDS");
  // ---- the weight image: global -> LDS once per workgroup --------------------------------------------------------
  for (int o = threadIdx.x * 16; o < lds_bytes<P>(); o += WPB * 64 * 16)
    *reinterpret_cast<u32x4*>(lds + o) = *reinterpret_cast<const u32x4*>(a.w + o);
  __syncthreads();

  Ctx c0;
  const int lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  c0.lds = lds;
  c0.lane16 = lane * 16;
  c0.h64 = h * 64;
  c0.I0 = ident_frag(lane, 0);
  c0.I1 = ident_frag(lane, 1);
  const int NS = a.nv - 1;
  const int64_t wave_id = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * WPB;
  const int64_t n_pts = a.n_dev ? (int64_t)*a.n_dev : a.n;
  const int64_t n_tiles = (n_pts + TILE - 1) / TILE;
  Stage slot;
  slot.slot = a.scratch + wave_id * slot_floats(NS);
  slot.lds_v = lds + lds_bytes<P>() + (threadIdx.x >> 6) * (lds_views<P>() * VIEW_BYTES) + lane * 16;
  slot.n_lds = lds_views<P>();
  const float* scal = reinterpret_cast<const float*>(lds + scal_off<P::BB>());
  const float s_abs = scal[0], b_vis = scal[1], b_vis2 = scal[2], b_rgb4 = scal[3];

  const Geo geo = make_geo(a, h);

#ifdef SURF_BLEND_TIMING
  unsigned long long tprev = __builtin_readcyclecounter(), tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = tprev, rt_begin = __builtin_amdgcn_s_memrealtime();
#endif
  for (int64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
    SURF_BT(0);
    const int64_t slot_i = tile * TILE + j;
    const int64_t sc = slot_i < n_pts ? slot_i : n_pts - 1;
    const int64_t i = a.idx ? (int64_t)a.idx[sc] : sc;
    const bool active = (slot_i < n_pts) && (!a.mask || a.mask[i] != 0);
    if (__ballot(active) == 0ull) continue;
    const float px = a.pts[i * 3 + 0], py = a.pts[i * 3 + 1], pz = a.pts[i * 3 + 2];

    float ax = a.cpos[0][0] - px, ay = a.cpos[0][1] - py, az = a.cpos[0][2] - pz;
    {
      const float nn = sqrtf(ax * ax + ay * ay + az * az) + 1e-6f;
      ax /= nn; ay /= nn; az /= nn;
    }
    int nvalid = 0;
    float emin = INFINITY;
    RegViews<P::REG_VIEWS> keep;
    // ------------------------------ pass 1: per view, projections + texel fetches + direction feature ----------------
#pragma unroll 1
    for (int v = 0; v < NS; ++v) {
      const Ctx c = opaque(c0);
      ViewState st;
      Fetch q;
      pass1_issue(a, geo, v + 1, px, py, pz, ax, ay, az, st, q);
      nvalid += pass1_finish<P>(c, geo, s_abs, st, q) ? 1 : 0;
      emin = fminf(emin, st.ex);
      slot_store(slot, v, lane, st, keep);
    }
    if (a.n_valid && active && h == 0) a.n_valid[i] = (uint8_t)nvalid;
    SURF_BT(1);

    // ------------------------------ pooling weights, weighted mean / variance (:76-86) ----------------------
    // w_v = (ex_v - min_v ex) mk_v / (sum + 1e-8); mean = sum_v w_v f_v; var = sum_v w_v (f_v - mean)^2: two sweeps
    // over the staged views (each lane reads back exactly what it wrote)
    float wsum = 0.f;
    float mv[24];
#pragma unroll
    for (int ch = 0; ch < 24; ++ch) mv[ch] = 0.f;
#pragma unroll 1
    for (int v = 0; v < NS; ++v) {
      ViewState st;
      slot_load(slot, v, lane, st, keep);
      const float w = (st.ex - emin) * st.mk;
      wsum += w;
#pragma unroll
      for (int ch = 0; ch < 12; ++ch) mv[ch] += st.floc[ch] * w;
    }
    const float winv = 1.0f / (wsum + 1e-8f);
#pragma unroll
    for (int ch = 0; ch < 12; ++ch) mv[ch] *= winv;
#pragma unroll 1
    for (int v = 0; v < NS; ++v) {
      ViewState st;
      slot_load(slot, v, lane, st, keep);
      const float w = (st.ex - emin) * st.mk * winv;
#pragma unroll
      for (int ch = 0; ch < 12; ++ch) { const float d = st.floc[ch] - mv[ch]; mv[12 + ch] += w * (d * d); }
    }
    SURF_BT(2);
    // view-independent part of base_fc.0: [mean(12) | var(12)] -> 64 (two tiles), initialised with the bias
    f32x16 G0a, G0b;
    {
      const Ctx c = opaque(c0);
      G0a = bias_row<P>(c, B_B0_T0);
      G0b = bias_row<P>(c, B_B0_T1);
      Frag fm[3];
      make_frags<P, 24, 3, 1>(c, mv, fm);
      mma_layer<P, L_B0S, 0, 3>(c, G0a, fm);
      mma_layer<P, L_B0S, 1, 3>(c, G0b, fm);
    }

    // ------------------------------ pass 2: per-view chain, online softmax over views (:88-116) -------------
    // Written for short live ranges (two wavefronts per SIMD = 256 registers): every activation tile is converted to
    // B fragments pair by pair as it leaves the accumulator and consumed by the next layer's k-steps at once.
    float Mx = -INFINITY, Zs = 0.f, o_r = 0.f, o_g = 0.f, o_b = 0.f;
    SURF_BT(3);
#pragma unroll 1
    for (int v = 0; v < NS; ++v) {
      const Ctx c = opaque(c0);
      ViewState st;
      slot_load(slot, v, lane, st, keep);
      const float wv = (st.ex - emin) * st.mk * winv;
      // base_fc.0 (view part) : 57 -> 64, on top of the view-independent part
      f32x16 a0 = G0a, a1 = G0b;
      {
        Frag fl[2];
        make_frags<P, 12, 2, 2>(c, st.floc, fl);
        mma_layer<P, L_B0V, 0, 2>(c, a0, fl);
        mma_layer<P, L_B0V, 1, 2>(c, a1, fl);
      }
      // ELU, base_fc.2 : 64 -> 32, two k-steps per input tile
      f32x16 ax2 = bias_row<P>(c, B_B2);
      {
        Frag hf[2];
        elu_make_frags<P, 16, 3>(c, a0, hf);
        mma_blk<P, L_B2, 0, 0>(c, ax2, hf[0]);
        mma_blk<P, L_B2, 1, 0>(c, ax2, hf[1]);
        SURF_PHASE();
        elu_make_frags<P, 16, 4>(c, a1, hf);
        mma_blk<P, L_B2, 2, 0>(c, ax2, hf[0]);
        mma_blk<P, L_B2, 3, 0>(c, ax2, hf[1]);
        SURF_PHASE();
      }
      float x[16];
      elu_rows<16>(ax2, x);
      SURF_BT(4);
      // vis_fc: 32 -> 32 (ELU) -> 33 (ELU)
      float vis;
      {
        f32x16 at;
#pragma unroll
        for (int r = 0; r < 16; ++r) at[r] = 0.f;
        {
          Frag xf[2];
          make_frags<P, 16, 2, 5>(c, x, xf);
          mma_layer<P, L_V0, 0, 2>(c, at, xf);
        }
        const f32x16 bt = bias_row<P>(c, B_V0);
        f32x16 ar = bias_row<P>(c, B_V2);
        float vraw = 0.f;
        {
          float t16[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) t16[r] = elu_t(fmaf(at[r], wv, bt[r]));
          const f32x16 dvis = dot_row<P>(c, D_VIS);
#pragma unroll
          for (int r = 0; r < 16; ++r) vraw = fmaf(dvis[r], t16[r], vraw);
          Frag tf[2];
          make_frags<P, 16, 2, 6>(c, t16, tf);
          mma_layer<P, L_V2, 0, 2>(c, ar, tf);
        }
        vraw += __shfl_xor(vraw, 32);
        vis = sigm(elu_x(vraw + b_vis)) * st.mk;
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] += elu_t(ar[r]);
      }
      // vis_fc2: 32 -> 32 (ELU) -> 1 (sigmoid).  Its input x vis and the x part of rgb_fc's input share ONE operand
      // split: a Linear commutes with the per-sample scale, W (x vis) + b = vis (W x) + b, so the matrix product runs on
      // the fragments of x itself and the scale is applied to the accumulator (16 FMAs instead of a second split).
      SURF_BT(5);
      Frag rf[3];
      make_frags<P, 16, 2, 7>(c, x, rf);
      float vis2;
      {
        f32x16 aw;
#pragma unroll
        for (int r = 0; r < 16; ++r) aw[r] = 0.f;
        mma_layer<P, L_W0, 0, 2>(c, aw, rf);
        const f32x16 bw = bias_row<P>(c, B_W0);
        const f32x16 dvis2 = dot_row<P>(c, D_VIS2);
        float v2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v2 = fmaf(dvis2[r], elu_t(fmaf(aw[r], vis, bw[r])), v2);
        v2 += __shfl_xor(v2, 32);
        vis2 = sigm(v2 + b_vis2) * st.mk;
      }
      SURF_BT(6);
      // rgb_fc: [x(32), vis, ray_diff(4)] = 37 -> 16 (ELU) -> 8 (ELU) -> 1
      float rr;
      {
        f32x16 a16 = bias_row<P>(c, B_R0);
        {
          const float extra[3] = {h ? st.rd[0] : vis2, h ? st.rd[2] : st.rd[1], h ? 0.f : st.rd[3]};
          make_frags<P, 3, 1, 8>(c, extra, rf + 2);
          mma_layer<P, L_R0, 0, 3>(c, a16, rf);
        }
        f32x16 a8 = bias_row<P>(c, B_R2);
        {
          Frag f8;
          elu_make_frags<P, 8, 9>(c, a16, &f8);
          mma_layer<P, L_R2, 0, 1>(c, a8, &f8);
        }
        const f32x16 drgb4 = dot_row<P>(c, D_RGB4);
        rr = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) rr = fmaf(drgb4[r], elu_t(a8[r]), rr);
        rr += __shfl_xor(rr, 32);
        rr += b_rgb4;
      }
      if (st.mk == 0.f) rr = -1e9f;
      const float Mn = fmaxf(Mx, rr);
      const float scl = expf(Mx - Mn);  // exp(-inf) = 0 on the first view
      const float e = expf(rr - Mn);
      Zs = Zs * scl + e;
      o_r = o_r * scl + e * st.rgb[0];
      o_g = o_g * scl + e * st.rgb[1];
      o_b = o_b * scl + e * st.rgb[2];
      Mx = Mn;
      SURF_BT(7);
    }
    if (active && h == 0) {
      a.color[i * 3 + 0] = o_r / Zs;
      a.color[i * 3 + 1] = o_g / Zs;
      a.color[i * 3 + 2] = o_b / Zs;
    }
  }
#ifdef SURF_BLEND_TIMING
  if (threadIdx.x == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&g_bphase[k], tacc[k]);
  if (threadIdx.x == 0 && blockIdx.x < 256) {
    g_bwg[blockIdx.x][0] = rt_begin;
    g_bwg[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    g_bphase[8] = __builtin_readcyclecounter() - t_begin;
    g_bphase[9] = __builtin_amdgcn_s_memrealtime() - rt_begin;
  }
#endif
}

int grid_blocks(int64_t n) {
  const int64_t tiles = (n + TILE - 1) / TILE;
  const int64_t blocks = (tiles + WPB - 1) / WPB;
  return (int)(blocks < 256 ? blocks : 256);  // one 8-wave workgroup per CU, persistent over tiles
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
  _Float16 hh = (_Float16)v;  // round to nearest even
  uint16_t b;
  memcpy(&b, &hh, 2);
  return b;
}
inline float f16_to_f(uint16_t b) {
  _Float16 hh;
  memcpy(&hh, &b, 2);
  return (float)hh;
}
template <class P> void split_host(float v, uint16_t* p);
template <> void split_host<BPolBf3>(float v, uint16_t* p) {
  p[0] = bf16_rne(v);
  float r = v - bf16_to_f(p[0]);
  p[1] = bf16_rne(r);
  r = r - bf16_to_f(p[1]);
  p[2] = bf16_rne(r);
}
template <> void split_host<BPolH2>(float v, uint16_t* p) {
  p[0] = f16_bits(v);
  p[1] = f16_bits(v - f16_to_f(p[0]));
}
// element (lane, k-slot i) of block `blk` of the image
template <class P> void put_weight(unsigned char* out, int blk, int lane, int i, float v) {
  uint16_t p[P::NP];
  split_host<P>(v, p);
  for (int pc = 0; pc < P::NP; ++pc) reinterpret_cast<uint16_t*>(out + (int64_t)blk * P::BB + pc * 1024 + lane * 16)[i] = p[pc];
}
template <> void put_weight<BPolF32>(unsigned char* out, int blk, int lane, int i, float v) {
  reinterpret_cast<float*>(out + (int64_t)blk * BPolF32::BB + lane * 32)[i] = v;
}

// feature held by accumulator register r of half h in 32-row tile tt (= k-slot (s, i) with r = 8 s + i)
inline int hk(int tt, int r, int h) { return 32 * tt + (r & 3) + 8 * (r >> 2) + 4 * h; }
// local channel index (register r of half h) -> channel of the 19-vector, -1 = pad
inline int loc_ch(int r, int h) { return h == 0 ? (r < 11 ? r : -1) : (r < 8 ? 11 + r : -1); }

// A blocks of one layer: row_of(tile, rho) = weight row (or -1), col_of(m, h) = weight column of k-slot m = 8 ks + i of
// lane half h (or -1)
template <class P, class RowF, class ColF>
void pack_layer(unsigned char* out, int L, const float* W, int ldw, RowF row_of, ColF col_of) {
  for (int ks = 0; ks < NKS[L]; ++ks)
    for (int t = 0; t < NT[L]; ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 8; ++i) {
          const int h = lane >> 5, rho = lane & 31;
          const int row = row_of(t, rho), col = col_of(8 * ks + i, h);
          const float v = (row >= 0 && col >= 0) ? (float)((double)W[row * ldw + col] * (double)LOG2E) : 0.f;
          put_weight<P>(out, blk_off(L) + ks * NT[L] + t, lane, i, v);
        }
}
template <class RowF>
void pack_rows(float* dst, const float* b, RowF feat_of, float scale) {  // [h][16] <- b[feat_of(r,h)]
  for (int h = 0; h < 2; ++h)
    for (int r = 0; r < 16; ++r) {
      const int f = feat_of(r, h);
      dst[h * 16 + r] = f >= 0 ? (float)((double)b[f] * (double)scale) : 0.f;
    }
}

template <class P>
int pack_weights(const float* raw, unsigned char* out) {
  if (!raw || !out) return SURF_E_ARG;
  memset(out, 0, image_bytes<P::BB>());
  float* bias = reinterpret_cast<float*>(out + bias_off<P::BB>());
  float* dots = reinterpret_cast<float*>(out + dot_off<P::BB>());
  float* scal = reinterpret_cast<float*>(out + scal_off<P::BB>());
  auto nat_row = [](int lim) { return [lim](int t, int rho) { const int f = 32 * t + rho; return f < lim ? f : -1; }; };
  // k-slot m of half h -> natural feature of the producing accumulator tile(s)
  auto nat_col = [](int lim) { return [lim](int m, int h) { const int f = hk(m / 16, m % 16, h); return f < lim ? f : -1; }; };
  auto natf = [](int lim) { return [lim](int r, int h) { const int f = hk(0, r, h); return f < lim ? f : -1; }; };
  // row map of the direction-feature output: D row rho -> (r, h_row) -> local channel
  auto dir_row = [](int t, int rho) { const int hr = (rho >> 2) & 1, r = (rho & 3) | ((rho >> 3) << 2); return loc_ch(r, hr); };

  // ray_dir_fc.0: 4 -> 16; k-slots 0, 1 of half h = (rd0 | rd1), (rd2 | rd3)
  pack_layer<P>(out, L_RD0, raw + R_RD0_W, 4, nat_row(16), [](int m, int h) { return m < 2 ? 2 * m + h : -1; });
  pack_rows(bias + B_RD0 * 32, raw + R_RD0_B, natf(16), LOG2E);
  // ray_dir_fc.2: 16 -> 19, output rows in local-channel order
  pack_layer<P>(out, L_RD2, raw + R_RD2_W, 16, dir_row, nat_col(16));
  pack_rows(bias + B_RD2 * 32, raw + R_RD2_B, [](int r, int h) { return loc_ch(r, h); }, LOG2E);
  // base_fc.0 shared part: [mean(12 slots) | var(12 slots)] -> 64
  pack_layer<P>(out, L_B0S, raw + R_B0_W, 57, nat_row(64), [](int m, int h) {
    const int grp = m / 12, s = m % 12;
    const int ch = (grp < 2 && s < 11) ? loc_ch(s, h) : -1;
    return ch >= 0 ? grp * DF + ch : -1;
  });
  pack_rows(bias + B_B0_T0 * 32, raw + R_B0_B, [](int r, int h) { return hk(0, r, h); }, LOG2E);
  pack_rows(bias + B_B0_T1 * 32, raw + R_B0_B, [](int r, int h) { return hk(1, r, h); }, LOG2E);
  // base_fc.0 view part: f (12 slots) -> 64
  pack_layer<P>(out, L_B0V, raw + R_B0_W, 57, nat_row(64), [](int m, int h) {
    const int ch = m < 11 ? loc_ch(m, h) : -1;
    return ch >= 0 ? 2 * DF + ch : -1;
  });
  // base_fc.2: 64 -> 32
  pack_layer<P>(out, L_B2, raw + R_B2_W, 64, nat_row(32), nat_col(64));
  pack_rows(bias + B_B2 * 32, raw + R_B2_B, natf(32), LOG2E);
  // vis_fc.0: 32 -> 32 ; vis_fc.2 rows 0..31 (x_res) as MFMA, row 32 (vis) as a per-lane dot
  pack_layer<P>(out, L_V0, raw + R_V0_W, 32, nat_row(32), nat_col(32));
  pack_rows(bias + B_V0 * 32, raw + R_V0_B, natf(32), LOG2E);
  pack_layer<P>(out, L_V2, raw + R_V2_W, 32, nat_row(32), nat_col(32));
  pack_rows(bias + B_V2 * 32, raw + R_V2_B, natf(32), LOG2E);
  pack_rows(dots + D_VIS * 32, raw + R_V2_W + 32 * 32, natf(32), 1.0f);
  // vis_fc2.0: 32 -> 32 ; vis_fc2.2: 32 -> 1 as a dot
  pack_layer<P>(out, L_W0, raw + R_W0_W, 32, nat_row(32), nat_col(32));
  pack_rows(bias + B_W0 * 32, raw + R_W0_B, natf(32), LOG2E);
  pack_rows(dots + D_VIS2 * 32, raw + R_W2_W, natf(32), 1.0f);
  // rgb_fc.0: [x(32) | vis | rd(4)] -> 16 ; k-step 2: slots 0..2 = (vis2 | rd0), (rd1 | rd2), (rd3 | -)
  pack_layer<P>(out, L_R0, raw + R_R0_W, 37, nat_row(16), [](int m, int h) {
    if (m < 16) return hk(0, m, h);
    if (m == 16) return h ? 33 : 32;
    if (m == 17) return h ? 35 : 34;
    if (m == 18) return h ? -1 : 36;
    return -1;
  });
  pack_rows(bias + B_R0 * 32, raw + R_R0_B, natf(16), LOG2E);
  // rgb_fc.2: 16 -> 8 ; rgb_fc.4: 8 -> 1 as a dot over registers 0..3 (feature 4 h + r)
  pack_layer<P>(out, L_R2, raw + R_R2_W, 16, nat_row(8), nat_col(16));
  pack_rows(bias + B_R2 * 32, raw + R_R2_B, natf(8), LOG2E);
  pack_rows(dots + D_RGB4 * 32, raw + R_R4_W, [](int r, int h) { return r < 4 ? 4 * h + r : -1; }, 1.0f);
  scal[0] = fabsf(raw[R_S]);
  scal[1] = raw[R_V2_B + 32];
  scal[2] = raw[R_W2_B];
  scal[3] = raw[R_R4_B];
  return 0;
}

template <class P>
int launch(const BlendArgs& a, hipStream_t st) {
  dim3 grid(grid_blocks(a.n)), block(WPB * 64);
  hipLaunchKernelGGL((blend_split_kernel<P>), grid, block, 0, st, a);
  return surf_check_launch();
}

}  // namespace

#ifdef SURF_BLEND_TIMING
extern "C" int surf_debug_blend_wg(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwg), sizeof(unsigned long long) * 512) == hipSuccess ? 0 : 100;
}
extern "C" int surf_debug_blend_phases(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bphase), sizeof(unsigned long long) * 10) != hipSuccess) return 100;
  if (reset) {
    unsigned long long z[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bphase), z, sizeof(z)) != hipSuccess) return 100;
  }
  return 0;
}
#endif

extern "C" int64_t surf_blend_split_packed_bytes(int precision) {
  if (precision == BPolBf3::ID) return image_bytes<BPolBf3::BB>();
  if (precision == BPolH2::ID) return image_bytes<BPolH2::BB>();
  if (precision == BPolF32::ID) return image_bytes<BPolF32::BB>();
  return SURF_E_ARG;
}

extern "C" int surf_blend_pack_weights_split(const float* h_raw, unsigned char* h_packed, int precision) {
  if (precision == BPolBf3::ID) return pack_weights<BPolBf3>(h_raw, h_packed);
  if (precision == BPolH2::ID) return pack_weights<BPolH2>(h_raw, h_packed);
  if (precision == BPolF32::ID) return pack_weights<BPolF32>(h_raw, h_packed);
  return SURF_E_ARG;
}

extern "C" int64_t surf_blend_split_scratch_bytes(int64_t n_points, int nv) {
  if (n_points <= 0 || nv < 2 || nv > SURF_MAX_VIEWS) return 0;
  return (int64_t)grid_blocks(n_points) * WPB * slot_floats(nv - 1) * (int64_t)sizeof(float);
}

static int blend_split_impl(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n, const int32_t* d_n,
                            const float* const* h_feats, const int* h_hw, int n_level, const float* imgs, int nv,
                            const float* h_intrs, const float* h_w2c, const float* h_c2w, const void* blend_w,
                            int precision, float* color, uint8_t* n_valid, void* scratch, void* stream) {
  if (!pts || !h_feats || !h_hw || !imgs || !h_intrs || !h_w2c || !h_c2w || !blend_w || !color || !scratch) return SURF_E_ARG;
  if (n <= 0 || nv < 2) return SURF_E_ARG;
  if (n_level != 4 || nv > SURF_MAX_VIEWS) return SURF_E_LIMIT;  // d_feature = 16 = 4 levels x 4 channels
  if (precision != BPolBf3::ID && precision != BPolH2::ID && precision != BPolF32::ID) return SURF_E_ARG;
  BlendArgs a;
  a.pts = pts; a.mask = mask; a.idx = idx; a.n = n; a.n_dev = d_n; a.imgs = imgs; a.w = (const unsigned char*)blend_w; a.color = color;
  a.n_valid = n_valid;
  a.scratch = (float*)scratch;
  a.nv = nv;
  for (int l = 0; l < 4; ++l) {
    if (!h_feats[l]) return SURF_E_ARG;
    a.feats[l] = h_feats[l];
    a.hw[2 * l] = h_hw[2 * l];
    a.hw[2 * l + 1] = h_hw[2 * l + 1];
    // pass 1 normalises texel coordinates by the reciprocal of (W - 1) / 2 (div_const): a one-texel-wide or -high pyramid
    // level makes that infinite and the taps NaN (IEEE division gave inf and a zero weight instead).  No shipped conf has one.
    if (h_hw[2 * l] < 2 || h_hw[2 * l + 1] < 2) return SURF_E_LIMIT;
  }
  for (int v = 0; v < SURF_MAX_VIEWS; ++v) {
    const int s = v < nv ? v : 0;
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) a.K[v][r * 3 + cc] = h_intrs[s * 16 + r * 4 + cc];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 4; ++cc) a.w2c[v][r * 4 + cc] = h_w2c[s * 16 + r * 4 + cc];
    for (int r = 0; r < 3; ++r) a.cpos[v][r] = h_c2w[s * 16 + r * 4 + 3];
  }
  if (precision == BPolBf3::ID) return launch<BPolBf3>(a, (hipStream_t)stream);
  if (precision == BPolH2::ID) return launch<BPolH2>(a, (hipStream_t)stream);
  return launch<BPolF32>(a, (hipStream_t)stream);
}

extern "C" int surf_blend_split(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n,
                                const float* const* h_feats, const int* h_hw, int n_level, const float* imgs, int nv,
                                const float* h_intrs, const float* h_w2c, const float* h_c2w, const void* blend_w,
                                int precision, float* color, uint8_t* n_valid, void* scratch, void* stream) {
  return blend_split_impl(pts, mask, idx, n, nullptr, h_feats, h_hw, n_level, imgs, nv, h_intrs, h_w2c, h_c2w, blend_w, precision,
                          color, n_valid, scratch, stream);
}

// The same with the number of idx entries read from DEVICE memory (d_n[0] <= n_capacity): no host round trip between the
// compaction and this launch.
extern "C" int surf_blend_split_dn(const float* pts, const int32_t* idx, int64_t n_capacity, const int32_t* d_n,
                                   const float* const* h_feats, const int* h_hw, int n_level, const float* imgs, int nv,
                                   const float* h_intrs, const float* h_w2c, const float* h_c2w, const void* blend_w,
                                   int precision, float* color, uint8_t* n_valid, void* scratch, void* stream) {
  if (!idx || !d_n) return SURF_E_ARG;
  return blend_split_impl(pts, nullptr, idx, n_capacity, d_n, h_feats, h_hw, n_level, imgs, nv, h_intrs, h_w2c, h_c2w, blend_w,
                          precision, color, n_valid, scratch, stream);
}
