// Shared pieces of the split-operand SDF kernels (sdf_mlp_split.hip, sdf_mlp_v2.hip): network shape and weight-stream
// tables, precision policies, scratch layout, softplus, sparse gather and positional encoding.  Included once per
// translation unit (everything lives in an anonymous namespace).
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {


constexpr int HID = 128, NE = 27, H2 = 101, TILE = 32;
constexpr int BWD_NT[6] = {1, 5, 5, 6, 5, 5};
constexpr int WPB = 4;
constexpr int MAX_KS = 12;

// ---- chunk stream ------------------------------------------------------------------------------------------------
// forward chunk (l, t): k-steps = [e s=0,1 (l = 0, 3)][phi s=0,1 (l >= 1)][hidden (tt, s) ...]: the hidden fragments of
// the previous layer's last tile are still being converted during the first k-steps of tile 0
constexpr int fwd_nh(int l) { return l == 0 ? 0 : (l == 3 ? 7 : 8); }
constexpr int fwd_ne(int l) { return (l == 0 || l == 3) ? 2 : 0; }
constexpr int fwd_np(int l) { return l == 0 ? 0 : 2; }
constexpr int fwd_nl(int l) { return fwd_ne(l) + fwd_np(l); }
constexpr int fwd_ks(int l) { return fwd_nh(l) + fwd_nl(l); }
constexpr int bwd_ks(int l) { return l == 2 ? 7 : 8; }
constexpr int N_FWD_CHUNKS = 24;
constexpr int n_bwd_chunks() { int n = 0; for (int l = 0; l < 6; ++l) n += BWD_NT[l]; return n; }
constexpr int N_BWD_CHUNKS = n_bwd_chunks();
constexpr int N_CHUNKS = N_FWD_CHUNKS + N_BWD_CHUNKS;

constexpr int MAX_PAD = 4;  // empty chunks that round the gradient stream up to a multiple of the ring length
// Round 5, "R0": the reverse sweep RECOMPUTES layer 0's pre-activations (= its softplus exponent arguments) instead of reading
// them back from the scratch slot: one extra chunk of 8 k-steps between the reverse layers 2 and 1 whose LDS image is the four
// forward chunks of layer 0 - they are contiguous at the start of the packed stream, so the chunk is only a second table entry
// with offset 0 (no packer change) - multiplied by the positional-encoding fragments (kept live) into four accumulators.
// 48 MFMAs per tile (+1.9 %) for 16 of the 40 KB a wavefront moves through its scratch slot per tile.
constexpr int R0_KS = 8;
struct ChunkTable {
  int off[N_CHUNKS + 1 + MAX_PAD + 1];  // byte offset into the packed stream
  int ks[N_CHUNKS + 1 + MAX_PAD + 1];
  int n;      // chunks of the gradient stream (N_CHUNKS, + 1 with R0)
  int total;  // bytes of the packed stream
};
constexpr ChunkTable make_chunks(int np, bool r0) {  // one k-step = np pieces x 64 lanes x 16 B
  ChunkTable c{};
  int n = 0, o = 0;
  for (int l = 0; l < 6; ++l)
    for (int t = 0; t < 4; ++t) { c.off[n] = o; c.ks[n] = fwd_ks(l); o += fwd_ks(l) * np * 1024; ++n; }
  for (int l = 5; l >= 0; --l) {
    if (r0 && l == 1) { c.off[n] = 0; c.ks[n] = R0_KS; ++n; }
    for (int t = 0; t < BWD_NT[l]; ++t) { c.off[n] = o; c.ks[n] = bwd_ks(l); o += bwd_ks(l) * np * 1024; ++n; }
  }
  c.n = n;
  c.total = o;
  for (; n <= N_CHUNKS + 1 + MAX_PAD; ++n) { c.off[n] = o; c.ks[n] = 0; }
  return c;
}
static_assert(fwd_ks(0) * 4 == R0_KS, "the recompute chunk is the four forward chunks of layer 0");
constexpr int fwd_chunk(int l, int t) { return l * 4 + t; }
constexpr int bwd_chunk(int l, int t, bool r0 = false) {
  int n = N_FWD_CHUNKS;
  for (int i = 5; i > l; --i) n += BWD_NT[i];
  return n + t + ((r0 && l <= 1) ? 1 : 0);
}
constexpr int r0_chunk() { return bwd_chunk(1, 0, false); }  // (its index when present: in front of reverse layer 1)
// fp32 tail of the packed buffer (floats): W6[0] in lane order, b6
constexpr int TAIL_W6H = 0;               // [h][64]
constexpr int TAIL_W6P = TAIL_W6H + 128;  // [h][16]
constexpr int TAIL_B6 = TAIL_W6P + 32;
constexpr int TAIL_FLOATS = TAIL_B6 + 4;

// per-wave scratch slot (floats): softplus exponent arguments of layers 0..4 ("exponent slices") + feature Jacobian
constexpr int SCR_S = 5 * 16 * 64 * 4;
constexpr int SCR_J = 12 * 64 * 4;
constexpr int SCR_SLOT = SCR_S + SCR_J;

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NP>
struct FragT { u32x4 p[NP]; };  // B-operand fragments of a 16-wide k-step: NP pieces x 4 dwords (8 x 16 bit)

// End of a slot: its results exist HERE.  VALU arithmetic is a pure value to the compiler, and instruction selection
// linearises pure values wherever it likes between operands and use (the scheduling barriers only bind the machine scheduler
// that runs afterwards): without the pin the tail of a slot drifts into the next gap's slot.
#ifndef SURF_SDF_PINS
#define SURF_SDF_PINS 2
#endif
#if SURF_SDF_PINS & 1
__device__ __forceinline__ void slot_pin(f32x2& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void slot_pin(f32x2& a, f32x2& b) { asm volatile("" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void slot_pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void slot_pin(uint32_t& v) { asm volatile("" : "+v"(v)); }
#else
__device__ __forceinline__ void slot_pin(f32x2&) {}
__device__ __forceinline__ void slot_pin(f32x2&, f32x2&) {}
__device__ __forceinline__ void slot_pin(float&) {}
__device__ __forceinline__ void slot_pin(uint32_t&) {}
#endif

// ---- precision policies ----------------------------------------------------------------------------------------------
#ifndef SURF_X_NOGATHER  // timing experiments only (wrong results)
#define SURF_X_NOGATHER 0
#endif
#ifndef SURF_X_NOSCRATCH
#define SURF_X_NOSCRATCH 0
#endif
#ifndef SURF_X_NOMMA
#define SURF_X_NOMMA 0
#endif
#ifndef SURF_X_NOLDS
#define SURF_X_NOLDS 0
#endif
#ifndef SURF_X_NOSOFTPLUS
#define SURF_X_NOSOFTPLUS 0
#endif
#ifndef SURF_X_NODMA  // no LDS-DMA inside the chunks (the ring keeps whatever the prologue loaded)
#define SURF_X_NODMA 0
#endif
#ifndef SURF_X_NOSPLIT  // activations "split" by plain bit copies (no conversion arithmetic)
#define SURF_X_NOSPLIT 0
#endif

// LDS ring length of the gradient kernels (chunks in flight + the one being read), per policy.  Measured (round 2, f16x2,
// half image): a fourth slot (one more chunk of slack before the counted vmcnt) is SLOWER, 47.4-48.6 vs 46.3-46.5 ms, and
// costs ~30-40 registers (bf16x3 then spills): the barrier wait is not store-acknowledge latency.  3 is the default.
#ifndef SURF_SDF_NSLOT_BF3
#define SURF_SDF_NSLOT_BF3 3
#endif
#ifndef SURF_SDF_NSLOT_H2
#define SURF_SDF_NSLOT_H2 3
#endif

#ifndef SURF_SDF_LDS_SLICES_BF3
#define SURF_SDF_LDS_SLICES_BF3 3
#endif
#ifndef SURF_SDF_LDS_SLICES_H2
#define SURF_SDF_LDS_SLICES_H2 5
#endif
// ... and in REGISTERS (16 per slice and lane) after those: what the gradient kernels' register files have left where the
// slices are live (the end of the forward sweep and the first layers of the reverse one are not where the pressure peaks).
// bf16x3: 10 (500 of 512 registers; 11 fit exactly or spill 28 bytes depending on the surrounding code, 12 spill - build.sh
// refuses such a build): 13 of the 20 slices never leave the CU.
// f16x2: 15 (504 registers): all 20 stay on the CU and only the feature Jacobian still makes the round trip.
// Measured (round 3, half image, same box): bf16x3 61.5 -> 58.6 ms with 10, f16x2 42.9 -> 37.7 ms with 15.
// (round 5, with R0: 9 - the positional-encoding fragments stay live through the reverse sweep, 24 registers; layers 4, 3, 2 on the
// CU, layer 1 through the scratch slot, layer 0 recomputed)
#ifndef SURF_SDF_REG_SLICES_BF3
#define SURF_SDF_REG_SLICES_BF3 9
#endif
#ifndef SURF_SDF_REG_SLICES_H2
#define SURF_SDF_REG_SLICES_H2 15
#endif

struct PolBf3 {
  // NA: accumulator chains (two independent ones measured no faster); PF: k-steps of LDS read-ahead (2, 3: no faster)
#ifndef SURF_SDF_NA_BF3
#define SURF_SDF_NA_BF3 1
#endif
  static constexpr int NP = 3, NA = SURF_SDF_NA_BF3, PF = 1;
  static constexpr int occ(bool) { return 1; }  // three-piece activations need the whole register file
  static constexpr int nslot(bool) { return SURF_SDF_NSLOT_BF3; }  // LDS ring length (36 KB slots)
  static constexpr int REG_SLICES = SURF_SDF_REG_SLICES_BF3;
  static constexpr int LDS_SLICES = SURF_SDF_LDS_SLICES_BF3;       // exponent slices per wavefront in the spare LDS (3 x 36 + 4 x 3 x 4 KB = 156 KB)
#ifndef SURF_SDF_DEEP_BF3
#define SURF_SDF_DEEP_BF3 1
#endif
  static constexpr bool DEEP = SURF_SDF_DEEP_BF3;   // the reverse sweep reads its exponent slices two chunks ahead
  static constexpr bool DEEPJ = true;  // feature Jacobian fetched under the last backward chunk
#ifndef SURF_SDF_R0_BF3
#define SURF_SDF_R0_BF3 1
#endif
  static constexpr bool R0 = SURF_SDF_R0_BF3;  // layer 0's exponent arguments recomputed by the reverse sweep (see make_chunks)
  static constexpr ChunkTable CH = make_chunks(NP, R0);
  struct Acc { f32x16 v[NA]; };
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    uint32_t u = __builtin_bit_cast(uint32_t, v);
    asm volatile("" : "+v"(u));  // keep the packed value: the residuals below come from its two halves
    return u;
  }
  static __device__ __forceinline__ void split(float a, float b, uint32_t (&p)[NP]) { surf_split3_bf16(a, b, p); }
  // The same split as mini-phases (see "conversion slots"): piece = pack(v) (round to nearest even), its exact value back
  // as two floats, v -= that.
  static __device__ __forceinline__ uint32_t pack(f32x2 v) { return pack2(v[0], v[1]); }
  static __device__ __forceinline__ f32x2 expand(uint32_t p) {
    f32x2 x;
    x[0] = __builtin_bit_cast(float, p << 16);
    x[1] = __builtin_bit_cast(float, p & 0xffff0000u);
    return x;
  }
  static constexpr bool FUSED_SUB = false;  // (no fp32 instruction reads a bf16 half in place)
  // Residuals of the split on the matrix pipe (K_MRES / K_PACKT), round 5 - measured and OFF: -35 % plain VALU instructions per tile
  // (7,993 -> 5,216) for +6.8 % MFMAs (2,538 -> 2,710), bit-identical results, and the gradient kernel got SLOWER, 114.4 -> 116.2 ms
  // same-box (forward-only 52.7 -> 53.9): the sweeps are not bound by the number of VALU instructions in their gaps (DESIGN K9b).
#ifndef SURF_SDF_MRES_BF3
#define SURF_SDF_MRES_BF3 0
#endif
  static constexpr bool MRES = SURF_SDF_MRES_BF3;
  // The network runs in units of the softplus exponent: pre-activations u = 100 log2(e) t, activations z = y 100 / ln 2
  // = max(u, 0) + log2(1 + 2^-|u|).  Because 100 log2(e) x ln(2) / 100 = 1, every hidden matrix is UNCHANGED (u' = W z + c b):
  // the packer scales only the biases and the input (positional-encoding / feature) columns by c = 100 log2 e and row 0 of
  // lin6 by ln 2 / 100 - and the kernel saves the multiply in front of every exponential (2 of ~25 instructions per pair).
  // Not for f16x2: activations 144 times larger would leave the fp16 range at |y| > 454.
  static constexpr bool PRESCALED = true;
  static __device__ __forceinline__ f32x2 sub_piece(f32x2 v, uint32_t p) { return v - expand(p); }
  // MFMA m of a k-step (NM per k-step, smallest terms first): piece of A, piece of B
  static constexpr int NM = 6;
  static __device__ __forceinline__ void mma_one(Acc& acc, const u32x4 (&a)[NP], const FragT<NP>& b, int m) {
    constexpr int X[6] = {2, 0, 1, 1, 0, 0}, Y[6] = {0, 2, 1, 0, 1, 0};
    acc.v[m & (NA - 1)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[X[m]]), __builtin_bit_cast(bf16x8, b.p[Y[m]]),
                                                                 acc.v[m & (NA - 1)], 0, 0, 0);
  }
  static __device__ __forceinline__ void mma(Acc& acc, const u32x4 (&a)[NP], const FragT<NP>& b) {
#pragma unroll
    for (int m = 0; m < NM; ++m) mma_one(acc, a, b, m);
  }
  static __device__ __forceinline__ f32x16 finish(const Acc& acc) {
    if (NA == 1) return acc.v[0];
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = acc.v[0][i] + acc.v[NA - 1][i];
    return r;
  }
};

struct PolH2 {
  static constexpr int NP = 2, NA = 1, PF = 1;
  // Workgroups per CU.  Forward-only fits 256 registers without spilling and gains from a second workgroup; the
  // gradient kernel at 256 registers spills ~250 dwords, runs slower than one workgroup with the whole register file
  // (51.0 vs 49.2 ms) and - with the two-chunk read-ahead enabled - FAILED the race screen (scripts/stress_sdf.py)
  // in every launch, for a reason not understood; at 512 registers nothing spills and the screen is clean.
  static constexpr int occ(bool grad) { return grad ? 1 : 2; }
  static constexpr int nslot(bool grad) { return grad ? SURF_SDF_NSLOT_H2 : 3; }
  static constexpr int REG_SLICES = SURF_SDF_REG_SLICES_H2;
  static constexpr int LDS_SLICES = SURF_SDF_LDS_SLICES_H2;        // 3 x 24 + 4 x 5 x 4 KB = 152 KB (gradient kernel: one workgroup per CU)  // 24 KB slots; two workgroups per CU forward-only
  static constexpr bool DEEP = true, DEEPJ = true;
  static constexpr bool R0 = false;  // all 20 slices already stay on the CU
  static constexpr ChunkTable CH = make_chunks(NP, R0);
  struct Acc { f32x16 v[NA]; };
  static __device__ __forceinline__ void split(float a, float b, uint32_t (&p)[NP]) {
    const f32x2 v = {a, b};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 r = v - __builtin_convertvector(h, f32x2);
    p[0] = __builtin_bit_cast(uint32_t, h);
    p[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
  }
  static __device__ __forceinline__ uint32_t pack(f32x2 v) {
    uint32_t p = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
    asm volatile("" : "+v"(p));
    return p;
  }
  static __device__ __forceinline__ f32x2 expand(uint32_t p) { return __builtin_convertvector(__builtin_bit_cast(f16x2, p), f32x2); }
  // v - (the two halves of p as floats), one v_fma_mix_f32 per element (an fp32 FMA that reads an fp16 half directly: no
  // conversion instruction): fma(half, -1, v) is exact in the half and rounds once, like the subtraction it replaces.
  static constexpr bool FUSED_SUB = true;
  static constexpr bool MRES = false;  // one residual, one v_fma_mix per element: nothing to move
  static constexpr bool PRESCALED = false;
  static __device__ __forceinline__ f32x2 sub_piece(f32x2 v, uint32_t p) {
    f32x2 r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(p), "v"(v[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(p), "v"(v[1]));
    return r;
  }
  static constexpr int NM = 3;
  static __device__ __forceinline__ void mma_one(Acc& acc, const u32x4 (&a)[NP], const FragT<NP>& b, int m) {
    constexpr int X[3] = {1, 0, 0}, Y[3] = {0, 1, 0};
    acc.v[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[X[m]]), __builtin_bit_cast(f16x8, b.p[Y[m]]), acc.v[0], 0, 0, 0);
  }
  static __device__ __forceinline__ void mma(Acc& acc, const u32x4 (&a)[NP], const FragT<NP>& b) {
#pragma unroll
    for (int m = 0; m < NM; ++m) mma_one(acc, a, b, m);
  }
  static __device__ __forceinline__ f32x16 finish(const Acc& acc) { return acc.v[0]; }
};

// Power-of-two operand scales (exact): weights and back-propagated deltas are stored x 2^8 so that their second fp16
// piece stays a normal number down to |v| ~ 5e-4 (activations are O(1) and stay unscaled; their second piece carries an
// absolute error <= 2^-25).  Accumulators therefore come out x W_SCALE (forward) and x W_SCALE x D_SCALE (backward).
template <class P> struct Scales { static constexpr float W = 1.0f, D = 1.0f; };
template <> struct Scales<PolH2> { static constexpr float W = 256.0f, D = 256.0f; };

template <class P> constexpr int stream_bytes() { return P::CH.total; }
template <class P> constexpr int slot_bytes() { return MAX_KS * P::NP * 1024; }
template <class P> constexpr int max_blocks(bool grad) { return 256 * P::occ(grad); }
// chunks per round: the gradient stream is padded with empty chunks to a multiple of the ring length, so that the slot of
// a chunk (index % ring length) continues across rounds
template <class P> constexpr int n_chunks(bool grad) {
  return grad ? (P::CH.n + P::nslot(true) - 1) / P::nslot(true) * P::nslot(true) : N_FWD_CHUNKS;
}

struct SdfArgs {
  const float* pts;
  const uint8_t* mask;
  const int32_t* idx;
  int64_t n;            // entries of idx / points (capacity when n_dev is set)
  const int32_t* n_dev;  // optional device-side count (<= n): no host round trip between the compaction and this launch
  const float* vols[SURF_MAX_STAGES];
  const int32_t* tables[SURF_MAX_STAGES];
  int dims[SURF_MAX_STAGES];
  const unsigned char* packed;
  float* sdf;
  float* grad;
  float* scratch;
  // lattice mode (pts == nullptr, forward only; surf_sdf_lattice_*): point i = (ax[i / (ny nz)], ay[(i / nz) % ny], az[i % nz]) from
  // the three axis arrays - no point tensor is written or read - and sdf[i] = out_sign * value (extract_geometry's u = -sdf)
  const float* lat_axes[3];
  int lat_ny, lat_nz;
  float out_sign;
};

__device__ __forceinline__ f32x4 bload(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// Cache policy of the exponent-slice / Jacobian scratch round trip (written once, read once ~40 us later by the same wave):
// SURF_X_SCRATCH_NT bit 0: stores non-temporal, bit 1: loads non-temporal (timing experiment of round 3).
#ifndef SURF_X_SCRATCH_NT
#define SURF_X_SCRATCH_NT 0
#endif
__device__ __forceinline__ f32x4 bload_scratch(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, (SURF_X_SCRATCH_NT & 2) ? 2 : 0));
}
// 16-byte scratch store.  gfx950: a VALU write to the data VGPRs right after `buffer_store_dwordx4 ... sN offen` corrupts
// lanes 12-15 of every 16 (see sdf_mlp.hip).  Store and pad are ONE asm statement so that nothing - not the scheduler,
// not a register-allocator copy or reload - can land between them.
__device__ __forceinline__ void bstore(rsrc_t r, int voff, int soff, f32x4 v) {
#if SURF_X_SCRATCH_NT & 1
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(r), "s"(soff) : "memory");
#else
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(r), "s"(soff) : "memory");
#endif
}

template <class P>
__device__ __forceinline__ void frag_set_pair(FragT<P::NP>& f, int pair /*0..3*/, float a, float b) {
  uint32_t p[P::NP];
  if (SURF_X_NOSPLIT) {
#pragma unroll
    for (int k = 0; k < P::NP; ++k) p[k] = __builtin_bit_cast(uint32_t, k & 1 ? a : b);
  } else {
    P::split(a, b, p);
  }
#pragma unroll
  for (int k = 0; k < P::NP; ++k) f.p[k][pair] = p[k];
}

// softplus(beta = 100, threshold = 20) and its derivative for a pair of pre-activations given x ACC_SCALE, in the
// overflow-free form  h = max(t, 0) + log(1 + exp(-|100 t|)) / 100,  h' = (t >= 0 ? 1 : exp(-|100 t|)) / (1 + exp(-|100 t|)),
// which equals torch's thresholded softplus to fp32 rounding (the linear branch differs from it by < 2^-33 relative).
// MODE 0: h only.  MODE 1: sv = h'.  MODE 2: sv = u = 100 log2(e) t, the exponent argument: the reverse sweep forms
// h' = 1 / (1 + 2^-u) from it under ITS MFMAs.  The forward tiles are bound by the issue slots of their
// conversion arithmetic (~42 issue cycles per 32-cycle MFMA with h' formed here, round-3 ISA count), the backward tiles
// have slots to spare (~22), so the compare / select / reciprocal / multiply of h' move there; the scratch round trip
// carries u instead of h' (same bytes).
template <int MODE, bool PRESCALED = false>
__device__ __forceinline__ void softplus_pair(f32x2 acc, float acc_scale_inv, f32x2& hv, f32x2& sv) {
  if (SURF_X_NOSOFTPLUS) {
    hv = acc * acc_scale_inv;
    sv = acc * 0.5f;
    return;
  }
  const f32x2 arg = PRESCALED ? acc : acc * (144.269504088896341f * acc_scale_inv);  // 100 log2(e) t
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(-__builtin_fabsf(arg[0]));
  e[1] = __builtin_amdgcn_exp2f(-__builtin_fabsf(arg[1]));
  const f32x2 d = e + 1.0f;
  f32x2 l;
  l[0] = __builtin_amdgcn_logf(d[0]);
  l[1] = __builtin_amdgcn_logf(d[1]);
  f32x2 m;
  m[0] = __builtin_amdgcn_fmed3f(acc[0], 0.0f, 3.0e38f);  // max(acc, 0) without the canonicalising extra v_max
  m[1] = __builtin_amdgcn_fmed3f(acc[1], 0.0f, 3.0e38f);
  if (acc_scale_inv != 1.0f) m = m * acc_scale_inv;
  if (PRESCALED) {  // (policy PRESCALED: the activation in exponent units)
    hv = m + l;
  } else {
    hv[0] = fmaf(l[0], 0.69314718055994531f * 0.01f, m[0]);
    hv[1] = fmaf(l[1], 0.69314718055994531f * 0.01f, m[1]);
  }
  if (MODE == 1) {
    f32x2 r, sel;
    r[0] = __builtin_amdgcn_rcpf(d[0]);
    r[1] = __builtin_amdgcn_rcpf(d[1]);
    sel[0] = acc[0] >= 0.0f ? 1.0f : e[0];
    sel[1] = acc[1] >= 0.0f ? 1.0f : e[1];
    sv = sel * r;
  } else if (MODE == 2) {
    sv = arg;
  }
}
// h' = 1 / (1 + 2^-u) of a stored exponent argument u = 100 log2(e) t (softplus_pair MODE 2), formed by the reverse sweep's
// mini-phases K_EXPN / K_ADD1 / K_RCP.  No branch and no overflow case: u << 0 gives 2^-u = inf and 1 / inf = 0, u >> 0 gives
// 2^-u = 0 and h' = 1; for u < 0 this is e / (1 + e) with e = 2^u divided through by e, i.e. the same value as MODE 1 to
// fp32 rounding.

// ---- conversion slots ------------------------------------------------------------------------------------------------------
// One wavefront per SIMD issues in order, and an MFMA that follows another within its 32 cycles holds the wave's issue until
// the matrix pipe is free: only instructions that stand BETWEEN two MFMAs in program order run in the first one's shadow
// (8 of the 32 cycles are the MFMA's own issue, plain VALU operations cost 4, transcendental ones 8, and so does every
// `s_nop` the compiler has to put between a result and a use that follows it too closely: MI355X_MICROARCH.md, instruction
// constants).  Measured on this kernel (round 3, half image): MFMA stream alone 36.3 ms, everything but the MFMAs 36.4 ms,
// both 63.8 ms when the compiler places the conversion arithmetic (runs of 5-6 bare MFMAs, then clumps of 10-25 VALU
// operations): the two hardly overlapped.  So the conversion work of a tile (softplus, operand split, stores) is cut into
// MINI-PHASES of two independent instructions (the same step for the two elements of a pair, 8-16 issue cycles) whose
// inputs were produced by an earlier mini-phase; the pairs of a tile form two streams (even / odd pairs, the odd one half a
// pair behind), and a compile-time plan (plan_gaps) deals the woven sequence out over the MFMA gaps of the chunk it hides
// under - at most one mini-phase of each stream per gap where the chunk is long enough, so that no instruction waits for
// its neighbour - around the gaps' fixed contents (LDS reads of the next k-step's A pieces, LDS-DMA issue).  run_chunk pins
// each gap with a scheduling barrier and every mini-phase pins its inputs and results (slot_pin).
enum MiniKind : int {
  K_ARG, K_EXP, K_ADD1, K_LOG, K_MAX, K_FMA,  // u = 100 log2(e) t | e = 2^-|u| | d = 1 + e | log2 d | max(t, 0) | h
  K_Y0, K_RCP, K_SEL, K_SIG,                  // layer 5: y0 += w6 h | 1 / d | (t >= 0 ? 1 : e) | h' w6
  K_EXPN, K_MULG,                             // reverse sweep: e = 2^-u | delta = G / d
  K_PACK, K_EXPAND, K_SUB,                    // operand split: piece i = pack(v) | its exact value | v -= that
  K_SUBP,                                     // ... or both in one where an fp32 instruction reads the packed halves (f16x2)
  // Matrix-pipe residuals (round 5, P::MRES): the tile's values stay an accumulator tile `vt`; after piece i of ALL pairs is packed,
  // vt -= (piece i) as two MFMAs with a constant "minus identity" A operand (one per 16-row k-step of the tile; bit-identical to
  // the VALU subtraction, scripts/microbench/mfma_residual.hip); piece i + 1 of every pair is then packed from vt.
  K_MRES,                                     // arg = 2 lvl + s: vt = mfma(-I_s, piece lvl of fragment s, vt)
  K_PACKT                                     // arg = lvl: piece lvl of pair q = pack(vt[2q], vt[2q + 1])
};
constexpr int MAX_MINI = 24, MAX_SLOTS = 8 * MAX_MINI;
struct MiniProg { int n; int kind[MAX_MINI]; int arg[MAX_MINI]; int cost[MAX_MINI]; };
constexpr void mini_add(MiniProg& mp, int kind, int arg, int cost) {
  mp.kind[mp.n] = kind; mp.arg[mp.n] = arg; mp.cost[mp.n] = cost; ++mp.n;
}
constexpr void mini_add_split(MiniProg& mp, int np, bool fused_sub, bool mres = false) {
  if (mres) {  // per pair only piece 0 (and the value into the tile); weave() appends the tile-wide tail
    mini_add(mp, K_PACK, 0, 4);
    return;
  }
  for (int i = 0; i < np; ++i) {
    mini_add(mp, K_PACK, i, 4);
    if (i + 1 < np && fused_sub) mini_add(mp, K_SUBP, i, 8);
    if (i + 1 < np && !fused_sub) { mini_add(mp, K_EXPAND, i, 8); mini_add(mp, K_SUB, i, 8); }
  }
}
// the woven sequence of a tile's `pairs` pairs: slot s = mini-phase j[s] of pair q[s] (stream q & 1)
// (kind / arg of a slot; q = -1: a tile-wide slot; j: position in the pair's own sequence, the tail's pieces continuing it)
struct SlotProg { int n; int q[MAX_SLOTS]; int j[MAX_SLOTS]; int cost[MAX_SLOTS]; int kind[MAX_SLOTS]; int arg[MAX_SLOTS]; int per_pair; };
// mres_np > 0: the per-pair programs end with piece 0 (K_PACK 0); the tail is, for every further piece lvl = 1 .. mres_np - 1:
// K_MRES (lvl - 1, s = 0), K_MRES (lvl - 1, s = 1), then K_PACKT lvl of pairs 0 .. pairs - 1.
constexpr SlotProg weave(MiniProg mp, int pairs, int store_cost, int mres_np = 0) {
  SlotProg sp{};
  const int per = (pairs / 2) * mp.n, lag = mp.n | 1;  // stream B starts `lag` half-steps after stream A
  int ia = 0, ib = 0;
  while (ia < per || ib < per) {
    const bool take_a = ib >= per || (ia < per && 2 * ia <= 2 * ib + lag);
    const int i = take_a ? ia++ : ib++;
    const int q = 2 * (i / mp.n) + (take_a ? 0 : 1), j = i % mp.n;
    sp.q[sp.n] = q; sp.j[sp.n] = j;
    sp.kind[sp.n] = mp.kind[j]; sp.arg[sp.n] = mp.arg[j];
    sp.cost[sp.n] = mp.cost[j] + ((j == mp.n - 1 && (q & 1)) ? store_cost : 0);
    ++sp.n;
  }
  sp.per_pair = mp.n;
  for (int lvl = 1; lvl < mres_np; ++lvl) {
    for (int s2 = 0; s2 < 2; ++s2) {
      sp.q[sp.n] = -1; sp.j[sp.n] = 0; sp.kind[sp.n] = K_MRES; sp.arg[sp.n] = 2 * (lvl - 1) + s2; sp.cost[sp.n] = 0;
      ++sp.n;
    }
    for (int q = 0; q < pairs; ++q) {
      sp.q[sp.n] = q; sp.j[sp.n] = mp.n + lvl - 1; sp.kind[sp.n] = K_PACKT; sp.arg[sp.n] = lvl; sp.cost[sp.n] = 4;
      ++sp.n;
    }
    sp.per_pair = mp.n + lvl;
  }
  return sp;
}

// Which slots run in which MFMA gap of a chunk.  Gap g = ks * NM + m follows MFMA m of k-step ks.
constexpr int MAX_GAPS = MAX_KS * 6, MAX_DMA = 12;
struct GapPlan {
  int first[MAX_GAPS + 1];  // slots [first[g], first[g + 1]) run in gap g
  int dma_gap[MAX_DMA];     // gap after which LDS-DMA piece k is issued
  int cap, per_stream;      // what the plan needed: issue cycles per gap beside the MFMA itself (24 hide completely), mini-phases
                            // of one stream per gap (1: no instruction of a gap depends on another one of it)
};
// nks k-steps of nm MFMAs; np LDS reads (A pieces of the next k-step, 8 cycles with the address copy) follow MFMAs 0..np-1
// of every k-step but the last; n_dma DMA issues (descriptor offset + M0 + the load: ~12 cycles); slots in order, all of
// them in gaps < deadline.
constexpr GapPlan plan_gaps(int nks, int nm, int np, int n_dma, int deadline, SlotProg sp) {
  GapPlan pl{};
  const int ng = nks * nm;
  int fixed[MAX_GAPS + 1] = {};
  for (int ks = 0; ks + 1 < nks; ++ks)
    for (int m = 0; m < np && m < nm; ++m) fixed[ks * nm + m] += 8;
  for (int k = 0; k < n_dma; ++k) {
    int g = 0;
    if (ng > 0) g = n_dma <= nks ? ((k * nks) / n_dma) * nm + nm - 1 : (k * ng) / n_dma;
    pl.dma_gap[k] = g;
    fixed[g] += 12;
  }
  if (deadline > ng) deadline = ng;
  int at[MAX_SLOTS] = {};
  int cap = 24, per = 1;
  for (bool done = sp.n == 0; !done;) {  // fewest mini-phases of one stream per gap first, then the smallest budget
    for (cap = 24; cap <= 32 + 8 * per && !done; cap += 4) {
      int g = 0, used = fixed[0], cnt[2] = {0, 0};
      int mres_gap = -1;
      bool ok = true, after_mres = false;
      for (int s = 0; s < sp.n && ok; ++s) {
        if (sp.kind[s] == K_MRES) {
          // a residual MFMA: at most one per gap (the second of a level depends on the first), behind what the gap already
          // holds; whatever follows it in the gap runs in ITS 32 cycles: a fresh budget
          if (mres_gap == g) {
            if (g + 1 >= deadline) { ok = false; break; }
            ++g; used = fixed[g]; cnt[0] = cnt[1] = 0;
          }
          at[s] = g;
          mres_gap = g;
          used = 0; cnt[0] = cnt[1] = 0;
          after_mres = true;
          continue;
        }
        if (sp.kind[s] == K_PACKT) {
          // reads the residual tile: not in the gap of the MFMA that writes it (its result is 32+ cycles away)
          if (after_mres && mres_gap == g) {
            if (g + 1 >= deadline) { ok = false; break; }
            ++g; used = fixed[g]; cnt[0] = cnt[1] = 0;
          }
          after_mres = false;
          while (used + sp.cost[s] > cap && g + 1 < deadline) { ++g; used = fixed[g]; cnt[0] = cnt[1] = 0; }
          if (used + sp.cost[s] > cap) ok = false;
          at[s] = g;
          used += sp.cost[s];
          continue;
        }
        const int st = sp.q[s] & 1;
        while ((used + sp.cost[s] > cap || cnt[st] >= per) && g + 1 < deadline) { ++g; used = fixed[g]; cnt[0] = cnt[1] = 0; }
        if (used + sp.cost[s] > cap || cnt[st] >= per) ok = false;
        at[s] = g;
        used += sp.cost[s];
        ++cnt[st];
      }
      done = ok;
    }
    if (!done) ++per;
    else cap -= 4;
    if (per > MAX_SLOTS) break;
  }
  for (int i = 0; i <= ng; ++i) {  // first[i] = number of slots placed in gaps < i
    int n = 0;
    for (int q = 0; q < sp.n; ++q) n += at[q] < i ? 1 : 0;
    pl.first[i] = n;
  }
  pl.cap = cap;
  pl.per_stream = per;
  return pl;
}
// What every plan is checked for at compile time (static_assert in FwdPlan / BwdPlan): each pair's mini-phases 0 .. n-1 appear
// exactly once and in order in the woven sequence; the slots are dealt out in order (first[] non-decreasing, from 0 to all of
// them) and all of them before the deadline gap; no gap holds more mini-phases of one stream than the plan reports.
// With matrix-pipe residuals (K_MRES / K_PACKT slots) also: the residual MFMAs of level lvl come after piece lvl of every pair
// and before piece lvl + 1 of any, s = 0 before s = 1 and in different gaps, and no piece is packed in the gap of the MFMA that
// forms its residual.
constexpr bool plan_ok(const GapPlan& pl, const SlotProg& sp, const MiniProg& mp, int ng, int deadline) {
  if (deadline > ng) deadline = ng;
  int next[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int mres_seen = 0;  // residual MFMAs so far: 2 lvl + s is the next one expected
  for (int s = 0; s < sp.n; ++s) {
    if (sp.kind[s] == K_MRES) {
      if (sp.arg[s] != mres_seen) return false;
      for (int q = 0; q < 8; ++q)  // piece lvl of every pair is packed, none of the next one
        if (next[q] != mp.n + sp.arg[s] / 2) return false;
      ++mres_seen;
      continue;
    }
    if (sp.q[s] < 0 || sp.q[s] >= 8 || sp.j[s] != next[sp.q[s]]) return false;
    if (sp.kind[s] == K_PACKT && (sp.j[s] != mp.n + sp.arg[s] - 1 || mres_seen != 2 * sp.arg[s])) return false;
    ++next[sp.q[s]];
  }
  for (int q = 0; q < 8; ++q)
    if (sp.n > 0 && next[q] != sp.per_pair) return false;
  if (pl.first[0] != 0 || (ng > 0 && pl.first[ng] != sp.n) || (sp.n > 0 && pl.first[deadline] != sp.n)) return false;
  for (int g = 0; g < ng; ++g) {
    if (pl.first[g] > pl.first[g + 1]) return false;
    int cnt[2] = {0, 0}, n_mres = 0;
    bool mres_here = false;
    for (int s = pl.first[g]; s < pl.first[g + 1]; ++s) {
      if (sp.kind[s] == K_MRES) { ++n_mres; mres_here = true; cnt[0] = cnt[1] = 0; continue; }
      if (sp.kind[s] == K_PACKT) { if (mres_here) return false; continue; }
      ++cnt[sp.q[s] & 1];
      if (cnt[0] > pl.per_stream || cnt[1] > pl.per_stream) return false;
    }
    if (n_mres > 1) return false;
  }
  return true;
}
// gap in which the last mini-phase of pair q runs
constexpr int pair_done_gap(const GapPlan& pl, const SlotProg& sp, int q, int ng) {
  int last = 0;
  for (int s = 0; s < sp.n; ++s)
    if (sp.q[s] == q) last = s;
  int g = 0;
  while (g < ng && pl.first[g + 1] <= last) ++g;
  return g;
}

// compile-time loops: f(IC<LO>{}), ..., f(IC<HI - 1>{}).  (`#pragma unroll` loops whose bounds come out of the plan tables were
// left as run-time loops over run-time register indices by the unroller.)
template <int I> struct IC { static constexpr int value = I; };
template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (LO < HI) {
    f(IC<LO>{});
    static_for<LO + 1, HI>(f);
  }
}

// state of one stream of mini-phases
template <int NP>
struct MiniState { f32x2 u, e, d, l, m, r, sel, x, val; uint32_t pc[NP]; };

// ---- gather / posenc (identical arithmetic to sdf_mlp.hip) -------------------------------------------------------------
// Sparse trilinear gather of this lane half's two pyramid levels: phi[7 sl + ch], and (GRAD) the feature Jacobian of each
// level straight to the wave's scratch slot (6 x 16 B per level: [ch][axis], 21 values + pad) to keep registers free.
// `mid()` runs between the issue of the row loads and their first use (the caller's positional encoding: a few hundred
// instructions that need no memory).
template <bool GRAD, class Ctx, class Mid>
__device__ __forceinline__ void gather_features(const SdfArgs& a, const Ctx& c, float px, float py, float pz, float (&phi)[16], Mid mid) {
#pragma unroll
  for (int ch = 0; ch < 16; ++ch) phi[ch] = 0.f;
  int rows[2][8];
  float tx[2], ty[2], tz[2], inv_vs[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    // (selects between kernel arguments read with compile-time indices: `a.dims[2 * c.h + sl]` is a per-lane global load from
    // the argument buffer, one more round trip in front of the table -> row chain)
    const int D = c.h ? a.dims[2 + sl] : a.dims[sl];
    const int32_t* __restrict__ table = c.h ? a.tables[2 + sl] : a.tables[sl];
    const float vs = 2.0f / ((float)D - 1.0f);
    inv_vs[sl] = 1.0f / vs;
    const float gx = (px + 1.0f) / vs, gy = (py + 1.0f) / vs, gz = (pz + 1.0f) / vs;
    const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
    tx[sl] = gx - fx; ty[sl] = gy - fy; tz[sl] = gz - fz;
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    // clamped corner coordinates; D <= 1024 (checked at launch), so 24-bit multiplies and 32-bit indices are exact
    int xs[2], ys[2], zs[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      xs[d] = min(max(x0 + d, 0), D - 1);
      ys[d] = min(max(y0 + d, 0), D - 1);
      zs[d] = min(max(z0 + d, 0), D - 1);
    }
    // unconditional loads (the launcher points the tables of absent levels at a valid one; their rows are discarded here):
    // a load under `D > 0 ? .. : -1` becomes a branch, and sixteen branches are sixteen serialised round trips
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned xy = __umul24(__umul24(xs[k >> 2], D) + ys[(k >> 1) & 1], D);
      rows[sl][k] = table[D > 0 ? xy + zs[k & 1] : 0u];
    }
  }
#pragma unroll
  for (int sl = 0; sl < 2; ++sl)
#pragma unroll
    for (int k = 0; k < 8; ++k) rows[sl][k] = (c.h ? a.dims[2 + sl] : a.dims[sl]) > 0 ? rows[sl][k] : -1;
  // all 32 row loads of the two levels are in flight before the first of them is used (the kernels are compiled with source-
  // order instruction selection, see build.sh: the order written here is the order issued)
  f32x4 f0[2][8], f1[2][8];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const float* __restrict__ vol = c.h ? a.vols[2 + sl] : a.vols[sl];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const f32x4* fr = reinterpret_cast<const f32x4*>(vol + (int64_t)max(rows[sl][k], 0) * 8);
      f0[sl][k] = fr[0];
      f1[sl][k] = fr[1];
    }
  }
  mid();
  // The trilinear sums on explicit pairs of channels (v_pk_mul / v_pk_add_f32: two lanes of a register pair per instruction;
  // the file is compiled without the SLP vectoriser, which would otherwise also pack the conversion arithmetic between the
  // MFMAs, where a packed instruction costs ~15 cycles more than two plain ones - scripts/microbench/mfma_issue_model.hip).
  // Pair p of a row = channels (2p, 2p + 1); channel 7 is the row's padding.
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    f32x2 ph[4], Jx[4], Jy[4], Jz[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) ph[p] = Jx[p] = Jy[p] = Jz[p] = f32x2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int dx = k >> 2, dy = (k >> 1) & 1, dz = k & 1;
      const float ok = rows[sl][k] >= 0 ? 1.0f : 0.0f;
      const float wx = dx ? tx[sl] : 1.0f - tx[sl];
      const float wy = dy ? ty[sl] : 1.0f - ty[sl];
      const float wz = dz ? tz[sl] : 1.0f - tz[sl];
      const float w = wx * wy * wz * ok;
      const f32x2 fp[4] = {{f0[sl][k][0], f0[sl][k][1]}, {f0[sl][k][2], f0[sl][k][3]}, {f1[sl][k][0], f1[sl][k][1]}, {f1[sl][k][2], f1[sl][k][3]}};
      float cx = 0.f, cy = 0.f, cz = 0.f;
      if (GRAD) {
        cx = ((dx ? 1.0f : -1.0f) * wy * wz) * (inv_vs[sl] * ok);
        cy = ((dy ? 1.0f : -1.0f) * wx * wz) * (inv_vs[sl] * ok);
        cz = ((dz ? 1.0f : -1.0f) * wx * wy) * (inv_vs[sl] * ok);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        ph[p] = ph[p] + fp[p] * w;
        if (GRAD) {
          Jx[p] = Jx[p] + fp[p] * cx;
          Jy[p] = Jy[p] + fp[p] * cy;
          Jz[p] = Jz[p] + fp[p] * cz;
        }
      }
    }
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) phi[7 * sl + ch] = ph[ch >> 1][ch & 1];
    if (GRAD) {
      float Jl[24];
#pragma unroll
      for (int ch = 0; ch < 7; ++ch) {
        Jl[3 * ch + 0] = Jx[ch >> 1][ch & 1];
        Jl[3 * ch + 1] = Jy[ch >> 1][ch & 1];
        Jl[3 * ch + 2] = Jz[ch >> 1][ch & 1];
      }
      Jl[21] = Jl[22] = Jl[23] = 0.f;
#pragma unroll
      for (int g = 0; g < 6; ++g) {
        const f32x4 v = {Jl[4 * g], Jl[4 * g + 1], Jl[4 * g + 2], Jl[4 * g + 3]};
        bstore(c.sr, c.svoff, SCR_S * 4 + (6 * sl + g) * 1024, v);
      }
    }
  }
}

// Positional encoding of this lane half (14 of the 27 channels + pad) from the three base (sin, cos) pairs: the
// 2x, 4x, 8x terms by exact double-angle steps.  The base pairs are kept for the epilogue's Jacobian diagonal.
struct SinCos3 { float s[3], c[3]; };
__device__ __forceinline__ SinCos3 sincos3(float x, float y, float z) {
  SinCos3 b;
  sincosf(x, &b.s[0], &b.c[0]);
  sincosf(y, &b.s[1], &b.c[1]);
  sincosf(z, &b.s[2], &b.c[2]);
  return b;
}
__device__ __forceinline__ void posenc_half(int h, float x, float y, float z, const SinCos3& b, float (&e)[16], float (&je)[14],
                                            bool want_j) {
  float all[28], jall[28];
  all[0] = x; all[1] = y; all[2] = z;
  jall[0] = jall[1] = jall[2] = 1.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float s = b.s[c], co = b.c[c];
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
template <class P>
__device__ __forceinline__ void local_frags(const float (&v)[16], FragT<P::NP> (&f)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) frag_set_pair<P>(f[s], pr, v[8 * s + 2 * pr], v[8 * s + 2 * pr + 1]);
}
}  // namespace
