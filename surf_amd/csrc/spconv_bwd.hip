// K5b  weight gradient of a sparse 3^3 convolution (first backward kernel of the volume-build side, row f2 / K12):
//     dW[k][ci][co] = sum over output sites i whose offset-k neighbour j exists of  x[j][ci] * dy[i][co]
// (the input gradient of a sparse convolution is itself a sparse convolution - submanifold with mirrored offsets, stride-2
// down <-> transposed up - with the transposed kernel slices, so it runs on spconv.hip's kernels: ops.spconv_backward).
// One workgroup sweeps tiles of 64 (wide layers) or 256 (thin layers) output sites; per kernel offset the gathered input rows and
// the dy rows sit in LDS and every thread owns 4 x 2 register blocks of that offset's slice (thin layers: one block on a
// 1/G share of the tile's sites; 0.75 LDS words per FMA instead of 2), accumulated in registers over the workgroup's tiles and added to dW with one float atomic per
// entry and workgroup.
#include <stdlib.h>

#include "common.h"

namespace {

enum { MODE_SUBM = 0, MODE_DOWN = 1, MODE_UP = 2 };

struct WgArgs {
  const float* x;            // (n_in, CIN)
  const int32_t* in_table;   // (Din^3)
  int Din;
  const int32_t* out_coords; // (n_out, 3)
  int64_t n_out;
  int mode;
  const float* dy;           // (n_out, COUT)
  float* dW;                 // (27, CIN, COUT), accumulated
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void spconv_wgrad_kernel(WgArgs a) {
  constexpr int TS = (CIN + COUT <= 48) ? 256 : 64;   // output sites per tile (LDS: TS (CIN + COUT + 6) floats)
  constexpr int NB = (CIN / 4) * (COUT / 2);          // 4 x 2 register blocks of the offset's (CIN, COUT) slice
  constexpr int PERB = NB >= 256 ? NB / 256 : 1;      // blocks per thread
  constexpr int G = NB >= 256 ? 1 : 256 / NB;         // site groups sharing one block (thin layers: every thread has work)
  constexpr int XS = CIN + 4, DS = COUT + 2;          // padded row strides (16-B / 8-B aligned vector reads)
  static_assert(NB >= 256 ? NB % 256 == 0 : 256 % NB == 0, "(CIN / 4) (COUT / 2) must divide or be a multiple of 256");
  __shared__ __attribute__((aligned(16))) float xs[TS * XS];
  __shared__ __attribute__((aligned(16))) float ds[TS * DS];
  __shared__ int rows[TS];
  __shared__ float red[NB >= 256 ? 1 : 256 * 8];
  const int D = a.Din;
  const int64_t n_tiles = (a.n_out + TS - 1) / TS;
  const int g = NB >= 256 ? 0 : threadIdx.x / NB;
  const int b0 = NB >= 256 ? threadIdx.x : threadIdx.x % NB;
  // offsets k = blockIdx.y (mod gridDim.y): the launch spreads the 27 offsets over the grid for the wide pairs (round 5) - one
  // workgroup walking all 27 for its tiles was a chain of 27 x tiles gather -> barrier -> FMA phases, ~200 us even for the
  // few-thousand-site coarse lattices
  for (int k = blockIdx.y; k < 27; k += gridDim.y) {
    const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
    float acc[PERB][4][2];
#pragma unroll
    for (int e = 0; e < PERB; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[e][q][0] = acc[e][q][1] = 0.f;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      __syncthreads();
      for (int t = threadIdx.x; t < TS; t += 256) {
        const int64_t i = tile * TS + t;
        int row = -1;
        if (i < a.n_out) {
          const int cx = a.out_coords[i * 3 + 0], cy = a.out_coords[i * 3 + 1], cz = a.out_coords[i * 3 + 2];
          int x, y, z;
          bool ok = true;
          if (a.mode == MODE_SUBM) { x = cx + ox; y = cy + oy; z = cz + oz; }
          else if (a.mode == MODE_DOWN) { x = 2 * cx + ox; y = 2 * cy + oy; z = 2 * cz + oz; }
          else {
            const int tx = cx - ox, ty = cy - oy, tz = cz - oz;
            ok = ((tx | ty | tz) & 1) == 0;
            x = tx >> 1; y = ty >> 1; z = tz >> 1;
          }
          ok = ok && x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D;
          if (ok) row = a.in_table[((int64_t)x * D + y) * D + z];
        }
        rows[t] = row;
      }
      __syncthreads();
      for (int e = threadIdx.x; e < TS * (CIN / 4); e += 256) {       // 16-B gathers of the neighbour rows
        const int s = e / (CIN / 4), c4 = e % (CIN / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (rows[s] >= 0) v = *reinterpret_cast<const f32x4*>(a.x + (int64_t)rows[s] * CIN + c4 * 4);
        *reinterpret_cast<f32x4*>(xs + s * XS + c4 * 4) = v;
      }
      for (int e = threadIdx.x; e < TS * COUT; e += 256) {
        const int s = e / COUT, c = e % COUT;
        const int64_t i = tile * TS + s;
        ds[s * DS + c] = (i < a.n_out && rows[s] >= 0) ? a.dy[i * COUT + c] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < PERB; ++e) {
        const int blk = b0 + 256 * e;
        const int ci0 = (blk / (COUT / 2)) * 4, co0 = (blk % (COUT / 2)) * 2;
#pragma unroll 4
        for (int t = g; t < TS; t += G) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + t * XS + ci0);
          const float d0 = ds[t * DS + co0], d1 = ds[t * DS + co0 + 1];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[e][q][0] = fmaf(xv[q], d0, acc[e][q][0]);
            acc[e][q][1] = fmaf(xv[q], d1, acc[e][q][1]);
          }
        }
      }
    }
    float* dst = a.dW + (int64_t)k * CIN * COUT;
    if (NB >= 256) {
#pragma unroll
      for (int e = 0; e < PERB; ++e) {
        const int blk = b0 + 256 * e;
        const int ci0 = (blk / (COUT / 2)) * 4, co0 = (blk % (COUT / 2)) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < 2; ++r)
            if (acc[e][q][r] != 0.f) atomicAdd(dst + (ci0 + q) * COUT + co0 + r, acc[e][q][r]);
      }
    } else {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        red[(q * 2 + 0) * 256 + threadIdx.x] = acc[0][q][0];
        red[(q * 2 + 1) * 256 + threadIdx.x] = acc[0][q][1];
      }
      __syncthreads();
      for (int o = threadIdx.x; o < NB * 8; o += 256) {                  // o = entry (0..7) x block
        const int ent = o / NB, blk = o % NB;
        float t = 0.f;
        for (int q = 0; q < G; ++q) t += red[ent * 256 + q * NB + blk];
        const int ci = (blk / (COUT / 2)) * 4 + ent / 2, co = (blk % (COUT / 2)) * 2 + (ent & 1);
        if (t != 0.f) atomicAdd(dst + ci * COUT + co, t);
      }
    }
  }
}

// ---- thin layers (C_in + C_out <= 48: the finest lattices, millions of sites), round 3 --------------------------------------
// The kernel above sweeps the output sites once per kernel offset: 27 passes over coords / dy and 27 gathers of 64-byte rows
// per site (15 GB per 5 M-site call, 3 TB/s: memory bound).  Neighbouring sites share almost all of their 27 neighbours - the
// 8 children of a parent reference 64 cells 216 times - so here a workgroup takes a tile of 128 consecutive sites ONCE:
//   B  every (site, offset) neighbour is looked up in the index table and its row id entered into an LDS hash set;
//   C  the DISTINCT rows of the tile (typically ~450 of 3,456 references) are loaded into LDS once;
//   D  every thread owns (offset k, 4 x 2 block of the (C_in, C_out) slice) pairs - 27 (C_in / 4)(C_out / 2) of them over 256
//      threads - and walks the tile's 128 sites for each: x row from the LDS cache, dy row from LDS, 8 FMAs; its accumulators
//      stay in registers across all tiles of the workgroup (no cross-thread reduction) and are added to dW once at the end.
// Traffic per site: 12 B coords + 4 C_out B dy + 108 B of table entries + ~4 distinct rows, instead of 27 x (16 + 4 C_out + 4 C_in p).
// A tile that references more distinct rows than the cache holds is processed in rounds over windows of the row list
// (coarse lattices listed along z: 3 x 3 x 130 cells); a reference that finds the hash set full is served by its own thread with
// global float atomics (> 2,048 distinct rows in one tile: practically never).  Used for the submanifold and transposed (up) convolutions, whose sites share
// neighbours; a stride-2 (down) convolution's 3^3 windows hardly overlap - there the kernel above is faster (1.9 vs 6.2 ms per
// step on <8,16>) and stays in use.
template <int CIN, int COUT>
struct ThinCfg {
// Round 5: 64-site tiles with a 256-row cache and 1,024 hash slots for C_in <= 16 (was 128 / 512 / 2,048): 37 instead of 75 KB of
// LDS, four instead of two workgroups per CU - <16,8> 7.79 -> 5.56 ms per training step; 32 sites 6.58; 512 hash slots overflow
// into the global-atomic fallback (<16,16> 1.3 -> 172 ms): profiles/r05_wgrad_thin_tiles.txt.
#ifndef SURF_THIN_TS
#define SURF_THIN_TS 64
#endif
#ifndef SURF_THIN_CAP
#define SURF_THIN_CAP 256
#endif
#ifndef SURF_THIN_HS
#define SURF_THIN_HS 1024
#endif
  static constexpr int TS = CIN <= 16 ? SURF_THIN_TS : 64;       // sites per tile (C_in = 32: the row cache holds fewer rows)
  // register block of a (offset, block) pair: 4 input x BC output channels.  4 x 4 where that still gives most threads a pair
  // (one LDS index + two 16-byte reads per 16 FMAs), 4 x 2 otherwise (<8,8>: 108 pairs of 4 x 4 would idle 148 threads)
  static constexpr int BC = (COUT % 4 == 0 && 27 * (CIN / 4) * (COUT / 4) >= 200) ? 4 : 2;
  static constexpr int NB = (CIN / 4) * (COUT / BC);
  static constexpr int NPAIR = 27 * NB;
  static constexpr int PER = (NPAIR + 255) / 256;
  static constexpr int CAP = CIN <= 16 ? SURF_THIN_CAP : 320;     // distinct neighbour rows cached per tile
  static constexpr int HS = SURF_THIN_HS;               // hash slots (>= the distinct rows of any realistic tile)
  static constexpr int XS = CIN + 4, DS = COUT + 4;     // padded row strides (16-byte aligned vector reads)
  static constexpr int NJ = (27 * TS + 255) / 256;      // (site, offset) references per thread
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void spconv_wgrad_thin_kernel(WgArgs a) {
  typedef ThinCfg<CIN, COUT> C;
  constexpr int TS = C::TS, XS = C::XS, DS = C::DS, CAP = C::CAP, HS = C::HS, PER = C::PER, NJ = C::NJ, BC = C::BC;
  __shared__ __attribute__((aligned(16))) float xs[(CAP + 1) * XS];   // row CAP = zeros (absent neighbour)
  __shared__ __attribute__((aligned(16))) float ds[TS * DS];
  __shared__ int hkey[HS];
  __shared__ unsigned short hidx[HS];
  __shared__ unsigned short nbr[27 * TS];
  __shared__ int rowlist[HS];
  __shared__ int n_rows;
  const int D = a.Din;
  const int p = threadIdx.x;
  const int64_t n_tiles = (a.n_out + TS - 1) / TS;
  // this thread's (offset, block) pairs
  int kk[PER], ci0[PER], co0[PER];
  float acc[PER][4][BC];
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int q = p + 256 * e;
    const int qq = q < C::NPAIR ? q : 0;
    kk[e] = qq / C::NB;
    const int blk = qq % C::NB;
    ci0[e] = (blk / (COUT / BC)) * 4;
    co0[e] = (blk % (COUT / BC)) * BC;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
#pragma unroll
      for (int r = 0; r < BC; ++r) acc[e][qd][r] = 0.f;
  }
  for (int e = p; e < XS; e += 256) xs[CAP * XS + e] = 0.f;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // ---- A: reset the set, stage dy
    for (int e = p; e < HS; e += 256) hkey[e] = -1;
    if (p == 0) n_rows = 0;
    for (int e = p; e < TS * COUT; e += 256) {
      const int s_ = e / COUT, c = e % COUT;
      const int64_t i = tile * TS + s_;
      ds[s_ * DS + c] = i < a.n_out ? a.dy[i * COUT + c] : 0.f;
    }
    __syncthreads();
    // ---- B1: reference q = p + 256 j = (offset q / TS, site q % TS): table lookup, row id into the hash set
    int my_row[NJ], my_slot[NJ];
    // Round 6 (submanifold / stride-2 windows): the three z-neighbours of a column come as one 12-byte table load
    // (common.h surf_table_column3) - 9 TS requests instead of 27 TS - staged through the row cache's LDS (free until phase C);
    // -2 marks the columns of a site on the lattice's z border: those entries are looked up one by one as before.
    const bool columns = SURF_SPCONV_TRIPLE && a.mode != MODE_UP;
    int* __restrict__ stage = reinterpret_cast<int*>(xs);         // 27 TS ints < CAP XS: the zero row at CAP XS stays untouched
    static_assert(27 * TS <= CAP * XS, "the staged table entries fit below the row cache's zero row");
    if (columns) {
      for (int q = p; q < 9 * TS; q += 256) {
        const int jc = q / TS, st = q % TS;
        const int64_t si = tile * TS + st;
        I3u t3 = {-1, -1, -1};
        if (si < a.n_out) {
          const int f = a.mode == MODE_DOWN ? 2 : 1;
          const int bx = f * a.out_coords[si * 3 + 0], by = f * a.out_coords[si * 3 + 1], bz = f * a.out_coords[si * 3 + 2];
          if (bz - 1 >= 0 && bz + 1 < D) t3 = surf_table_column3(a.in_table, D, bx + jc % 3 - 1, by + jc / 3 - 1, bz);
          else t3 = I3u{-2, -2, -2};
        }
        stage[jc * TS + st] = t3.a;
        stage[(9 + jc) * TS + st] = t3.b;
        stage[(18 + jc) * TS + st] = t3.c;
      }
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {                                // all table lookups first: independent loads in flight together
      const int q = p + 256 * j;
      const int k = q / TS, st = q % TS;
      const int64_t si = tile * TS + st;
      int row = -1;
      const int staged = (columns && k < 27) ? stage[q] : -2;
      if (staged != -2) row = staged;
      else if (k < 27 && si < a.n_out) {
        const int cx = a.out_coords[si * 3 + 0], cy = a.out_coords[si * 3 + 1], cz = a.out_coords[si * 3 + 2];
        const int ox = k % 3 - 1, oy = (k / 3) % 3 - 1, oz = k / 9 - 1;
        int x, y, z;
        bool ok = true;
        if (a.mode == MODE_SUBM) { x = cx + ox; y = cy + oy; z = cz + oz; }
        else if (a.mode == MODE_DOWN) { x = 2 * cx + ox; y = 2 * cy + oy; z = 2 * cz + oz; }
        else {
          const int tx = cx - ox, ty = cy - oy, tz = cz - oz;
          ok = ((tx | ty | tz) & 1) == 0;
          x = tx >> 1; y = ty >> 1; z = tz >> 1;
        }
        ok = ok && x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D;
        if (ok) row = a.in_table[((int64_t)x * D + y) * D + z];
      }
      my_row[j] = row;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {                                // then the set insertions
      const int row = my_row[j];
      int slot = -1;
      if (row >= 0) {
        unsigned hsh = ((unsigned)row * 2654435761u) >> 21;
#pragma unroll 1
        for (int probe = 0; probe < 48; ++probe) {
          const int s_ = (int)((hsh + probe) & (HS - 1));
          int cur = hkey[s_];
          if (cur == -1) { cur = atomicCAS(&hkey[s_], -1, row); if (cur == -1) cur = row; }
          if (cur == row) { slot = s_; break; }
        }
      }
      my_slot[j] = slot;
    }
    __syncthreads();
    // ---- B2: dense indices for the distinct rows
    for (int e = p; e < HS; e += 256) {
      const int key = hkey[e];
      if (key >= 0) {
        const int idx = atomicAdd(&n_rows, 1);
        hidx[e] = (unsigned short)idx;
        rowlist[idx] = key;
      }
    }
    __syncthreads();
    // ---- B3: references -> dense row indices.  A reference that found no slot (a full set: > 2,048 distinct rows in one tile)
    // is served here and now by its own thread with C_in C_out global float atomics: always correct, practically never taken
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int q = p + 256 * j;
      const int k = q / TS, st = q % TS;
      if (k < 27) {
        int ref = 0xffff;                                         // absent
        if (my_row[j] >= 0) {
          if (my_slot[j] >= 0) ref = (int)hidx[my_slot[j]];
          else {
            float* dst = a.dW + (int64_t)k * CIN * COUT;
            const float* xr = a.x + (int64_t)my_row[j] * CIN;
            for (int ci = 0; ci < CIN; ++ci) {
              const float xv = xr[ci];
              for (int co = 0; co < COUT; ++co) atomicAdd(dst + ci * COUT + co, xv * ds[st * DS + co]);
            }
          }
        }
        nbr[q] = (unsigned short)ref;
      }
    }
    // ---- C + D in rounds of CAP cached rows (one round unless the tile references more distinct rows than the cache holds)
    const int total = n_rows;
    for (int base = 0; base < total || base == 0; base += CAP) {
      const int nr = total - base < CAP ? total - base : CAP;
      if (base > 0) __syncthreads();                              // the previous round's reads of xs
      for (int e = p; e < nr * (CIN / 4); e += 256) {
        const int idx = e / (CIN / 4), c4 = e % (CIN / 4);
        *reinterpret_cast<f32x4*>(xs + idx * XS + c4 * 4) =
            *reinterpret_cast<const f32x4*>(a.x + (int64_t)rowlist[base + idx] * CIN + c4 * 4);
      }
      __syncthreads();
#pragma unroll 2
      for (int t = 0; t < TS; ++t) {
#pragma unroll
        for (int e = 0; e < PER; ++e) {
          const int idx = (int)nbr[kk[e] * TS + t] - base;
          const int ref = (unsigned)idx < (unsigned)CAP ? idx : CAP;        // outside this round's window (or absent): zeros
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + ref * XS + ci0[e]);
          float dv[BC];
          if constexpr (BC == 4) {
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(ds + t * DS + co0[e]);
            dv[0] = d4[0]; dv[1] = d4[1]; dv[2] = d4[2]; dv[3] = d4[3];
          } else {
            dv[0] = ds[t * DS + co0[e]];
            dv[1] = ds[t * DS + co0[e] + 1];
          }
#pragma unroll
          for (int qd = 0; qd < 4; ++qd)
#pragma unroll
            for (int r = 0; r < BC; ++r) acc[e][qd][r] = fmaf(xv[qd], dv[r], acc[e][qd][r]);
        }
      }
      if (total == 0) break;
    }
    __syncthreads();
  }
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    if (p + 256 * e < C::NPAIR) {
      float* dst = a.dW + (int64_t)kk[e] * CIN * COUT;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
#pragma unroll
        for (int r = 0; r < BC; ++r)
          if (acc[e][qd][r] != 0.f) atomicAdd(dst + (ci0[e] + qd) * COUT + co0[e] + r, acc[e][qd][r]);
    }
  }
}

}  // namespace

#define WG_CASES(X) X(8, 8) X(16, 8) X(8, 16) X(16, 16) X(16, 32) X(32, 32) X(32, 64) X(64, 64) X(64, 32) X(32, 16)

extern "C" int surf_spconv_wgrad(const float* x, int cin, const int32_t* in_table, int D_in, const int32_t* out_coords,
                                 int64_t n_out, int mode, const float* dy, int cout, float* dW, void* stream) {
  if (!x || !in_table || !out_coords || !dy || !dW || n_out <= 0 || D_in < 1 || mode < 0 || mode > 2) return SURF_E_ARG;
  WgArgs a;
  a.x = x; a.in_table = in_table; a.Din = D_in; a.out_coords = out_coords; a.n_out = n_out; a.mode = mode; a.dy = dy; a.dW = dW;
  const bool thin = cin + cout <= 48 && mode != MODE_DOWN && !getenv("SURF_WGRAD_CLASSIC");    // (env: A/B timing of the round-2 kernel)
  const int ts = thin ? 128 : ((cin + cout <= 48) ? 256 : 64);
  const int64_t tiles = (n_out + ts - 1) / ts;
  const unsigned grid = (unsigned)(tiles < 1024 ? tiles : 1024);
  // the per-offset kernel (wide pairs, stride-2 thin pairs): one offset per workgroup row, >= 4 tiles per workgroup where there
  // are that many (each workgroup ends with one atomic per entry of its offset's slice), <= 256 x 27 workgroups
#ifndef SURF_WGRAD_WIDE_KY
#define SURF_WGRAD_WIDE_KY 27
#endif
  const int64_t gx = tiles / 4 < 1 ? 1 : (tiles / 4 > 256 ? 256 : tiles / 4);
  const dim3 wide_grid = SURF_WGRAD_WIDE_KY > 1 ? dim3((unsigned)gx, SURF_WGRAD_WIDE_KY) : dim3(grid);
#define X(CI, CO)                                                                                                \
  if (cin == CI && cout == CO) {                                                                                 \
    if constexpr (CI + CO <= 48) {                                                                               \
      if (thin) {                                                                                                \
        /* one workgroup per ThinCfg<CI, CO>::TS-site tile up to 1,024 (the kernel strides over the rest) */      \
        const int64_t ttiles = (n_out + ThinCfg<CI, CO>::TS - 1) / ThinCfg<CI, CO>::TS;                           \
        hipLaunchKernelGGL((spconv_wgrad_thin_kernel<CI, CO>), dim3((unsigned)(ttiles < 1024 ? ttiles : 1024)), dim3(256), 0, \
                           (hipStream_t)stream, a);                                                              \
        return surf_check_launch();                                                                              \
      }                                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL((spconv_wgrad_kernel<CI, CO>), wide_grid, dim3(256), 0, (hipStream_t)stream, a);           \
    return surf_check_launch();                                                                                  \
  }
  WG_CASES(X)
#undef X
  return SURF_E_LIMIT;
}
