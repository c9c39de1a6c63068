// First-hit face ids of a triangle mesh seen from a pinhole camera, by z-buffer rasterisation.
//
// Used by the mesh cleaning of the evaluation (utils/clean_mesh.py:37-108, evaluation/clean_mesh.py:187-262 of the
// reference, which casts one ray per (up-scaled) mask pixel with trimesh + pyembree and keeps the faces that are some ray's
// first hit).  The first hit of the ray through a sample position is the nearest triangle covering that position, so the
// same face set comes out of a z-buffer: one thread per triangle, perspective-correct depth at every covered sample,
// 64-bit atomicMin on (depth bits << 32 | face id).  Sample (i, j) sits at pixel (j (w-1)/(Wup-1), i (h-1)/(Hup-1)), the
// reference's torch.linspace(0, w-1, w*upscale) lattice.  Offline tool: HBM-atomic bound, not part of the render hot path.
#include <math.h>

#include "common.h"

namespace {

struct RasterArgs {
  const float* vertices;
  const int32_t* faces;
  int64_t nf;
  float K[9], w2c[12];
  int h, w, Hup, Wup;
  unsigned long long* zbuf;
};

__global__ __launch_bounds__(256) void raster_kernel(RasterArgs a) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= a.nf) return;
  float sx[3], sy[3], iz[3];
  const float kx = (float)(a.Wup - 1) / (float)(a.w - 1), ky = (float)(a.Hup - 1) / (float)(a.h - 1);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int64_t v = a.faces[f * 3 + k];
    const float X = a.vertices[v * 3 + 0], Y = a.vertices[v * 3 + 1], Z = a.vertices[v * 3 + 2];
    const float cx = a.w2c[0] * X + a.w2c[1] * Y + a.w2c[2] * Z + a.w2c[3];
    const float cy = a.w2c[4] * X + a.w2c[5] * Y + a.w2c[6] * Z + a.w2c[7];
    const float cz = a.w2c[8] * X + a.w2c[9] * Y + a.w2c[10] * Z + a.w2c[11];
    if (!(cz > 1e-6f)) return;  // behind the camera: no near clipping (cameras look at the scene from outside)
    const float u = (a.K[0] * cx + a.K[1] * cy + a.K[2] * cz) / cz, v2 = (a.K[3] * cx + a.K[4] * cy + a.K[5] * cz) / cz;
    sx[k] = u * kx;
    sy[k] = v2 * ky;
    iz[k] = 1.0f / cz;
  }
  const float area = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sx[2] - sx[0]) * (sy[1] - sy[0]);
  if (area == 0.f) return;
  const int x0 = max(0, (int)ceilf(fminf(sx[0], fminf(sx[1], sx[2])))), x1 = min(a.Wup - 1, (int)floorf(fmaxf(sx[0], fmaxf(sx[1], sx[2]))));
  const int y0 = max(0, (int)ceilf(fminf(sy[0], fminf(sy[1], sy[2])))), y1 = min(a.Hup - 1, (int)floorf(fmaxf(sy[0], fmaxf(sy[1], sy[2]))));
  const float inv_area = 1.0f / area;
  for (int y = y0; y <= y1; ++y)
    for (int x = x0; x <= x1; ++x) {
      const float px = (float)x, py = (float)y;
      const float b0 = ((sx[1] - px) * (sy[2] - py) - (sx[2] - px) * (sy[1] - py)) * inv_area;
      const float b1 = ((sx[2] - px) * (sy[0] - py) - (sx[0] - px) * (sy[2] - py)) * inv_area;
      const float b2 = 1.0f - b0 - b1;
      if (b0 < 0.f || b1 < 0.f || b2 < 0.f) continue;
      const float z = 1.0f / (b0 * iz[0] + b1 * iz[1] + b2 * iz[2]);
      const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)(unsigned)f;
      atomicMin(&a.zbuf[(int64_t)y * a.Wup + x], key);
    }
}

}  // namespace

extern "C" int surf_raster_first_hit(const float* vertices, const int32_t* faces, int64_t n_faces, const float* h_K,
                                     const float* h_w2c, int h, int w, int Hup, int Wup, unsigned long long* zbuf, void* stream) {
  if (!vertices || !faces || !h_K || !h_w2c || !zbuf || n_faces <= 0) return SURF_E_ARG;
  if (h < 2 || w < 2 || Hup < 2 || Wup < 2) return SURF_E_ARG;
  RasterArgs a;
  a.vertices = vertices; a.faces = faces; a.nf = n_faces; a.h = h; a.w = w; a.Hup = Hup; a.Wup = Wup; a.zbuf = zbuf;
  for (int i = 0; i < 9; ++i) a.K[i] = h_K[i];
  for (int i = 0; i < 12; ++i) a.w2c[i] = h_w2c[i];
  hipLaunchKernelGGL(raster_kernel, dim3((unsigned)((n_faces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return surf_check_launch();
}
