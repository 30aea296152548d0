// segnet.hip — what Cellpose does immediately before and after the network forward:
// percentile normalisation, 224-px overlapped tiling and sigmoid-taper blending.
//
// Reference call site: `model.eval(pixels, normalize=True, ...)` at
// src/aliby/segment/dispatch.py:208-215 (cellpose 4.0.6 not vendored; restated from
// cellpose.transforms.normalize99 / pad_image_ND / make_tiles / average_tiles, U-Net family:
// bsize=224, tile_overlap=0.1, zero padding to a multiple of 16 plus 8 px per side).
//
// All three are streaming passes (HBM-bound).  The 1st/99th percentiles are exact order statistics:
// a 65536-bin histogram of the uint16 plane + a scan, interpolated like numpy.percentile("linear").
#include "common.h"

typedef unsigned short u16;

// ---- histogram ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hist_u16(const u16* __restrict__ img, size_t plane, int* __restrict__ hist) {
  const int f = blockIdx.y;
  const u16* p = img + (size_t)f * plane;
  int* h = hist + (size_t)f * 65536;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x)
    atomicAdd(&h[p[i]], 1);
}

// one workgroup (1024 threads) per image: cumulative histogram -> order statistics -> percentiles
__global__ __launch_bounds__(1024) void k_percentiles(const int* __restrict__ hist, size_t plane, double qlo, double qhi,
                                                      double* __restrict__ out) {
  __shared__ long long part[1024];
  __shared__ double vals[4];
  const int f = blockIdx.x, t = threadIdx.x;
  const int* h = hist + (size_t)f * 65536;
  long long c = 0;
  for (int k = 0; k < 64; ++k) c += h[t * 64 + k];
  part[t] = c;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const long long v = (t >= o) ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  const long long before = part[t] - c;  // pixels with value < t*64
  // ranks wanted: floor(pos) and floor(pos)+1 for both percentiles
  const double n1 = (double)(plane - 1);
  const double pos[2] = {n1 * (qlo / 100.0), n1 * (qhi / 100.0)};
  for (int q = 0; q < 2; ++q) {
    const long long lo = (long long)floor(pos[q]);
    long long hi = lo + 1;
    if (hi > (long long)plane - 1) hi = (long long)plane - 1;
    const long long want[2] = {lo, hi};
    for (int w = 0; w < 2; ++w) {
      const long long r = want[w];
      if (r >= before && r < before + c) {
        long long run = before;
        for (int k = 0; k < 64; ++k) {
          run += h[t * 64 + k];
          if (r < run) { vals[q * 2 + w] = (double)(t * 64 + k); break; }
        }
      }
    }
  }
  __syncthreads();
  if (t == 0) {
    for (int q = 0; q < 2; ++q) {
      const double a = vals[q * 2], b = vals[q * 2 + 1];
      const double tt = pos[q] - floor(pos[q]);
      out[f * 2 + q] = a + (b - a) * tt;
    }
  }
}

__global__ void k_normalize99(const u16* __restrict__ img, size_t plane, int F, const double* __restrict__ pct,
                              float* __restrict__ out) {
  const size_t total = plane * F;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t f = i / plane;
    const double x01 = pct[f * 2], x99 = pct[f * 2 + 1];
    float v = 0.0f;
    if (x99 - x01 > 1e-3) v = (float)(((double)img[i] - x01) / (x99 - x01));
    out[i] = v;
  }
}

// ---- tiling --------------------------------------------------------------------------------------
struct TileGeom {
  int F, Y, X;          // normalised images [F,Y,X]
  int ypad1, xpad1;     // zero padding before
  int Ly, Lx;           // padded size
  int by, bx;           // tile size
  int ny, nx;
  int nchan;            // network input channels (channel 0 = image, the rest zero)
};

__global__ void k_make_tiles(const float* __restrict__ img, TileGeom g, const int* __restrict__ ystart,
                             const int* __restrict__ xstart, float* __restrict__ tiles) {
  const size_t per_tile = (size_t)g.nchan * g.by * g.bx;
  const size_t total = (size_t)g.F * g.ny * g.nx * per_tile;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t t = i / per_tile, rem = i % per_tile;
    const int ch = (int)(rem / ((size_t)g.by * g.bx));
    const int r = (int)((rem / g.bx) % g.by), c = (int)(rem % g.bx);
    const int f = (int)(t / (g.ny * g.nx)), k = (int)(t % (g.ny * g.nx));
    float v = 0.0f;
    if (ch == 0) {
      const int y = ystart[k / g.nx] + r - g.ypad1, x = xstart[k % g.nx] + c - g.xpad1;
      if (y >= 0 && y < g.Y && x >= 0 && x < g.X) v = img[((size_t)f * g.Y + y) * g.X + x];
    }
    tiles[i] = v;
  }
}

// net output tiles [F*ny*nx, 3, by, bx] -> dP [F,2,Y,X], cellprob [F,Y,X]: taper-weighted average,
// accumulated in tile order (float32), cropped back to the unpadded image
__global__ void k_average_tiles(const float* __restrict__ ytiles, TileGeom g, const int* __restrict__ ystart,
                                const int* __restrict__ xstart, const float* __restrict__ taper,
                                float* __restrict__ dP, float* __restrict__ prob) {
  const size_t P = (size_t)g.Y * g.X;
  const size_t total = (size_t)g.F * P;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int f = (int)(i / P);
    const int y = (int)((i % P) / g.X), x = (int)(i % g.X);
    const int yp = y + g.ypad1, xp = x + g.xpad1;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, nav = 0.0f;
    for (int j = 0; j < g.ny; ++j) {
      const int r = yp - ystart[j];
      if (r < 0 || r >= g.by) continue;
      for (int q = 0; q < g.nx; ++q) {
        const int c = xp - xstart[q];
        if (c < 0 || c >= g.bx) continue;
        const float m = taper[r * g.bx + c];
        const size_t base = (((size_t)f * g.ny * g.nx + (size_t)j * g.nx + q) * 3) * g.by * g.bx + (size_t)r * g.bx + c;
        a0 = a0 + ytiles[base] * m;
        a1 = a1 + ytiles[base + (size_t)g.by * g.bx] * m;
        a2 = a2 + ytiles[base + 2 * (size_t)g.by * g.bx] * m;
        nav = nav + m;
      }
    }
    dP[((size_t)f * 2 + 0) * P + (i % P)] = a0 / nav;
    dP[((size_t)f * 2 + 1) * P + (i % P)] = a1 / nav;
    prob[i] = a2 / nav;
  }
}

// Z max-projection of one channel: pixels [F,C,Z,Y,X] u16 -> [F,Y,X]  (dispatch.py:192,199-206)
__global__ void k_select_project(const u16* __restrict__ px, int F, int C, int Z, size_t plane, int channel,
                                 u16* __restrict__ out) {
  const size_t total = (size_t)F * plane;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t f = i / plane, p = i % plane;
    const u16* src = px + (((size_t)f * C + channel) * Z) * plane + p;
    u16 m = src[0];
    for (int z = 1; z < Z; ++z) { const u16 v = src[(size_t)z * plane]; m = v > m ? v : m; }
    out[i] = m;
  }
}

static inline int grid_for(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

extern "C" {

int aliby_select_project_u16(aliby_ctx* ctx, const uint16_t* pixels, int F, int C, int Z, int Y, int X, int channel,
                             uint16_t* out, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  ARG_CHECK(F >= 0 && C > 0 && Z > 0 && Y > 0 && X > 0, "bad shape");
  ARG_CHECK(channel >= 0 && channel < C, "channel out of range");
  if (F == 0) return ALIBY_OK;
  ARG_CHECK(pixels && out, "NULL argument");
  const size_t plane = (size_t)Y * X;
  hipLaunchKernelGGL(k_select_project, dim3(grid_for(plane * F)), dim3(256), 0, as_stream(stream), pixels, F, C, Z, plane,
                     channel, out);
  KERNEL_CHECK();
  return ALIBY_OK;
}

int aliby_normalize99_u16(aliby_ctx* ctx, const uint16_t* img, int F, int Y, int X, double lower, double upper,
                          float* out, double* percentiles_dev, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  ARG_CHECK(F >= 0 && Y > 0 && X > 0, "bad shape");
  ARG_CHECK(lower >= 0 && upper <= 100 && lower < upper, "0 <= lower < upper <= 100");
  if (F == 0) return ALIBY_OK;
  ARG_CHECK(img && out && percentiles_dev, "NULL argument");
  const size_t plane = (size_t)Y * X;
  int rc = aliby_ensure_scratch(ctx, sizeof(int) * 65536 * (size_t)F);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  int* hist = (int*)ctx->scratch;
  HIP_TRY(hipMemsetAsync(hist, 0, sizeof(int) * 65536 * (size_t)F, s));
  int bx = (int)((plane + 255) / 256);
  if (bx > 512) bx = 512;
  hipLaunchKernelGGL(k_hist_u16, dim3(bx, F), dim3(256), 0, s, img, plane, hist);
  KERNEL_CHECK();
  hipLaunchKernelGGL(k_percentiles, dim3(F), dim3(1024), 0, s, hist, plane, lower, upper, percentiles_dev);
  KERNEL_CHECK();
  hipLaunchKernelGGL(k_normalize99, dim3(grid_for(plane * F)), dim3(256), 0, s, img, plane, F, percentiles_dev, out);
  KERNEL_CHECK();
  return ALIBY_OK;
}

int aliby_make_tiles(aliby_ctx* ctx, const float* img, int F, int Y, int X, int ypad1, int xpad1, int Ly, int Lx,
                     int by, int bx, int ny, int nx, const int32_t* ystart_dev, const int32_t* xstart_dev,
                     int nchan, float* tiles, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (F == 0) return ALIBY_OK;
  ARG_CHECK(img && ystart_dev && xstart_dev && tiles, "NULL argument");
  ARG_CHECK(by > 0 && bx > 0 && ny > 0 && nx > 0 && nchan > 0 && by <= Ly && bx <= Lx, "bad tile geometry");
  TileGeom g{F, Y, X, ypad1, xpad1, Ly, Lx, by, bx, ny, nx, nchan};
  const size_t total = (size_t)F * ny * nx * nchan * by * bx;
  hipLaunchKernelGGL(k_make_tiles, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), img, g, ystart_dev, xstart_dev, tiles);
  KERNEL_CHECK();
  return ALIBY_OK;
}

int aliby_average_tiles(aliby_ctx* ctx, const float* ytiles, int F, int Y, int X, int ypad1, int xpad1, int Ly,
                        int Lx, int by, int bx, int ny, int nx, const int32_t* ystart_dev,
                        const int32_t* xstart_dev, const float* taper_dev, float* dP, float* cellprob,
                        void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (F == 0) return ALIBY_OK;
  ARG_CHECK(ytiles && ystart_dev && xstart_dev && taper_dev && dP && cellprob, "NULL argument");
  ARG_CHECK(by > 0 && bx > 0 && ny > 0 && nx > 0, "bad tile geometry");
  TileGeom g{F, Y, X, ypad1, xpad1, Ly, Lx, by, bx, ny, nx, 3};
  hipLaunchKernelGGL(k_average_tiles, dim3(grid_for((size_t)F * Y * X)), dim3(256), 0, as_stream(stream), ytiles, g,
                     ystart_dev, xstart_dev, taper_dev, dP, cellprob);
  KERNEL_CHECK();
  return ALIBY_OK;
}

}  // extern "C"
