// segnet.hip — what Cellpose does immediately before and after the network forward:
// percentile normalisation, 224-px overlapped tiling and sigmoid-taper blending.
//
// Reference call site: `model.eval(pixels, normalize=True, ...)` at
// src/aliby/segment/dispatch.py:208-215 (cellpose 4.0.6 not vendored; restated from
// cellpose.transforms.normalize99 / pad_image_ND / make_tiles / average_tiles, U-Net family:
// bsize=224, tile_overlap=0.1, zero padding to a multiple of 16 plus 8 px per side).
//
// All three are streaming passes (HBM-bound).  The 1st/99th percentiles are exact order statistics:
// found by a two-pass radix select over the uint16 plane (below), interpolated like numpy.percentile("linear").
#include "common.h"

typedef unsigned short u16;

// ---- exact percentiles by two-pass radix select -----------------------------------------------------
// Pass 1: 256-bin histogram of the HIGH byte (LDS-privatised; a wave whose pixels share one key — a fluorescence
// background — adds once).  k_pick_buckets locates the (at most 4)
// buckets holding ranks floor(pos), floor(pos)+1 of both percentiles.  Pass 2: 256-bin histogram of the LOW
// byte inside those buckets.  k_percentiles_radix reads the order statistics and interpolates like
// numpy.percentile(method="linear").
struct alignas(16) u16x8s { u16 v[8]; };

__device__ __forceinline__ void wave_hist_add(int* h, int key, bool valid) {
  const int lane = threadIdx.x & 63;
  const unsigned long long active = __ballot(valid);
  if (!active) return;
  // one key for the whole wave (background): one atomic; otherwise every lane adds its own — merging key by key costs a
  // ballot round trip per distinct key, and a textured image has ~15 distinct high bytes among a wave's 64 pixels
  const int lead = __ffsll((long long)active) - 1;
  const int k = __shfl(key, lead, 64);
  const unsigned long long m = __ballot(valid && key == k);
  if (m == active) {
    if (lane == lead) atomicAdd(&h[k], (int)__popcll(m));
  } else if (valid) {
    atomicAdd(&h[key], 1);
  }
}

__global__ __launch_bounds__(256) void k_hist_hi(const u16* __restrict__ img, size_t plane, int* __restrict__ hist_hi) {
  __shared__ int h[256];
  const int f = blockIdx.y;
  const u16* p = img + (size_t)f * plane;
  h[threadIdx.x] = 0;
  __syncthreads();
  const size_t nvec = plane / 8;
  const bool aligned = ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x; i0 < (aligned ? nvec : 0); i0 += (size_t)gridDim.x * blockDim.x) {
    const size_t i = i0 + threadIdx.x;
    u16x8s v;
    const bool ok = i < nvec;
    if (ok) v = *reinterpret_cast<const u16x8s*>(p + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) wave_hist_add(h, ok ? (v.v[k] >> 8) : 0, ok);
  }
  const size_t tail0 = aligned ? nvec * 8 : 0;
  for (size_t i0 = tail0 + (size_t)blockIdx.x * blockDim.x; i0 < plane; i0 += (size_t)gridDim.x * blockDim.x) {
    const size_t i = i0 + threadIdx.x;
    const bool ok = i < plane;
    wave_hist_add(h, ok ? (p[i] >> 8) : 0, ok);
  }
  __syncthreads();
  const int c = h[threadIdx.x];
  if (c) atomicAdd(&hist_hi[f * 256 + threadIdx.x], c);
}

// one workgroup (256 threads) per image: prefix of the high-byte histogram -> buckets of the 4 wanted ranks
__global__ __launch_bounds__(256) void k_pick_buckets(const int* __restrict__ hist_hi, size_t plane, double qlo, double qhi,
                                                      int* __restrict__ bucket, long long* __restrict__ before) {
  __shared__ long long pre[256];
  const int f = blockIdx.x, t = threadIdx.x;
  const long long c = hist_hi[f * 256 + t];
  pre[t] = c;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const long long v = (t >= o) ? pre[t - o] : 0;
    __syncthreads();
    pre[t] += v;
    __syncthreads();
  }
  const long long b4 = pre[t] - c;
  const double n1 = (double)(plane - 1);
  const double pos[2] = {n1 * (qlo / 100.0), n1 * (qhi / 100.0)};
  for (int q = 0; q < 2; ++q) {
    const long long lo = (long long)floor(pos[q]);
    long long hi = lo + 1;
    if (hi > (long long)plane - 1) hi = (long long)plane - 1;
    const long long want[2] = {lo, hi};
    for (int w = 0; w < 2; ++w)
      if (c > 0 && want[w] >= b4 && want[w] < b4 + c) { bucket[f * 4 + q * 2 + w] = t; before[f * 4 + q * 2 + w] = b4; }
  }
}

__global__ __launch_bounds__(256) void k_hist_lo(const u16* __restrict__ img, size_t plane, const int* __restrict__ bucket,
                                                 int* __restrict__ hist_lo) {
  __shared__ int h[4 * 256];
  const int f = blockIdx.y;
  const u16* p = img + (size_t)f * plane;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) h[i] = 0;
  int b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) b[k] = bucket[f * 4 + k];
  // duplicates among the 4 buckets are served by the first slot holding that bucket
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x) {
    const int v = p[i], hb = v >> 8;
    int slot = -1;
#pragma unroll
    for (int k = 3; k >= 0; --k) if (hb == b[k]) slot = k;
    if (slot >= 0) atomicAdd(&h[slot * 256 + (v & 255)], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) { const int c = h[i]; if (c) atomicAdd(&hist_lo[f * 1024 + i], c); }
}

__global__ __launch_bounds__(64) void k_percentiles_radix(const int* __restrict__ hist_lo, const int* __restrict__ bucket,
                                                          const long long* __restrict__ before, size_t plane, double qlo,
                                                          double qhi, double* __restrict__ out) {
  const int f = blockIdx.x;
  if (threadIdx.x != 0) return;
  const double n1 = (double)(plane - 1);
  const double pos[2] = {n1 * (qlo / 100.0), n1 * (qhi / 100.0)};
  double vals[4];
  for (int q = 0; q < 2; ++q) {
    const long long lo = (long long)floor(pos[q]);
    long long hi = lo + 1;
    if (hi > (long long)plane - 1) hi = (long long)plane - 1;
    const long long want[2] = {lo, hi};
    for (int w = 0; w < 2; ++w) {
      const int k = q * 2 + w, bk = bucket[f * 4 + k];
      int slot = k;
      for (int j = 3; j >= 0; --j) if (bucket[f * 4 + j] == bk) slot = j;  // first slot holding this bucket
      long long run = before[f * 4 + k];
      int low = 255;
      for (int j = 0; j < 256; ++j) { run += hist_lo[f * 1024 + slot * 256 + j]; if (want[w] < run) { low = j; break; } }
      vals[k] = (double)((bk << 8) | low);
    }
  }
  for (int q = 0; q < 2; ++q) {
    const double a = vals[q * 2], b = vals[q * 2 + 1], tt = pos[q] - floor(pos[q]);
    out[f * 2 + q] = a + (b - a) * tt;
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
  // scratch: hist_hi int[F*256] | hist_lo int[F*1024] | bucket int[F*4] | before i64[F*4]
  const size_t b_hi = sizeof(int) * 256 * (size_t)F, b_lo = sizeof(int) * 1024 * (size_t)F, b_bk = sizeof(int) * 4 * (size_t)F;
  const size_t b_bf = sizeof(long long) * 4 * (size_t)F;
  int rc = aliby_ensure_scratch(ctx, b_hi + b_lo + ((b_bk + 15) & ~(size_t)15) + b_bf);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  int* hist_hi = (int*)ctx->scratch;
  int* hist_lo = hist_hi + 256 * (size_t)F;
  int* bucket = hist_lo + 1024 * (size_t)F;
  long long* before = (long long*)((unsigned char*)ctx->scratch + b_hi + b_lo + ((b_bk + 15) & ~(size_t)15));
  HIP_TRY(hipMemsetAsync(hist_hi, 0, b_hi + b_lo, s));
  int bx = (int)((plane / 8 + 255) / 256);
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(k_hist_hi, dim3(bx, F), dim3(256), 0, s, img, plane, hist_hi);
  KERNEL_CHECK();
  hipLaunchKernelGGL(k_pick_buckets, dim3(F), dim3(256), 0, s, hist_hi, plane, lower, upper, bucket, before);
  KERNEL_CHECK();
  hipLaunchKernelGGL(k_hist_lo, dim3(bx, F), dim3(256), 0, s, img, plane, bucket, hist_lo);
  KERNEL_CHECK();
  hipLaunchKernelGGL(k_percentiles_radix, dim3(F), dim3(64), 0, s, hist_lo, bucket, before, plane, lower, upper, percentiles_dev);
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
