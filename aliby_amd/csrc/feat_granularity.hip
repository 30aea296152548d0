// feat_granularity.hip — cp_measure "granularity" (CellProfiler MeasureGranularity; bound at
// src/extraction/core/functions/loaders.py:71-73 like every core measurement, outside the builder's default list).
//
// The reference evaluates it once per object on a full-frame single-object label image (extract.py:147-153).  Everything up
// to the per-object means is a property of the IMAGE (with CellProfiler's default, whole-frame image mask): the 4x
// subsampled frame, its background (erode + dilate with disk(10) at another 4x), and the 16 erode-by-disk(1) /
// reconstruct-by-dilation rounds of the granular spectrum.  So one call does that once per (tile, channel) and finishes
// every object of every tile with 17 per-object mean kernels:
//
//   sample (bilinear, map_coordinates order 1)  ->  background (masked erode, masked dilate, bilinear back, subtract, clamp)
//   -> for i in 1..L:  ero = erode(ero, disk(1));  rec = reconstruct(ero under pix)  [Jacobi sweeps to the fixed point];
//                      mean_i(object) = mean over the object's pixels of rec resized to the frame (bilinear);
//                      Granularity_i = (mean_{i-1} - mean_i) * 100 / max(mean_0, eps),  mean_0 on the ORIGINAL pixels.
//
// float64 throughout (CellProfiler images are float64); per-object sums in a fixed order (deterministic).  Restated in
// oracle/granularity_restated.py; its primitives are pinned against scikit-image 0.18.3, the measurement as a whole is
// PARITY UNPINNED (cp_measure is not available offline).
#include "common.h"

typedef unsigned short u16;

namespace {

struct GranGeom {
  int F, Y, X;    // frame
  int sh, sw;     // subsampled
  int bh, bw;     // background grid
};

// scipy.ndimage.map_coordinates(order=1, mode="constant", cval=0): linear interpolation inside [0, n-1], 0 outside
template <typename LoadF>
__device__ __forceinline__ double bilinear(LoadF load, int n0, int n1, double y, double x) {
  if (!(y >= 0.0 && y <= (double)(n0 - 1) && x >= 0.0 && x <= (double)(n1 - 1))) return 0.0;
  const int y0 = (int)floor(y), x0 = (int)floor(x);
  const int y1 = min(y0 + 1, n0 - 1), x1 = min(x0 + 1, n1 - 1);
  const double fy = y - (double)y0, fx = x - (double)x0;
  const double a = load(y0, x0), b = load(y0, x1), c = load(y1, x0), d = load(y1, x1);
  return (a * (1.0 - fx) + b * fx) * (1.0 - fy) + (c * (1.0 - fx) + d * fx) * fy;
}

template <typename T>
__global__ void k_gran_sample_frame(const T* __restrict__ planes, int C, int channel, GranGeom g, double inv, double* __restrict__ dst) {
  const size_t total = (size_t)g.F * g.sh * g.sw, plane = (size_t)g.Y * g.X;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % g.sw), y = (int)((i / g.sw) % g.sh), f = (int)(i / ((size_t)g.sw * g.sh));
    const T* p = planes + ((size_t)f * C + channel) * plane;
    dst[i] = bilinear([&](int yy, int xx) { return (double)px_load<T>(p, (size_t)yy * g.X + xx); }, g.Y, g.X, (double)y * inv, (double)x * inv);
  }
}

// mask of the subsampled frame when the image mask is "objects": bilinear sample of (labels > 0) > 0.9 (mask_order = 1)
__global__ void k_gran_sample_mask(const u16* __restrict__ labels, GranGeom g, double inv, unsigned char* __restrict__ dst) {
  const size_t total = (size_t)g.F * g.sh * g.sw, plane = (size_t)g.Y * g.X;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % g.sw), y = (int)((i / g.sw) % g.sh), f = (int)(i / ((size_t)g.sw * g.sh));
    const u16* p = labels + (size_t)f * plane;
    dst[i] = bilinear([&](int yy, int xx) { return p[(size_t)yy * g.X + xx] ? 1.0 : 0.0; }, g.Y, g.X, (double)y * inv, (double)x * inv) > 0.9;
  }
}

// generic [F,h,w] double -> [F,h2,w2] double resample at (i, j) * scale (order 1)
__global__ void k_gran_resample(const double* __restrict__ src, int F, int h, int w, int h2, int w2, double sy, double sx,
                                double* __restrict__ dst) {
  const size_t total = (size_t)F * h2 * w2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % w2), y = (int)((i / w2) % h2), f = (int)(i / ((size_t)w2 * h2));
    const double* p = src + (size_t)f * h * w;
    dst[i] = bilinear([&](int yy, int xx) { return p[(size_t)yy * w + xx]; }, h, w, (double)y * sy, (double)x * sx);
  }
}
__global__ void k_gran_resample_mask(const unsigned char* __restrict__ src, int F, int h, int w, int h2, int w2, double sy, double sx,
                                     unsigned char* __restrict__ dst) {
  const size_t total = (size_t)F * h2 * w2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % w2), y = (int)((i / w2) % h2), f = (int)(i / ((size_t)w2 * h2));
    const unsigned char* p = src + (size_t)f * h * w;
    dst[i] = bilinear([&](int yy, int xx) { return (double)p[(size_t)yy * w + xx]; }, h, w, (double)y * sy, (double)x * sx) > 0.9;
  }
}

__device__ __forceinline__ int reflect(int i, int n) {  // ndimage mode="reflect": d c b a | a b c d | d c b a
  while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i;
  return i;
}

// grey erosion (DILATE = false) / dilation of src restricted to the mask (pixels outside the mask count as 0, as in
// `tmp = zeros; tmp[mask] = src[mask]`), disk(radius) footprint, reflect border
template <bool DILATE>
__global__ void k_gran_morph(const double* __restrict__ src, const unsigned char* __restrict__ msk, int F, int h, int w, int radius,
                             double* __restrict__ dst) {
  const size_t total = (size_t)F * h * w;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const size_t base = i - (size_t)y * w - x;
    double best = DILATE ? -INFINITY : INFINITY;
    for (int dy = -radius; dy <= radius; ++dy) {
      const int yy = reflect(y + dy, h);
      for (int dx = -radius; dx <= radius; ++dx) {
        if (dy * dy + dx * dx > radius * radius) continue;
        const int xx = reflect(x + dx, w);
        const size_t j = base + (size_t)yy * w + xx;
        const double v = (!msk || msk[j]) ? src[j] : 0.0;
        best = DILATE ? fmax(best, v) : fmin(best, v);
      }
    }
    dst[i] = best;
  }
}

// pix = max(sub - bilinear(back), 0); ero = pix inside the mask, 0 outside
__global__ void k_gran_subtract(const double* __restrict__ sub, const double* __restrict__ back, const unsigned char* __restrict__ msk,
                                GranGeom g, int resized, double sy, double sx, double* __restrict__ pix, double* __restrict__ ero) {
  const size_t total = (size_t)g.F * g.sh * g.sw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % g.sw), y = (int)((i / g.sw) % g.sh), f = (int)(i / ((size_t)g.sw * g.sh));
    double b;
    if (resized) {
      const double* p = back + (size_t)f * g.bh * g.bw;
      b = bilinear([&](int yy, int xx) { return p[(size_t)yy * g.bw + xx]; }, g.bh, g.bw, (double)y * sy, (double)x * sx);
    } else {
      b = back[i];
    }
    const double v = fmax(sub[i] - b, 0.0);
    pix[i] = v;
    ero[i] = (!msk || msk[i]) ? v : 0.0;
  }
}

// one Jacobi sweep of the reconstruction by dilation with disk(1) (the 4-neighbourhood + centre), bounded by pix
__global__ void k_gran_recon_sweep(const double* __restrict__ in, const double* __restrict__ pix, int F, int h, int w,
                                   double* __restrict__ out, int* __restrict__ changed) {
  const size_t total = (size_t)F * h * w;
  int any = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)((i / w) % h);
    double m = in[i];
    if (y > 0) m = fmax(m, in[i - w]);
    if (y + 1 < h) m = fmax(m, in[i + w]);
    if (x > 0) m = fmax(m, in[i - 1]);
    if (x + 1 < w) m = fmax(m, in[i + 1]);
    m = fmin(m, pix[i]);
    any |= m != in[i];
    out[i] = m;
  }
  if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(changed, 1);
}

struct GranMeanArgs {
  const u16* labels;
  const aliby_object* tab;
  int n_obj;
  GranGeom g;
  const void* planes;  // STEP0: the original pixels (typed)
  int C, channel;
  const double* rec;   // STEP > 0: [F, sh, sw]
  double sy, sx;       // frame -> subsampled coordinates ((sh - 1) / (Y - 1), ...)
  double* prev;        // [n_obj] mean of the previous step (updated)
  double* start;       // [n_obj] max(mean_0, eps) (written by step 0)
  double* out;         // feature matrix
  int ld, col;         // column of this step
};

// one wave per object: mean over the object's pixels of the (resized) image, lanes stride the bbox, fixed reduction order
template <typename T, bool STEP0>
__global__ __launch_bounds__(64) void k_gran_means(GranMeanArgs a) {
  const int lane = threadIdx.x;
  const size_t plane = (size_t)a.g.Y * a.g.X;
  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    double s = 0.0;
    if (o.area > 0) {
      const u16* lab = a.labels + (size_t)o.tile * plane;
      const int h = o.y1 - o.y0, w = o.x1 - o.x0;
      const u16 L = (u16)o.label;
      const T* px = STEP0 ? reinterpret_cast<const T*>(a.planes) + ((size_t)o.tile * a.C + a.channel) * plane : nullptr;
      const double* rec = STEP0 ? nullptr : a.rec + (size_t)o.tile * a.g.sh * a.g.sw;
      for (int i = lane; i < h * w; i += 64) {
        const int yy = o.y0 + i / w, xx = o.x0 + i % w;
        const size_t idx = (size_t)yy * a.g.X + xx;
        if (lab[idx] != L) continue;
        if (STEP0) s += (double)px_load<T>(px, idx);
        else s += bilinear([&](int y2, int x2) { return rec[(size_t)y2 * a.g.sw + x2]; }, a.g.sh, a.g.sw, (double)yy * a.sy, (double)xx * a.sx);
      }
    }
    s = wave_sum(s);
    if (lane == 0) {
      const double mean = o.area > 0 ? s / (double)o.area : NAN;
      if (STEP0) {
        a.prev[oi] = mean;
        a.start[oi] = fmax(mean, 2.220446049250313e-16);
      } else {
        a.out[(size_t)oi * a.ld + a.col] = (a.prev[oi] - mean) * 100.0 / a.start[oi];
        a.prev[oi] = mean;
      }
    }
  }
}

inline unsigned grid_for(size_t n) { return (unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384); }

}  // namespace

static void gran_geometry(int F, int Y, int X, double subsample_size, double image_sample_size, GranGeom* g) {
  g->F = F; g->Y = Y; g->X = X;
  // numpy: new_shape = shape * subsample_size; mgrid[0:new_shape] has ceil(new_shape) points
  g->sh = subsample_size < 1 ? (int)ceil(Y * subsample_size) : Y;
  g->sw = subsample_size < 1 ? (int)ceil(X * subsample_size) : X;
  g->bh = image_sample_size < 1 ? (int)ceil(g->sh * image_sample_size) : g->sh;
  g->bw = image_sample_size < 1 ? (int)ceil(g->sw * image_sample_size) : g->sw;
}

static size_t gran_bytes(const GranGeom& g, int n_obj) {
  const size_t ns = (size_t)g.F * g.sh * g.sw, nb = (size_t)g.F * g.bh * g.bw;
  // 4 x [ns] doubles (sub -> pix, ero, two reconstruction buffers), 2 x [nb] doubles, per-object means, flag, masks
  return 4 * ns * 8 + 2 * nb * 8 + 2 * (size_t)n_obj * 8 + 256 + ns + nb + 64;
}

extern "C" size_t aliby_granularity_workspace_bytes(int F, int Y, int X, int n_obj, double subsample_size, double image_sample_size) {
  if (F <= 0 || Y <= 0 || X <= 0 || n_obj < 0 || !(subsample_size > 0) || !(image_sample_size > 0)) return 0;
  GranGeom g;
  gran_geometry(F, Y, X, subsample_size, image_sample_size, &g);
  return gran_bytes(g, n_obj);
}

extern "C" int aliby_features_granularity(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C, int Y,
                                          int X, int channel, const aliby_object* table_dev, int n_obj, double subsample_size,
                                          double image_sample_size, int element_size, int spectrum_length, int image_mask_objects,
                                          void* workspace, size_t workspace_bytes, double* out, int ld, int col0, void* stream_) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && planes && table_dev && out, "NULL argument");
  ARG_CHECK(F > 0 && Y > 1 && X > 1, "bad shape");
  ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32, "dtype must be ALIBY_U16 or ALIBY_F32");
  ARG_CHECK(channel >= 0 && channel < C, "channel out of range");
  ARG_CHECK(subsample_size > 0 && subsample_size <= 1 && image_sample_size > 0 && image_sample_size <= 1, "sample sizes must be in (0, 1]");
  ARG_CHECK(element_size >= 1 && element_size <= 64 && spectrum_length >= 1 && spectrum_length <= 64, "element_size / spectrum length out of range");
  ARG_CHECK(col0 >= 0 && col0 + spectrum_length <= ld, "columns exceed row stride");
  hipStream_t s = as_stream(stream_);
  GranGeom g;
  gran_geometry(F, Y, X, subsample_size, image_sample_size, &g);
  ARG_CHECK(g.sh > 1 && g.sw > 1 && g.bh > 1 && g.bw > 1, "frame too small for these sample sizes");
  const size_t ns = (size_t)F * g.sh * g.sw, nb = (size_t)F * g.bh * g.bw;
  ARG_CHECK(workspace != nullptr && workspace_bytes >= gran_bytes(g, n_obj), "workspace smaller than aliby_granularity_workspace_bytes()");
  ARG_CHECK(((uintptr_t)workspace & 7) == 0, "workspace must be 8-byte aligned");
  unsigned char* w = (unsigned char*)workspace;
  double* bufA = (double*)w;           w += ns * 8;
  double* bufB = (double*)w;           w += ns * 8;
  double* bufC = (double*)w;           w += ns * 8;
  double* bufD = (double*)w;           w += ns * 8;
  double* backA = (double*)w;          w += nb * 8;
  double* backB = (double*)w;          w += nb * 8;
  double* prev = (double*)w;           w += (size_t)n_obj * 8;
  double* start = (double*)w;          w += (size_t)n_obj * 8;
  int* flag = (int*)w;                 w += 256;
  unsigned char* msk = image_mask_objects ? w : nullptr;  w += ns;
  unsigned char* bmsk = image_mask_objects ? w : nullptr;

  // 1. subsample (bufA = sub)
  const double inv = subsample_size < 1 ? 1.0 / subsample_size : 1.0;
  if (dtype == ALIBY_U16) hipLaunchKernelGGL((k_gran_sample_frame<u16>), dim3(grid_for(ns)), dim3(256), 0, s, (const u16*)planes, C, channel, g, inv, bufA);
  else hipLaunchKernelGGL((k_gran_sample_frame<float>), dim3(grid_for(ns)), dim3(256), 0, s, (const float*)planes, C, channel, g, inv, bufA);
  if (msk) hipLaunchKernelGGL(k_gran_sample_mask, dim3(grid_for(ns)), dim3(256), 0, s, labels, g, inv, msk);
  // 2. background: subsample again, masked erode, masked dilate, resize back, subtract, clamp (bufB = pix, bufC = ero)
  const double binv = image_sample_size < 1 ? 1.0 / image_sample_size : 1.0;
  const double* back_src = bufA;
  if (image_sample_size < 1) {
    hipLaunchKernelGGL(k_gran_resample, dim3(grid_for(nb)), dim3(256), 0, s, bufA, F, g.sh, g.sw, g.bh, g.bw, binv, binv, backA);
    if (msk) hipLaunchKernelGGL(k_gran_resample_mask, dim3(grid_for(nb)), dim3(256), 0, s, msk, F, g.sh, g.sw, g.bh, g.bw, binv, binv, bmsk);
    back_src = backA;
  } else if (msk) {
    bmsk = msk;
  }
  double* eroded = image_sample_size < 1 ? backB : bufC;
  double* dilated = image_sample_size < 1 ? backA : bufD;
  hipLaunchKernelGGL((k_gran_morph<false>), dim3(grid_for(nb)), dim3(256), 0, s, back_src, msk ? bmsk : nullptr, F, g.bh, g.bw, element_size, eroded);
  hipLaunchKernelGGL((k_gran_morph<true>), dim3(grid_for(nb)), dim3(256), 0, s, eroded, msk ? bmsk : nullptr, F, g.bh, g.bw, element_size, dilated);
  const double sy = (double)(g.bh - 1) / (double)(g.sh - 1), sx = (double)(g.bw - 1) / (double)(g.sw - 1);
  hipLaunchKernelGGL(k_gran_subtract, dim3(grid_for(ns)), dim3(256), 0, s, bufA, dilated, msk, g, image_sample_size < 1 ? 1 : 0, sy, sx, bufB, bufC);
  KERNEL_CHECK();
  double* pix = bufB;
  double* ero = bufC;
  double* spare[2] = {bufA, bufD};  // (sub is no longer needed; with image_sample_size == 1 bufD held the dilated image, consumed above)
  // 3. the objects' starting means, on the original pixels
  GranMeanArgs m;
  m.labels = labels; m.tab = table_dev; m.n_obj = n_obj; m.g = g; m.planes = planes; m.C = C; m.channel = channel; m.rec = nullptr;
  m.sy = (double)(g.sh - 1) / (double)(Y - 1); m.sx = (double)(g.sw - 1) / (double)(X - 1);
  m.prev = prev; m.start = start; m.out = out; m.ld = ld; m.col = col0;
  const unsigned og = (unsigned)(n_obj < 65535 ? n_obj : 65535);
  if (dtype == ALIBY_U16) hipLaunchKernelGGL((k_gran_means<u16, true>), dim3(og), dim3(64), 0, s, m);
  else hipLaunchKernelGGL((k_gran_means<float, true>), dim3(og), dim3(64), 0, s, m);
  // 4. the spectrum
  // Jacobi sweeps move a value one pixel per sweep along a geodesic path, which in a serpentine mask can be as long as half the
  // image has pixels: that is the hard cap (reaching it without convergence is reported, never returned as a result).
  const long long max_total = (long long)g.sh * g.sw / 2 + 64;
  for (int step = 1; step <= spectrum_length; ++step) {
    double* next = spare[0];
    hipLaunchKernelGGL((k_gran_morph<false>), dim3(grid_for(ns)), dim3(256), 0, s, ero, msk, F, g.sh, g.sw, 1, next);
    spare[0] = ero;
    ero = next;
    // reconstruction: rec0 = ero (<= pix), Jacobi sweeps in chunks of 16 until a whole chunk changes nothing
    const double* cur = ero;
    double* pp[2] = {spare[0], spare[1]};
    int which = 0, sweeps = 0;
    for (;;) {
      HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), s));
      for (int k = 0; k < 16; ++k) {
        hipLaunchKernelGGL(k_gran_recon_sweep, dim3(grid_for(ns)), dim3(256), 0, s, cur, pix, F, g.sh, g.sw, pp[which], flag);
        cur = pp[which];
        which ^= 1;
      }
      sweeps += 16;
      int changed = 0;
      HIP_TRY(hipMemcpyAsync(&changed, flag, sizeof(int), hipMemcpyDeviceToHost, s));
      int rc2 = aliby_wait_stream(s);
      if (rc2 != ALIBY_OK) return rc2;
      if (!changed) break;
      if (sweeps >= max_total) {
        aliby_set_error("granularity: the reconstruction of spectrum step %d did not converge within %d sweeps on a %d x %d image", step,
                        sweeps, g.sh, g.sw);
        return ALIBY_ERR_TOO_LARGE;
      }
    }
    KERNEL_CHECK();
    m.rec = cur;
    m.col = col0 + step - 1;
    hipLaunchKernelGGL((k_gran_means<u16, false>), dim3(og), dim3(64), 0, s, m);  // (T is unused when STEP0 is false)
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}
