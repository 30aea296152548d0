// traps.hip — image-sized building blocks of ALCATRAS trap detection, float64 like scikit-image.
//
// Reference: src/aliby/tile/process_traps.py — segment_traps (24-137: rescale, rank entropy, Otsu, closing,
// clear_border, label, regionprops) and identify_trap_locations (140-218: rescale, rotate, match_template,
// peak_local_max).  Runs once per position on one frame; nothing here is on the per-tile hot path, so the kernels are
// one-thread-per-output-pixel and favour the exact operation order of the library calls they stand in for
// (aliby_amd/tile/traps.py sequences them; oracle/traps_restated.py is the CPU statement of the same arithmetic).
#include "common.h"

namespace {

__device__ __forceinline__ int mirror_index(long long i, int n) {  // scipy 'mirror' / skimage warp 'reflect'
  if (n == 1) return 0;
  const long long period = 2LL * (n - 1);
  if (i < 0) i = -i;
  i %= period;
  return (int)(i >= n ? period - i : i);
}

// One axis of scipy.ndimage.gaussian_filter(mode='mirror'); truncate != 0 reproduces an integer array's C cast.
__global__ void k_gauss1d(const double* __restrict__ in, double* __restrict__ out, int H, int W, int axis,
                          const double* __restrict__ w, int radius, int truncate) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  double acc = 0.0;
  for (int k = -radius; k <= radius; ++k) {
    const double v = axis == 0 ? in[(size_t)mirror_index((long long)y + k, H) * W + x]
                               : in[(size_t)y * W + mirror_index((long long)x + k, W)];
    acc += w[k + radius] * v;
  }
  out[(size_t)y * W + x] = truncate ? trunc(acc) : acc;
}

// skimage _warp_fast, order 1: output (r, c) reads the input at (x, y) = M (c, r, 1); mode 0 = constant, 1 = reflect.
__global__ void k_warp(const double* __restrict__ in, int H, int W, double* __restrict__ out, int OH, int OW,
                       double m0, double m1, double m2, double m3, double m4, double m5, int mode, double cval) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= OW) return;
  const double x = m0 * c + m1 * r + m2;
  const double y = m3 * c + m4 * r + m5;
  const double y0 = floor(y), x0 = floor(x), y1 = ceil(y), x1 = ceil(x);
  const double dy = y - y0, dx = x - x0;
  auto px = [&](double rr, double cc) -> double {
    long long ri = (long long)rr, ci = (long long)cc;
    if (mode == 1) return in[(size_t)mirror_index(ri, H) * W + mirror_index(ci, W)];
    if (ri < 0 || ri >= H || ci < 0 || ci >= W) return cval;
    return in[(size_t)ri * W + ci];
  };
  const double top = (1 - dx) * px(y0, x0) + dx * px(y0, x1);
  const double bottom = (1 - dx) * px(y1, x0) + dx * px(y1, x1);
  out[(size_t)r * OW + c] = (1 - dy) * top + dy * bottom;
}

// filters.rank.entropy with a disk footprint: Shannon entropy (bits) of the grey levels of the in-image neighbours.
__global__ void k_rank_entropy(const uint8_t* __restrict__ in, int H, int W, int radius, double* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int r2 = radius * radius;
  int pop = 0;
  for (int dy = -radius; dy <= radius; ++dy)
    for (int dx = -radius; dx <= radius; ++dx)
      if (dy * dy + dx * dx <= r2 && y + dy >= 0 && y + dy < H && x + dx >= 0 && x + dx < W) ++pop;
  double e = 0.0;
  // every distinct level is handled at its first occurrence in raster order of the footprint
  for (int dy = -radius; dy <= radius; ++dy)
    for (int dx = -radius; dx <= radius; ++dx) {
      if (dy * dy + dx * dx > r2 || y + dy < 0 || y + dy >= H || x + dx < 0 || x + dx >= W) continue;
      const uint8_t v = in[(size_t)(y + dy) * W + x + dx];
      int count = 0;
      bool first = true;
      for (int ey = -radius; ey <= radius && first; ++ey)
        for (int ex = -radius; ex <= radius; ++ex) {
          if (ey * ey + ex * ex > r2 || y + ey < 0 || y + ey >= H || x + ex < 0 || x + ex >= W) continue;
          if (in[(size_t)(y + ey) * W + x + ex] != v) continue;
          if (ey < dy || (ey == dy && ex < dx)) { first = false; break; }
          ++count;
        }
      if (!first) continue;
      const double p = (double)count / (double)pop;
      e -= p * log(p) / 0.6931471805599453;
    }
  out[(size_t)y * W + x] = e;
}

// k x k maximum / minimum with replicated borders; the window spans offsets [lo, hi] on both axes.
__global__ void k_morph(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W, int lo, int hi, int is_max) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  int v = is_max ? 0 : 255;
  for (int dy = lo; dy <= hi; ++dy)
    for (int dx = lo; dx <= hi; ++dx) {
      const int yy = min(max(y + dy, 0), H - 1), xx = min(max(x + dx, 0), W - 1);
      const int p = in[(size_t)yy * W + xx];
      v = is_max ? max(v, p) : min(v, p);
    }
  out[(size_t)y * W + x] = (uint8_t)v;
}

// 8-connected labelling by minimum propagation: label = 1 + raster index of the component's first pixel.
__global__ void k_label_init(const uint8_t* __restrict__ bw, int n, int32_t* __restrict__ lab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lab[i] = bw[i] ? i + 1 : 0;
}
__global__ void k_label_step(int32_t* __restrict__ lab, int H, int W, int* __restrict__ changed) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const size_t i = (size_t)y * W + x;
  int32_t cur = lab[i];
  if (cur == 0) return;
  int32_t best = cur;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const int32_t o = lab[(size_t)yy * W + xx];
      if (o != 0 && o < best) best = o;
    }
  if (best < cur) {
    // jump to the root's current label as well: roots are pixels whose label is their own index + 1
    const int32_t root = lab[best - 1];
    if (root != 0 && root < best) best = root;
    atomicMin(&lab[i], best);
    *changed = 1;
  }
}

// Raw moments of every component, exact in 64-bit integers: count, sum y, sum x, sum yy, sum xx, sum xy, touches border.
__global__ void k_region_sums(const int32_t* __restrict__ lab, int H, int W, unsigned long long* __restrict__ sums) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int32_t l = lab[(size_t)y * W + x];
  if (l == 0) return;
  unsigned long long* s = sums + (size_t)(l - 1) * 7;
  atomicAdd(s + 0, 1ULL);
  atomicAdd(s + 1, (unsigned long long)y);
  atomicAdd(s + 2, (unsigned long long)x);
  atomicAdd(s + 3, (unsigned long long)y * y);
  atomicAdd(s + 4, (unsigned long long)x * x);
  atomicAdd(s + 5, (unsigned long long)x * y);
  if (y == 0 || y == H - 1 || x == 0 || x == W - 1) atomicAdd(s + 6, 1ULL);
}

// feature.match_template on the median-padded image P: normalised cross-correlation of the th x tw window whose first
// pixel is (y + d0 + 1, x + d1 + 1).
__global__ void k_match(const double* __restrict__ P, int PW, const double* __restrict__ T, int th, int tw,
                        double* __restrict__ out, int H, int W, int d0, int d1, double t_mean, double t_ssd) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  double s = 0.0, s2 = 0.0, st = 0.0;
  const double* base = P + (size_t)(y + d0 + 1) * PW + x + d1 + 1;
  for (int i = 0; i < th; ++i) {
    const double* row = base + (size_t)i * PW;
    const double* trow = T + (size_t)i * tw;
    for (int j = 0; j < tw; ++j) {
      const double p = row[j];
      s += p;
      s2 += p * p;
      st += p * trow[j];
    }
  }
  const double num = st - s * t_mean;
  double den = (s2 - s * s / (double)(th * tw)) * t_ssd;
  den = sqrt(fmax(den, 0.0));
  out[(size_t)y * W + x] = den > 2.220446049250313e-16 ? num / den : 0.0;
}

// scipy.ndimage.maximum_filter1d(mode='constant', cval=0)
__global__ void k_maxfilter1d(const double* __restrict__ in, double* __restrict__ out, int H, int W, int axis, int radius) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int n = axis == 0 ? H : W, at = axis == 0 ? y : x;
  double v = (at - radius < 0 || at + radius >= n) ? 0.0 : -INFINITY;
  for (int k = max(at - radius, 0); k <= min(at + radius, n - 1); ++k)
    v = fmax(v, axis == 0 ? in[(size_t)k * W + x] : in[(size_t)y * W + k]);
  out[(size_t)y * W + x] = v;
}

inline dim3 grid_for(int W, int H) { return dim3((W + 127) / 128, H); }

}  // namespace

int aliby_trap_gauss1d(aliby_ctx* ctx, const double* in, double* out, int H, int W, int axis, const double* weights_dev,
                       int radius, int truncate, void* stream) {
  ARG_CHECK(ctx && in && out && weights_dev && in != out, "buffers");
  ARG_CHECK(H > 0 && W > 0 && (axis == 0 || axis == 1) && radius >= 0, "geometry");
  k_gauss1d<<<grid_for(W, H), 128, 0, as_stream(stream)>>>(in, out, H, W, axis, weights_dev, radius, truncate);
  KERNEL_CHECK();
  return ALIBY_OK;
}

int aliby_trap_warp(aliby_ctx* ctx, const double* in, int H, int W, double* out, int OH, int OW, const double* matrix6_host,
                    int mode, double cval, void* stream) {
  ARG_CHECK(ctx && in && out && matrix6_host, "buffers");
  ARG_CHECK(H > 0 && W > 0 && OH > 0 && OW > 0 && (mode == 0 || mode == 1), "geometry");
  const double* m = matrix6_host;
  k_warp<<<grid_for(OW, OH), 128, 0, as_stream(stream)>>>(in, H, W, out, OH, OW, m[0], m[1], m[2], m[3], m[4], m[5], mode, cval);
  KERNEL_CHECK();
  return ALIBY_OK;
}

int aliby_trap_entropy(aliby_ctx* ctx, const uint8_t* in, int H, int W, int radius, double* out, void* stream) {
  ARG_CHECK(ctx && in && out, "buffers");
  ARG_CHECK(H > 0 && W > 0 && radius >= 0 && radius <= 32, "geometry");
  k_rank_entropy<<<grid_for(W, H), 128, 0, as_stream(stream)>>>(in, H, W, radius, out);
  KERNEL_CHECK();
  return ALIBY_OK;
}

int aliby_trap_morph(aliby_ctx* ctx, const uint8_t* in, uint8_t* out, int H, int W, int lo, int hi, int is_max, void* stream) {
  ARG_CHECK(ctx && in && out && in != out, "buffers");
  ARG_CHECK(H > 0 && W > 0 && lo <= 0 && hi >= 0, "geometry");
  k_morph<<<grid_for(W, H), 128, 0, as_stream(stream)>>>(in, out, H, W, lo, hi, is_max);
  KERNEL_CHECK();
  return ALIBY_OK;
}

// labels: int32 [H,W], 0 = background, otherwise 1 + raster index of the component's first pixel (8-connected)
int aliby_trap_label(aliby_ctx* ctx, const uint8_t* bw, int H, int W, int32_t* labels, void* stream) {
  ARG_CHECK(ctx && bw && labels, "buffers");
  ARG_CHECK(H > 0 && W > 0 && (long long)H * W < (1LL << 31), "geometry");
  hipStream_t s = as_stream(stream);
  int rc = aliby_ensure_scratch(ctx, sizeof(int));
  if (rc != ALIBY_OK) return rc;
  int* changed = (int*)ctx->scratch;
  const int n = H * W;
  k_label_init<<<(n + 255) / 256, 256, 0, s>>>(bw, n, labels);
  KERNEL_CHECK();
  for (int round = 0; round < 1 << 16; ++round) {
    HIP_TRY(hipMemsetAsync(changed, 0, sizeof(int), s));
    for (int k = 0; k < 8; ++k) k_label_step<<<grid_for(W, H), 128, 0, s>>>(labels, H, W, changed);
    KERNEL_CHECK();
    int host = 0;
    HIP_TRY(hipMemcpyAsync(&host, changed, sizeof(int), hipMemcpyDeviceToHost, s));
    rc = aliby_wait_stream(s);
    if (rc != ALIBY_OK) return rc;
    if (!host) return ALIBY_OK;
  }
  aliby_set_error("labelling did not converge");
  return ALIBY_ERR_HIP;
}

// sums: uint64 [H*W, 7] zeroed by the caller; row l-1 receives the raw moments of label l
int aliby_trap_region_sums(aliby_ctx* ctx, const int32_t* labels, int H, int W, uint64_t* sums, void* stream) {
  ARG_CHECK(ctx && labels && sums, "buffers");
  k_region_sums<<<grid_for(W, H), 128, 0, as_stream(stream)>>>(labels, H, W, (unsigned long long*)sums);
  KERNEL_CHECK();
  return ALIBY_OK;
}

int aliby_trap_match_template(aliby_ctx* ctx, const double* padded, int PH, int PW, const double* templ, int th, int tw,
                              double* out, int H, int W, double t_mean, double t_ssd, void* stream) {
  ARG_CHECK(ctx && padded && templ && out, "buffers");
  ARG_CHECK(PH == H + 2 * th && PW == W + 2 * tw && th > 0 && tw > 0, "padded image must be the image plus one template on every side");
  k_match<<<grid_for(W, H), 128, 0, as_stream(stream)>>>(padded, PW, templ, th, tw, out, H, W, (th - 1) / 2, (tw - 1) / 2, t_mean, t_ssd);
  KERNEL_CHECK();
  return ALIBY_OK;
}

int aliby_trap_maxfilter1d(aliby_ctx* ctx, const double* in, double* out, int H, int W, int axis, int radius, void* stream) {
  ARG_CHECK(ctx && in && out && in != out, "buffers");
  ARG_CHECK(H > 0 && W > 0 && (axis == 0 || axis == 1) && radius >= 0, "geometry");
  k_maxfilter1d<<<grid_for(W, H), 128, 0, as_stream(stream)>>>(in, out, H, W, axis, radius);
  KERNEL_CHECK();
  return ALIBY_OK;
}
