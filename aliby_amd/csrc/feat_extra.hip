// feat_extra.hip — the remaining in-repo metrics of the reference that round 1 left to the CPU oracle:
//
//   cell.ratio            src/extraction/core/functions/cell.py:268-279   median over the object of channel0 / channel1
//                         (true division, float64), NaN when any channel-1 pixel of the object is 0
//   trap.imBackground     src/extraction/core/functions/trap.py:6-23      median of the tile's pixels under NO cell mask
//   trap.background_max5  trap.py:26-43                                   mean of the five largest such pixels
//
// ratio: one workgroup per object, the ratios gathered into LDS and bitonic-sorted (np.median = middle element, or the
// mean of the two middle ones).  The trap metrics are order statistics over up to a whole tile: exact radix selection on
// the values' order-preserving 32-bit keys, 8 bits per pass with a 256-bin LDS histogram (uint16 pixels need the two low
// bytes only), one workgroup per tile.
#include "common.h"

typedef unsigned short u16;

namespace {

__device__ __forceinline__ unsigned key_of(u16 v) { return v; }
__device__ __forceinline__ unsigned key_of(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving
}
__device__ __forceinline__ double value_of_key(unsigned k, u16) { return (double)k; }
__device__ __forceinline__ double value_of_key(unsigned k, float) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return (double)__uint_as_float(u);
}

struct RatioArgs {
  const u16* labels;
  const void* planes;
  int F, C, Y, X, ch0, ch1;
  const aliby_object* tab;
  int n_obj, cap;  // cap: power of two >= the largest area
  double* out;     // [n_obj]
};

template <typename T>
__global__ __launch_bounds__(256) void k_cell_ratio(RatioArgs a) {
  extern __shared__ double vals[];
  __shared__ int s_n, s_zero;
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;
  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    if (tid == 0) { s_n = 0; s_zero = 0; }
    __syncthreads();
    if (o.area > 0) {
      const u16* lab = a.labels + (size_t)o.tile * plane;
      const T* p0 = reinterpret_cast<const T*>(a.planes) + ((size_t)o.tile * a.C + a.ch0) * plane;
      const T* p1 = reinterpret_cast<const T*>(a.planes) + ((size_t)o.tile * a.C + a.ch1) * plane;
      const int h = o.y1 - o.y0, w = o.x1 - o.x0;
      const u16 L = (u16)o.label;
      for (int i = tid; i < h * w; i += blockDim.x) {
        const size_t idx = (size_t)(o.y0 + i / w) * a.X + (o.x0 + i % w);
        if (lab[idx] != L) continue;
        const float num = px_load<T>(p0, idx), den = px_load<T>(p1, idx);
        if (den == 0.0f) s_zero = 1;
        // NumPy's true division: float64 for integer pixels, float32 for float32 pixels
        vals[atomicAdd(&s_n, 1)] = sizeof(T) == 2 ? (double)num / (double)den : (double)(num / den);
      }
    }
    __syncthreads();
    const int n = s_n;
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (int i = n + tid; i < n2; i += blockDim.x) vals[i] = INFINITY;  // pad: sorts to the end
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < n2; i += blockDim.x) {
          const int l = i ^ j;
          if (l > i) {
            const double x = vals[i], y = vals[l];
            const bool up = (i & k) == 0;
            if ((x > y) == up) { vals[i] = y; vals[l] = x; }
          }
        }
        __syncthreads();
      }
    if (tid == 0) {
      double r = NAN;
      if (n > 0 && !s_zero) r = (n & 1) ? vals[n >> 1] : 0.5 * (vals[(n >> 1) - 1] + vals[n >> 1]);
      a.out[oi] = r;
    }
    __syncthreads();
  }
}

struct TrapArgs {
  const u16* labels;
  const void* planes;
  int F, C, Y, X, channel;
  double* out;  // [F, 2]: imBackground, background_max5
};

// k-th smallest key (0-based) among the tile's background pixels: MSB-first radix selection
template <typename T>
__device__ unsigned select_kth(const u16* lab, const T* px, size_t npix, unsigned long long k, unsigned* hist, int first_byte) {
  unsigned prefix = 0, mask = 0;
  for (int byte = first_byte; byte >= 0; --byte) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int sh = 8 * byte;
    for (size_t i = threadIdx.x; i < npix; i += blockDim.x) {
      if (lab[i]) continue;
      const unsigned key = key_of(px[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> sh) & 255u], 1u);
    }
    __syncthreads();
    // every thread walks the 256 bins the same way (cheap, and no broadcast needed)
    unsigned long long run = 0;
    int bin = 255;
    for (int b = 0; b < 256; ++b) {
      if (k < run + hist[b]) { bin = b; break; }
      run += hist[b];
    }
    k -= run;
    prefix |= (unsigned)bin << sh;
    mask |= 255u << sh;
    __syncthreads();
  }
  return prefix;
}

template <typename T>
__global__ __launch_bounds__(1024) void k_trap_background(TrapArgs a) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_cnt;
  const int f = blockIdx.x;
  const size_t npix = (size_t)a.Y * a.X;
  const u16* lab = a.labels + (size_t)f * npix;
  const T* px = reinterpret_cast<const T*>(a.planes) + ((size_t)f * a.C + a.channel) * npix;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  unsigned long long mine = 0;
  for (size_t i = threadIdx.x; i < npix; i += blockDim.x) mine += lab[i] == 0;
  atomicAdd(&s_cnt, mine);
  __syncthreads();
  const unsigned long long n = s_cnt;
  constexpr int FIRST = sizeof(T) == 2 ? 1 : 3;
  double med = NAN, top = NAN;
  if (n > 0) {
    // np.median: the middle element, or the mean of the two middle ones
    const double hi = value_of_key(select_kth<T>(lab, px, npix, n >> 1, hist, FIRST), T());
    med = hi;
    if ((n & 1) == 0) med = 0.5 * (hi + value_of_key(select_kth<T>(lab, px, npix, (n >> 1) - 1, hist, FIRST), T()));
    // np.mean(np.sort(x)[-5:]): the five largest (all of them when there are fewer)
    const int m = n < 5 ? (int)n : 5;
    double s = 0;
    for (int j = 0; j < m; ++j) s += value_of_key(select_kth<T>(lab, px, npix, n - 1 - j, hist, FIRST), T());
    top = s / (double)m;
  }
  if (threadIdx.x == 0) {
    a.out[2 * f] = med;
    a.out[2 * f + 1] = top;
  }
}

}  // namespace

extern "C" int aliby_features_cell_ratio(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C, int Y,
                                         int X, int channel0, int channel1, const aliby_object* table_dev, int n_obj, int max_area,
                                         double* out, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && planes && table_dev && out, "NULL argument");
  ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32, "dtype must be ALIBY_U16 or ALIBY_F32");
  ARG_CHECK(channel0 >= 0 && channel0 < C && channel1 >= 0 && channel1 < C, "channel out of range");
  int cap = 64;
  while (cap < max_area) cap <<= 1;
  const size_t need = (size_t)cap * 8;
  if (need > 144 * 1024) {
    aliby_set_error("cell.ratio: an object of %d pixels does not fit the %d KiB of LDS this kernel sorts in", max_area, 144);
    return ALIBY_ERR_TOO_LARGE;
  }
  RatioArgs a;
  a.labels = labels; a.planes = planes; a.F = F; a.C = C; a.Y = Y; a.X = X; a.ch0 = channel0; a.ch1 = channel1;
  a.tab = table_dev; a.n_obj = n_obj; a.cap = cap; a.out = out;
  hipStream_t s = as_stream(stream);
  if (dtype == ALIBY_U16) {
    if (need > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_cell_ratio<u16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL((k_cell_ratio<u16>), dim3(n_obj), dim3(256), need, s, a);
  } else {
    if (need > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_cell_ratio<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL((k_cell_ratio<float>), dim3(n_obj), dim3(256), need, s, a);
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}

extern "C" int aliby_features_trap_background(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C,
                                              int Y, int X, int channel, double* out, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (F == 0) return ALIBY_OK;
  ARG_CHECK(labels && planes && out, "NULL argument");
  ARG_CHECK(F > 0 && Y > 0 && X > 0, "bad shape");
  ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32, "dtype must be ALIBY_U16 or ALIBY_F32");
  ARG_CHECK(channel >= 0 && channel < C, "channel out of range");
  TrapArgs a;
  a.labels = labels; a.planes = planes; a.F = F; a.C = C; a.Y = Y; a.X = X; a.channel = channel; a.out = out;
  hipStream_t s = as_stream(stream);
  if (dtype == ALIBY_U16) hipLaunchKernelGGL((k_trap_background<u16>), dim3(F), dim3(1024), 0, s, a);
  else hipLaunchKernelGGL((k_trap_background<float>), dim3(F), dim3(1024), 0, s, a);
  KERNEL_CHECK();
  return ALIBY_OK;
}
