// common.h — shared host/device helpers for libaliby_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "../../include/aliby_hip.h"

#define WAVE 64

struct aliby_ctx {
  int device;
  int cu_count;
  int lds_bytes;
  size_t hbm_bytes;
  char name[128];
  // small device scratch owned by the context (per-tile counters etc.)
  void* scratch;
  size_t scratch_bytes;
};

void aliby_set_error(const char* fmt, ...);
int aliby_ensure_scratch(aliby_ctx* ctx, size_t bytes);
int aliby_wait_stream(hipStream_t s);

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess) {                                                       \
      aliby_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),      \
                      __FILE__, __LINE__);                                         \
      return ALIBY_ERR_HIP;                                                        \
    }                                                                              \
  } while (0)

#define ARG_CHECK(cond, msg)                                  \
  do {                                                        \
    if (!(cond)) {                                            \
      aliby_set_error("invalid argument: %s (%s)", msg, #cond); \
      return ALIBY_ERR_INVALID;                               \
    }                                                         \
  } while (0)

#define KERNEL_CHECK()                                                         \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      aliby_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), \
                      __FILE__, __LINE__);                                     \
      return ALIBY_ERR_HIP;                                                    \
    }                                                                          \
  } while (0)

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// Workgroup size for one-workgroup-per-object kernels: a ~450-pixel nucleus keeps a single wave busy; four
// waves would spend their time in barriers.  `work` = pixels the workgroup loops over (bbox or area).
static inline int aliby_pick_block(long long work) { return work <= 2048 ? 64 : (work <= 8192 ? 128 : 256); }

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
#ifdef __HIPCC__

template <typename T>
__device__ __forceinline__ float px_load(const T* p, size_t i);
template <>
__device__ __forceinline__ float px_load<uint16_t>(const uint16_t* p, size_t i) { return (float)p[i]; }
template <>
__device__ __forceinline__ float px_load<float>(const float* p, size_t i) { return p[i]; }

// ---- wave-wide reductions by DPP ------------------------------------------------------------------------------------------
// A reduction step is one VALU instruction reading a neighbour lane through the data-parallel-primitives path (row_shr within a
// 16-lane row, row_bcast between rows) instead of a ds_bpermute round trip through the LDS crossbar per step: the per-object
// kernels are chains of dozens of dependent wave reductions over a few hundred pixels, so the latency of a reduction is what
// they wait for.  Scheme (gfx9 DPP): Hillis-Steele inclusive scan inside each row (shifts 1, 2, 4, 8; lanes without a source
// take the identity), row 0's total into row 1 and row 2's into row 3 (row_bcast:15), lane 31's into rows 2-3 (row_bcast:31):
// lane 63 holds the total, which v_readlane broadcasts.  Fixed order: deterministic.  The total is returned in EVERY lane.
#define ALIBY_DPP_ROW_SHR(n) (0x110 + (n))
#define ALIBY_DPP_ROW_BCAST15 0x142
#define ALIBY_DPP_ROW_BCAST31 0x143

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_take(int ident, int v) {
  return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float ident, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_take(long long ident, long long v) {
  const int lo = __builtin_amdgcn_update_dpp((int)ident, (int)v, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(ident >> 32), (int)(v >> 32), CTRL, ROW_MASK, 0xf, false);
  return ((long long)hi << 32) | (unsigned int)lo;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double ident, double v) {
  return __longlong_as_double(dpp_take<CTRL, ROW_MASK>(__double_as_longlong(ident), __double_as_longlong(v)));
}
__device__ __forceinline__ int wave_last(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ float wave_last(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ long long wave_last(long long v) {
  const int lo = __builtin_amdgcn_readlane((int)v, 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
  return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ double wave_last(double v) { return __longlong_as_double(wave_last(__double_as_longlong(v))); }

template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce_dpp(T v, T ident, Op op) {
  v = op(v, dpp_take<ALIBY_DPP_ROW_SHR(1), 0xf>(ident, v));
  v = op(v, dpp_take<ALIBY_DPP_ROW_SHR(2), 0xf>(ident, v));
  v = op(v, dpp_take<ALIBY_DPP_ROW_SHR(4), 0xf>(ident, v));
  v = op(v, dpp_take<ALIBY_DPP_ROW_SHR(8), 0xf>(ident, v));
  v = op(v, dpp_take<ALIBY_DPP_ROW_BCAST15, 0xa>(ident, v));
  v = op(v, dpp_take<ALIBY_DPP_ROW_BCAST31, 0xc>(ident, v));
  return wave_last(v);
}

__device__ __forceinline__ double wave_sum(double v) { return wave_reduce_dpp(v, 0.0, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ long long wave_sum(long long v) { return wave_reduce_dpp(v, 0LL, [](long long a, long long b) { return a + b; }); }
__device__ __forceinline__ int wave_sum(int v) { return wave_reduce_dpp(v, 0, [](int a, int b) { return a + b; }); }
__device__ __forceinline__ float wave_min(float v) { return wave_reduce_dpp(v, INFINITY, [](float a, float b) { return fminf(a, b); }); }
__device__ __forceinline__ float wave_max(float v) { return wave_reduce_dpp(v, -INFINITY, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce_dpp(v, (double)-INFINITY, [](double a, double b) { return fmax(a, b); }); }
__device__ __forceinline__ int wave_max(int v) { return wave_reduce_dpp(v, INT_MIN, [](int a, int b) { return max(a, b); }); }
__device__ __forceinline__ int wave_min(int v) { return wave_reduce_dpp(v, INT_MAX, [](int a, int b) { return min(a, b); }); }

// Block-wide reductions.  `red` is an LDS array of >= blockDim.x/64 elements of T.
// Deterministic: fixed tree inside the wave, then wave 0 folds the per-wave partials
// in wave order.  Result is broadcast to every thread.
#define DEFINE_BLOCK_REDUCE(NAME, T, WOP, IDENT, COMBINE)                     \
  __device__ __forceinline__ T NAME(T v, T* red) {                            \
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;      \
    const int nw = (blockDim.x + WAVE - 1) / WAVE;                            \
    v = WOP(v);                                                               \
    if (nw == 1) { /* single-wave workgroup: no LDS round trip */             \
      __syncthreads();                                                        \
      return v; /* (wave reductions return the total in every lane) */        \
    }                                                                         \
    __syncthreads();                                                          \
    if (lane == 0) red[wid] = v;                                              \
    __syncthreads();                                                          \
    T r = IDENT;                                                              \
    for (int i = 0; i < nw; ++i) { T o = red[i]; r = COMBINE; }               \
    return r;                                                                 \
  }
DEFINE_BLOCK_REDUCE(block_sum_f64, double, wave_sum, 0.0, r + o)
DEFINE_BLOCK_REDUCE(block_sum_i64, long long, wave_sum, 0LL, r + o)
DEFINE_BLOCK_REDUCE(block_sum_i32, int, wave_sum, 0, r + o)
DEFINE_BLOCK_REDUCE(block_min_f32, float, wave_min, INFINITY, fminf(r, o))
DEFINE_BLOCK_REDUCE(block_max_f32, float, wave_max, -INFINITY, fmaxf(r, o))
DEFINE_BLOCK_REDUCE(block_max_f64, double, wave_max, -INFINITY, fmax(r, o))
DEFINE_BLOCK_REDUCE(block_max_i32, int, wave_max, INT_MIN, max(r, o))
DEFINE_BLOCK_REDUCE(block_min_i32, int, wave_min, INT_MAX, min(r, o))

__device__ __forceinline__ float sort_lo(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float sort_hi(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double sort_lo(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double sort_hi(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ unsigned int sort_lo(unsigned int a, unsigned int b) { return min(a, b); }
__device__ __forceinline__ unsigned int sort_hi(unsigned int a, unsigned int b) { return max(a, b); }
__device__ __forceinline__ int sort_lo(int a, int b) { return min(a, b); }
__device__ __forceinline__ int sort_hi(int a, int b) { return max(a, b); }

// The cross-lane half of a bitonic stage (partner distance j < 64) for R register-resident elements per lane: element
// i = 64 r + lane meets lane ^ j of the same register.  `k` is the run length of the merge the stage belongs to.
template <typename T, int R>
__device__ __forceinline__ void wave_sort_lanes(T (&v)[R], int k, int lane) {
  // (not unrolled, here and in the callers' k loops: fully unrolled, the chains of min / max selects send LLVM's InstCombine
  // into minutes of compile time per instantiation)
#pragma clang loop unroll(disable)
  for (int j = (k < 64 ? k : 64) >> 1; j > 0; j >>= 1) {
    const bool lower = (lane & j) == 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const T o = __shfl_xor(v[r], j, WAVE);
      const bool up = (((r << 6) | lane) & k) == 0;
      // keep the smaller of (mine, partner's) when lower == up, else the larger: compare + select (as the LDS loop does; min /
      // max intrinsics here cost minutes of InstCombine time per instantiation)
      const bool partner_smaller = o < v[r];
      v[r] = (partner_smaller == (lower == up)) ? o : v[r];
    }
  }
}

// Bitonic sort of 64 R elements by ONE wave with the elements in registers (a[64 r + lane] <-> v[r]): partner distances
// below 64 are lane shuffles, distances of 64 and more are compare-exchanges between a lane's own registers.  About a fifth
// of the instructions of the LDS loop below, which is what bounds the one-wave-per-object kernels (16 lanes per SIMD: four
// cycles per wave instruction).  No NaNs expected (min / max instead of compare-and-swap).
template <typename T, int R>
__device__ __forceinline__ void wave_bitonic_sort_regs(T* a) {
  const int lane = threadIdx.x & (WAVE - 1);
  T v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = a[(r << 6) + lane];
#pragma clang loop unroll(disable)
  for (int k = 2; k <= 64; k <<= 1) wave_sort_lanes<T, R>(v, k, lane);
#pragma unroll
  for (int kr = 2; kr <= R; kr <<= 1) {  // merges of run length k = 64 kr
#pragma unroll
    for (int jr = kr >> 1; jr > 0; jr >>= 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r & jr) continue;
        const T x = v[r], y = v[r | jr];
        const bool up = (r & kr) == 0;  // compile-time: bit k of i = 64 r + lane is a bit of r
        const bool sw = (x > y) == up;
        v[r] = sw ? y : x;
        v[r | jr] = sw ? x : y;
      }
    }
    wave_sort_lanes<T, R>(v, kr << 6, lane);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) a[(r << 6) + lane] = v[r];
}

// Ascending bitonic sort of `n2` (power of two) elements; works on LDS or global scratch.  All threads of the block must
// call it.  A single-wave workgroup sorts up to 1024 elements in registers (wave_bitonic_sort_regs).
template <typename T>
__device__ __forceinline__ void block_bitonic_sort(T* a, int n2) {
  if (blockDim.x <= WAVE && n2 <= 1024) {
    __syncthreads();
    if (n2 > 64) {
      switch (n2 >> 6) {
        case 2: wave_bitonic_sort_regs<T, 2>(a); break;
        case 4: wave_bitonic_sort_regs<T, 4>(a); break;
        case 8: wave_bitonic_sort_regs<T, 8>(a); break;
        default: {
          // 1024 elements: both halves in registers, then one merge through LDS (a 16-register instantiation takes LLVM's
          // InstCombine minutes to compile).  Two ascending runs merge as: compare i with 1023 - i, then half-cleaners 256 .. 1.
          wave_bitonic_sort_regs<T, 8>(a);
          wave_bitonic_sort_regs<T, 8>(a + 512);
          __syncthreads();
          const int lane = threadIdx.x & (WAVE - 1);
          for (int i = lane; i < 512; i += WAVE) {
            const T x = a[i], y = a[1023 - i];
            if (x > y) { a[i] = y; a[1023 - i] = x; }
          }
          for (int j = 256; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = lane; t < 512; t += WAVE) {
              const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
              const T x = a[i], y = a[p];
              if (x > y) { a[i] = y; a[p] = x; }
            }
          }
          break;
        }
      }
    } else {
      // fewer than a wave's worth: pad in registers
      const int lane = threadIdx.x & (WAVE - 1);
      T v[1];
      T keep = a[lane < n2 ? lane : 0];
      // lanes >= n2 must hold something that sorts to the end: the largest element present
      T big = keep;
#pragma unroll
      for (int o = WAVE / 2; o > 0; o >>= 1) big = sort_hi(big, __shfl_xor(big, o, WAVE));
      v[0] = lane < n2 ? keep : big;
#pragma clang loop unroll(disable)
      for (int k = 2; k <= 64; k <<= 1) wave_sort_lanes<T, 1>(v, k, lane);
      if (lane < n2) a[lane] = v[0];
    }
    __syncthreads();
    return;
  }
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      // one compare-exchange per thread and iteration: pair t of the stage is (i, i | j) with bit j of i clear
      for (int t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int p = i | j;
        T x = a[i], y = a[p];
        const bool up = ((i & k) == 0);
        if ((x > y) == up) { a[i] = y; a[p] = x; }
      }
    }
  }
  __syncthreads();
}

// sum a K-vector of doubles over the block (blockDim.x <= 256); every thread gets the K sums in v[].
// `lds` needs >= 4*K doubles.  Deterministic (fixed tree, waves folded in order).
template <int K>
__device__ __forceinline__ void block_sum_vec_all(double (&v)[K], double* lds) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (nw == 1) {  // single-wave workgroup: shuffles only
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);  // (the total, in every lane)
    return;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = wave_sum(v[k]);
    if (lane == 0) lds[wid * K + k] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double r = 0;
    for (int i = 0; i < nw; ++i) r += lds[i * K + k];
    v[k] = r;
  }
}

// Order-preserving compaction slot for one "chunk" (one element per thread, in thread order):
// returns base + number of set flags among lower-numbered threads; `base` is advanced by the chunk
// total for every thread.  `wsum` is an LDS int[4].
__device__ __forceinline__ int block_compact_slot(bool flag, int& base, int* wsum) {
  const unsigned long long bal = __ballot(flag);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  const int before = __popcll(bal & ((1ull << lane) - 1ull));
  __syncthreads();
  if (lane == 0) wsum[wid] = __popcll(bal);
  __syncthreads();
  int off = 0, tot = 0;
  for (int i = 0; i < nw; ++i) { const int c = wsum[i]; if (i < wid) off += c; tot += c; }
  const int pos = base + off + before;
  base += tot;
  return pos;
}

// In-place inclusive scan of n ints (LDS or global); all threads call.  `part` is an LDS int[256].
__device__ __forceinline__ void block_inclusive_scan(int* a, int n, int* part) {
  const int t = threadIdx.x, nt = blockDim.x;
  const int per = (n + nt - 1) / nt;
  const int lo = min(t * per, n), hi = min(lo + per, n);
  int s = 0;
  for (int i = lo; i < hi; ++i) { s += a[i]; a[i] = s; }
  // exclusive prefix of the per-thread totals: shuffle scan inside the wave, LDS only across waves
  const int lane = t & 63, wid = t >> 6, nw = (nt + 63) >> 6;
  int incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
  int off = incl - s;
  __syncthreads();
  if (nw > 1) {
    if (lane == 63) part[wid] = incl;
    __syncthreads();
    for (int i = 0; i < wid; ++i) off += part[i];
  }
  if (off) for (int i = lo; i < hi; ++i) a[i] += off;
  __syncthreads();
}

__device__ __forceinline__ int next_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

#endif  // __HIPCC__
