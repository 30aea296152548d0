// feat_intensity3d.hip — per-object intensity statistics over a labelled Z-stack [F, Z, Y, X] (round 3; an EXTENSION beyond
// what the reference wires).
//
// BASELINE config 5 asks for "3D Cellpose + 3D intensity features".  As shipped, the reference collapses a 3-D label result to
// 2-D (src/aliby/segment/dispatch.py:218-223) and max-projects the pixels before every feature call (pipe_builder.py:120,
// distributors.py:22-24), so nothing 3-D ever reaches cp_measure; SURVEY.md §8(d).5 therefore asks for two numbers, the
// reference-faithful projected one and a true-3-D one "as a parity-unpinned extension".  This is the feature half of the
// second: the moment-based statistics of CellProfiler's MeasureObjectIntensity on a volume (the names cp_measure's
// `intensity` uses in 2-D): Volume (voxel count), IntegratedIntensity, MeanIntensity, StdIntensity (population), MinIntensity,
// MaxIntensity, CenterMassIntensity_X/Y/Z (intensity-weighted) and Center_X/Y/Z (geometric), 12 columns per object and channel.
//
// One pass over (labels, pixels): uint16 in, every sum an exact 64-bit integer (sum v, sum v^2, sum x v, ... < 2^63 for any
// object that fits a stack), so the result does not depend on the order of the atomics: run-to-run deterministic.  A lane walks
// 16 consecutive voxels of a row and flushes its running sums whenever the label changes: one set of atomics per run of equal
// labels, not per voxel.  HBM-bound: (2 + 2) bytes per voxel.
#include "common.h"

typedef unsigned short u16;
typedef unsigned long long u64;

#define I3_ACC 10  // n, sum v, sum v^2, sum x v, sum y v, sum z v, sum x, sum y, sum z  (+ 1 spare); min / max live in their own arrays

namespace {

struct Run {
  u64 n, s, s2, xv, yv, zv, sx, sy, sz;
  unsigned mn, mx;
  __device__ void reset() { n = s = s2 = xv = yv = zv = sx = sy = sz = 0; mn = 0xffffffffu; mx = 0u; }
  __device__ void add(unsigned v, unsigned x, unsigned y, unsigned z) {
    n += 1; s += v; s2 += (u64)v * v; xv += (u64)x * v; yv += (u64)y * v; zv += (u64)z * v; sx += x; sy += y; sz += z;
    mn = v < mn ? v : mn; mx = v > mx ? v : mx;
  }
};

__device__ __forceinline__ void flush(const Run& r, u64* acc, unsigned* mn, unsigned* mx) {
  atomicAdd(&acc[0], r.n); atomicAdd(&acc[1], r.s); atomicAdd(&acc[2], r.s2); atomicAdd(&acc[3], r.xv); atomicAdd(&acc[4], r.yv);
  atomicAdd(&acc[5], r.zv); atomicAdd(&acc[6], r.sx); atomicAdd(&acc[7], r.sy); atomicAdd(&acc[8], r.sz);
  atomicMin(mn, r.mn);
  atomicMax(mx, r.mx);
}

// labels [F, Z, Y, X]; pixels [F, C, Z, Y, X], channel c; offsets[f] = first row of stack f; row = offsets[f] + label - 1
__global__ __launch_bounds__(256) void k_intensity3d(const u16* __restrict__ labels, const u16* __restrict__ pixels, int F, int C, int Z, int Y,
                                                     int X, int c, const int* __restrict__ offsets, u64* __restrict__ acc,
                                                     unsigned* __restrict__ vmin, unsigned* __restrict__ vmax) {
  const size_t vol = (size_t)Z * Y * X;
  const int segs = (X + 15) / 16;  // 16-voxel segments per row
  const size_t total = (size_t)F * Z * Y * segs;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int sg = (int)(i % segs);
    size_t rest = i / segs;
    const int y = (int)(rest % Y);
    rest /= Y;
    const int z = (int)(rest % Z), f = (int)(rest / Z);
    const size_t row = (size_t)f * vol + ((size_t)z * Y + y) * X;
    const u16* lb = labels + row;
    const u16* px = pixels + ((size_t)f * C + c) * vol + ((size_t)z * Y + y) * X;
    const int x0 = sg * 16, x1 = min(X, x0 + 16);
    const int base = offsets[f], nrows = offsets[f + 1] - base;
    Run r;
    r.reset();
    unsigned cur = 0;
    for (int x = x0; x < x1; ++x) {
      const unsigned L = lb[x];
      if (L != cur) {
        if (cur && (int)cur <= nrows) flush(r, acc + (size_t)(base + cur - 1) * I3_ACC, vmin + base + cur - 1, vmax + base + cur - 1);
        r.reset();
        cur = L;
      }
      if (L) r.add(px[x], (unsigned)x, (unsigned)y, (unsigned)z);
    }
    if (cur && (int)cur <= nrows) flush(r, acc + (size_t)(base + cur - 1) * I3_ACC, vmin + base + cur - 1, vmax + base + cur - 1);
  }
}

__global__ void k_intensity3d_finish(const u64* __restrict__ acc, const unsigned* __restrict__ vmin, const unsigned* __restrict__ vmax, int n,
                                     double* __restrict__ out, int ld, int col0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64* a = acc + (size_t)i * I3_ACC;
  double* o = out + (size_t)i * ld + col0;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  const double cnt = (double)a[0], s = (double)a[1];
  if (a[0] == 0) {  // a label without voxels (cannot come from a sequential relabel; kept NaN like the 2-D block)
    for (int k = 0; k < 12; ++k) o[k] = nan;
    o[0] = 0.0;
    return;
  }
  const double mean = s / cnt;
  // population variance from exact integer sums: (n * sum v^2 - (sum v)^2) / n^2, the numerator in 128-bit integer arithmetic
  const unsigned __int128 num = (unsigned __int128)a[0] * a[2] - (unsigned __int128)a[1] * a[1];
  const double var = (double)num / (cnt * cnt);
  o[0] = cnt;
  o[1] = s;
  o[2] = mean;
  o[3] = sqrt(var);
  o[4] = (double)vmin[i];
  o[5] = (double)vmax[i];
  o[6] = a[1] ? (double)a[3] / s : nan;  // CenterMassIntensity_X / _Y / _Z
  o[7] = a[1] ? (double)a[4] / s : nan;
  o[8] = a[1] ? (double)a[5] / s : nan;
  o[9] = (double)a[6] / cnt;             // Center_X / _Y / _Z
  o[10] = (double)a[7] / cnt;
  o[11] = (double)a[8] / cnt;
}

}  // namespace

extern "C" int aliby_features_intensity3d(aliby_ctx* ctx, const uint16_t* labels, const uint16_t* pixels, int F, int C, int Z, int Y, int X,
                                          int channel, const int32_t* offsets_host, double* out, int ld, int col0, void* stream) {
  ARG_CHECK(ctx && labels && pixels && offsets_host && out, "intensity3d: null argument");
  ARG_CHECK(F > 0 && C > 0 && Z > 0 && Y > 0 && X > 0 && channel >= 0 && channel < C, "intensity3d: bad shape");
  ARG_CHECK(offsets_host[0] == 0 && col0 >= 0 && ld >= col0 + 12, "intensity3d: bad offsets / output stride");
  // (x, y, z < 65536 and at most 2^32 voxels per object keep every sum below 2^63)
  ARG_CHECK(X <= 65536 && Y <= 65536 && Z <= 65536 && (size_t)Z * Y * X <= (1ull << 32), "intensity3d: stack too large for exact 64-bit sums");
  const int n = offsets_host[F];
  if (n <= 0) return ALIBY_OK;
  hipStream_t s = as_stream(stream);
  const size_t acc_bytes = sizeof(u64) * (size_t)n * I3_ACC;
  int rc = aliby_ensure_scratch(ctx, acc_bytes + 2 * sizeof(unsigned) * (size_t)n + sizeof(int) * (size_t)(F + 1) + 64);
  if (rc) return rc;
  u64* acc = (u64*)ctx->scratch;
  unsigned* vmin = (unsigned*)((char*)ctx->scratch + acc_bytes);
  unsigned* vmax = vmin + n;
  int* d_off = (int*)(vmax + n);
  HIP_TRY(hipMemsetAsync(acc, 0, acc_bytes, s));
  HIP_TRY(hipMemsetAsync(vmin, 0xFF, sizeof(unsigned) * (size_t)n, s));
  HIP_TRY(hipMemsetAsync(vmax, 0, sizeof(unsigned) * (size_t)n, s));
  HIP_TRY(hipMemcpyAsync(d_off, offsets_host, sizeof(int) * (size_t)(F + 1), hipMemcpyHostToDevice, s));
  const size_t total = (size_t)F * Z * Y * ((X + 15) / 16);
  const unsigned grid = (unsigned)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
  hipLaunchKernelGGL(k_intensity3d, dim3(grid), dim3(256), 0, s, labels, pixels, F, C, Z, Y, X, channel, d_off, acc, vmin, vmax);
  KERNEL_CHECK();
  hipLaunchKernelGGL(k_intensity3d_finish, dim3((n + 255) / 256), dim3(256), 0, s, acc, vmin, vmax, n, out, ld, col0);
  KERNEL_CHECK();
  // the offsets live in ctx scratch: they must be consumed before the host reuses it
  { const int rcw = aliby_wait_stream(s); if (rcw) return rcw; }
  return ALIBY_OK;
}
