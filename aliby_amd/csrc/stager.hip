// stager.hip — TCZYX tile stager: crop / pad (a4) and z-reduction (a5).
//
//   crop/pad : Tiler.get_tp_channel + if_out_of_bounds_pad (src/aliby/tile/tiler.py:335-366,601-650)
//              with the slices of Tile.as_range (src/aliby/tile/tiles.py:151-166).
//   reduce_z : extraction/core/functions/distributors.py:6-24 — `ufunc.reduce(pixels, axis)`.
//
// Both are pure streaming copies (HBM-bound): one 16-byte load/store per lane where rows allow it.
// The padded border follows numpy.pad(mode="median") exactly: axis by axis, one median per line
// (rounded half-to-even for integer data), the x pass seeing the rows the y pass just created.
#include "common.h"

typedef unsigned short u16;

struct CropArgs {
  const u16* stack;  // [C,Z,Y,X]
  int C, Z, Y, X;
  const int* rects;  // [F,4] y0,x0,h,w (device)
  const int* flags;  // [F] 1 = NaN tile (skip)
  int F, h, w;
  u16* out;          // [F,C,Z,h,w]
};

// interior copy: out[f,c,z,r,q] = stack[c,z,y0+r,x0+q] where inside the image
__global__ void k_crop_copy(CropArgs a) {
  const int f = blockIdx.z;
  const int cz = blockIdx.y;  // c*Z+z
  if (a.flags[f]) return;
  const int y0 = a.rects[f * 4 + 0], x0 = a.rects[f * 4 + 1];
  const u16* src = a.stack + (size_t)cz * a.Y * a.X;
  u16* dst = a.out + ((size_t)f * a.C * a.Z + cz) * (size_t)a.h * a.w;
  const int n = a.h * a.w;
  // rows of whole 16-byte groups on both sides (tile width, frame width and the rect's x origin multiples of 8 pixels, both
  // base pointers 16-byte aligned): one uint4 per lane; anything else, and groups that straddle the frame's edge, go pixel by pixel
  const bool vec = (a.w % 8 == 0) && (a.X % 8 == 0) && (x0 % 8 == 0) && ((((size_t)(const void*)src) | ((size_t)(void*)dst)) % 16 == 0);
  if (vec) {
    const int ng = n / 8, wg = a.w / 8;
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += gridDim.x * blockDim.x) {
      const int r = g / wg, q = (g - r * wg) * 8;
      const int y = y0 + r, x = x0 + q;
      if (y < 0 || y >= a.Y) continue;
      if (x >= 0 && x + 8 <= a.X) {
        *reinterpret_cast<uint4*>(dst + (size_t)g * 8) = *reinterpret_cast<const uint4*>(src + (size_t)y * a.X + x);
      } else {
        for (int k = 0; k < 8; ++k)
          if (x + k >= 0 && x + k < a.X) dst[(size_t)g * 8 + k] = src[(size_t)y * a.X + x + k];
      }
    }
    return;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int r = i / a.w, q = i % a.w;
    const int y = y0 + r, x = x0 + q;
    if (y >= 0 && y < a.Y && x >= 0 && x < a.X) dst[i] = src[(size_t)y * a.X + x];
  }
}

__device__ __forceinline__ u16 median_round_u16(const u16* sorted, int n) {
  if (n & 1) return sorted[n / 2];
  const unsigned s = (unsigned)sorted[n / 2 - 1] + (unsigned)sorted[n / 2];
  unsigned m = s >> 1;
  if (s & 1) m += (m & 1);  // x.5 -> nearest even (numpy.round)
  return (u16)m;
}

// pass over axis y: one block per (f, cz, column x inside the original x-range)
__global__ void k_pad_y(CropArgs a) {
  extern __shared__ u16 line[];
  const int f = blockIdx.z, cz = blockIdx.y;
  if (a.flags[f]) return;
  const int y0 = a.rects[f * 4 + 0], x0 = a.rects[f * 4 + 1];
  const int ya = max(0, -y0), yb = min(a.h, a.Y - y0);  // original rows [ya,yb) in tile coords
  const int xa = max(0, -x0), xb = min(a.w, a.X - x0);
  if (ya == 0 && yb == a.h) return;
  const int q = xa + blockIdx.x;
  if (q >= xb) return;
  u16* dst = a.out + ((size_t)f * a.C * a.Z + cz) * (size_t)a.h * a.w;
  const int n = yb - ya;
  const int n2 = next_pow2(n);
  for (int i = threadIdx.x; i < n2; i += blockDim.x) line[i] = (i < n) ? dst[(size_t)(ya + i) * a.w + q] : (u16)0xFFFF;
  // 0xFFFF padding sorts last; real 0xFFFF values are indistinguishable but equal, so medians are unaffected
  block_bitonic_sort(line, n2);
  const u16 m = median_round_u16(line, n);
  for (int r = threadIdx.x; r < a.h; r += blockDim.x)
    if (r < ya || r >= yb) dst[(size_t)r * a.w + q] = m;
}

// pass over axis x: one block per (f, cz, row r of the full padded height)
__global__ void k_pad_x(CropArgs a) {
  extern __shared__ u16 line[];
  const int f = blockIdx.z, cz = blockIdx.y;
  if (a.flags[f]) return;
  const int x0 = a.rects[f * 4 + 1];
  const int xa = max(0, -x0), xb = min(a.w, a.X - x0);
  if (xa == 0 && xb == a.w) return;
  const int r = blockIdx.x;
  u16* dst = a.out + ((size_t)f * a.C * a.Z + cz) * (size_t)a.h * a.w + (size_t)r * a.w;
  const int n = xb - xa;
  const int n2 = next_pow2(n);
  for (int i = threadIdx.x; i < n2; i += blockDim.x) line[i] = (i < n) ? dst[xa + i] : (u16)0xFFFF;
  block_bitonic_sort(line, n2);
  const u16 m = median_round_u16(line, n);
  for (int q = threadIdx.x; q < a.w; q += blockDim.x)
    if (q < xa || q >= xb) dst[q] = m;
}

// ---------------------------------------------------------------------------------------------
// reduce_z
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO, int OP>
__global__ void k_reduce_z(const TI* __restrict__ in, size_t outer, int Z, size_t inner, TO* __restrict__ out) {
  const size_t total = outer * inner;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t o = i / inner, k = i % inner;
    const TI* p = in + o * (size_t)Z * inner + k;
    if (OP == ALIBY_RED_MAX) {
      TI m = p[0];
      for (int z = 1; z < Z; ++z) { TI v = p[(size_t)z * inner]; m = v > m ? v : m; }
      out[i] = (TO)m;
    } else if (OP == ALIBY_RED_ADD) {
      // numpy: add.reduce upcasts small unsigned ints (no wrap); float32 stays float32 (left fold)
      if (sizeof(TI) == 2) {
        unsigned long long s = 0;
        for (int z = 0; z < Z; ++z) s += (unsigned long long)p[(size_t)z * inner];
        out[i] = (TO)s;
      } else {
        float s = (float)p[0];
        for (int z = 1; z < Z; ++z) s += (float)p[(size_t)z * inner];
        out[i] = (TO)s;
      }
    } else {  // true-divide left fold; integers divide in float64 like numpy
      if (sizeof(TI) == 2) {
        double s = (double)p[0];
        for (int z = 1; z < Z; ++z) s = s / (double)p[(size_t)z * inner];
        out[i] = (TO)s;
      } else {
        float s = (float)p[0];
        for (int z = 1; z < Z; ++z) s = s / (float)p[(size_t)z * inner];
        out[i] = (TO)s;
      }
    }
  }
}

extern "C" {

int aliby_crop_pad_u16(aliby_ctx* ctx, const uint16_t* stack, int C, int Z, int Y, int X,
                       const int32_t* rects, int F, int h, int w, uint16_t* out,
                       int32_t* nan_flags_host, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  ARG_CHECK(C > 0 && Z > 0 && Y > 0 && X > 0 && F >= 0 && h > 0 && w > 0, "bad shape");
  if (F == 0) return ALIBY_OK;
  ARG_CHECK(stack && rects && out && nan_flags_host, "NULL argument");
  ARG_CHECK((size_t)C * Z <= 65535 && F <= 65535, "grid dimension overflow");
  bool any_y = false, any_x = false;
  for (int f = 0; f < F; ++f) {
    const int y0 = rects[f * 4 + 0], x0 = rects[f * 4 + 1];
    ARG_CHECK(rects[f * 4 + 2] == h && rects[f * 4 + 3] == w, "all tiles must share (h,w)");
    // if_out_of_bounds_pad: padding = [[yb, ya], [xb, xa]]; `(padding / 0.25 > tile_shape).any()`
    // broadcasts tile_shape=(h,w) over the LAST axis, i.e. before-pads compare with h, after-pads with w.
    const int pyb = y0 < 0 ? -y0 : 0, pya = (y0 + h > Y) ? (y0 + h - Y) : 0;
    const int pxb = x0 < 0 ? -x0 : 0, pxa = (x0 + w > X) ? (x0 + w - X) : 0;
    const bool nan = (pyb * 4 > h) || (pya * 4 > w) || (pxb * 4 > h) || (pxa * 4 > w);
    nan_flags_host[f] = nan ? 1 : 0;
    if (!nan) { any_y |= (pyb || pya); any_x |= (pxb || pxa); }
  }
  int rc = aliby_ensure_scratch(ctx, sizeof(int) * (size_t)F * 5);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  int* d_rects = (int*)ctx->scratch;
  int* d_flags = d_rects + (size_t)F * 4;
  HIP_TRY(hipMemcpyAsync(d_rects, rects, sizeof(int) * (size_t)F * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_flags, nan_flags_host, sizeof(int) * (size_t)F, hipMemcpyHostToDevice, s));
  CropArgs a;
  a.stack = stack; a.C = C; a.Z = Z; a.Y = Y; a.X = X; a.rects = d_rects; a.flags = d_flags;
  a.F = F; a.h = h; a.w = w; a.out = out;
  int bx = (h * w + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(k_crop_copy, dim3(bx, C * Z, F), dim3(256), 0, s, a);
  KERNEL_CHECK();
  if (any_y) {
    int n2 = 1; while (n2 < h) n2 <<= 1;
    hipLaunchKernelGGL(k_pad_y, dim3(w, C * Z, F), dim3(128), n2 * sizeof(u16), s, a);
    KERNEL_CHECK();
  }
  if (any_x) {
    int n2 = 1; while (n2 < w) n2 <<= 1;
    hipLaunchKernelGGL(k_pad_x, dim3(h, C * Z, F), dim3(128), n2 * sizeof(u16), s, a);
    KERNEL_CHECK();
  }
  // rects/flags live in ctx scratch: make sure the copies are consumed before the host reuses it
  { const int rcw = aliby_wait_stream(s); if (rcw) return rcw; }
  return ALIBY_OK;
}

int aliby_reduce_z(aliby_ctx* ctx, const void* in, int dtype, size_t outer, int Z, size_t inner, int op,
                   void* out, int out_dtype, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  ARG_CHECK(Z > 0, "Z must be positive");
  if (outer == 0 || inner == 0) return ALIBY_OK;
  ARG_CHECK(in && out, "NULL argument");
  ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32, "bad dtype");
  if (op != ALIBY_RED_MAX && op != ALIBY_RED_ADD && op != ALIBY_RED_DIV) {
    aliby_set_error("reduce_z: operator %d is an invalid reducer (only ufuncs max/add/div)", op);
    return ALIBY_ERR_UNSUPPORTED;
  }
  const int numpy_out = (dtype != ALIBY_U16) ? ALIBY_F32 : (op == ALIBY_RED_ADD ? ALIBY_U64 : ALIBY_F64);
  ARG_CHECK(op == ALIBY_RED_MAX ? out_dtype == dtype : (out_dtype == ALIBY_F32 || out_dtype == numpy_out),
            "out_dtype: max keeps the dtype; add/div produce f32, or NumPy's own u64 (u16 add) / f64 (u16 div)");
  hipStream_t s = as_stream(stream);
  const size_t total = outer * inner;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  dim3 g((unsigned)blocks), b(256);
#define LAUNCH(TI, TO, OP) hipLaunchKernelGGL((k_reduce_z<TI, TO, OP>), g, b, 0, s, (const TI*)in, outer, Z, inner, (TO*)out)
  if (dtype == ALIBY_U16) {
    if (op == ALIBY_RED_MAX) LAUNCH(u16, u16, ALIBY_RED_MAX);
    else if (op == ALIBY_RED_ADD) { if (out_dtype == ALIBY_U64) LAUNCH(u16, unsigned long long, ALIBY_RED_ADD); else LAUNCH(u16, float, ALIBY_RED_ADD); }
    else { if (out_dtype == ALIBY_F64) LAUNCH(u16, double, ALIBY_RED_DIV); else LAUNCH(u16, float, ALIBY_RED_DIV); }
  } else {
    if (op == ALIBY_RED_MAX) LAUNCH(float, float, ALIBY_RED_MAX);
    else if (op == ALIBY_RED_ADD) LAUNCH(float, float, ALIBY_RED_ADD);
    else LAUNCH(float, float, ALIBY_RED_DIV);
  }
#undef LAUNCH
  KERNEL_CHECK();
  return ALIBY_OK;
}

}  // extern "C"
