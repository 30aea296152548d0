// nn_fused.hip — fused pointwise stages of the segmentation U-Net (bf16, NHWC / channels_last).
//
// The network's convolutions stay with PyTorch-ROCm (MIOpen/CK implicit GEMM on MFMA, north_star);
// everything BETWEEN them is pointwise and HBM-bound, and in eager PyTorch costs one full read+write
// pass per op (BatchNorm, ReLU, residual add, style add, nearest upsample: ~43 % of the forward time in
// the round-1 profile).  One kernel does all of it in a single pass:
//
//     sum = A (+ B)                                   A, B optionally read through a 2x nearest upsample
//     act = relu?( scale[c] * sum + shift[n, c] )     shift carries BN's bias and the style vector
//
// writing `sum` and/or `act`.  16-byte (8 x bf16) loads/stores per lane, fp32 math.
#include "common.h"

typedef unsigned short bf16_t;
struct alignas(16) bf16x8 { bf16_t v[8]; };

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                         // round to nearest even
  return (bf16_t)(u >> 16);
}

struct FusedArgs {
  const bf16_t* A;
  const bf16_t* B;      // may be NULL
  bf16_t* SUM;          // may be NULL
  bf16_t* ACT;          // may be NULL
  const float* scale;   // [C]      (NULL -> 1)
  const float* shift;   // [N, C] or [C] (shift_per_sample = 0)
  int N, H, W, C;       // output shape (NHWC)
  int upA, upB;         // read A / B at (h/2, w/2) of a [N, H/2, W/2, C] tensor
  int relu, shift_per_sample;
};

__global__ __launch_bounds__(256) void k_fused_act(FusedArgs a) {
  const int C8 = a.C >> 3;
  const size_t total = (size_t)a.N * a.H * a.W * C8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    size_t p = i / C8;
    const int w = (int)(p % a.W);
    p /= a.W;
    const int h = (int)(p % a.H);
    const int n = (int)(p / a.H);
    const size_t o = i * 8;
    size_t ia = o, ib = o;
    if (a.upA) ia = ((((size_t)n * (a.H >> 1) + (h >> 1)) * (a.W >> 1) + (w >> 1)) * C8 + c8) * 8;
    if (a.upB) ib = ((((size_t)n * (a.H >> 1) + (h >> 1)) * (a.W >> 1) + (w >> 1)) * C8 + c8) * 8;
    const bf16x8 va = *reinterpret_cast<const bf16x8*>(a.A + ia);
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = bf2f(va.v[k]);
    if (a.B) {
      const bf16x8 vb = *reinterpret_cast<const bf16x8*>(a.B + ib);
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += bf2f(vb.v[k]);
    }
    if (a.SUM) {
      bf16x8 r;
#pragma unroll
      for (int k = 0; k < 8; ++k) r.v[k] = f2bf(s[k]);
      *reinterpret_cast<bf16x8*>(a.SUM + o) = r;
    }
    if (a.ACT) {
      const float* sh = a.shift + (a.shift_per_sample ? (size_t)n * a.C : 0) + c8 * 8;
      bf16x8 r;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float v = s[k] * (a.scale ? a.scale[c8 * 8 + k] : 1.0f) + sh[k];
        if (a.relu) v = fmaxf(v, 0.0f);
        r.v[k] = f2bf(v);
      }
      *reinterpret_cast<bf16x8*>(a.ACT + o) = r;
    }
  }
}

// first layer: float32 NCHW tiles (2 channels) -> bf16 NHWC padded to 8 channels, with BN+ReLU variant
__global__ void k_tiles_to_nhwc8(const float* __restrict__ x, int N, int Cin, int H, int W, const float* scale,
                                 const float* shift, bf16_t* __restrict__ raw, bf16_t* __restrict__ act) {
  const size_t total = (size_t)N * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / ((size_t)H * W), p = i % ((size_t)H * W);
    bf16x8 r, q;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = 0.0f;
      if (k < Cin) v = x[(n * Cin + k) * (size_t)H * W + p];
      r.v[k] = f2bf(v);
      float t = (k < Cin) ? fmaxf(v * scale[k] + shift[k], 0.0f) : 0.0f;
      q.v[k] = f2bf(t);
    }
    *reinterpret_cast<bf16x8*>(raw + i * 8) = r;
    *reinterpret_cast<bf16x8*>(act + i * 8) = q;
  }
}

extern "C" {

int aliby_nn_fused_act_bf16(aliby_ctx* ctx, const void* A, const void* B, void* SUM, void* ACT, const float* scale,
                            const float* shift, int N, int H, int W, int C, int upA, int upB, int relu,
                            int shift_per_sample, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  ARG_CHECK(A && (SUM || ACT), "A and at least one output are required");
  ARG_CHECK(N > 0 && H > 0 && W > 0 && C > 0 && (C % 8) == 0, "C must be a multiple of 8");
  ARG_CHECK(!ACT || shift, "shift is required when ACT is written");
  ARG_CHECK(!(upA || upB) || ((H % 2) == 0 && (W % 2) == 0), "upsampled reads need even H and W");
  FusedArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.SUM = (bf16_t*)SUM; a.ACT = (bf16_t*)ACT;
  a.scale = scale; a.shift = shift; a.N = N; a.H = H; a.W = W; a.C = C; a.upA = upA; a.upB = upB;
  a.relu = relu; a.shift_per_sample = shift_per_sample;
  const size_t total = (size_t)N * H * W * (C / 8);
  size_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_fused_act, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
  KERNEL_CHECK();
  return ALIBY_OK;
}

int aliby_nn_tiles_to_nhwc8_bf16(aliby_ctx* ctx, const float* tiles, int N, int Cin, int H, int W, const float* scale,
                                 const float* shift, void* raw, void* act, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  ARG_CHECK(tiles && scale && shift && raw && act, "NULL argument");
  ARG_CHECK(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 8, "1 <= Cin <= 8");
  const size_t total = (size_t)N * H * W;
  size_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_tiles_to_nhwc8, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), tiles, N, Cin, H, W, scale,
                     shift, (bf16_t*)raw, (bf16_t*)act);
  KERNEL_CHECK();
  return ALIBY_OK;
}

}  // extern "C"
