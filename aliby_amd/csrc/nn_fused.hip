// nn_fused.hip — fused pointwise stages of the segmentation U-Net (bf16, NHWC / channels_last).
//
// The network's convolutions stay with PyTorch-ROCm (MIOpen/CK implicit GEMM on MFMA, north_star);
// everything BETWEEN them is pointwise and HBM-bound, and in eager PyTorch costs one full read+write
// pass per op (conv bias, BatchNorm, ReLU, residual add, style add, nearest upsample: ~45 % of the
// forward time in the round-1 profile).  One kernel does all of it in a single pass:
//
//     sum = A (+ B) (+ bias[c])                       A, B optionally read through a 2x nearest upsample;
//                                                     bias = the biases of the convolutions that made A/B
//     act = relu?( scale[c] * sum + shift[n, c] )     shift carries BN's bias and the style vector
//
// writing `sum` and/or `act`.  One workgroup per output row (n, h): no integer division per element,
// 16-byte (8 x bf16) loads/stores per lane, fp32 math.
#include "common.h"
#include <algorithm>

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
  const float* bias;    // [C] added to the sum (may be NULL)
  const float* scale;   // [C] (NULL -> 1)
  const float* shift;   // [N, C] or [C] (shift_per_sample = 0)
  int N, H, W, C;       // output shape (NHWC)
  int upA, upB;         // read A / B at (h/2, w/2) of a [N, H/2, W/2, C] tensor
  int relu, shift_per_sample;
  int c8_shift;         // log2(C/8) when C/8 is a power of two, else -1
};

__global__ __launch_bounds__(256) void k_fused_act(FusedArgs a) {
  const int C8 = a.C >> 3;
  const int h = blockIdx.x, n = blockIdx.y;
  const int rowv = a.W * C8;  // 16-byte vectors per output row
  const size_t orow = ((size_t)n * a.H + h) * (size_t)rowv;
  const size_t arow = a.upA ? ((size_t)n * (a.H >> 1) + (h >> 1)) * (size_t)((a.W >> 1) * C8) : orow;
  const size_t brow = a.upB ? ((size_t)n * (a.H >> 1) + (h >> 1)) * (size_t)((a.W >> 1) * C8) : orow;
  const bf16x8* A = reinterpret_cast<const bf16x8*>(a.A) + arow;
  const bf16x8* B = a.B ? reinterpret_cast<const bf16x8*>(a.B) + brow : nullptr;
  bf16x8* S = a.SUM ? reinterpret_cast<bf16x8*>(a.SUM) + orow : nullptr;
  bf16x8* T = a.ACT ? reinterpret_cast<bf16x8*>(a.ACT) + orow : nullptr;
  const float* sh = a.shift ? a.shift + (size_t)n * a.shift_per_sample : nullptr;  // 0 = shared, else row stride
  for (int i = threadIdx.x; i < rowv; i += blockDim.x) {
    int w, c8;
    if (a.c8_shift >= 0) { w = i >> a.c8_shift; c8 = i & (C8 - 1); }
    else { w = i / C8; c8 = i - w * C8; }
    const int ia = a.upA ? (w >> 1) * C8 + c8 : i;
    const bf16x8 va = A[ia];
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = bf2f(va.v[k]);
    if (B) {
      const int ib = a.upB ? (w >> 1) * C8 + c8 : i;
      const bf16x8 vb = B[ib];
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += bf2f(vb.v[k]);
    }
    if (a.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(a.bias + c8 * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(a.bias + c8 * 8 + 4);
      s[0] += b0.x; s[1] += b0.y; s[2] += b0.z; s[3] += b0.w;
      s[4] += b1.x; s[5] += b1.y; s[6] += b1.z; s[7] += b1.w;
    }
    if (S) {
      bf16x8 r;
#pragma unroll
      for (int k = 0; k < 8; ++k) r.v[k] = f2bf(s[k]);
      S[i] = r;
    }
    if (T) {
      float sc[8] = {1, 1, 1, 1, 1, 1, 1, 1};
      if (a.scale) {
        const float4 q0 = *reinterpret_cast<const float4*>(a.scale + c8 * 8);
        const float4 q1 = *reinterpret_cast<const float4*>(a.scale + c8 * 8 + 4);
        sc[0] = q0.x; sc[1] = q0.y; sc[2] = q0.z; sc[3] = q0.w; sc[4] = q1.x; sc[5] = q1.y; sc[6] = q1.z; sc[7] = q1.w;
      }
      const float4 t0 = *reinterpret_cast<const float4*>(sh + c8 * 8);
      const float4 t1 = *reinterpret_cast<const float4*>(sh + c8 * 8 + 4);
      const float tt[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      bf16x8 r;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float v = s[k] * sc[k] + tt[k];
        if (a.relu) v = fmaxf(v, 0.0f);
        r.v[k] = f2bf(v);
      }
      T[i] = r;
    }
  }
}

// first layer: float32 NCHW tiles (Cin <= 8 channels) -> bf16 NHWC padded to 8 channels, raw + relu(bn(x))
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

// network output: bf16 NHWC [N,H,W,Cpad] (+ bias) -> float32 NCHW [N,Cout,H,W]
__global__ void k_nhwc_to_nchw_f32(const bf16_t* __restrict__ y, int N, int H, int W, int Cpad, int Cout,
                                   const float* __restrict__ bias, float* __restrict__ out) {
  const size_t P = (size_t)H * W, total = (size_t)N * P;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / P, p = i % P;
    for (int c = 0; c < Cout; ++c) out[(n * Cout + c) * P + p] = bf2f(y[i * Cpad + c]) + (bias ? bias[c] : 0.0f);
  }
}

// (the output head lives in nn_conv.hip since round 3: k_out_head_mfma, on the matrix cores)

// style vector of the network and everything derived from it, one workgroup per sample:
//   style[n, c]  = mean over (y, x) of X[n, y, x, c]            (cellpose `make_style`: global average pool ...
//   style[n, :] /= sqrt(sum_c style[n, c]^2)                     ... L2-normalised)
//   shifts[n, j] = b[j] + sum_c style[n, c] * Wt[c, j]            (the per-sample shifts of every styled unit of the up path:
//                                                                 `batchconvstyle.full` folded with its BatchNorm, batched)
__global__ __launch_bounds__(256) void k_style(const bf16_t* __restrict__ x, int P, int C, const float* __restrict__ wt,
                                               const float* __restrict__ b, int J, float* __restrict__ style,
                                               float* __restrict__ shifts) {
  // grid (N, ceil(J/256)): every workgroup rebuilds the sample's style (the feature map is small and L2-resident) and
  // produces its 256-wide slice of the shifts; workgroup (n, 0) also writes the style vector.
  extern __shared__ float sv[];  // [C] style, then [groups][C] partial sums, then [8] reduction scratch
  const int C8 = C >> 3, groups = blockDim.x / C8;  // C8 lanes cover a pixel (16-byte loads), `groups` pixels at a time
  float* part = sv + C;
  float* red = part + groups * C;
  const int n = blockIdx.x, tid = threadIdx.x;
  const bf16x8* xn = reinterpret_cast<const bf16x8*>(x) + (size_t)n * P * C8;
  if (tid < groups * C8) {
    const int c8 = tid % C8, g = tid / C8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = g; p < P; p += groups) {
      const bf16x8 v = xn[(size_t)p * C8 + c8];
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += bf2f(v.v[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) part[g * C + c8 * 8 + k] = s[k];
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int g = 0; g < groups; ++g) s += part[g * C + c];  // fixed order: deterministic
    sv[c] = s / (float)P;
  }
  __syncthreads();
  float q = 0.f;
  for (int c = tid; c < C; c += blockDim.x) q += sv[c] * sv[c];
  for (int o = 32; o > 0; o >>= 1) q += __shfl_down(q, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = q;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
  const float inv = 1.0f / sqrtf(tot);
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    sv[c] *= inv;
    if (blockIdx.y == 0) style[(size_t)n * C + c] = sv[c];
  }
  __syncthreads();
  const int j = blockIdx.y * blockDim.x + tid;
  if (j < J) {
    float acc = b[j];
#pragma unroll 8
    for (int c = 0; c < C; ++c) acc += sv[c] * wt[(size_t)c * J + j];
    shifts[(size_t)n * J + j] = acc;
  }
}

extern "C" {

int aliby_nn_fused_act_bf16(aliby_ctx* ctx, const void* A, const void* B, void* SUM, void* ACT, const float* bias,
                            const float* scale, const float* shift, int N, int H, int W, int C, int upA, int upB,
                            int relu, int shift_per_sample, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  ARG_CHECK(A && (SUM || ACT), "A and at least one output are required");
  ARG_CHECK(N > 0 && H > 0 && W > 0 && C > 0 && (C % 8) == 0, "C must be a multiple of 8");
  ARG_CHECK(N <= 65535, "N exceeds the grid limit");
  ARG_CHECK(!ACT || shift, "shift is required when ACT is written");
  ARG_CHECK(!(upA || upB) || ((H % 2) == 0 && (W % 2) == 0), "upsampled reads need even H and W");
  FusedArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.SUM = (bf16_t*)SUM; a.ACT = (bf16_t*)ACT;
  a.bias = bias; a.scale = scale; a.shift = shift; a.N = N; a.H = H; a.W = W; a.C = C; a.upA = upA; a.upB = upB;
  a.relu = relu;
  a.shift_per_sample = shift_per_sample == 1 ? C : shift_per_sample;  // 1 = contiguous [N, C]; >1 = row stride in floats
  const int C8 = C / 8;
  a.c8_shift = -1;
  for (int k = 0; k < 16; ++k) if ((1 << k) == C8) a.c8_shift = k;
  hipLaunchKernelGGL(k_fused_act, dim3(H, N), dim3(256), 0, as_stream(stream), a);
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

int aliby_nn_nhwc_to_nchw_f32(aliby_ctx* ctx, const void* y, int N, int H, int W, int Cpad, int Cout, const float* bias,
                              float* out, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  ARG_CHECK(y && out, "NULL argument");
  ARG_CHECK(N > 0 && H > 0 && W > 0 && Cout > 0 && Cout <= Cpad, "bad shape");
  const size_t total = (size_t)N * H * W;
  size_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_nhwc_to_nchw_f32, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const bf16_t*)y, N, H, W, Cpad,
                     Cout, bias, out);
  KERNEL_CHECK();
  return ALIBY_OK;
}

int aliby_nn_style_bf16(aliby_ctx* ctx, const void* x, int N, int H, int W, int C, const float* wt, const float* b, int J,
                        float* style, float* shifts, void* stream) {
  ARG_CHECK(ctx && x && wt && b && style && shifts, "style: null argument");
  ARG_CHECK(N > 0 && H > 0 && W > 0 && J > 0 && C >= 8 && C % 8 == 0 && C <= 2048, "style: C must be a multiple of 8, <= 2048");
  const int groups = 256 / (C / 8) > 0 ? 256 / (C / 8) : 0;
  ARG_CHECK(groups >= 1, "style: C too wide for one workgroup");
  hipLaunchKernelGGL(k_style, dim3(N, (J + 255) / 256), dim3(256), sizeof(float) * ((size_t)C + (size_t)groups * C + 8), static_cast<hipStream_t>(stream),
                     static_cast<const bf16_t*>(x), H * W, C, wt, b, J, style, shifts);
  KERNEL_CHECK();
  return ALIBY_OK;
}

}  // extern "C"
