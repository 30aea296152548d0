// nn_conv1x1.hip — the U-Net's 1x1 projections (bf16 NHWC) as a hand-written MFMA GEMM, and its 2-channel first layer.
//
// cellpose's residual blocks add `proj(x)` = BatchNorm -> Conv1x1 of the block input to the block's first two
// convolutions (reference call site: the network `model.eval` runs, src/aliby/segment/dispatch.py:208-215).  With the
// BatchNorm folded into the weights (host side) a projection is a plain GEMM  OUT[p, co] = sum_ci W[co, ci] X[p, ci] + b[co]
// over all N*H*W pixels p: HBM-bound (CIN + COUT channels of bf16 per pixel against 2*CIN*COUT flops).  Levels 0-1 fuse it
// into conv1 (nn_conv.hip); this kernel serves the deep / up blocks, where the inputs are small.
//
//   * a workgroup takes 128 (64 at 256 input channels) consecutive pixels x up to 128 output channels; its X tile is read once with 16-byte
//     coalesced loads and stored to LDS as channel-octet planes [octet][pixel] (the layout nn_conv.hip uses), so every
//     B fragment is one conflict-free ds_read_b128;
//   * the wave's weight fragments (CIN/16 x 4 VGPRs) stay in registers; fragment order and the output-channel
//     permutation are those of aliby_nn_pack_conv1x1_bf16, so a lane ends with 16 contiguous channels of one pixel.
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ unsigned p1_pack2(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}

struct P1Args {
  const uint4* in;    // [P, CIN] bf16
  const uint4* wpk;   // [COUT/32][CIN/16][64] fragments
  const float* bias;  // [COUT] or NULL
  uint4* out;         // [P, COUT] bf16
  size_t P;
  int COUT;
};

template <int CIN>
__global__ __launch_bounds__(256) void k_conv1x1(P1Args a) {
  constexpr int KC = CIN / 16, NPL = CIN / 8, TP = CIN >= 256 ? 64 : 128, PP = TP + 1;  // X tile <= 33 KB of LDS
  __shared__ uint4 planes[NPL * PP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 31, hh = lane >> 5;
  const int ncb = min(a.COUT / 32 - (int)blockIdx.y * 4, 4);  // output-channel blocks of this workgroup (1, 2 or 4)
  const int cb = wave % ncb, pg0 = wave / ncb, pgs = 4 / ncb;  // this wave: block cb, 32-pixel groups pg0, pg0 + pgs, ...
  const int cbg = blockIdx.y * 4 + cb;
  const size_t p0 = (size_t)blockIdx.x * TP;

  bf16x8_t wfrag[KC];
  {
    const bf16x8_t* wp = reinterpret_cast<const bf16x8_t*>(a.wpk) + (size_t)cbg * KC * 64 + lane;
#pragma unroll
    for (int k = 0; k < KC; ++k) wfrag[k] = wp[k * 64];
  }
  // ---- X tile -> LDS planes (rows past the end are clamped: computed, never stored)
#pragma unroll
  for (int it = 0; it < (TP * NPL) / 256; ++it) {
    const int u = tid + it * 256, pix = u / NPL, oct = u % NPL;
    const size_t gp = min(p0 + pix, a.P - 1);
    planes[oct * PP + pix] = a.in[gp * NPL + oct];
  }
  __syncthreads();
  const int c0 = cbg * 32 + hh * 16;
  float4 b4[4] = {};
  if (a.bias) {
    const float4* bp = reinterpret_cast<const float4*>(a.bias + c0);
#pragma unroll
    for (int q = 0; q < 4; ++q) b4[q] = bp[q];
  }
  const bf16x8_t* L = reinterpret_cast<const bf16x8_t*>(planes) + hh * PP + px;
  for (int g = pg0; g < TP / 32; g += pgs) {
    f32x16_t acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[4 * q] = b4[q].x; acc[4 * q + 1] = b4[q].y; acc[4 * q + 2] = b4[q].z; acc[4 * q + 3] = b4[q].w; }
#pragma unroll
    for (int k = 0; k < KC; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[k], L[2 * k * PP + g * 32], acc, 0, 0, 0);
    const size_t gp = p0 + g * 32 + px;
    if (gp < a.P) {
      uint4* op = a.out + gp * (a.COUT / 8) + (c0 >> 3);
      op[0] = make_uint4(p1_pack2(acc[0], acc[1]), p1_pack2(acc[2], acc[3]), p1_pack2(acc[4], acc[5]), p1_pack2(acc[6], acc[7]));
      op[1] = make_uint4(p1_pack2(acc[8], acc[9]), p1_pack2(acc[10], acc[11]), p1_pack2(acc[12], acc[13]), p1_pack2(acc[14], acc[15]));
    }
  }
}

// ---- first layer: float32 NCHW tiles with Cin <= 2 channels -> c0 = conv3x3(bf16(relu(scale*x + shift))) as bf16
// NHWC[32] (no bias: it rides in the next unit's shift) and the raw input as bf16 NHWC[8] (the projection's input).
// The 9*Cin <= 18 taps of a pixel are laid out as K = 32 (k = 12*c + 4*ty + tx, tx = 3 and k >= 24 carry zero weights),
// which makes the LDS offset of a tap linear in (c, ty, tx): two k-steps of the 32x32x16 MFMA per 32 pixels, the im2col
// fragment gathered straight from the activated window in LDS.  HBM-bound: 8 bytes read, 64 + 16 written per pixel.
struct FirstArgs {
  const float* x;       // [N, Cin, H, W]
  const float* scale;   // [8] (first Cin used)
  const float* shift;   // [8]
  const float* w;       // [32][Cin][9] float32 (bf16-representable values)
  unsigned short* raw;  // [N, H, W, 8]
  unsigned short* c0;   // [N, H, W, 32]
  int N, Cin, H, W;
};

__device__ __forceinline__ constexpr int first_tap_offset(int k, int plane, int lw) {  // k -> offset in the LDS window
  return (k / 12) * plane + ((k % 12) / 4) * lw + (k % 4);
}

__global__ __launch_bounds__(256) void k_first_conv(FirstArgs a) {
  constexpr int TW = 32, TH = 8, LW = TW + 3, LH = TH + 2, PLANE = LH * LW;  // one spare column: tx = 3 reads stay inside
  __shared__ unsigned short act[2 * PLANE + 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 31, hh = lane >> 5;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, n = blockIdx.z;
  const size_t plane = (size_t)a.H * a.W;
  // ---- weights as MFMA A fragments: row m <-> output channel 16*((m>>2)&1) + (m&3) + 4*(m>>3) (as in nn_conv.hip)
  bf16x8_t wfrag[2];
  {
    const int m = lane & 31, co = 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3);
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      unsigned r4[4];
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int k = 16 * kc + 8 * hh + 2 * j2 + e, c = k / 12, ty = (k % 12) / 4, tx = k % 4;
          v[e] = (c < a.Cin && tx < 3 && k < 24) ? a.w[((size_t)co * a.Cin + c) * 9 + ty * 3 + tx] : 0.f;
        }
        r4[j2] = p1_pack2(v[0], v[1]);
      }
      wfrag[kc] = __builtin_bit_cast(bf16x8_t, make_uint4(r4[0], r4[1], r4[2], r4[3]));
    }
  }
  // ---- activated window as bf16 (zero padding applies to the ACTIVATED tensor)
  for (int i = tid; i < 2 * PLANE + 8; i += 256) {
    const int c = i / PLANE, r = i % PLANE, ly = r / LW, lx = r % LW;
    const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
    float t = 0.f;
    if (c < a.Cin && i < 2 * PLANE && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
      t = fmaxf(a.x[((size_t)n * a.Cin + c) * plane + (size_t)gy * a.W + gx] * a.scale[c] + a.shift[c], 0.f);
    act[i] = (unsigned short)(p1_pack2(t, 0.f) & 0xffffu);
  }
  __syncthreads();
  const int gx = x0 + px;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int ly = wave * 2 + rr, gy = y0 + ly;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      unsigned r4[4];
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) {
        unsigned short e2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int k_lo = 16 * kc + 2 * j2 + e, k_hi = k_lo + 8;  // this lane's k for hh = 0 / 1 (compile-time pair)
          const int off = hh ? first_tap_offset(k_hi < 24 ? k_hi : 0, PLANE, LW) : first_tap_offset(k_lo < 24 ? k_lo : 0, PLANE, LW);
          e2[e] = act[off + ly * LW + px];  // k >= 24 meets a zero weight: any in-range value will do
        }
        r4[j2] = (unsigned)e2[0] | ((unsigned)e2[1] << 16);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[kc], __builtin_bit_cast(bf16x8_t, make_uint4(r4[0], r4[1], r4[2], r4[3])), acc, 0, 0, 0);
    }
    if (gx < a.W && gy < a.H) {
      const size_t p = (size_t)n * plane + (size_t)gy * a.W + gx;
      uint4* op = reinterpret_cast<uint4*>(a.c0) + p * 4 + hh * 2;  // lane holds channels 16*hh .. 16*hh + 15
      op[0] = make_uint4(p1_pack2(acc[0], acc[1]), p1_pack2(acc[2], acc[3]), p1_pack2(acc[4], acc[5]), p1_pack2(acc[6], acc[7]));
      op[1] = make_uint4(p1_pack2(acc[8], acc[9]), p1_pack2(acc[10], acc[11]), p1_pack2(acc[12], acc[13]), p1_pack2(acc[14], acc[15]));
      if (hh == 0) {  // raw copy, 8 channels (zeros past Cin)
        const float v0 = a.x[((size_t)n * a.Cin) * plane + (size_t)gy * a.W + gx];
        const float v1 = a.Cin > 1 ? a.x[((size_t)n * a.Cin + 1) * plane + (size_t)gy * a.W + gx] : 0.f;
        reinterpret_cast<uint4*>(a.raw)[p] = make_uint4(p1_pack2(v0, v1), 0u, 0u, 0u);
      }
    }
  }
}

template <int CIN>
int launch_1x1(P1Args& a, hipStream_t s) {
  constexpr int TP = CIN >= 256 ? 64 : 128;
  dim3 grid((unsigned)((a.P + TP - 1) / TP), (unsigned)((a.COUT / 32 + 3) / 4));
  hipLaunchKernelGGL((k_conv1x1<CIN>), grid, dim3(256), 0, s, a);
  KERNEL_CHECK();
  return ALIBY_OK;
}

}  // namespace

extern "C" int aliby_nn_conv1x1_bf16(aliby_ctx* ctx, const void* in, const void* wpk, const float* bias, void* out, int N,
                                     int H, int W, int CIN, int COUT, void* stream) {
  ARG_CHECK(ctx && in && wpk && out, "conv1x1: null argument");
  ARG_CHECK(N > 0 && H > 0 && W > 0, "conv1x1: empty shape");
  ARG_CHECK(COUT > 0 && COUT % 32 == 0 && (COUT / 32 <= 4 ? (COUT / 32 == 1 || COUT / 32 == 2 || COUT / 32 == 4) : COUT % 128 == 0),
            "conv1x1: COUT must be 32, 64 or a multiple of 128");
  P1Args a;
  a.in = static_cast<const uint4*>(in);
  a.wpk = static_cast<const uint4*>(wpk);
  a.bias = bias;
  a.out = static_cast<uint4*>(out);
  a.P = (size_t)N * H * W;
  a.COUT = COUT;
  hipStream_t s = as_stream(stream);
  switch (CIN) {
    case 32: return launch_1x1<32>(a, s);
    case 64: return launch_1x1<64>(a, s);
    case 128: return launch_1x1<128>(a, s);
    case 256: return launch_1x1<256>(a, s);
    default:
      aliby_set_error("conv1x1: unsupported CIN=%d (32, 64, 128, 256)", CIN);
      return ALIBY_ERR_UNSUPPORTED;
  }
}

extern "C" int aliby_nn_first_conv_bf16(aliby_ctx* ctx, const float* tiles, int N, int Cin, int H, int W, const float* scale,
                                        const float* shift, const float* w_oihw, void* raw8, void* c0, void* stream) {
  ARG_CHECK(ctx && tiles && scale && shift && w_oihw && raw8 && c0, "first_conv: null argument");
  ARG_CHECK(N > 0 && H > 0 && W > 0 && Cin >= 1 && Cin <= 2, "first_conv: Cin must be 1 or 2");
  FirstArgs a;
  a.x = tiles; a.scale = scale; a.shift = shift; a.w = w_oihw;
  a.raw = static_cast<unsigned short*>(raw8);
  a.c0 = static_cast<unsigned short*>(c0);
  a.N = N; a.Cin = Cin; a.H = H; a.W = W;
  dim3 grid((unsigned)((W + 31) / 32), (unsigned)((H + 7) / 8), (unsigned)N);
  hipLaunchKernelGGL(k_first_conv, grid, dim3(256), 0, as_stream(stream), a);
  KERNEL_CHECK();
  return ALIBY_OK;
}
