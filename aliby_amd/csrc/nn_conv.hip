// nn_conv.hip — the U-Net's 3x3 convolution unit as ONE hand-written MFMA kernel (bf16, NHWC).
//
// Every convolution unit of the Cellpose residual U-Net (reference call site: cellpose's CPnet as driven by
// src/aliby/segment/dispatch.py:55-63) is   BatchNorm -> ReLU -> Conv3x3 (+ bias) (+ residual / skip add).
// At the two high-resolution levels (32 / 64 channels) that work is HBM-bound (144-288 FLOP/B against a
// machine balance of ~500), so the pointwise stages must not cost their own passes over HBM.  This kernel
// fuses them around an implicit GEMM on the matrix cores:
//
//   prologue  a = bf16( relu( scale[c] * IN[n, y>>UP, x>>UP, c] + shift[n, c] ) ), zero outside the image
//             staged ONCE per workgroup tile into LDS as 16-byte channel-octet planes [octet][row][col]
//   GEMM      D[cout, pixel] += Wt[cout, (tap, c)] * a[(tap, c), pixel]   v_mfma_f32_32x32x16_bf16,
//             the packed weights of the wave's 32 output channels live in VGPRs for the whole launch
//             (persistent workgroups), the pixel operand is one conflict-free ds_read_b128 per MFMA
//   epilogue  OUT = bf16( D + bias[cout] + RES[n, y>>RU, x>>RU, cout] )  written as 32-byte runs per lane
//
// Operand roles: the weights are the MFMA "A" operand (row m = output channel) and the pixels the "B"
// operand (column n = pixel), so that a lane ends up with 16 output channels of ONE pixel; the row order of
// the weights is permuted at pack time so that those 16 channels are contiguous in memory
// (MFMA row (reg&3) + 8*(reg>>2) + 4*(lane>>5)  <->  channel 16*(lane>>5) + reg).
#include "common.h"
#include <utility>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

namespace {

// compile-time loop: every index is a constant expression, so register arrays are never indexed dynamically
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ float cv_bf2f(unsigned h) { return __uint_as_float(h << 16); }
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned cv_pack2(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  const bf16x2_t b = __builtin_convertvector(f, bf16x2_t);
  return __builtin_bit_cast(unsigned, b);
}
__device__ __forceinline__ unsigned cv_f2bf(float f) { return cv_pack2(f, 0.f) & 0xffffu; }

typedef short s16x2_t __attribute__((ext_vector_type(2)));
// The prologue on one channel octet: bf16 -> relu(scale * x + shift) -> bf16, two channels per instruction
// (v_pk_fma_f32, v_cvt_pk_bf16_f32; ReLU as v_pk_max_i16 on the bf16 bit patterns: a set sign bit is a negative
// int16), and zeroed outside the image (`keep` = 0 there: the convolution pads the ACTIVATED tensor).
__device__ __forceinline__ uint4 conv_act8(uint4 v, const f32x2_t (&sc)[4], const f32x2_t (&sh)[4], unsigned keep) {
  const unsigned w4[4] = {v.x, v.y, v.z, v.w};
  unsigned r4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2_t x = {__uint_as_float(w4[q] << 16), __uint_as_float(w4[q] & 0xffff0000u)};
    const f32x2_t y = __builtin_elementwise_fma(sc[q], x, sh[q]);
    s16x2_t b = __builtin_bit_cast(s16x2_t, __builtin_convertvector(y, bf16x2_t));
    const s16x2_t zero = {0, 0};
    b = __builtin_elementwise_max(b, zero);
    r4[q] = __builtin_bit_cast(unsigned, b) & keep;
  }
  return make_uint4(r4[0], r4[1], r4[2], r4[3]);
}

struct ConvArgs {
  const uint4* in;     // [N, H>>UP, W>>UP, CIN] bf16
  const uint4* wpk;    // packed weights, see k_pack_conv3x3
  uint4* out;          // [N, H, W, COUT] bf16
  uint4* pool;         // [N, H/2, W/2, COUT] bf16: 2x2 max pool of OUT, or NULL
  const float* scale;  // [CIN]
  const float* shift;  // [N, CIN] (shift_stride = CIN) or [CIN] (shift_stride = 0)
  const float* bias;   // [COUT] or NULL
  const uint4* res;    // [N, H>>res_up, W>>res_up, COUT] bf16 or NULL
  int shift_stride, res_up;
  int cs, coff;        // input pixel stride and first staged octet, in 16-byte units (a channel slice of a wider tensor)
  int ocs, ocoff;      // the same for OUT / RES / POOL (an output-channel slice of a wider tensor)
  int N, H, W;
  int G;               // packed launches: images laid side by side per tile row (G * W = 224); 1 otherwise
  int tiles_x, tiles_y, ntiles;
  const uint4* pin;    // fused 1x1 projection: raw input [N, H, W, pcs*8] bf16 (NULL: none)
  const uint4* pwpk;   // its packed weights ([cout block][k-step][lane]); the projection's bias rides in `bias`
  int pcs;             // 16-byte units per projection input pixel
  // fused output head (the network's last unit): y[n, o, p] = hbias[o] + sum_c hw[o, c] * bf16(relu(hscale[c] * OUT[n, p, c] + hshift[c]))
  const float* hscale;
  const float* hshift;
  const float* hw;     // [hO, COUT]
  const float* hbias;  // [hO]
  float* hout;         // [N, hO, H, W] float32, or NULL: no head
  int hO;
  // second unit of a fused pair (k_conv_pair32): OUT = conv_B(act_B(conv_A(act_A(IN)) + bias_A)) + bias_B + RES
  const uint4* wpk2;
  const float* scale2;
  const float* shift2;
  const float* bias2;
  int shift2_stride;
  // first layer as the producer of a pair (k_conv_first_pair): float32 NCHW tiles, <= 2 channels
  const float* fx;      // [N, fcin, H, W]
  const float* fw;      // [32][fcin][9] float32 (bf16-representable values)
  const float* fscale;  // [fcin]
  const float* fshift;  // [fcin]
  int fcin;
  unsigned long long* trace;  // diagnostics: per-phase shader-clock stamps of workgroup 0 (NULL in production)
};

// phase stamps for the latency analysis in DESIGN.md (wave 0 of workgroup 0 only; compiled in, never taken when
// trace == NULL)
#define CONV_STAMP(slot_)                                                                   \
  do {                                                                                      \
    if (a.trace && blockIdx.x == 0 && tid == 0 && stamp_tile < 16)                          \
      a.trace[stamp_tile * 8 + (slot_)] = __builtin_amdgcn_s_memtime();                     \
  } while (0)


// ------------------------------------------------------------------------------------------------
// Pieces shared by the two kernels below (register-staged and LDS-DMA window staging).
// ------------------------------------------------------------------------------------------------

// residual rows of one pass: R rows x 32 bytes per lane, from clamped addresses (rows / columns past the image edge
// are loaded but never stored)
template <int R>
__device__ __forceinline__ void conv_load_res(const ConvArgs& a, int n, int irow, int row0, int gx, int c0, uint4 (&rr)[R][2]) {
  const int RH = a.H >> a.res_up, RW = a.W >> a.res_up;
  const uint4* resN = a.res + (size_t)n * RH * RW * a.ocs + a.ocoff;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int cy = min(row0 + r, a.H - 1), cx = min(gx, a.W - 1);
    const unsigned off = (unsigned)((((irow + cy) >> a.res_up) * RW + (cx >> a.res_up)) * a.ocs + (c0 >> 3));
    rr[r][0] = resN[off];
    rr[r][1] = resN[off + 1];
  }
}

// accumulators start at bias + residual, so the epilogue is convert + store
template <int R>
__device__ __forceinline__ void conv_seed(f32x16_t (&acc)[R], const float4 (&b4)[4], const uint4 (&rr)[R][2], bool has_res) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[r][4 * q] = b4[q].x; acc[r][4 * q + 1] = b4[q].y; acc[r][4 * q + 2] = b4[q].z; acc[r][4 * q + 3] = b4[q].w;
    }
    if (has_res) {
      const unsigned rw[8] = {rr[r][0].x, rr[r][0].y, rr[r][0].z, rr[r][0].w, rr[r][1].x, rr[r][1].y, rr[r][1].z, rr[r][1].w};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        acc[r][2 * q] += cv_bf2f(rw[q] & 0xffffu);
        acc[r][2 * q + 1] += cv_bf2f(rw[q] >> 16);
      }
    }
  }
}

// The implicit GEMM of one pass.  A pixel fragment (input row ir, column offset dx, k-step kc) serves the up to three
// output rows r = ir - dy it is a tap of, so it is read from LDS once: (R+2)*3*KC ds_read_b128 for 9*KC*R MFMAs.  The
// reads run DEPTH fragments ahead of their MFMAs through a register ring; the sched_barrier keeps the compiler from
// hoisting them further (it would spill the resident weights: register budget in ConvCfg / DmaCfg).
template <int R, int KC, int DEPTH, int PLANE, int LW>
__device__ __forceinline__ void conv_mfma(const bf16x8_t* L, const bf16x8_t (&wfrag)[9 * KC], f32x16_t (&acc)[R]) {
  constexpr int NF = (R + 2) * 3 * KC;
  bf16x8_t ring[DEPTH];
  auto frag = [&](int f) { return L[2 * (f % KC) * PLANE + (f / (3 * KC)) * LW + (f / KC) % 3]; };
#pragma unroll
  for (int f = 0; f < DEPTH - 1; ++f) ring[f] = frag(f);
  static_for<NF>([&](auto fc) {
    constexpr int f = decltype(fc)::value;
    if constexpr (f + DEPTH - 1 < NF) ring[(f + DEPTH - 1) % DEPTH] = frag(f + DEPTH - 1);
    constexpr int ir = f / (3 * KC), dx = (f / KC) % 3, kc = f % KC;
    static_for<3>([&](auto dc) {
      constexpr int dy = decltype(dc)::value, r = ir - dy;
      if constexpr (r >= 0 && r < R)
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[(dy * 3 + dx) * KC + kc], ring[f % DEPTH], acc[r], 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// epilogue: fp32 -> bf16, 32 contiguous bytes per lane and row
template <int R>
__device__ __forceinline__ void conv_store(const ConvArgs& a, int n, int irow, int row0, int gx, int c0, const f32x16_t (&acc)[R]) {
  if (gx >= a.W) return;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int gy = row0 + r;
    if (gy >= a.H) continue;
    uint4* op = a.out + (size_t)n * a.H * a.W * a.ocs + a.ocoff + (unsigned)(((irow + gy) * a.W + gx) * a.ocs + (c0 >> 3));
    op[0] = make_uint4(cv_pack2(acc[r][0], acc[r][1]), cv_pack2(acc[r][2], acc[r][3]),
                       cv_pack2(acc[r][4], acc[r][5]), cv_pack2(acc[r][6], acc[r][7]));
    op[1] = make_uint4(cv_pack2(acc[r][8], acc[r][9]), cv_pack2(acc[r][10], acc[r][11]),
                       cv_pack2(acc[r][12], acc[r][13]), cv_pack2(acc[r][14], acc[r][15]));
  }
}

// ------------------------------------------------------------------------------------------------
// The network's output head on the matrix cores (round 3).  y[o] = hbias[o] + sum_c hw[o, c] * act(OUT[c]), act = bf16(relu(
// hscale[c] * bf16(OUT[c]) + hshift[c])), 32 channels -> O <= 4 maps.  Rounds 1-2 summed the 96 products per pixel on the vector
// unit, which cost the last unit half as much again as its whole 3x3 convolution and made the fused pair lose to two launches;
// as two k-steps of the 32x32x16 MFMA (A = the head's weights in rows 0..O-1, B = the activated pixel) it is 2 of 20 MFMAs.
// A lane holds channels 16 hh .. 16 hh + 15 of its pixel; the B fragment of k-step kc wants channels 16 kc + 8 hh .. + 7, so
// the two lanes of a pixel swap one octet each (one 16-byte exchange through lane ^ 32).  All three forms of the head —
// k_out_head, the last unit's epilogue, the fused pair's consumers — go through this function: same bits.
struct HeadW { bf16x8_t w[2]; };

__device__ __forceinline__ HeadW head_weights(const float* hw, int hO, int lane) {  // hw [hO][32] float32 (rounded to bf16 here)
  HeadW h;
  const int m = lane & 31, hh = lane >> 5, o = 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3);
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    unsigned r4[4];
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
      const int c = 16 * kc + 8 * hh + 2 * j2;
      r4[j2] = cv_pack2(o < hO ? hw[o * 32 + c] : 0.f, o < hO ? hw[o * 32 + c + 1] : 0.f);
    }
    h.w[kc] = __builtin_bit_cast(bf16x8_t, make_uint4(r4[0], r4[1], r4[2], r4[3]));
  }
  return h;
}

// lo / hi: this lane's 16 channels of the unit's output as bf16 (two octets); sc / sh [0..31]: the head's BatchNorm affine.
// Returns the MFMA result: outputs 0..3 are registers 0..3 of the hh == 0 lanes.
__device__ __forceinline__ f32x16_t head_apply(uint4 lo, uint4 hi, const float* sc, const float* sh, const HeadW& hw, int hh) {
  f32x2_t s2[4], h2[4];
  const float* sp = sc + 16 * hh;
  const float* hp = sh + 16 * hh;
#pragma unroll
  for (int k = 0; k < 4; ++k) { s2[k] = f32x2_t{sp[2 * k], sp[2 * k + 1]}; h2[k] = f32x2_t{hp[2 * k], hp[2 * k + 1]}; }
  const uint4 alo = conv_act8(lo, s2, h2, 0xffffffffu);
#pragma unroll
  for (int k = 0; k < 4; ++k) { s2[k] = f32x2_t{sp[8 + 2 * k], sp[8 + 2 * k + 1]}; h2[k] = f32x2_t{hp[8 + 2 * k], hp[8 + 2 * k + 1]}; }
  const uint4 ahi = conv_act8(hi, s2, h2, 0xffffffffu);
  const uint4 send = hh ? alo : ahi;
  const uint4 recv = make_uint4(__shfl_xor(send.x, 32), __shfl_xor(send.y, 32), __shfl_xor(send.z, 32), __shfl_xor(send.w, 32));
  const uint4 b0 = hh ? recv : alo, b1 = hh ? ahi : recv;
  f32x16_t y;
#pragma unroll
  for (int k = 0; k < 16; ++k) y[k] = 0.f;
  y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hw.w[0], __builtin_bit_cast(bf16x8_t, b0), y, 0, 0, 0);
  y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hw.w[1], __builtin_bit_cast(bf16x8_t, b1), y, 0, 0, 0);
  return y;
}

__device__ __forceinline__ f32x16_t head_from_acc(const f32x16_t& acc, const float* sc, const float* sh, const HeadW& hw, int hh) {
  const uint4 lo = make_uint4(cv_pack2(acc[0], acc[1]), cv_pack2(acc[2], acc[3]), cv_pack2(acc[4], acc[5]), cv_pack2(acc[6], acc[7]));
  const uint4 hi = make_uint4(cv_pack2(acc[8], acc[9]), cv_pack2(acc[10], acc[11]), cv_pack2(acc[12], acc[13]), cv_pack2(acc[14], acc[15]));
  return head_apply(lo, hi, sc, sh, hw, hh);
}

template <int CIN, int COUT, bool TALL = false, bool PACK = false, bool HEAD = false>
struct ConvCfg {
  static constexpr int KC = CIN / 16;    // MFMA k-steps per tap
  static constexpr int NPL = CIN / 8;    // 16-byte channel-octet planes in LDS
  static constexpr int NCB = COUT / 32;  // blocks of 32 output channels
  // register budget per wave (256 VGPRs at 2 waves/SIMD): 9*KC*4 for the weights + 16*R accumulators
  // + the next tile's staging loads + the residual prefetch
  // (HEAD: two rows per pass — the head's constants and sums need the registers of two rows of accumulators)
  static constexpr int R = (CIN >= 64 || HEAD) ? 2 : 4;            // output rows per wave and pass
  // row passes per tile (accumulators reused); TALL doubles them: a taller tile has less halo per output row, taken
  // when the image height is a multiple of the taller tile
  // (packed launches: 14-row tiles — 28, 56 and 112 are multiples of 14 — so a 28-row level is 504 tiles for the chip's 512
  // workgroup slots and the halo is 2 rows in 16)
  static constexpr int PASSES = (PACK && COUT >= 128) ? 7 : ((CIN >= 64 && COUT >= 64) ? 2 : 1) * (TALL ? 2 : 1) * (HEAD ? 2 : 1);
  static constexpr int RG = 4 / NCB;                               // row groups per workgroup (4 waves)
  // PACK: one more window column, the zero column that separates two images meeting inside the 32-pixel block
  static constexpr int TH = RG * R * PASSES, TW = 32, LH = TH + 2, LW = TW + 2 + (PACK ? 1 : 0);
  static constexpr int RAW = LH * LW;
  // plane pitch in 16-byte slots, chosen so that the 8-lane groups of ds_write_b128 (lanes = NPL octets x
  // 8/NPL pixels) land on 8 distinct slots of the 128-byte bank window
  static constexpr int PLANE = NPL == 4 ? RAW + (10 - RAW % 8) % 8 : (RAW | 1);
  static constexpr int LDS_BYTES = NPL * PLANE * 16;
  static constexpr int PIX_PER_IT = 256 / NPL;
  static constexpr int ITERS = (RAW + PIX_PER_IT - 1) / PIX_PER_IT;  // staging loads per thread and tile
#ifndef C3_B32T
#define C3_B32T 1  // diagnostics: batches of the 32-channel TALL window
#endif
#ifndef C3_B64T
#define C3_B64T 4  // batches of the 64 -> 64 TALL window (20 loads per thread; measured 2 / 3 / 4 batches: 394 / 320 / 294 us per 288-tile
                   // launch — ten or seven loads at once spill the resident weights)
#endif
  static constexpr int NBATCH = (PACK && COUT >= 128) ? 3 : (CIN >= 64 ? ((TALL && COUT == 64) ? C3_B64T : 2) : (TALL ? C3_B32T : 1));
  static constexpr int BATCH = (ITERS + NBATCH - 1) / NBATCH;  // staging loads in flight per thread
  static constexpr int DEPTH = (CIN >= 64 && COUT >= 64) ? 4 : 6;  // pixel fragments in flight LDS -> VGPR ahead of their MFMAs
  static constexpr int DEPTH_POOL = (CIN >= 64 && COUT >= 64) ? 3 : 6;  // the pooling epilogue needs a few registers more
};

// PK > 0: the residual block's 1x1 projection of the block input (cellpose `resdown.proj`, BatchNorm folded into its
// weights) is PK extra k-steps of the same accumulation: its raw input tile (no halo, no activation) is staged beside
// the window and the projected tensor never exists in HBM.
//
// PACK: the deep levels' images are 28 / 56 / 112 pixels wide, which leaves a 32-pixel MFMA block 12.5 % empty.  A packed
// launch lays G = 224 / W images side by side (a view: image g of a group starts G rows of pointers further, nothing is
// copied) and cuts that 224-pixel row into seven full blocks.  A block then holds the end of one image and the start of
// the next: the staging puts one zero column between them in LDS (the right halo of the first and the left halo of the
// second), lanes past the seam read one slot further, and every lane keeps its own (image, column) for the residual,
// the store and the pooled output.
//
// HEAD: the network's output head rides in the last unit's epilogue — BatchNorm + ReLU + 1x1 convolution to <= 3 channels +
// NHWC -> NCHW float32, from the accumulators: a lane holds 16 of a pixel's 32 channels, its partner (lane ^ 32) the other
// 16.  Products are summed octet by octet, then octet pairs, then the two halves: the order of k_out_head, so both paths give
// the same bits.  OUT may be NULL then (nothing else reads the last unit's output: 0.9 GB per forward not written, not re-read).
#ifndef C3_HACK
#define C3_HACK 0  // diagnostics (scripts/phase_builds.sh): compile-time phase switches of this kernel, timing only
#endif
template <int CIN, int COUT, bool UP, bool POOL, int PK, bool TALL, bool PACK = false, bool HEAD = false>
__global__ __launch_bounds__(256, 2) void k_conv3x3(ConvArgs a) {
  static_assert(!PACK || (!UP && PK == 0), "packed launches: plain input, no fused projection");
  static_assert(!HEAD || (COUT == 32 && !POOL && PK == 0 && !PACK), "the fused head is built for the 32-channel last unit");
  using cfg = ConvCfg<CIN, COUT, TALL, PACK, HEAD>;
  constexpr int PPLANE = (cfg::TH * cfg::TW) | 1;  // slot pitch of a projection-input octet plane
  constexpr int KC = cfg::KC, NPL = cfg::NPL, NCB = cfg::NCB, R = cfg::R, TH = cfg::TH, TW = cfg::TW;
  constexpr int LW = cfg::LW, RAW = cfg::RAW, PLANE = cfg::PLANE, PASSES = cfg::PASSES;
  constexpr int PIX_PER_IT = cfg::PIX_PER_IT, ITERS = cfg::ITERS, BATCH = cfg::BATCH;
  extern __shared__ uint4 lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = wave % NCB, rg = wave / NCB;
  const int px = lane & 31, hh = lane >> 5;

  // ---- weights of this wave's 32 output channels: 9*KC fragments, resident for the whole launch
  bf16x8_t wfrag[9 * KC];
  {
    const bf16x8_t* wp = reinterpret_cast<const bf16x8_t*>(a.wpk) + (size_t)cb * 9 * KC * 64 + lane;
#pragma unroll
    for (int i = 0; i < 9 * KC; ++i) {
      if constexpr (C3_HACK & 256) { wfrag[i] = bf16x8_t{}; wfrag[i][0] = (__bf16)(float)(i + lane); } else wfrag[i] = wp[i * 64];
    }
  }
  bf16x8_t pfrag[PK > 0 ? PK : 1];
  if constexpr (PK > 0) {
    const bf16x8_t* pp = reinterpret_cast<const bf16x8_t*>(a.pwpk) + (size_t)cb * PK * 64 + lane;
#pragma unroll
    for (int i = 0; i < PK; ++i) pfrag[i] = pp[i * 64];
  }
  uint4* const ldsP = lds + NPL * PLANE;
  // bias: once per launch into LDS (a global load per pass sat in front of every pass's first MFMA: an exposed L2 round trip,
  // 3-5 % of the launch by phase elimination); visible to every wave after the first tile's barriers
  __shared__ float4 sbias[COUT / 4];
  if (tid < COUT / 4) sbias[tid] = a.bias ? reinterpret_cast<const float4*>(a.bias)[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
  __shared__ float hconst[HEAD ? 2 * 32 + 4 : 1];  // head: scale, shift, bias (the weights are MFMA fragments in registers)
  HeadW hwf;
  if constexpr (HEAD) {
    hwf = head_weights(a.hw, a.hO, lane);
    if (tid < 32) {
      hconst[tid] = a.hscale[tid];
      hconst[32 + tid] = a.hshift[tid];
      if (tid < 4) hconst[64 + tid] = tid < a.hO ? a.hbias[tid] : 0.f;
    }
    // (visible to every wave after the first tile's barriers)
  }
  // ---- staging role of this thread: a fixed channel octet
  const int pl = tid % NPL, pix0 = tid / NPL;
  const int ly0 = pix0 / LW, lx0 = pix0 - ly0 * LW;  // window coordinates of this thread's first pixel
  f32x2_t sc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sc[k] = f32x2_t{a.scale[pl * 8 + 2 * k], a.scale[pl * 8 + 2 * k + 1]};

  const int IH = UP ? a.H >> 1 : a.H, IW = UP ? a.W >> 1 : a.W;
  // XCD-aware persistent schedule: workgroup b lives on XCD b%8; each XCD walks one contiguous eighth of
  // the tile list so that neighbouring tiles (shared halos) meet in the same L2.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int per_xcd = (a.ntiles + 7) >> 3;
  const int t_end = min(a.ntiles, (xcd + 1) * per_xcd);

  int stamp_tile = 0;
  for (int tile = xcd * per_xcd + slot; tile < t_end; tile += nslots, ++stamp_tile) {
    CONV_STAMP(0);
    const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
    const int ty = tyn % a.tiles_y, n = tyn / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    // packed: the block starts at column cst of image g0 of group n; the seam (if any) is before block pixel kb
    int g0 = 0, cst = x0, kb = 1 << 20, nimg = n;
    if constexpr (PACK) {
      g0 = x0 / a.W;
      cst = x0 - g0 * a.W;
      kb = a.W - cst;
      nimg = n * a.G;
    }

    f32x2_t sh[4], sh1[PACK ? 4 : 1];
    {
      const float* sp = a.shift + (size_t)(nimg + g0) * a.shift_stride + pl * 8;
#pragma unroll
      for (int k = 0; k < 4; ++k) sh[k] = f32x2_t{sp[2 * k], sp[2 * k + 1]};
      if constexpr (PACK) {
        const float* sq = a.shift + (size_t)(nimg + min(g0 + 1, a.G - 1)) * a.shift_stride + pl * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) sh1[k] = f32x2_t{sq[2 * k], sq[2 * k + 1]};
      }
    }
    // ---- residual rows and bias seed the accumulators: requested now, unpacked after the prologue
    const int second = (PACK && px >= kb) ? 1 : 0;  // this lane's pixel belongs to image g0 + 1
    const int gx = PACK ? (second ? px - kb : cst + px) : x0 + px;
    const int irow = PACK ? (g0 + second) * a.H : 0;
    const int c0 = cb * 32 + hh * 16;
    // (pass 0 before the prologue, pass p+1 while pass p is on the matrix cores: one pass worth of registers)
    uint4 rr[R][2];
    auto load_res = [&](int pass) { conv_load_res<R>(a, nimg, irow, y0 + (rg * PASSES + pass) * R, gx, c0, rr); };
    if (a.res && !(C3_HACK & 2)) load_res(0);
    CONV_STAMP(1);
    if constexpr (!(C3_HACK & 128)) __syncthreads();  // every wave is done reading the previous tile's planes
    CONV_STAMP(2);
    // ---- stage the raw window: BATCH 16-byte loads per thread in flight at once (unconditional, from clamped
    // addresses: no divergent branch around a load), then the prologue (BatchNorm affine + style shift + ReLU,
    // bf16; the convolution's zero padding is applied AFTER the activation) into the LDS planes
    const uint4* inN = a.in + (size_t)nimg * IH * IW * a.cs + a.coff;  // uniform base (SGPR pair) + 32-bit lane offsets
    int p0 = pix0, qy = ly0, qx = lx0;
    asm volatile("" : "+v"(p0), "+v"(qy), "+v"(qx));  // recompute the window coordinates per tile: 3 registers, not 2*ITERS
#pragma unroll
    for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
      uint4 v[BATCH];
      unsigned inside = 0, later = 0;
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        if (it0 + u >= ITERS) break;
        // pixel p0 + it*PIX_PER_IT of the window raster, without a division: constant row / column advance + one wrap
        int lx = qx + ((it0 + u) * PIX_PER_IT) % LW, ly = qy + ((it0 + u) * PIX_PER_IT) / LW;
        if (lx >= LW) { lx -= LW; ly += 1; }
        const int gy = y0 - 1 + ly;
        const int cy = min(max(gy, 0), a.H - 1);
        if constexpr (PACK) {
          // window slot -> block pixel -1..32 (slot `gapi` is the seam's zero column, or the unused last slot) -> (image, column)
          const int gapi = kb < 32 ? kb + 1 : TW + 2;
          const int bp = lx - 1 - (lx > gapi ? 1 : 0);
          const int snd = bp >= kb ? 1 : 0;
          const int c = snd ? bp - kb : cst + bp;
          // (a halo pixel must lie in the image of the block pixel it borders: kb == 32 puts the right halo in the next image)
          const bool ok = lx != gapi && c >= 0 && c < a.W && !(bp == TW && kb == TW) && (unsigned)gy < (unsigned)a.H;
          const int img = (snd && ok) ? g0 + 1 : g0;
          inside |= (unsigned)ok << u;
          later |= (unsigned)snd << u;
          if constexpr (C3_HACK & 1) v[u] = make_uint4(cy, c, img, 3);
          else v[u] = inN[(unsigned)(((img * a.H + cy) * IW + min(max(c, 0), a.W - 1)) * a.cs + pl)];
        } else {
          const int gxi = x0 - 1 + lx;
          inside |= (unsigned)((unsigned)gy < (unsigned)a.H && (unsigned)gxi < (unsigned)a.W) << u;
          const int cx = min(max(gxi, 0), a.W - 1);
          if constexpr (C3_HACK & 1) v[u] = make_uint4(cy, cx, 2, 3);
          else v[u] = inN[(unsigned)(((UP ? cy >> 1 : cy) * IW + (UP ? cx >> 1 : cx)) * a.cs + pl)];
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        if (it0 + u >= ITERS) break;
        uint4 o;
        if constexpr (PACK) {  // the style shift is per image
          f32x2_t shs[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) shs[k] = ((later >> u) & 1u) ? sh1[k] : sh[k];
          o = conv_act8(v[u], sc, shs, 0u - ((inside >> u) & 1u));
        } else {
          if constexpr (C3_HACK & 16) o = v[u]; else o = conv_act8(v[u], sc, sh, 0u - ((inside >> u) & 1u));
        }
        if (it0 + u == ITERS - 1 && p0 + (it0 + u) * PIX_PER_IT >= RAW) continue;  // only the last round can run past the window
        if constexpr (C3_HACK & 32) { if (o.x == 0x12345u) lds[pl * PLANE + p0 + (it0 + u) * PIX_PER_IT] = o; }
        else lds[pl * PLANE + p0 + (it0 + u) * PIX_PER_IT] = o;
      }
    }
    if constexpr (PK > 0) {  // the projection's raw input tile: TH x TW pixels, 2*PK channel octets, no halo
      constexpr int POCT = 2 * PK, PUNITS = TH * TW * POCT;
      const uint4* pN = a.pin + (size_t)n * a.H * a.W * a.pcs;
#pragma unroll
      for (int it = 0; it < (PUNITS + 255) / 256; ++it) {
        const int u = tid + it * 256;
        if (u >= PUNITS) break;
        const int oct = u % POCT, pix = u / POCT;
        const int cy = min(y0 + pix / TW, a.H - 1), cx = min(x0 + pix % TW, a.W - 1);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (oct < a.pcs) v = pN[(unsigned)((cy * a.W + cx) * a.pcs + oct)];  // channels past the tensor's are zero
        ldsP[oct * PPLANE + pix] = v;
      }
    }
    CONV_STAMP(3);
    if constexpr (!(C3_HACK & 128)) __syncthreads();
    CONV_STAMP(4);

#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      const int rbase = (rg * PASSES + pass) * R;  // first output row of this wave and pass inside the tile
      f32x16_t acc[R];
      {
        float4 b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b4[q] = (C3_HACK & 64) ? make_float4(0.f, 0.f, 0.f, 0.f) : sbias[(c0 >> 2) + q];
        conv_seed<R>(acc, b4, rr, a.res != nullptr && !(C3_HACK & 2));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (pass + 1 < PASSES && a.res && !(C3_HACK & 2)) load_res(pass + 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(C3_HACK & 8))
      conv_mfma<R, KC, (POOL ? cfg::DEPTH_POOL : cfg::DEPTH), PLANE, LW>(
          reinterpret_cast<const bf16x8_t*>(lds) + hh * PLANE + rbase * LW + px + second, wfrag, acc);
      if constexpr (PK > 0) {
        const bf16x8_t* LP = reinterpret_cast<const bf16x8_t*>(ldsP) + hh * PPLANE + rbase * TW + px;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int k = 0; k < PK; ++k)
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pfrag[k], LP[2 * k * PPLANE + r * TW], acc[r], 0, 0, 0);
      }
      if (pass == PASSES - 1) CONV_STAMP(5);
      if constexpr (HEAD) {
        __builtin_amdgcn_sched_barrier(0);  // (nothing of the head is hoisted into the MFMA loop: it would spill the weights)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const f32x16_t y = head_from_acc(acc[r], hconst, hconst + 32, hwf, hh);
          const int gy = y0 + rbase + r;
          if (hh == 0 && gx < a.W && gy < a.H) {
            float* hp = a.hout + (size_t)n * a.hO * a.H * a.W + (size_t)gy * a.W + gx;
            const float yo[4] = {y[0], y[1], y[2], y[3]};
            for (int o = 0; o < a.hO; ++o) hp[(size_t)o * a.H * a.W] = yo[o] + hconst[64 + o];
          }
        }
      }
      if ((!HEAD || a.out) && (!(C3_HACK & 4) || acc[0][3] == 12345.f)) conv_store<R>(a, nimg, irow, y0 + rbase, gx, c0, acc);
      // ---- the next level's input, max_pool2d(OUT, 2, 2), straight from the accumulators: row pairs are in this
      // wave's registers, column pairs are neighbouring lanes (max commutes with the bf16 rounding)
      if constexpr (POOL) {
        const int PH = a.H >> 1, PW = a.W >> 1;
#pragma unroll
        for (int r = 0; r < R; r += 2) {
          const int gy = y0 + rbase + r;
          const bool writer = (px & 1) == 0 && gx + 1 < a.W && gy + 1 < a.H;
          uint4* pp = a.pool + (size_t)nimg * PH * PW * a.ocs + a.ocoff + (unsigned)((((irow + gy) >> 1) * PW + (gx >> 1)) * a.ocs + (c0 >> 3));
#pragma unroll
          for (int half = 0; half < 2; ++half) {  // 8 channels at a time: few live registers beside the resident weights
            unsigned pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float v0 = fmaxf(acc[r][half * 8 + 2 * q], acc[r + 1][half * 8 + 2 * q]);
              const float v1 = fmaxf(acc[r][half * 8 + 2 * q + 1], acc[r + 1][half * 8 + 2 * q + 1]);
              pk[q] = cv_pack2(fmaxf(v0, __shfl_xor(v0, 1)), fmaxf(v1, __shfl_xor(v1, 1)));
            }
            if (writer) pp[half] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
          }
        }
      }
    }
    CONV_STAMP(6);
  }
}

// ------------------------------------------------------------------------------------------------
// Variant with the raw window travelling global -> LDS by LDS-DMA (global_load_lds_dwordx4), one tile
// AHEAD of the MFMA loop that consumes it: no VGPRs hold the prefetch, so it fits beside the resident
// weights.  Two LDS images per workgroup: R (raw bf16 window, pixel-major = lane-linear, as the DMA
// writes it) and A (the activated channel-octet planes the MFMA loop reads).  Per tile:
//     wait DMA(t), barrier | prologue R -> A | barrier | seed accumulators | issue DMA(t+1) | MFMA | store
template <int CIN, int COUT, bool UP, bool PACK = false>
struct DmaCfg {
  static constexpr int KC = CIN / 16, NPL = CIN / 8, NCB = COUT / 32;
  // (a CIN = 128 instantiation — 288 weight VGPRs, one wave per SIMD, two row passes — was measured at the same
  // ~760 TFLOP/s as two 64-channel K-slices and dropped; WAVES_PER_SIMD / PASSES keep the knobs it needed)
  static constexpr int WAVES_PER_SIMD = CIN >= 128 ? 1 : 2;
  static constexpr int R = CIN >= 64 ? 2 : 4;      // output rows per wave and pass
  static constexpr int PASSES = (CIN >= 128 || COUT >= 128) ? 2 : 1;  // row passes per tile (accumulators reused): a taller tile, less halo
  static constexpr int RG = 4 / NCB;
  static constexpr int TH = RG * R * PASSES, TW = 32, LH = TH + 2, LW = TW + 2 + (PACK ? 1 : 0);  // PACK: + the seam's zero column
  static constexpr int RAW = LH * LW;
  static constexpr int PLANE = NPL == 4 ? RAW + (10 - RAW % 8) % 8 : (RAW | 1);
  static constexpr int A_SLOTS = NPL * PLANE;
  static constexpr int PIX_PER_IT = 256 / NPL;
  static constexpr int ITERS = (RAW + PIX_PER_IT - 1) / PIX_PER_IT;
  // raw window (at the input resolution)
  static constexpr int RLH = UP ? TH / 2 + 2 : LH, RLW = UP ? TW / 2 + 2 : LW;
  static constexpr int RUNITS = RLH * RLW * NPL;            // 16-byte units
  static constexpr int GIT = (RUNITS + 255) / 256;          // DMA instructions per wave and tile
  static constexpr int R_SLOTS = GIT * 256;
  static constexpr int LDS_BYTES = (A_SLOTS + R_SLOTS) * 16;
  static constexpr int DEPTH = (CIN >= 64 && COUT >= 64) ? 4 : 6;
};

// PACK (upsampled input only): the packed launch of k_conv3x3, with the raw half-resolution window switching images at the seam.
template <int CIN, int COUT, bool UP, bool PACK = false>
__global__ __launch_bounds__(256, (DmaCfg<CIN, COUT, UP, PACK>::WAVES_PER_SIMD)) void k_conv3x3_dma(ConvArgs a) {
  static_assert(!PACK || UP, "the packed LDS-DMA variant is written for upsampled input");
  using cfg = DmaCfg<CIN, COUT, UP, PACK>;
  constexpr int KC = cfg::KC, NPL = cfg::NPL, NCB = cfg::NCB, R = cfg::R, TH = cfg::TH, TW = cfg::TW;
  constexpr int LW = cfg::LW, RAW = cfg::RAW, PLANE = cfg::PLANE, RLW = cfg::RLW, PASSES = cfg::PASSES;
  constexpr int PIX_PER_IT = cfg::PIX_PER_IT, ITERS = cfg::ITERS, GIT = cfg::GIT;
  extern __shared__ uint4 lds[];
  uint4* const ldsA = lds;
  uint4* const ldsR = lds + cfg::A_SLOTS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = wave % NCB, rg = wave / NCB;
  const int px = lane & 31, hh = lane >> 5;
  const int c0 = cb * 32 + hh * 16;

  bf16x8_t wfrag[9 * KC];
  {
    const bf16x8_t* wp = reinterpret_cast<const bf16x8_t*>(a.wpk) + (size_t)cb * 9 * KC * 64 + lane;
#pragma unroll
    for (int i = 0; i < 9 * KC; ++i) wfrag[i] = wp[i * 64];
  }
  const int pl = tid % NPL, pix0 = tid / NPL;
  const int ly0 = pix0 / LW, lx0 = pix0 - ly0 * LW;
  f32x2_t sc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sc[k] = f32x2_t{a.scale[pl * 8 + 2 * k], a.scale[pl * 8 + 2 * k + 1]};

  const int IH = UP ? a.H >> 1 : a.H, IW = UP ? a.W >> 1 : a.W;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int per_xcd = (a.ntiles + 7) >> 3;
  const int t_end = min(a.ntiles, (xcd + 1) * per_xcd);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  auto issue_dma = [&](int tile) {
    const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
    const int ty = tyn % a.tiles_y, n = tyn / a.tiles_y;
    const int ry0 = UP ? (ty * TH >> 1) - 1 : ty * TH - 1, rx0 = UP ? (tx * TW >> 1) - 1 : tx * TW - 1;
    int g0 = 0, cst = 0, half = 1 << 20;
    if constexpr (PACK) {
      g0 = (tx * TW) / a.W;
      cst = tx * TW - g0 * a.W;
      half = (a.W - cst) >> 1;  // raw columns 0..half come from image g0 (left halo + its last columns), the rest from g0 + 1
    }
    const uint4* inN = a.in + (size_t)(PACK ? n * a.G : n) * IH * IW * a.cs + a.coff;
    int t0 = tid;
    asm volatile("" : "+v"(t0));
#pragma unroll
    for (int k = 0; k < GIT; ++k) {
      const int unit = t0 + k * 256;  // lane-linear: unit u lands at ldsR[u]
      const int rpix = unit / NPL, oct = unit % NPL;
      const int rly = rpix / RLW, rlx = rpix - rly * RLW;
      const int iy = min(max(ry0 + rly, 0), IH - 1);
      int ix, img = 0;
      if constexpr (PACK) {
        const bool first = rlx <= half;
        ix = min(max(first ? (cst >> 1) - 1 + rlx : rlx - half - 1, 0), IW - 1);
        img = first ? g0 : min(g0 + 1, a.G - 1);
      } else {
        ix = min(max(rx0 + rlx, 0), IW - 1);  // clamped: masked later
      }
      const uint4* g = inN + (unsigned)(((img * IH + iy) * IW + ix) * a.cs + oct);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(ldsR + k * 256 + wave_u * 64), 16, 0, 0);
    }
  };

  int tile = xcd * per_xcd + slot, stamp_tile = 0;
  if (tile < t_end) issue_dma(tile);
  for (; tile < t_end; tile += nslots, ++stamp_tile) {
    CONV_STAMP(0);
    const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
    const int ty = tyn % a.tiles_y, n = tyn / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    int g0 = 0, cst = x0, kb = 1 << 20, nimg = n;
    if constexpr (PACK) {
      g0 = x0 / a.W;
      cst = x0 - g0 * a.W;
      kb = a.W - cst;
      nimg = n * a.G;
    }
    const int second = (PACK && px >= kb) ? 1 : 0;
    const int gx = PACK ? (second ? px - kb : cst + px) : x0 + px;
    const int irow = PACK ? (g0 + second) * a.H : 0;

    // ---- everything this tile needs from global memory besides the window: requested before the DMA wait
    f32x2_t sh[4], sh1[PACK ? 4 : 1];
    {
      const float* sp = a.shift + (size_t)(nimg + g0) * a.shift_stride + pl * 8;
#pragma unroll
      for (int k = 0; k < 4; ++k) sh[k] = f32x2_t{sp[2 * k], sp[2 * k + 1]};
      if constexpr (PACK) {
        const float* sq = a.shift + (size_t)(nimg + min(g0 + 1, a.G - 1)) * a.shift_stride + pl * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) sh1[k] = f32x2_t{sq[2 * k], sq[2 * k + 1]};
      }
    }
    float4 b4[4] = {};
    if (a.bias) {
      const float4* bp = reinterpret_cast<const float4*>(a.bias + c0);
#pragma unroll
      for (int q = 0; q < 4; ++q) b4[q] = bp[q];
    }
    uint4 rr[R][2];
    auto load_res = [&](int pass) { conv_load_res<R>(a, nimg, irow, y0 + (rg * PASSES + pass) * R, gx, c0, rr); };
    if (a.res && !(C3_HACK & 2)) load_res(0);
    CONV_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the window has landed in R
    __syncthreads();                                   // ... and everyone's; the previous tile's reads of A are over
    CONV_STAMP(2);
    // ---- prologue R -> A
    {
      const int ry0 = UP ? (y0 >> 1) - 1 : y0 - 1, rx0 = UP ? (x0 >> 1) - 1 : x0 - 1;
      int p0 = pix0, qy = ly0, qx = lx0;
      asm volatile("" : "+v"(p0), "+v"(qy), "+v"(qx));
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        int lx = qx + (it * PIX_PER_IT) % LW, ly = qy + (it * PIX_PER_IT) / LW;
        if (lx >= LW) { lx -= LW; ly += 1; }
        const int gy = y0 - 1 + ly, gxi = x0 - 1 + lx;
        const int pix = p0 + it * PIX_PER_IT;
        unsigned keep;
        int rpix;
        uint4 o;
        if constexpr (PACK) {
          // window slot -> block pixel -1..32 (see k_conv3x3) -> (image, column) -> raw half-resolution column
          const int gapi = kb < 32 ? kb + 1 : TW + 2;
          const int bp = lx - 1 - (lx > gapi ? 1 : 0);
          const int snd = bp >= kb ? 1 : 0;
          const int c = snd ? bp - kb : cst + bp;
          keep = 0u - (unsigned)(lx != gapi && c >= 0 && c < a.W && !(bp == TW && kb == TW) && (unsigned)gy < (unsigned)a.H);
          const int rcol = snd ? (kb >> 1) + 1 + (c >> 1) : ((cst + bp) >> 1) - ((cst >> 1) - 1);
          rpix = min(max(((gy >> 1) - ry0) * RLW + rcol, 0), cfg::RLH * RLW - 1);
          f32x2_t shs[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) shs[k] = snd ? sh1[k] : sh[k];
          o = conv_act8(ldsR[rpix * NPL + pl], sc, shs, keep);
        } else {
          keep = 0u - (unsigned)((unsigned)gy < (unsigned)a.H && (unsigned)gxi < (unsigned)a.W);
          // (the last round can run past the window: its raw index is clamped, its value never stored)
          rpix = UP ? min(((gy >> 1) - ry0) * RLW + ((gxi >> 1) - rx0), cfg::RLH * RLW - 1) : min(pix, RAW - 1);
          o = conv_act8(ldsR[rpix * NPL + pl], sc, sh, keep);
        }
        if (it == ITERS - 1 && pix >= RAW) continue;
        ldsA[pl * PLANE + pix] = o;
      }
    }
    CONV_STAMP(3);
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      const int rbase = (rg * PASSES + pass) * R;
      f32x16_t acc[R];
      conv_seed<R>(acc, b4, rr, a.res != nullptr);
      __builtin_amdgcn_sched_barrier(0);
      if (pass == 0 && tile + nslots < t_end) issue_dma(tile + nslots);  // R is free: everyone passed the barrier after the prologue
      if (pass + 1 < PASSES && a.res) load_res(pass + 1);                   // consumed after this pass's MFMA loop
      __builtin_amdgcn_sched_barrier(0);
      if (pass == 0) CONV_STAMP(4);
      conv_mfma<R, KC, cfg::DEPTH, PLANE, LW>(reinterpret_cast<const bf16x8_t*>(ldsA) + hh * PLANE + rbase * LW + px + second, wfrag, acc);
      if (pass == PASSES - 1) CONV_STAMP(5);
      conv_store<R>(a, nimg, irow, y0 + rbase, gx, c0, acc);
    }
    CONV_STAMP(6);
  }
}

// ------------------------------------------------------------------------------------------------
// Two units of a 32-channel residual block in one launch (cellpose resdown / resup: x + conv3(conv2(x))): the tensor
// between them — 0.9 GB written and read back per 288-tile forward at 224 x 224 — stays in LDS.
//
// Wave-specialised: an 8-wave workgroup, one per CU.  Waves 0-3 (producers) hold unit A's weights: they stage the 18 x 34
// input window of a tile through A's prologue, compute the 16 x 32 intermediate block (one MFMA block wide: tiles are 30
// output pixels wide so that the one-pixel halo fills the block), round it to bf16 exactly where the two-launch path
// stores it, run it through unit B's prologue (BatchNorm affine + style shift + ReLU, zero outside the image) and write it
// into one of two sets of channel-octet planes.  Waves 4-7 (consumers) hold unit B's weights and turn the previous tile's
// planes into 14 x 30 output pixels (+ residual, + pooled output) meanwhile.  One weight set per wave: no spills, and the
// consumers' MFMAs overlap the producers' load latency.  Every wave passes the same two barriers per tile:
//     producers:  stage window(t)        | S1 | unit A -> planes[t & 1]   | S2 |
//     consumers:  unit B pass 0 of t - 1 | S1 | unit B pass 1 of t - 1    | S2 |
// Same bits as the two launches.
// ------------------------------------------------------------------------------------------------
struct PairCfg {
  static constexpr int KC = 2, NPL = 4, R = 2, PASSES = 2;
  static constexpr int TH = 14, TWO = 30, MH = 16, MW = 32, IH = 18, IW = 34;
  static constexpr int RAW_IN = IH * IW, PLANE_IN = RAW_IN + (10 - RAW_IN % 8) % 8;
  static constexpr int RAW_MID = MH * MW, PLANE_MID = RAW_MID + (10 - RAW_MID % 8) % 8;
  static constexpr int PIX_PER_IT = 256 / NPL, ITERS = (RAW_IN + PIX_PER_IT - 1) / PIX_PER_IT;
  static constexpr int MID_SLOTS = NPL * PLANE_MID + 8;
  static constexpr int SLOTS = NPL * PLANE_IN + 2 * MID_SLOTS;
  static constexpr int LDS_BYTES = SLOTS * 16;
  static constexpr int DEPTH = 6;
};

#ifndef CP_HACK
#define CP_HACK 0  // diagnostics (scripts/phase_builds.sh): compile-time phase switches of the pair kernel, timing only
#endif
template <bool POOL, bool HEAD>
__global__ __launch_bounds__(512, 1) void k_conv_pair32(ConvArgs a) {
  static_assert(!(POOL && HEAD), "either the pooled output (down block) or the output head (last unit)");
  using cfg = PairCfg;
  constexpr int KC = cfg::KC, NPL = cfg::NPL, R = cfg::R, PASSES = cfg::PASSES, TH = cfg::TH, TWO = cfg::TWO;
  constexpr int IW_ = cfg::IW, MW = cfg::MW, PLANE_IN = cfg::PLANE_IN, PLANE_MID = cfg::PLANE_MID, RAW_IN = cfg::RAW_IN;
  constexpr int PIX_PER_IT = cfg::PIX_PER_IT, ITERS = cfg::ITERS;
  extern __shared__ uint4 lds[];
  uint4* const ldsIn = lds;
  uint4* const ldsMid0 = lds + NPL * PLANE_IN;
  __shared__ __align__(16) float tabB[64];  // unit B's scale[32], shift[32] of the producers' current tile
  __shared__ float hconst[HEAD ? 2 * 32 + 4 : 1];  // output head: scale, shift, bias (see k_conv3x3)

  const int consumer = threadIdx.x >> 8;            // wave-uniform: waves 4-7
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;  // role-local thread / wave index
  const int px = lane & 31, hh = lane >> 5;
  const int c0 = hh * 16;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int per_xcd = (a.ntiles + 7) >> 3;
  const int t_begin = xcd * per_xcd + slot, t_end = min(a.ntiles, (xcd + 1) * per_xcd);
  const int my_tiles = t_begin < t_end ? (t_end - t_begin + nslots - 1) / nslots : 0;  // the same for both roles

  if (!consumer) {
    // =============================================================== producers: window -> unit A -> B's planes
    bf16x8_t wA[9 * KC];
    {
      const bf16x8_t* pa = reinterpret_cast<const bf16x8_t*>(a.wpk) + lane;
#pragma unroll
      for (int i = 0; i < 9 * KC; ++i) wA[i] = pa[i * 64];
    }
    const int pl = tid % NPL, pix0 = tid / NPL;
    const int ly0 = pix0 / IW_, lx0 = pix0 - ly0 * IW_;
    f32x2_t sc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sc[k] = f32x2_t{a.scale[pl * 8 + 2 * k], a.scale[pl * 8 + 2 * k + 1]};
    // the window of tile t + 1 is requested while unit A works on tile t: its loads are in flight behind the MFMA loop
    uint4 v[ITERS];
    unsigned inside = 0;
    auto request = [&](int tile) {
      const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
      const int ty = tyn % a.tiles_y, n = tyn / a.tiles_y;
      const int y0 = ty * TH, x0 = tx * TWO;
      const uint4* inN = a.in + (size_t)n * a.H * a.W * a.cs + a.coff;
      inside = 0;
#pragma unroll
      for (int u = 0; u < ITERS; ++u) {
        int lx = lx0 + (u * PIX_PER_IT) % IW_, ly = ly0 + (u * PIX_PER_IT) / IW_;
        if (lx >= IW_) { lx -= IW_; ly += 1; }
        const int gy = y0 - 2 + ly, gxi = x0 - 2 + lx;
        inside |= (unsigned)((unsigned)gy < (unsigned)a.H && (unsigned)gxi < (unsigned)a.W) << u;
        const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gxi, 0), a.W - 1);
        if constexpr (CP_HACK & 1) v[u] = make_uint4(u, 1, 2, 3); else v[u] = inN[(unsigned)((cy * a.W + cx) * a.cs + pl)];
      }
    };
    if (my_tiles > 0) request(t_begin);
    for (int it = 0; it < my_tiles; ++it) {
      const int tile = t_begin + it * nslots;
      const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
      const int ty = tyn % a.tiles_y, n = tyn / a.tiles_y;
      const int y0 = ty * TH, x0 = tx * TWO;
      uint4* const ldsMid = ldsMid0 + (it & 1) * cfg::MID_SLOTS;
      f32x2_t sh[4];
      {
        const float* sp = a.shift + (size_t)n * a.shift_stride + pl * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) sh[k] = f32x2_t{sp[2 * k], sp[2 * k + 1]};
      }
      if (tid < 32) {  // (read by the producers only, after S1)
        tabB[tid] = a.scale2[tid];
        tabB[32 + tid] = a.shift2[(size_t)n * a.shift2_stride + tid];
      }
#pragma unroll
      for (int u = 0; u < ITERS; ++u) {
        const uint4 o = conv_act8(v[u], sc, sh, 0u - ((inside >> u) & 1u));
        if (u == ITERS - 1 && pix0 + u * PIX_PER_IT >= RAW_IN) continue;
        if constexpr (!(CP_HACK & 64)) ldsIn[pl * PLANE_IN + pix0 + u * PIX_PER_IT] = o;
      }
      __syncthreads();  // S1: the window is complete
      if (it + 1 < my_tiles) request(tile + nslots);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        const int mbase = (wave * PASSES + pass) * R;  // first intermediate row of this wave and pass (image row y0 - 1 + mbase)
        f32x16_t acc[R];
        {
          float4 b4[4] = {};
          if (a.bias) {
            const float4* bp = reinterpret_cast<const float4*>(a.bias + c0);
#pragma unroll
            for (int q = 0; q < 4; ++q) b4[q] = bp[q];
          }
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[r][4 * q] = b4[q].x; acc[r][4 * q + 1] = b4[q].y; acc[r][4 * q + 2] = b4[q].z; acc[r][4 * q + 3] = b4[q].w; }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(CP_HACK & 8)) conv_mfma<R, KC, cfg::DEPTH, PLANE_IN, IW_>(reinterpret_cast<const bf16x8_t*>(ldsIn) + hh * PLANE_IN + mbase * IW_ + px, wA, acc);
        __builtin_amdgcn_sched_barrier(0);
        // bf16 as the two-launch path stores it, then B's prologue; zero where the intermediate pixel lies outside the image
        const int mgx = x0 - 1 + px;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int mgy = y0 - 1 + mbase + r;
          const unsigned keep = 0u - (unsigned)((unsigned)mgy < (unsigned)a.H && (unsigned)mgx < (unsigned)a.W);
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            f32x2_t s2[4], h2[4];
            const float4* tp = reinterpret_cast<const float4*>(tabB + c0 + 8 * half);
            const float4 sa = tp[0], sb = tp[1], ha = tp[8], hb = tp[9];
            s2[0] = f32x2_t{sa.x, sa.y}; s2[1] = f32x2_t{sa.z, sa.w}; s2[2] = f32x2_t{sb.x, sb.y}; s2[3] = f32x2_t{sb.z, sb.w};
            h2[0] = f32x2_t{ha.x, ha.y}; h2[1] = f32x2_t{ha.z, ha.w}; h2[2] = f32x2_t{hb.x, hb.y}; h2[3] = f32x2_t{hb.z, hb.w};
            const uint4 raw = make_uint4(cv_pack2(acc[r][8 * half + 0], acc[r][8 * half + 1]), cv_pack2(acc[r][8 * half + 2], acc[r][8 * half + 3]),
                                         cv_pack2(acc[r][8 * half + 4], acc[r][8 * half + 5]), cv_pack2(acc[r][8 * half + 6], acc[r][8 * half + 7]));
            if constexpr (!(CP_HACK & 32)) ldsMid[(2 * hh + half) * PLANE_MID + (mbase + r) * MW + px] = conv_act8(raw, s2, h2, keep);
          }
        }
      }
      __syncthreads();  // S2: planes[it & 1] are complete; the window (and tabB) may be overwritten
    }
    __syncthreads();  // the consumers' last tile: its S1 ...
    __syncthreads();  // ... and S2
  } else {
    // =============================================================== consumers: B's planes -> unit B -> OUT
    bf16x8_t wB[9 * KC];
    {
      const bf16x8_t* pb = reinterpret_cast<const bf16x8_t*>(a.wpk2) + lane;
#pragma unroll
      for (int i = 0; i < 9 * KC; ++i) wB[i] = pb[i * 64];
    }
    HeadW hwf;
    if constexpr (HEAD) {
      hwf = head_weights(a.hw, a.hO, lane);
      if (tid < 32) {
        hconst[tid] = a.hscale[tid];
        hconst[32 + tid] = a.hshift[tid];
        if (tid < 4) hconst[64 + tid] = tid < a.hO ? a.hbias[tid] : 0.f;
      }  // (visible to the consumers after the two fill barriers below)
    }
    float4 b4[4] = {};
    if (a.bias2) {
      const float4* bp = reinterpret_cast<const float4*>(a.bias2 + c0);
#pragma unroll
      for (int q = 0; q < 4; ++q) b4[q] = bp[q];
    }
    __syncthreads();  // the producers' first tile: its S1 ...
    __syncthreads();  // ... and S2
    // (measured and NOT adopted, round 3: requesting a pass's residual rows one pass ahead instead of right before they seed the
    // accumulators — 8.03 against 7.75 ms per step for the down-block pair: the consumers are not what the launch waits for)
    for (int it = 0; it < my_tiles; ++it) {
      const int tile = t_begin + it * nslots;
      const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
      const int ty = tyn % a.tiles_y, n = tyn / a.tiles_y;
      const int y0 = ty * TH, x0 = tx * TWO;
      const uint4* const ldsMid = ldsMid0 + (it & 1) * cfg::MID_SLOTS;
      const int gx = (px < TWO && x0 + px < a.W) ? x0 + px : a.W;  // a.W: this lane stores nothing
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        const int rbase = (wave * PASSES + pass) * R;
        if (rbase < TH) {  // (the last wave's second pass has no rows)
          uint4 rr[R][2];
          if (a.res && !(CP_HACK & 2)) conv_load_res<R>(a, n, 0, y0 + rbase, min(gx, a.W - 1), c0, rr);
          f32x16_t acc[R];
          conv_seed<R>(acc, b4, rr, a.res != nullptr && !(CP_HACK & 2));
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!(CP_HACK & 16)) conv_mfma<R, KC, cfg::DEPTH, PLANE_MID, MW>(reinterpret_cast<const bf16x8_t*>(ldsMid) + hh * PLANE_MID + rbase * MW + px, wB, acc);
          if constexpr (HEAD) {  // the output head from the accumulators (head_apply: the same bits as k_out_head)
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const f32x16_t y = head_from_acc(acc[r], hconst, hconst + 32, hwf, hh);
              const int gy = y0 + rbase + r;
              if (hh == 0 && gx < a.W && gy < a.H) {
                float* hp = a.hout + (size_t)n * a.hO * a.H * a.W + (size_t)gy * a.W + gx;
                const float yo[4] = {y[0], y[1], y[2], y[3]};
                for (int o = 0; o < a.hO; ++o) hp[(size_t)o * a.H * a.W] = yo[o] + hconst[64 + o];
              }
            }
          }
          if ((!HEAD || a.out) && (!(CP_HACK & 4) || acc[0][3] == 12345.f)) conv_store<R>(a, n, 0, y0 + rbase, gx, c0, acc);
          if constexpr (POOL) {
            const int PH = a.H >> 1, PW = a.W >> 1;
            const int gy = y0 + rbase;
            const bool writer = (px & 1) == 0 && gx + 1 < a.W && gy + 1 < a.H;
            uint4* pp = a.pool + (size_t)n * PH * PW * a.ocs + a.ocoff + (unsigned)(((gy >> 1) * PW + (min(gx, a.W - 1) >> 1)) * a.ocs + (c0 >> 3));
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              unsigned pk[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float v0 = fmaxf(acc[0][half * 8 + 2 * q], acc[1][half * 8 + 2 * q]);
                const float v1 = fmaxf(acc[0][half * 8 + 2 * q + 1], acc[1][half * 8 + 2 * q + 1]);
                pk[q] = cv_pack2(fmaxf(v0, __shfl_xor(v0, 1)), fmaxf(v1, __shfl_xor(v1, 1)));
              }
              if (writer && (!(CP_HACK & 4) || acc[0][3] == 12345.f)) pp[half] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
          }
        }
        __syncthreads();  // pass 0: the producers' S1 of tile it + 1 (or their drain); pass 1: S2
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The network's first two units in one launch: c0 = conv3x3(act0(tiles)) (the 2-channel first layer, K = 18) feeds
// x1 = conv3x3(act1(c0)) + proj(tiles) + bias without ever existing in HBM — the separate first-layer launch wrote 0.93 GB of c0
// and 0.23 GB of raw input per 288-tile forward for the next launch to read back.  Same wave-specialised shape as k_conv_pair32:
// waves 0-3 (producers) turn the float32 window of tile t + 1 into the activated first-layer input (bf16, zero outside the
// image), compute the 16 x 32 block of c0 it implies — two MFMA k-steps per 32 pixels, the im2col fragment gathered from LDS as
// k_first_conv does (k = 12 c + 4 ty + tx) but from a pair-packed window, two ds_read_b32 per four taps — round it to bf16
// where the two-launch path stores it, run it through unit 1's prologue and write unit 1's channel-octet planes; they also
// leave the tile's raw pixels (the projection's input) in LDS.  Waves 4-7 (consumers) hold unit 1's weights and the
// projection's: 9 x 2 + 1 k-steps per output row, in the order of k_conv3x3's fused projection.  Same bits as the two launches.
struct FirstPairCfg {
  static constexpr int KC = 2, NPL = 4, R = 2, PASSES = 2;
  static constexpr int TH = 14, TWO = 30, MH = 16, MW = 32, AH = 18, AW = 36;
  static constexpr int RAW_MID = MH * MW, PLANE_MID = RAW_MID + (10 - RAW_MID % 8) % 8;
  static constexpr int MID_SLOTS = NPL * PLANE_MID + 8;     // 16-byte slots
  static constexpr int P0_WORDS = 2 * AH * AW;              // pair-packed activated window, 32-bit words
  static constexpr int RAWP_WORDS = MH * MW;                // raw pixels of the tile as (ch0, ch1) bf16 pairs
  static constexpr int VALS = 2 * AH * AW, ITERS = (VALS + 255) / 256;
  static constexpr int LDS_BYTES = 2 * MID_SLOTS * 16 + P0_WORDS * 4 + 2 * RAWP_WORDS * 4;
  static constexpr int DEPTH = 6;
};

__global__ __launch_bounds__(512, 1) void k_conv_first_pair(ConvArgs a) {
  using cfg = FirstPairCfg;
  constexpr int KC = cfg::KC, R = cfg::R, PASSES = cfg::PASSES, TH = cfg::TH, TWO = cfg::TWO, MW = cfg::MW;
  constexpr int AH = cfg::AH, AW = cfg::AW, PLANE_MID = cfg::PLANE_MID, ITERS = cfg::ITERS;
  extern __shared__ uint4 lds[];
  uint4* const ldsMid0 = lds;
  unsigned* const P0 = reinterpret_cast<unsigned*>(lds + 2 * cfg::MID_SLOTS);
  unsigned* const rawP0 = P0 + cfg::P0_WORDS;
  __shared__ __align__(16) float tabB[64];  // unit 1's scale[32], shift[32]

  const int consumer = threadIdx.x >> 8;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 31, hh = lane >> 5;
  const int c0 = hh * 16;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int per_xcd = (a.ntiles + 7) >> 3;
  const int t_begin = xcd * per_xcd + slot, t_end = min(a.ntiles, (xcd + 1) * per_xcd);
  const int my_tiles = t_begin < t_end ? (t_end - t_begin + nslots - 1) / nslots : 0;
  const size_t plane = (size_t)a.H * a.W;

  if (!consumer) {
    // =============================================================== producers: float32 window -> c0 -> unit 1's planes
    // first-layer weights as MFMA A fragments (row m <-> channel 16*((m>>2)&1) + (m&3) + 4*(m>>3), as k_first_conv)
    bf16x8_t wF[2];
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
            v[e] = (c < a.fcin && tx < 3 && k < 24) ? a.fw[((size_t)co * a.fcin + c) * 9 + ty * 3 + tx] : 0.f;
          }
          r4[j2] = cv_pack2(v[0], v[1]);
        }
        wF[kc] = __builtin_bit_cast(bf16x8_t, make_uint4(r4[0], r4[1], r4[2], r4[3]));
      }
    }
    // im2col: this lane's two tap groups per k-step as word offsets into P0 (k0 = 16 kc + 8 hh; group g covers k0 + 4g .. + 3,
    // i.e. (c, ty) = ((k0 + 4g) / 12, ((k0 + 4g) % 12) / 4) and tx = 0..3); groups with k >= 24 meet zero weights and read as zero
    int goff[2][2];
    unsigned gmask[2][2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int k = 16 * kc + 8 * hh + 4 * g;
        const bool live = k < 24 && k / 12 < a.fcin;
        goff[kc][g] = live ? (k / 12) * AH * AW + ((k % 12) / 4) * AW : 0;
        gmask[kc][g] = live ? 0xffffffffu : 0u;
      }
    if (tid < 32) {
      tabB[tid] = a.scale2[tid];
      tabB[32 + tid] = a.shift2[tid];  // (shared by every sample: the first block carries no style shift)
    }
    const float fs0 = a.fscale[0], fh0 = a.fshift[0];
    const float fs1 = a.fcin > 1 ? a.fscale[1] : 0.f, fh1 = a.fcin > 1 ? a.fshift[1] : 0.f;
    // the float32 window of tile t + 1 is requested while tile t's c0 is on the matrix cores
    float v[ITERS];
    unsigned inside = 0;
    auto request = [&](int tile) {
      const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
      const int ty = tyn % a.tiles_y, n = tyn / a.tiles_y;
      const int y0 = ty * TH, x0 = tx * TWO;
      inside = 0;
#pragma unroll
      for (int u = 0; u < ITERS; ++u) {
        const int i = tid + 256 * u;
        const int c = i / (AH * AW), rem = i - c * (AH * AW), r = rem / AW, q = rem - r * AW;
        const int gy = y0 - 2 + r, gx = x0 - 2 + q;
        const bool ok = i < cfg::VALS && c < a.fcin && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        inside |= (unsigned)ok << u;
        const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1), cc = min(c, a.fcin - 1);
        v[u] = a.fx[((size_t)n * a.fcin + cc) * plane + (size_t)cy * a.W + cx];
      }
    };
    if (my_tiles > 0) request(t_begin);
    unsigned short* const P0h = reinterpret_cast<unsigned short*>(P0);
    for (int it = 0; it < my_tiles; ++it) {
      const int tile = t_begin + it * nslots;
      const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
      const int ty = tyn % a.tiles_y;
      const int y0 = ty * TH, x0 = tx * TWO;
      uint4* const ldsMid = ldsMid0 + (it & 1) * cfg::MID_SLOTS;
      unsigned short* const rawPh = reinterpret_cast<unsigned short*>(rawP0 + (it & 1) * cfg::RAWP_WORDS);
      // ---- activated window, pair-packed: word (c, r, q) = act(q) | act(q + 1) << 16 (k_first_conv's arithmetic: separate multiply
      // and add, ReLU in float, then round to bf16; zero outside the image), and the tile's raw pixels for the projection
#pragma unroll
      for (int u = 0; u < ITERS; ++u) {
        const int i = tid + 256 * u;
        if (i >= cfg::VALS) break;
        const int c = i / (AH * AW), rem = i - c * (AH * AW), r = rem / AW, q = rem - r * AW;
        const bool ok = (inside >> u) & 1u;
        const float t = ok ? fmaxf(v[u] * (c ? fs1 : fs0) + (c ? fh1 : fh0), 0.f) : 0.f;
        const unsigned short b = (unsigned short)(cv_pack2(t, 0.f) & 0xffffu);
        P0h[2 * i] = b;                       // low half of word (c, r, q)
        if (q > 0) P0h[2 * (i - 1) + 1] = b;  // high half of word (c, r, q - 1)
        if (q == AW - 1) P0h[2 * i + 1] = 0;
        // raw pixel (image row y0 + r - 2, column x0 + q - 2) of the output tile: rows 2..17 of the window, columns 2..33
        if (r >= 2 && q >= 2 && q < 2 + MW)
          rawPh[2 * ((r - 2) * MW + (q - 2)) + c] = (unsigned short)(cv_pack2(ok ? v[u] : 0.f, 0.f) & 0xffffu);
      }
      if (a.fcin < 2) {  // the second channel of the raw pairs is zero
        for (int i = tid; i < cfg::RAWP_WORDS; i += 256) rawPh[2 * i + 1] = 0;
      }
      __syncthreads();  // S1: the window is complete
      if (it + 1 < my_tiles) request(tile + nslots);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        const int mbase = (wave * PASSES + pass) * R;  // first c0 row of this wave and pass (image row y0 - 1 + mbase)
        const int mgx = x0 - 1 + px;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int m = mbase + r;
          f32x16_t acc;
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
          for (int kc = 0; kc < 2; ++kc) {
            const unsigned* w0 = P0 + goff[kc][0] + m * AW + px;
            const unsigned* w1 = P0 + goff[kc][1] + m * AW + px;
            const uint4 bfrag = make_uint4(w0[0] & gmask[kc][0], w0[2] & gmask[kc][0], w1[0] & gmask[kc][1], w1[2] & gmask[kc][1]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wF[kc], __builtin_bit_cast(bf16x8_t, bfrag), acc, 0, 0, 0);
          }
          // bf16 as the first-layer launch stores c0, then unit 1's prologue; zero where the c0 pixel lies outside the image
          const int mgy = y0 - 1 + m;
          const unsigned keep = 0u - (unsigned)((unsigned)mgy < (unsigned)a.H && (unsigned)mgx < (unsigned)a.W);
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            f32x2_t s2[4], h2[4];
            const float4* tp = reinterpret_cast<const float4*>(tabB + c0 + 8 * half);
            const float4 sa = tp[0], sb = tp[1], ha = tp[8], hb = tp[9];
            s2[0] = f32x2_t{sa.x, sa.y}; s2[1] = f32x2_t{sa.z, sa.w}; s2[2] = f32x2_t{sb.x, sb.y}; s2[3] = f32x2_t{sb.z, sb.w};
            h2[0] = f32x2_t{ha.x, ha.y}; h2[1] = f32x2_t{ha.z, ha.w}; h2[2] = f32x2_t{hb.x, hb.y}; h2[3] = f32x2_t{hb.z, hb.w};
            const uint4 raw = make_uint4(cv_pack2(acc[8 * half + 0], acc[8 * half + 1]), cv_pack2(acc[8 * half + 2], acc[8 * half + 3]),
                                         cv_pack2(acc[8 * half + 4], acc[8 * half + 5]), cv_pack2(acc[8 * half + 6], acc[8 * half + 7]));
            ldsMid[(2 * hh + half) * PLANE_MID + m * MW + px] = conv_act8(raw, s2, h2, keep);
          }
        }
      }
      __syncthreads();  // S2: planes[it & 1] and rawP[it & 1] are complete; the window may be overwritten
    }
    __syncthreads();  // the consumers' last tile: its S1 ...
    __syncthreads();  // ... and S2
  } else {
    // =============================================================== consumers: unit 1's planes (+ raw pixels) -> OUT
    bf16x8_t wB[9 * KC];
    {
      const bf16x8_t* pb = reinterpret_cast<const bf16x8_t*>(a.wpk2) + lane;
#pragma unroll
      for (int i = 0; i < 9 * KC; ++i) wB[i] = pb[i * 64];
    }
    const bf16x8_t pfrag = (reinterpret_cast<const bf16x8_t*>(a.pwpk) + lane)[0];
    float4 b4[4] = {};
    if (a.bias2) {
      const float4* bp = reinterpret_cast<const float4*>(a.bias2 + c0);
#pragma unroll
      for (int q = 0; q < 4; ++q) b4[q] = bp[q];
    }
    __syncthreads();  // the producers' first tile: its S1 ...
    __syncthreads();  // ... and S2
    for (int it = 0; it < my_tiles; ++it) {
      const int tile = t_begin + it * nslots;
      const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
      const int ty = tyn % a.tiles_y, n = tyn / a.tiles_y;
      const int y0 = ty * TH, x0 = tx * TWO;
      const uint4* const ldsMid = ldsMid0 + (it & 1) * cfg::MID_SLOTS;
      const unsigned* const rawP = rawP0 + (it & 1) * cfg::RAWP_WORDS;
      const int gx = (px < TWO && x0 + px < a.W) ? x0 + px : a.W;  // a.W: this lane stores nothing
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        const int rbase = (wave * PASSES + pass) * R;
        if (rbase < TH) {  // (the last wave's second pass has no rows)
          f32x16_t acc[R];
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[r][4 * q] = b4[q].x; acc[r][4 * q + 1] = b4[q].y; acc[r][4 * q + 2] = b4[q].z; acc[r][4 * q + 3] = b4[q].w; }
          __builtin_amdgcn_sched_barrier(0);
          conv_mfma<R, KC, cfg::DEPTH, PLANE_MID, MW>(reinterpret_cast<const bf16x8_t*>(ldsMid) + hh * PLANE_MID + rbase * MW + px, wB, acc);
          // the projection of the raw pixels: one more k-step (channels 0-1 real, the rest of the 16 zero)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const unsigned w = hh ? 0u : rawP[(rbase + r) * MW + px];
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pfrag, __builtin_bit_cast(bf16x8_t, make_uint4(w, 0u, 0u, 0u)), acc[r], 0, 0, 0);
          }
          conv_store<R>(a, n, 0, y0 + rbase, gx, c0, acc);
        }
        __syncthreads();  // pass 0: the producers' S1 of tile it + 1 (or their drain); pass 1: S2
      }
    }
  }
}

// The output head as its own launch (the form the fused variants are checked against, and the fallback when the last unit is
// not the 32 -> 32 shape): x bf16 NHWC [P, 32] -> float32 [N, O, H*W].  One lane pair per pixel, 32 pixels per wave step.
__global__ __launch_bounds__(256) void k_out_head_mfma(const uint4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ w, const float* __restrict__ bias, int O, size_t N, size_t P,
                                                       float* __restrict__ out) {
  __shared__ float hconst[2 * 32 + 4];
  const int tid = threadIdx.x, lane = tid & 63, px = lane & 31, hh = lane >> 5;
  if (tid < 32) {
    hconst[tid] = scale[tid];
    hconst[32 + tid] = shift[tid];
    if (tid < 4) hconst[64 + tid] = tid < O ? bias[tid] : 0.f;
  }
  const HeadW hwf = head_weights(w, O, lane);
  __syncthreads();
  const size_t total = N * P, nblk = (total + 31) / 32;
  const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + tid) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t b = wave0; b < nblk; b += nwaves) {
    const size_t pix = b * 32 + px;
    const bool ok = pix < total;
    const size_t q = ok ? pix : total - 1;
    const uint4 lo = x[q * 4 + 2 * hh], hi = x[q * 4 + 2 * hh + 1];
    const f32x16_t y = head_apply(lo, hi, hconst, hconst + 32, hwf, hh);
    if (ok && hh == 0) {
      const size_t n = pix / P, p = pix - n * P;
      const float yo[4] = {y[0], y[1], y[2], y[3]};
      for (int o = 0; o < O; ++o) out[(n * O + o) * P + p] = yo[o] + hconst[64 + o];
    }
  }
}

// Packed layout: [cout block cb][tap][k-step kc][lane][8 bf16]; lane l holds the MFMA A fragment
// A[row m = l&31][k = 8*(l>>5) + j] = W[cb*32 + chan(m)][cin = 16*kc + 8*(l>>5) + j][tap], with
// chan(m) = 16*((m>>2)&1) + (m&3) + 4*(m>>3) (see the header comment).
__global__ void k_pack_conv3x3(const float* w, int cout, int cin_src, int cin, int taps, unsigned short* out, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
  size_t rest = i >> 9;
  const int kcn = cin / 16;
  const int kc = (int)(rest % kcn);
  rest /= kcn;
  const int tap = (int)(rest % taps), cb = (int)(rest / taps);
  const int m = lane & 31, h = lane >> 5;
  const int co = cb * 32 + 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3);
  const int ci = 16 * kc + 8 * h + j;
  float v = 0.f;
  if (ci < cin_src && co < cout) v = w[((size_t)co * cin_src + ci) * taps + tap];
  out[i] = (unsigned short)cv_f2bf(v);
}

unsigned long long* g_conv_trace = nullptr;

template <int CIN, int COUT, bool UP, bool PACK = false>
int launch_conv_dma(aliby_ctx* ctx, ConvArgs& a, hipStream_t stream) {
  using cfg = DmaCfg<CIN, COUT, UP, PACK>;
  a.G = PACK ? 224 / a.W : 1;
  a.tiles_x = PACK ? 224 / cfg::TW : (a.W + cfg::TW - 1) / cfg::TW;
  a.tiles_y = (a.H + cfg::TH - 1) / cfg::TH;
  const long long nt = (long long)(a.N / a.G) * a.tiles_x * a.tiles_y;
  ARG_CHECK(nt < INT_MAX, "conv3x3: too many tiles");
  a.ntiles = (int)nt;
  static bool attr_done = false;
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_dma<CIN, COUT, UP, PACK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, cfg::LDS_BYTES));
    attr_done = true;
  }
  const int per_xcd = (a.ntiles + 7) / 8;
  const int cap = 32 * cfg::WAVES_PER_SIMD;  // workgroups per XCD: 32 CUs x (1 or 2) resident workgroups
  const int nslots = per_xcd < cap ? per_xcd : cap;
  hipLaunchKernelGGL((k_conv3x3_dma<CIN, COUT, UP, PACK>), dim3(8 * nslots), dim3(256), cfg::LDS_BYTES, stream, a);
  KERNEL_CHECK();
  return ALIBY_OK;
}

template <int CIN, int COUT, bool UP, bool TALL, bool PACK = false>
int launch_conv_reg(aliby_ctx* ctx, ConvArgs& a, hipStream_t stream) {
  using cfg = ConvCfg<CIN, COUT, TALL, PACK>;
  a.G = PACK ? 224 / a.W : 1;
  a.tiles_x = PACK ? 224 / cfg::TW : (a.W + cfg::TW - 1) / cfg::TW;
  a.tiles_y = (a.H + cfg::TH - 1) / cfg::TH;
  const long long nt = (long long)(a.N / a.G) * a.tiles_x * a.tiles_y;
  ARG_CHECK(nt < INT_MAX, "conv3x3: too many tiles");
  a.ntiles = (int)nt;
  static bool attr_done = false;
  constexpr int PKV = PACK ? 0 : (!UP && CIN == 32 && COUT == 32) ? 1 : ((!UP && CIN == 64 && COUT == 64) ? 2 : 0);  // projection input: 16 / 32 channels
  constexpr int P_BYTES = PKV ? 2 * PKV * ((cfg::TH * cfg::TW) | 1) * 16 : 0;
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3<CIN, COUT, UP, false, 0, TALL, PACK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, cfg::LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3<CIN, COUT, UP, true, 0, TALL, PACK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, cfg::LDS_BYTES));
    if constexpr (PKV > 0)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3<CIN, COUT, UP, false, PKV, TALL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, cfg::LDS_BYTES + P_BYTES));
    attr_done = true;
  }
  const int per_xcd = (a.ntiles + 7) / 8;
  const int nslots = per_xcd < 64 ? per_xcd : 64;  // 2 workgroups per CU, 32 CUs per XCD
  if (a.pin) {
    if constexpr (PKV > 0) {
      ARG_CHECK(!a.pool && a.pcs >= 1 && a.pcs <= 2 * PKV, "conv3x3: fused projection: unsupported input width");
      hipLaunchKernelGGL((k_conv3x3<CIN, COUT, UP, false, PKV, TALL>), dim3(8 * nslots), dim3(256), cfg::LDS_BYTES + P_BYTES, stream, a);
    } else {
      aliby_set_error("conv3x3: fused projection is built for (32,32) and (64,64) without upsampling only");
      return ALIBY_ERR_UNSUPPORTED;
    }
  } else if (a.hout) {
    if constexpr (COUT == 32 && CIN == 32 && !UP && !PACK) {
      static bool head_attr = false;
      if (!head_attr) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3<CIN, COUT, UP, false, 0, TALL, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, cfg::LDS_BYTES));
        head_attr = true;
      }
      ARG_CHECK(!a.pool && a.hO >= 1 && a.hO <= 3, "conv3x3: fused head: 1..3 output channels, no pooled output");
      hipLaunchKernelGGL((k_conv3x3<CIN, COUT, UP, false, 0, TALL, false, true>), dim3(8 * nslots), dim3(256), cfg::LDS_BYTES, stream, a);
    } else {
      aliby_set_error("conv3x3: the fused output head is built for the (32,32) unit only");
      return ALIBY_ERR_UNSUPPORTED;
    }
  } else if (a.pool) {
    hipLaunchKernelGGL((k_conv3x3<CIN, COUT, UP, true, 0, TALL, PACK>), dim3(8 * nslots), dim3(256), cfg::LDS_BYTES, stream, a);
  } else {
    hipLaunchKernelGGL((k_conv3x3<CIN, COUT, UP, false, 0, TALL, PACK>), dim3(8 * nslots), dim3(256), cfg::LDS_BYTES, stream, a);
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}

template <int CIN, int COUT, bool UP>
int launch_conv(aliby_ctx* ctx, ConvArgs& a, hipStream_t stream) {
  // measured per shape (scripts/bench_conv.py): the DMA pipeline wins where the raw window is small (upsampled
  // input); ALIBY_CONV_DMA=0/1 forces one variant for A/B runs
  static const bool use_dma = [] { const char* e = getenv("ALIBY_CONV_DMA"); return e ? atoi(e) != 0 : UP; }();
  if (use_dma && !a.pool && !a.pin && !a.hout) {  // the pooled output is an epilogue of the register-staged variant
    if constexpr (UP && COUT >= 64) {
      static const int pack_up = [] { const char* e = getenv("ALIBY_CONV_PACK"); return e ? atoi(e) : 7; }();  // bit 2: upsampled shapes
      if ((pack_up & 4) && (a.W == 56 || a.W == 112) && a.N % (224 / a.W) == 0 && a.H % DmaCfg<CIN, COUT, UP, true>::TH == 0)
        return launch_conv_dma<CIN, COUT, UP, true>(ctx, a, stream);
    }
    return launch_conv_dma<CIN, COUT, UP>(ctx, a, stream);
  }
  // a taller tile where it divides the image and fits two workgroups per CU (measured: 64->128 at 56 rows +10 %,
  // 32->32 at 224 rows +3..7 %: the fixed per-tile latency chain is amortised over twice the pixels)
  constexpr bool CAN_TALL = (CIN >= 64 && COUT >= 128) || (CIN == 32 && COUT == 32);
  // images narrower than the 224-pixel level-0 tile are packed side by side so that no MFMA column runs empty
  if constexpr (!UP && COUT >= 64) {
    static const int pack_on = [] { const char* e = getenv("ALIBY_CONV_PACK"); return e ? atoi(e) : 7; }();  // bit 0: 128-cout shapes, bit 1: 64-cout
    if ((pack_on & (COUT >= 128 ? 1 : 2)) && !a.pin && (a.W == 28 || a.W == 56 || a.W == 112) && a.N % (224 / a.W) == 0 &&
        a.H % ConvCfg<CIN, COUT, false, true>::TH == 0) {
      if constexpr (CIN == 64 && COUT == 64) {
        static const bool tall64 = [] { const char* e = getenv("ALIBY_CONV_TALL64"); return e ? atoi(e) != 0 : true; }();
        if (tall64 && a.H % ConvCfg<CIN, COUT, true, true>::TH == 0) return launch_conv_reg<CIN, COUT, UP, true, true>(ctx, a, stream);
      }
      return launch_conv_reg<CIN, COUT, UP, false, true>(ctx, a, stream);
    }
  }
  if (CAN_TALL && !a.pin && a.H % (2 * ConvCfg<CIN, COUT, false>::TH) == 0) return launch_conv_reg<CIN, COUT, UP, CAN_TALL>(ctx, a, stream);
  return launch_conv_reg<CIN, COUT, UP, false>(ctx, a, stream);
}

}  // namespace

struct HeadArgs { const float *scale, *shift, *w, *bias; float* out; int channels; };

static int conv3x3_entry(aliby_ctx* ctx, const void* in, const void* wpk, void* out, const float* scale,
                         const float* shift, int shift_per_sample, const float* bias, const void* res,
                         int res_up, int N, int H, int W, int CIN, int COUT, int in_up,
                         int in_channels, int in_channel0, int out_channels, int out_channel0,
                         void* pool_out, const void* proj_in, const void* proj_wpk, int proj_channels, void* stream_,
                         const HeadArgs* head = nullptr) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ARG_CHECK(ctx && in && wpk && (out || head) && scale && shift, "conv3x3: null argument");
  ARG_CHECK(N > 0 && H > 0 && W > 0, "conv3x3: empty shape");
  ARG_CHECK(!in_up || ((H & 1) == 0 && (W & 1) == 0), "conv3x3: upsampled input needs even H, W");
  ARG_CHECK(!res || !res_up || ((H & 1) == 0 && (W & 1) == 0), "conv3x3: upsampled residual needs even H, W");
  ConvArgs a;
  a.in = static_cast<const uint4*>(in);
  a.wpk = static_cast<const uint4*>(wpk);
  a.out = static_cast<uint4*>(out);
  a.scale = scale;
  a.shift = shift;
  a.bias = bias;
  a.res = static_cast<const uint4*>(res);
  a.shift_stride = shift_per_sample == 1 ? CIN : shift_per_sample;  // 1 = contiguous [N, CIN]; >1 = row stride in floats
  a.res_up = res_up ? 1 : 0;
  a.pool = static_cast<uint4*>(pool_out);
  ARG_CHECK(!pool_out || ((H & 1) == 0 && (W & 1) == 0), "conv3x3: pooled output needs even H, W");
  if (in_channels <= 0) in_channels = CIN;  // 0 = the input tensor has exactly CIN channels
  ARG_CHECK(in_channels % 8 == 0 && in_channel0 % 8 == 0 && in_channel0 >= 0 && in_channel0 + CIN <= in_channels,
            "conv3x3: channel slice must be octet aligned and inside the input tensor");
  if (out_channels <= 0) out_channels = COUT;  // 0 = the output tensor has exactly COUT channels
  ARG_CHECK(out_channels % 8 == 0 && out_channel0 % 8 == 0 && out_channel0 >= 0 && out_channel0 + COUT <= out_channels,
            "conv3x3: output channel slice must be octet aligned and inside the output tensor");
  a.ocs = out_channels / 8;
  a.ocoff = out_channel0 / 8;
  a.trace = g_conv_trace;
  a.hscale = head ? head->scale : nullptr;
  a.hshift = head ? head->shift : nullptr;
  a.hw = head ? head->w : nullptr;
  a.hbias = head ? head->bias : nullptr;
  a.hout = head ? head->out : nullptr;
  a.hO = head ? head->channels : 0;
  a.pin = static_cast<const uint4*>(proj_in);
  a.pwpk = static_cast<const uint4*>(proj_wpk);
  a.pcs = proj_channels / 8;
  ARG_CHECK(!proj_in || (proj_wpk && proj_channels > 0 && proj_channels % 8 == 0), "conv3x3: fused projection needs packed weights and an octet-aligned width");
  a.cs = in_channels / 8;
  a.coff = in_channel0 / 8;
  a.N = N; a.H = H; a.W = W;
  if (CIN == 32 && COUT == 32 && !in_up) return launch_conv<32, 32, false>(ctx, a, stream);
  if (CIN == 64 && COUT == 32 && in_up) return launch_conv<64, 32, true>(ctx, a, stream);
  if (CIN == 64 && COUT == 64 && !in_up) return launch_conv<64, 64, false>(ctx, a, stream);
  if (CIN == 64 && COUT == 64 && in_up) return launch_conv<64, 64, true>(ctx, a, stream);
  if (CIN == 64 && COUT == 128) return in_up ? launch_conv<64, 128, true>(ctx, a, stream) : launch_conv<64, 128, false>(ctx, a, stream);
  if (CIN == 32 && COUT == 64 && !in_up) return launch_conv<32, 64, false>(ctx, a, stream);
  aliby_set_error("conv3x3: unsupported (CIN=%d, COUT=%d, upsample=%d) combination", CIN, COUT, in_up);
  return ALIBY_ERR_UNSUPPORTED;
}

extern "C" int aliby_nn_conv3x3_bf16(aliby_ctx* ctx, const void* in, const void* wpk, void* out, const float* scale,
                                     const float* shift, int shift_per_sample, const float* bias, const void* res,
                                     int res_up, int N, int H, int W, int CIN, int COUT, int in_up,
                                     int in_channels, int in_channel0, int out_channels, int out_channel0,
                                     void* pool_out, void* stream) {
  return conv3x3_entry(ctx, in, wpk, out, scale, shift, shift_per_sample, bias, res, res_up, N, H, W, CIN, COUT, in_up, in_channels,
                       in_channel0, out_channels, out_channel0, pool_out, nullptr, nullptr, 0, stream);
}

extern "C" int aliby_nn_conv3x3_proj_bf16(aliby_ctx* ctx, const void* in, const void* wpk, void* out, const float* scale,
                                          const float* shift, int shift_per_sample, const float* bias, int N, int H, int W,
                                          int CIN, int COUT, const void* proj_in, const void* proj_wpk, int proj_channels,
                                          void* stream) {
  ARG_CHECK(proj_in && proj_wpk, "conv3x3_proj: NULL projection operand");
  return conv3x3_entry(ctx, in, wpk, out, scale, shift, shift_per_sample, bias, nullptr, 0, N, H, W, CIN, COUT, 0, 0, 0, 0, 0, nullptr,
                       proj_in, proj_wpk, proj_channels, stream);
}

extern "C" int aliby_nn_conv3x3_head_bf16(aliby_ctx* ctx, const void* in, const void* wpk, void* out_or_null, const float* scale,
                                          const float* shift, int shift_per_sample, const float* bias, const void* res, int res_up,
                                          int N, int H, int W, int CIN, int COUT, const float* head_scale, const float* head_shift,
                                          const float* head_w, const float* head_bias, int head_channels, float* head_out,
                                          void* stream) {
  ARG_CHECK(head_scale && head_shift && head_w && head_bias && head_out, "conv3x3_head: NULL head operand");
  ARG_CHECK(CIN == 32 && COUT == 32, "conv3x3_head: built for the 32 -> 32 unit");
  const HeadArgs head = {head_scale, head_shift, head_w, head_bias, head_out, head_channels};
  return conv3x3_entry(ctx, in, wpk, out_or_null, scale, shift, shift_per_sample, bias, res, res_up, N, H, W, CIN, COUT, 0, 0, 0, 0, 0,
                       nullptr, nullptr, nullptr, 0, stream, &head);
}

extern "C" int aliby_nn_conv3x3_pair_bf16(aliby_ctx* ctx, const void* in, const void* wpk_a, const void* wpk_b, void* out_or_null,
                                          const float* scale_a, const float* shift_a, int shift_a_per_sample, const float* bias_a,
                                          const float* scale_b, const float* shift_b, int shift_b_per_sample, const float* bias_b,
                                          const void* res, int N, int H, int W, void* pool_out, const float* head_scale,
                                          const float* head_shift, const float* head_w, const float* head_bias, int head_channels,
                                          float* head_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ARG_CHECK(ctx && in && wpk_a && wpk_b && (out_or_null || head_out) && scale_a && shift_a && scale_b && shift_b, "conv3x3_pair: null argument");
  ARG_CHECK(N > 0 && H > 0 && W > 0, "conv3x3_pair: empty shape");
  ARG_CHECK(!pool_out || ((H & 1) == 0 && (W & 1) == 0), "conv3x3_pair: pooled output needs even H, W");
  ARG_CHECK(!head_out || (head_scale && head_shift && head_w && head_bias && head_channels >= 1 && head_channels <= 3 && !pool_out),
            "conv3x3_pair: the fused head needs its operands, 1..3 channels and no pooled output");
  ARG_CHECK(!pool_out || out_or_null, "conv3x3_pair: the pooled output comes with the full output");
  ConvArgs a = {};
  a.in = static_cast<const uint4*>(in);
  a.wpk = static_cast<const uint4*>(wpk_a);
  a.wpk2 = static_cast<const uint4*>(wpk_b);
  a.out = static_cast<uint4*>(out_or_null);
  a.pool = static_cast<uint4*>(pool_out);
  a.scale = scale_a; a.shift = shift_a; a.bias = bias_a;
  a.scale2 = scale_b; a.shift2 = shift_b; a.bias2 = bias_b;
  a.shift_stride = shift_a_per_sample == 1 ? 32 : shift_a_per_sample;
  a.shift2_stride = shift_b_per_sample == 1 ? 32 : shift_b_per_sample;
  a.res = static_cast<const uint4*>(res);
  a.res_up = 0;
  a.cs = 4; a.coff = 0; a.ocs = 4; a.ocoff = 0;
  a.N = N; a.H = H; a.W = W; a.G = 1;
  a.hscale = head_scale; a.hshift = head_shift; a.hw = head_w; a.hbias = head_bias; a.hout = head_out; a.hO = head_out ? head_channels : 0;
  a.tiles_x = (W + PairCfg::TWO - 1) / PairCfg::TWO;
  a.tiles_y = (H + PairCfg::TH - 1) / PairCfg::TH;
  const long long nt = (long long)N * a.tiles_x * a.tiles_y;
  ARG_CHECK(nt < INT_MAX, "conv3x3_pair: too many tiles");
  a.ntiles = (int)nt;
  static bool attr_done = false;
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_pair32<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, PairCfg::LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_pair32<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, PairCfg::LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_pair32<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PairCfg::LDS_BYTES));
    attr_done = true;
  }
  const int per_xcd = (a.ntiles + 7) / 8;
  const int nslots = per_xcd < 32 ? per_xcd : 32;  // one 8-wave workgroup per CU, 32 CUs per XCD
  const dim3 grid(8 * nslots), block(512);
  if (head_out) hipLaunchKernelGGL((k_conv_pair32<false, true>), grid, block, PairCfg::LDS_BYTES, stream, a);
  else if (pool_out) hipLaunchKernelGGL((k_conv_pair32<true, false>), grid, block, PairCfg::LDS_BYTES, stream, a);
  else hipLaunchKernelGGL((k_conv_pair32<false, false>), grid, block, PairCfg::LDS_BYTES, stream, a);
  KERNEL_CHECK();
  return ALIBY_OK;
}

extern "C" int aliby_nn_first_pair_bf16(aliby_ctx* ctx, const float* tiles, int N, int Cin, int H, int W, const float* scale0,
                                        const float* shift0, const float* w_oihw, const void* wpk1, const float* scale1,
                                        const float* shift1, const float* bias1, const void* proj_wpk, void* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ARG_CHECK(ctx && tiles && scale0 && shift0 && w_oihw && wpk1 && scale1 && shift1 && bias1 && proj_wpk && out, "first_pair: null argument");
  ARG_CHECK(N > 0 && H > 0 && W > 0 && Cin >= 1 && Cin <= 2, "first_pair: Cin must be 1 or 2");
  ConvArgs a = {};
  a.fx = tiles; a.fw = w_oihw; a.fscale = scale0; a.fshift = shift0; a.fcin = Cin;
  a.wpk2 = static_cast<const uint4*>(wpk1);
  a.scale2 = scale1; a.shift2 = shift1; a.bias2 = bias1; a.shift2_stride = 0;
  a.pwpk = static_cast<const uint4*>(proj_wpk);
  a.out = static_cast<uint4*>(out);
  a.cs = 4; a.coff = 0; a.ocs = 4; a.ocoff = 0;
  a.N = N; a.H = H; a.W = W; a.G = 1;
  a.tiles_x = (W + FirstPairCfg::TWO - 1) / FirstPairCfg::TWO;
  a.tiles_y = (H + FirstPairCfg::TH - 1) / FirstPairCfg::TH;
  const long long nt = (long long)N * a.tiles_x * a.tiles_y;
  ARG_CHECK(nt < INT_MAX, "first_pair: too many tiles");
  a.ntiles = (int)nt;
  static bool attr_done = false;
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_first_pair), hipFuncAttributeMaxDynamicSharedMemorySize, FirstPairCfg::LDS_BYTES));
    attr_done = true;
  }
  const int per_xcd = (a.ntiles + 7) / 8;
  const int nslots = per_xcd < 32 ? per_xcd : 32;  // one 8-wave workgroup per CU, 32 CUs per XCD
  hipLaunchKernelGGL(k_conv_first_pair, dim3(8 * nslots), dim3(512), FirstPairCfg::LDS_BYTES, stream, a);
  KERNEL_CHECK();
  return ALIBY_OK;
}

extern "C" int aliby_nn_out_head_bf16(aliby_ctx* ctx, const void* x, const float* scale, const float* shift, const float* w,
                                      const float* bias, int N, int H, int W, int C, int O, float* out, void* stream) {
  ARG_CHECK(ctx && x && scale && shift && w && bias && out, "out_head: null argument");
  ARG_CHECK(C == 32 && O >= 1 && O <= 4, "out_head: C must be 32 and 1 <= O <= 4");
  ARG_CHECK(N > 0 && H > 0 && W > 0, "out_head: empty shape");
  const size_t P = (size_t)H * W, total = (size_t)N * P;
  const size_t waves = (total + 31) / 32;
  const unsigned blocks = (unsigned)((waves + 3) / 4 < 4096 ? (waves + 3) / 4 : 4096);
  hipLaunchKernelGGL(k_out_head_mfma, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const uint4*>(x), scale, shift, w,
                     bias, O, (size_t)N, P, out);
  KERNEL_CHECK();
  return ALIBY_OK;
}

extern "C" int aliby_debug_conv_trace(aliby_ctx* ctx, void* stamps_dev) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  g_conv_trace = static_cast<unsigned long long*>(stamps_dev);
  return ALIBY_OK;
}

static int pack_entry(aliby_ctx* ctx, const float* w, int COUT, int CIN_src, int CIN, int taps, void* wpk, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ARG_CHECK(ctx && w && wpk, "pack_conv: null argument");
  ARG_CHECK(COUT > 0 && COUT % 32 == 0 && CIN % 16 == 0 && CIN_src > 0 && CIN_src <= CIN, "pack_conv: COUT must be a multiple of 32 and CIN of 16");
  const size_t total = (size_t)COUT * CIN * taps;
  hipLaunchKernelGGL(k_pack_conv3x3, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, COUT, CIN_src, CIN, taps,
                     static_cast<unsigned short*>(wpk), total);
  KERNEL_CHECK();
  return ALIBY_OK;
}

extern "C" int aliby_nn_pack_conv3x3_bf16(aliby_ctx* ctx, const float* w_oihw, int COUT, int CIN_src, int CIN, void* wpk, void* stream) {
  return pack_entry(ctx, w_oihw, COUT, CIN_src, CIN, 9, wpk, stream);
}

extern "C" int aliby_nn_pack_conv1x1_bf16(aliby_ctx* ctx, const float* w_oi, int COUT, int CIN_src, int CIN, void* wpk, void* stream) {
  return pack_entry(ctx, w_oi, COUT, CIN_src, CIN, 1, wpk, stream);
}
