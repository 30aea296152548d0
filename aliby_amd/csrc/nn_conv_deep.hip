// nn_conv_deep.hip — the U-Net's 3x3 convolution unit at the deep levels (128 / 256 channels, 56 x 56 and 28 x 28 maps)
// as ONE launch with the whole K = 9 * CIN reduction accumulated in fp32 registers.
//
// Same unit as nn_conv.hip (cellpose's batchconv / batchconvstyle as driven by segment/dispatch.py:208-215):
//     OUT = conv3x3( relu(scale[c] * IN[n, y>>up, x>>up, c] + shift[n, c]) ) + bias + RES[n, y>>ru, x>>ru, :]
// Round 1 ran these layers as K-slices (64 input channels per launch) x N-slices (128 output channels) of the 64 -> 128
// kernel, each launch adding to the previous one through a bf16 tensor in HBM: a 256 -> 256 convolution was 8 launches that
// moved ~5x its algorithmic bytes and rounded its partial sums to 8 bits of mantissa three times.  Here:
//
//   * positions, not rows: the N images of a launch are stacked into one tall zero-separated image of width LW = W + 2 and
//     period HP = H + 1 rows (one shared zero row between images, one zero column left and right), flattened row-major.
//     A 3x3 tap is then a CONSTANT offset (dy-1)*LW + (dx-1) in that flat index, so a workgroup tile is simply RUN = 224
//     consecutive output positions (7 MFMA blocks of 32) and its input window RUN + 2*LW + 2 consecutive positions — no
//     per-row bookkeeping, no seam handling for 28- and 56-pixel images (cost: the zero row / columns are computed and
//     dropped, 1 - W/(W+2) * H/(H+1) = 10 % at 28, 5 % at 56);
//   * a wave owns 32 output channels x 7 position blocks = 7 accumulators (112 VGPRs) for the WHOLE reduction; a workgroup
//     is 4 waves = 128 output channels (layers with 256 are two tiles per run, adjacent in the schedule so that the second
//     finds the window in L2);
//   * K loop over 64-channel slices: the slice's window is staged once into LDS as channel-octet planes with the prologue
//     (BatchNorm affine + style shift + ReLU, zero padding after the activation — conv_act8 of nn_conv.hip) applied on the
//     way, then 4 k-steps x 9 taps x 7 blocks of v_mfma_f32_32x32x16_bf16 run from it;
//   * weights are NOT resident (9 * CIN/16 fragments per wave do not fit beside the accumulators): each (tap, k-step)
//     fragment — 1 KiB per wave, already in fragment order in the packed array, L2-resident — is loaded straight into VGPRs
//     through a 3-deep ring, seven MFMAs of cover each;
//   * two workgroups per CU (<= 256 VGPRs, 44 KiB of LDS each) overlap each other's staging and MFMA phases.
#include "common.h"
#ifndef DC_REQ_AT
#define DC_REQ_AT 27  // weight fragment (of 36 per slice) before which the next slice's window is requested
#endif
#ifndef DC_HACK
#define DC_HACK 0  // diagnostics (scripts/deep_phases.sh): compile-time phase switches, timing only, results wrong
#endif
#include <stdlib.h>
#include <utility>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

namespace {

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ float dc_bf2f(unsigned h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ unsigned dc_pack2(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
// prologue on one channel octet (same arithmetic, same rounding as conv_act8 in nn_conv.hip)
__device__ __forceinline__ uint4 dc_act8(uint4 v, const f32x2_t (&sc)[4], const f32x2_t (&sh)[4], unsigned keep) {
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

struct DeepArgs {
  const uint4* in;     // [N, H>>up, W>>up, CIN] bf16
  const uint4* wpk;    // aliby_nn_pack_conv3x3_bf16 over the full CIN: [cout block][tap][k-step][lane][8]
  uint4* out;          // [N, H, W, COUT] bf16
  const float* scale;  // [CIN]
  const float* shift;  // [N, CIN] (shift_stride = row stride) or [CIN] (0)
  const float* bias;   // [COUT] or NULL
  const uint4* res;    // [N, H>>res_up, W>>res_up, COUT] bf16 or NULL
  int shift_stride, res_up;
  int N, H, W, COUT;
  int LW, HP;          // W + 2, H + 1
  int q_begin, q_end;  // flat output positions [q_begin, q_end) of the tall image
  int nruns, nhalf, ntiles;
  int stagger;         // the second workgroup of every CU starts this many x 64 x 64 cycles late (see the kernel)
  unsigned long long* trace;  // diagnostics: shader-clock stamps of workgroup 0's first tiles (NULL in production)
};

#define DC_STAMP(slot_)                                                                                  \
  do {                                                                                                   \
    if (a.trace && blockIdx.x == 0 && tid == 0 && stamp_tile < 8 && (slot_) < 32)                        \
      a.trace[stamp_tile * 32 + (slot_)] = __builtin_amdgcn_s_memtime();                                 \
  } while (0)

constexpr int DC_NB = 7;              // position blocks of 32 per wave
constexpr int DC_RUN = DC_NB * 32;    // output positions per tile
constexpr int DC_WMAX = 56;           // widest image the LDS window is sized for
constexpr int DC_WIN_MAX = DC_RUN + 2 * (DC_WMAX + 2) + 2;
constexpr int DC_PLANE = DC_WIN_MAX | 1;       // odd slot pitch: the 8 octets of a position land in 8 distinct 16-byte slots
constexpr int DC_ITERS = (DC_WIN_MAX + 31) / 32;  // staging rounds: 32 positions x 8 octets per round of 256 threads
// planes + per-thread input offsets + prologue constants + per-thread output / residual offsets
constexpr int DC_LDS_BYTES = 8 * DC_PLANE * 16 + DC_ITERS * 256 * 4 + 3 * 256 * 4 + 2 * DC_NB * 256 * 4;
#ifndef DC_WDEPTH_
#define DC_WDEPTH_ 3
#endif
#ifndef DC_PDEPTH_
#define DC_PDEPTH_ 4
#endif
constexpr int DC_WDEPTH = DC_WDEPTH_;  // weight fragments in flight
constexpr int DC_PDEPTH = DC_PDEPTH_;  // pixel fragments in flight

template <int CIN, bool UP>
__global__ __launch_bounds__(256, 2) void k_conv3x3_deep(DeepArgs a) {
  constexpr int S = CIN / 64, KCT = CIN / 16;  // K slices, k-steps in the packed array
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, cb = tid >> 6;
  const int px = lane & 31, hh = lane >> 5;
  const int pl = tid & 7, pos0 = tid >> 3;  // staging role: a fixed channel octet of position pos0 + 32 * round
  const int IH = UP ? a.H >> 1 : a.H, IW = UP ? a.W >> 1 : a.W;
  const int LW = a.LW, HP = a.HP;
  const int WIN = DC_RUN + 2 * LW + 2;
  const int cs = CIN / 8;
  int* const goff = reinterpret_cast<int*>(lds + 8 * DC_PLANE) + tid;  // goff[it * 256]: this thread's input offsets of the tile

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int per_xcd = (a.ntiles + 7) >> 3;
  const int t_end = min(a.ntiles, (xcd + 1) * per_xcd);
  if (slot >= (nslots >> 1))
    for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(64);

  int stamp_tile = 0;
  for (int tile = xcd * per_xcd + slot; tile < t_end; tile += nslots, ++stamp_tile) {
    DC_STAMP(0);
    const int half = tile % a.nhalf, run = tile / a.nhalf;
    const int q0 = a.q_begin + run * DC_RUN;
    const int p_first = q0 - LW - 1;  // flat index of window position 0
    // ---- this thread's window positions -> input pixel offsets, once per tile (they do not depend on the K slice):
    // (r, c) of the first position by division, the following ones by stepping 32 positions at a time
    unsigned inside = 0, later = 0;
    const int n0 = max(0, ((p_first / LW) - 1) / HP);  // image of the window's first row (clamped)
    {
      const int P0 = p_first + pos0;  // >= -1
      int r = P0 >= 0 ? P0 / LW : -1, c = P0 - r * LW;
      int n = r >= 1 ? (r - 1) / HP : 0, y = r >= 1 ? (r - 1) - n * HP : r - 1;  // (y < 0: rows above the first image)
#pragma unroll
      for (int it = 0; it < DC_ITERS; ++it) {
        const bool ok = y >= 0 && y < a.H && c >= 1 && c <= a.W && n < a.N;
        inside |= (unsigned)ok << it;
        later |= (unsigned)(n > n0) << it;
        const int nn = min(n, a.N - 1), yy = min(max(y, 0), a.H - 1), xx = min(max(c - 1, 0), a.W - 1);
        goff[it * 256] = ((nn * IH + (UP ? yy >> 1 : yy)) * IW + (UP ? xx >> 1 : xx)) * cs + pl;
        c += 32;
        while (c >= LW) {  // (at most two wraps: LW >= 30 at the levels this kernel serves)
          c -= LW;
          if (++y == HP) { y = 0; ++n; }
        }
      }
    }
    const int n1 = min(n0 + 1, a.N - 1);
    // The raw window of a slice travels global -> registers while the matrix cores still work on the previous slice (the
    // loads are issued before the last k-step's 63 MFMAs): a slice's staging phase is then prologue + LDS writes only.
    uint4 v[DC_ITERS];
    auto request = [&](int s) {
      const uint4* inS = a.in + s * 8;
#pragma unroll
      for (int it = 0; it < DC_ITERS; ++it) { if constexpr (!(DC_HACK & 32)) v[it] = inS[(unsigned)goff[it * 256]]; else v[it] = make_uint4(it, 1, 2, 3); }
    };
    request(0);

    // ---- accumulators start at bias + residual (as in nn_conv.hip): the residual's loads are in flight together with the first
    // window's, behind the first staging phase, instead of twice exposed in the epilogue, which is then convert + store
    const int c0 = half * 128 + cb * 32 + hh * 16;
    const int ocs = a.COUT / 8;
    f32x16_t acc[DC_NB];
    {
      float b16[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) b16[k] = a.bias ? a.bias[c0 + k] : 0.f;
      if (a.res && !(DC_HACK & 64)) {
        const int RH = a.H >> a.res_up, RW = a.W >> a.res_up;
        uint4 rr[DC_NB][2];
        {
          const int q = q0 + px;
          int r = q / LW, c = q - r * LW;
          int n = (r - 1) / HP, y = (r - 1) - n * HP;  // r >= 1 always: q >= q_begin = LW
#pragma unroll
          for (int b = 0; b < DC_NB; ++b) {
            // clamped: positions that are not stored (zero row / columns, past the last image) read some valid address
            const int nn = min(n, a.N - 1), yy = min(y, a.H - 1), xx = min(max(c - 1, 0), a.W - 1);
            const unsigned roff = (unsigned)(((nn * RH + (yy >> a.res_up)) * RW + (xx >> a.res_up)) * ocs + (c0 >> 3));
            rr[b][0] = a.res[roff];
            rr[b][1] = a.res[roff + 1];
            c += 32;
            while (c >= LW) {
              c -= LW;
              if (++y == HP) { y = 0; ++n; }
            }
          }
        }
#pragma unroll
        for (int b = 0; b < DC_NB; ++b) {
          const unsigned rw[8] = {rr[b][0].x, rr[b][0].y, rr[b][0].z, rr[b][0].w, rr[b][1].x, rr[b][1].y, rr[b][1].z, rr[b][1].w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc[b][2 * j] = b16[2 * j] + dc_bf2f(rw[j] & 0xffffu);
            acc[b][2 * j + 1] = b16[2 * j + 1] + dc_bf2f(rw[j] >> 16);
          }
        }
      } else {
#pragma unroll
        for (int b = 0; b < DC_NB; ++b)
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[b][k] = b16[k];
      }
    }
    const bf16x8_t* wbase = reinterpret_cast<const bf16x8_t*>(a.wpk) + (size_t)(half * 4 + cb) * 9 * KCT * 64 + lane;
    DC_STAMP(1);

    for (int s = 0; s < S; ++s) {
      // the slice's first weight fragments are requested before the staging phase: they do not depend on it, and their
      // L2 latency would otherwise sit between the second barrier and the first MFMA of every slice
      const bf16x8_t* wp = wbase + (size_t)(4 * s) * 64;  // fragment (tap, kc) of this slice at wp[(tap * KCT + kc) * 64]
      constexpr int NW = 36;  // weight fragments per slice, in (kc, tap) order
      bf16x8_t wring[DC_WDEPTH];
      auto wfrag = [&](int i) { if constexpr (DC_HACK & 2) return wp[0]; else return wp[((i % 9) * KCT + (i / 9)) * 64]; };
#pragma unroll
      for (int i = 0; i < DC_WDEPTH - 1; ++i) wring[i] = wfrag(i);
      // the slice's prologue constants for this thread's octet
      f32x2_t sc[4], sh[4], sh1[4];
      {
        const int ch = s * 64 + pl * 8;
        const float* sp0 = a.shift + (size_t)n0 * a.shift_stride + ch;
        const float* sp1 = a.shift + (size_t)n1 * a.shift_stride + ch;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          sc[k] = f32x2_t{a.scale[ch + 2 * k], a.scale[ch + 2 * k + 1]};
          sh[k] = f32x2_t{sp0[2 * k], sp0[2 * k + 1]};
          sh1[k] = f32x2_t{sp1[2 * k], sp1[2 * k + 1]};
        }
      }
      __syncthreads();  // every wave is done reading the previous slice's planes
      DC_STAMP(2 + 4 * s);
      // ---- stage: prologue (BatchNorm affine + style shift + ReLU, zero outside the image) into the octet planes
#pragma unroll
      for (int it = 0; it < DC_ITERS; ++it) {
        f32x2_t shs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) shs[k] = ((later >> it) & 1u) ? sh1[k] : sh[k];
        const uint4 o = (DC_HACK & 8) ? v[it] : dc_act8(v[it], sc, shs, 0u - ((inside >> it) & 1u));
        const int wp_ = pos0 + 32 * it;
        if (wp_ < WIN) lds[pl * DC_PLANE + wp_] = o;
      }
      DC_STAMP(3 + 4 * s);
      __syncthreads();
      DC_STAMP(4 + 4 * s);

      // ---- 4 k-steps x 9 taps x 7 blocks; weight fragment (tap, k-step) comes from global through the ring
      const bf16x8_t* L = reinterpret_cast<const bf16x8_t*>(lds) + hh * DC_PLANE + px;
      const int row_off[3] = {0, LW, 2 * LW};
      constexpr int NF = NW * DC_NB;  // pixel fragments (= MFMAs) per slice
      bf16x8_t pring[DC_PDEPTH];
      auto pfrag = [&](int f) {
        const int i = f / DC_NB, b = f % DC_NB, kc = i / 9, tap = i % 9;
        if constexpr (DC_HACK & 4) return L[f & 3]; else return L[2 * kc * DC_PLANE + b * 32 + row_off[tap / 3] + tap % 3];
      };
#pragma unroll
      for (int f = 0; f < DC_PDEPTH - 1; ++f) pring[f] = pfrag(f);
      sfor<NW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i == DC_REQ_AT) {  // request the next slice's window (see DC_REQ_AT)
          if (s + 1 < S) request(s + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (i + DC_WDEPTH - 1 < NW) wring[(i + DC_WDEPTH - 1) % DC_WDEPTH] = wfrag(i + DC_WDEPTH - 1);
        sfor<DC_NB>([&](auto bc) {
          constexpr int b = decltype(bc)::value, f = i * DC_NB + b;
#ifdef DC_HACK_HALF_LDS
          if constexpr (f % 2 == 0 && f + 2 < NF) pring[(f + 2) % DC_PDEPTH] = pfrag(f + 2);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wring[i % DC_WDEPTH], pring[(f & ~1) % DC_PDEPTH], acc[b], 0, 0, 0);
#else
          if constexpr (f + DC_PDEPTH - 1 < NF) pring[(f + DC_PDEPTH - 1) % DC_PDEPTH] = pfrag(f + DC_PDEPTH - 1);
          if constexpr (DC_HACK & 128) {  // timing only: the same FLOPs and operand reads as two v_mfma_f32_16x16x32_bf16 (the clock the chip holds depends on the shape)
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            f32x4_t q0 = {acc[b][0], acc[b][1], acc[b][2], acc[b][3]}, q1 = {acc[b][4], acc[b][5], acc[b][6], acc[b][7]};
            q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wring[i % DC_WDEPTH], pring[f % DC_PDEPTH], q0, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wring[i % DC_WDEPTH], pring[f % DC_PDEPTH], q1, 0, 0, 0);
            acc[b][0] = q0[0]; acc[b][1] = q0[1]; acc[b][2] = q0[2]; acc[b][3] = q0[3];
            acc[b][4] = q1[0]; acc[b][5] = q1[1]; acc[b][6] = q1[2]; acc[b][7] = q1[3];
          } else if constexpr (!(DC_HACK & 1)) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wring[i % DC_WDEPTH], pring[f % DC_PDEPTH], acc[b], 0, 0, 0);
          else { acc[b][0] += (float)(__builtin_bit_cast(uint4, wring[i % DC_WDEPTH]).x + __builtin_bit_cast(uint4, pring[f % DC_PDEPTH]).x); }
#endif
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      DC_STAMP(5 + 4 * s);
    }

    // ---- epilogue: fp32 -> bf16, 32 contiguous bytes per lane and position.  Positions by stepping (no divisions)
    __builtin_amdgcn_sched_barrier(0);  // nothing of the epilogue is hoisted into the MFMA loop (it would spill there)
    {
      const int q = q0 + px;
      int r = q / LW, c = q - r * LW;
      int n = (r - 1) / HP, y = (r - 1) - n * HP;  // r >= 1 always: q >= q_begin = LW
#pragma unroll
      for (int b = 0; b < DC_NB; ++b) {
        const int x = c - 1;
        const bool ok = q0 + b * 32 + px < a.q_end && y < a.H && x >= 0 && x < a.W && n < a.N;
        if (ok && (!(DC_HACK & 16) || acc[b][3] == 12345.f)) {
          uint4* op = a.out + (unsigned)(((n * a.H + y) * a.W + x) * ocs + (c0 >> 3));
          op[0] = make_uint4(dc_pack2(acc[b][0], acc[b][1]), dc_pack2(acc[b][2], acc[b][3]), dc_pack2(acc[b][4], acc[b][5]),
                             dc_pack2(acc[b][6], acc[b][7]));
          op[1] = make_uint4(dc_pack2(acc[b][8], acc[b][9]), dc_pack2(acc[b][10], acc[b][11]), dc_pack2(acc[b][12], acc[b][13]),
                             dc_pack2(acc[b][14], acc[b][15]));
        }
        c += 32;
        while (c >= LW) {
          c -= LW;
          if (++y == HP) { y = 0; ++n; }
        }
      }
    }
    DC_STAMP(2 + 4 * S);
  }
}

template <int CIN, bool UP>
int launch_deep(DeepArgs& a, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_deep<CIN, UP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                DC_LDS_BYTES));
    attr_done = true;
  }
  const int per_xcd = (a.ntiles + 7) / 8;
  static const int cap = [] { const char* e = getenv("ALIBY_DEEP_SLOTS"); return e ? atoi(e) : 64; }();  // 2 workgroups per CU, 32 CUs per XCD
  const int nslots = per_xcd < cap ? per_xcd : cap;
  hipLaunchKernelGGL((k_conv3x3_deep<CIN, UP>), dim3(8 * nslots), dim3(256), DC_LDS_BYTES, stream, a);
  KERNEL_CHECK();
  return ALIBY_OK;
}

// ------------------------------------------------------------------------------------------------
// The same unit on v_mfma_f32_16x16x32_bf16 (round 3).  Under a dense MFMA stream the chip holds a higher clock on this shape
// than on 32x32x16 (MI355X_MICROARCH.md "Clocks under load" item 7; timing-only check in this kernel: +4.5 %), at the same
// cycles per FLOP and the same operand traffic.  What changes against k_conv3x3_deep:
//   * a wave's 32 output channels x 224 positions are 2 x 14 tiles of 16 x 16; a k-step is 32 channels: per (tap, k-step) two
//     weight fragments (channel halves a = 0, 1) and, per block of 32 positions, two pixel fragments (position halves h): each
//     pixel fragment feeds two MFMAs, each weight fragment fourteen;
//   * fragment layout (A[row l & 15][k = 8 (l >> 4) + j], B[k][col l & 15]): lane l reads the octet plane of ITS k-group, so a
//     ds_read_b128 touches four planes.  Its four 16-lane service groups pair k-groups (0, 1) and (2, 3): the two planes of a
//     pair must sit a multiple of 16 slots apart for the group to cover all 64 banks once.  With a plane pitch = 4 mod 16 that
//     holds for planes p and p + 4, so k-step jj of a 64-channel slice takes the octets (jj*2, jj*2 + 4, jj*2 + 1, jj*2 + 5) as
//     its k-groups 0..3 — the packed weights follow (k_pack_deep16).  The staging writes (8 octets of one position per 8 lanes)
//     are 2-way conflicted with this pitch instead of conflict-free: 11 writes per thread and slice against 252 reads;
//   * output rows: row 4 g + i of channel half a is output channel 8 g + 4 a + i of the wave's 32, so a lane (position c, row
//     group g) holds 8 CONTIGUOUS channels of each of its positions: one 16-byte store / residual load per position.
// Another summation order than the 32x32x16 kernel (32 channels per instruction): the two kernels agree to fp32 rounding, not
// bit for bit; exact on integer data (tests/test_gpu_conv.py).
// MEASURED SLOWER than k_conv3x3_deep (5.71 against 4.75 ms per forward; DESIGN.md 3.2: the 2-way write conflicts this pitch
// costs the staging phase, and 124-156 bytes of scratch per lane at 128 / 256 input channels): kept behind the C ABI
// (aliby_nn_conv3x3_deep16_bf16) with its tests, not used by the network.
constexpr int DC_PLANE16 = ((DC_WIN_MAX + 15 - 4) / 16) * 16 + 4 >= DC_WIN_MAX ? ((DC_WIN_MAX + 15 - 4) / 16) * 16 + 4 : ((DC_WIN_MAX + 15 - 4) / 16) * 16 + 20;
static_assert(DC_PLANE16 % 16 == 4 && DC_PLANE16 >= DC_WIN_MAX, "plane pitch of the 16x16x32 kernel");
constexpr int DC16_LDS_BYTES = 8 * DC_PLANE16 * 16 + DC_ITERS * 256 * 4;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int CIN, bool UP>
__global__ __launch_bounds__(256, 2) void k_conv3x3_deep16(DeepArgs a) {
  constexpr int S = CIN / 64, K32T = CIN / 32;  // K slices, 32-channel k-steps in the packed array
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, cb = tid >> 6;
  const int c16 = lane & 15, kg = lane >> 4;  // MFMA role: column (position) / k-group = output row group
  const int pl = tid & 7, pos0 = tid >> 3;    // staging role: a fixed channel octet of position pos0 + 32 * round
  const int IH = UP ? a.H >> 1 : a.H, IW = UP ? a.W >> 1 : a.W;
  const int LW = a.LW, HP = a.HP;
  const int WIN = DC_RUN + 2 * LW + 2;
  const int cs = CIN / 8;
  int* const goff = reinterpret_cast<int*>(lds + 8 * DC_PLANE16) + tid;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int per_xcd = (a.ntiles + 7) >> 3;
  const int t_end = min(a.ntiles, (xcd + 1) * per_xcd);

  for (int tile = xcd * per_xcd + slot; tile < t_end; tile += nslots) {
    const int half = tile % a.nhalf, run = tile / a.nhalf;
    const int q0 = a.q_begin + run * DC_RUN;
    const int p_first = q0 - LW - 1;
    unsigned inside = 0, later = 0;
    const int n0 = max(0, ((p_first / LW) - 1) / HP);
    {
      const int P0 = p_first + pos0;
      int r = P0 >= 0 ? P0 / LW : -1, c = P0 - r * LW;
      int n = r >= 1 ? (r - 1) / HP : 0, y = r >= 1 ? (r - 1) - n * HP : r - 1;
#pragma unroll
      for (int it = 0; it < DC_ITERS; ++it) {
        const bool ok = y >= 0 && y < a.H && c >= 1 && c <= a.W && n < a.N;
        inside |= (unsigned)ok << it;
        later |= (unsigned)(n > n0) << it;
        const int nn = min(n, a.N - 1), yy = min(max(y, 0), a.H - 1), xx = min(max(c - 1, 0), a.W - 1);
        goff[it * 256] = ((nn * IH + (UP ? yy >> 1 : yy)) * IW + (UP ? xx >> 1 : xx)) * cs + pl;
        c += 32;
        while (c >= LW) {
          c -= LW;
          if (++y == HP) { y = 0; ++n; }
        }
      }
    }
    const int n1 = min(n0 + 1, a.N - 1);
    uint4 v[DC_ITERS];
    auto request = [&](int s) {
      const uint4* inS = a.in + s * 8;
#pragma unroll
      for (int it = 0; it < DC_ITERS; ++it) v[it] = inS[(unsigned)goff[it * 256]];
    };
    request(0);

    // ---- accumulators start at bias + residual; lane (c16, kg) holds channels c0 .. c0 + 7 of positions q0 + 32 b + 16 h + c16
    const int c0 = half * 128 + cb * 32 + kg * 8;
    const int ocs = a.COUT / 8;
    f32x4_t acc[DC_NB][2][2];  // [block][channel half a][position half h]
    int qbase = q0 + c16;  // (re-read through an empty asm before the epilogue: its 14 positions are then computed again there
    // instead of being kept in 40 registers across the MFMA loops, which is what common-subexpression elimination does otherwise)
    auto where = [&](int b, int h, int& n, int& y, int& x) {  // image, row, column of this lane's position (b, h)
      const int q = qbase + b * 32 + h * 16;
      const int r = q / LW;  // >= 1: q >= q_begin = LW
      x = q - r * LW - 1;
      n = (r - 1) / HP;
      y = (r - 1) - n * HP;
    };
    {
      float b8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) b8[k] = a.bias ? a.bias[c0 + k] : 0.f;
      if (a.res) {
        const int RH = a.H >> a.res_up, RW = a.W >> a.res_up;
        uint4 rr[DC_NB][2];
#pragma unroll
        for (int b = 0; b < DC_NB; ++b)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            int n, y, x;
            where(b, h, n, y, x);
            const int nn = min(n, a.N - 1), yy = min(y, a.H - 1), xx = min(max(x, 0), a.W - 1);  // (clamped: not every position is stored)
            rr[b][h] = a.res[(unsigned)(((nn * RH + (yy >> a.res_up)) * RW + (xx >> a.res_up)) * ocs + (c0 >> 3))];
          }
#pragma unroll
        for (int b = 0; b < DC_NB; ++b)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const unsigned rw[4] = {rr[b][h].x, rr[b][h].y, rr[b][h].z, rr[b][h].w};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              acc[b][0][h][2 * j] = b8[2 * j] + dc_bf2f(rw[j] & 0xffffu);
              acc[b][0][h][2 * j + 1] = b8[2 * j + 1] + dc_bf2f(rw[j] >> 16);
              acc[b][1][h][2 * j] = b8[4 + 2 * j] + dc_bf2f(rw[2 + j] & 0xffffu);
              acc[b][1][h][2 * j + 1] = b8[4 + 2 * j + 1] + dc_bf2f(rw[2 + j] >> 16);
            }
          }
      } else {
#pragma unroll
        for (int b = 0; b < DC_NB; ++b)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[b][0][h][k] = b8[k]; acc[b][1][h][k] = b8[4 + k]; }
      }
    }
    // packed weights: [cout block][tap][32-channel k-step][a][lane][8]
    const bf16x8_t* wbase = reinterpret_cast<const bf16x8_t*>(a.wpk) + (size_t)(half * 4 + cb) * 9 * K32T * 2 * 64 + lane;

    for (int s = 0; s < S; ++s) {
      const bf16x8_t* wp = wbase + (size_t)(2 * s) * 2 * 64;  // fragment (tap, jj, a) of this slice at wp[((tap * K32T + jj) * 2 + a) * 64]
      constexpr int NW = 18;  // weight steps per slice, in (jj, tap) order; two fragments each
      constexpr int WD = 2;  // weight steps in flight: a step is 28 MFMAs = 448 cycles, the cover two 7-MFMA steps give the 32x32x16 kernel
      bf16x8_t wring[WD][2];
      auto wfrag = [&](int i, int aa) { return wp[(((i % 9) * K32T + (i / 9)) * 2 + aa) * 64]; };
#pragma unroll
      for (int i = 0; i < WD - 1; ++i) { wring[i][0] = wfrag(i, 0); wring[i][1] = wfrag(i, 1); }
      f32x2_t sc[4], sh[4], sh1[4];
      {
        const int ch = s * 64 + pl * 8;
        const float* sp0 = a.shift + (size_t)n0 * a.shift_stride + ch;
        const float* sp1 = a.shift + (size_t)n1 * a.shift_stride + ch;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          sc[k] = f32x2_t{a.scale[ch + 2 * k], a.scale[ch + 2 * k + 1]};
          sh[k] = f32x2_t{sp0[2 * k], sp0[2 * k + 1]};
          sh1[k] = f32x2_t{sp1[2 * k], sp1[2 * k + 1]};
        }
      }
      __syncthreads();  // every wave is done reading the previous slice's planes
#pragma unroll
      for (int it = 0; it < DC_ITERS; ++it) {
        f32x2_t shs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) shs[k] = ((later >> it) & 1u) ? sh1[k] : sh[k];
        const uint4 o = dc_act8(v[it], sc, shs, 0u - ((inside >> it) & 1u));
        const int wp_ = pos0 + 32 * it;
        if (wp_ < WIN) lds[pl * DC_PLANE16 + wp_] = o;
      }
      __syncthreads();

      // k-group kg of k-step jj reads octet plane 2 jj + (kg >> 1) + 4 (kg & 1) (see the header comment)
      const bf16x8_t* L = reinterpret_cast<const bf16x8_t*>(lds) + ((kg >> 1) + 4 * (kg & 1)) * DC_PLANE16 + c16;
      const int row_off[3] = {0, LW, 2 * LW};
      constexpr int NPF = 2 * DC_NB;     // pixel fragments per weight step
      constexpr int NF = NW * NPF;       // pixel fragments per slice (two MFMAs each)
      constexpr int PD = 3;  // pixel fragments in flight (two MFMAs each)
      bf16x8_t pring[PD];
      auto pfrag = [&](int f) {
        const int i = f / NPF, bh = f % NPF, jj = i / 9, tap = i % 9;
        return L[2 * jj * DC_PLANE16 + (bh >> 1) * 32 + (bh & 1) * 16 + row_off[tap / 3] + tap % 3];
      };
#pragma unroll
      for (int f = 0; f < PD - 1; ++f) pring[f] = pfrag(f);
      sfor<NW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i == NW - 5) {  // request the next slice's window (as DC_REQ_AT = 27 of 36 in the 32x32x16 kernel)
          if (s + 1 < S) request(s + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (i + WD - 1 < NW) {
          wring[(i + WD - 1) % WD][0] = wfrag(i + WD - 1, 0);
          wring[(i + WD - 1) % WD][1] = wfrag(i + WD - 1, 1);
        }
        sfor<NPF>([&](auto bc) {
          constexpr int bh = decltype(bc)::value, f = i * NPF + bh, b = bh >> 1, h = bh & 1;
          if constexpr (f + PD - 1 < NF) pring[(f + PD - 1) % PD] = pfrag(f + PD - 1);
          acc[b][0][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wring[i % WD][0], pring[f % PD], acc[b][0][h], 0, 0, 0);
          acc[b][1][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wring[i % WD][1], pring[f % PD], acc[b][1][h], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }

    // ---- epilogue: fp32 -> bf16, 16 contiguous bytes (8 channels) per lane and position
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(qbase));
#pragma unroll
    for (int b = 0; b < DC_NB; ++b)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int n, y, x;
        where(b, h, n, y, x);
        const bool ok = q0 + b * 32 + h * 16 + c16 < a.q_end && y < a.H && x >= 0 && x < a.W && n < a.N;
        if (ok)
          a.out[(unsigned)(((n * a.H + y) * a.W + x) * ocs + (c0 >> 3))] =
              make_uint4(dc_pack2(acc[b][0][h][0], acc[b][0][h][1]), dc_pack2(acc[b][0][h][2], acc[b][0][h][3]),
                         dc_pack2(acc[b][1][h][0], acc[b][1][h][1]), dc_pack2(acc[b][1][h][2], acc[b][1][h][3]));
      }
  }
}

template <int CIN, bool UP>
int launch_deep16(DeepArgs& a, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_deep16<CIN, UP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                DC16_LDS_BYTES));
    attr_done = true;
  }
  const int per_xcd = (a.ntiles + 7) / 8;
  const int nslots = per_xcd < 64 ? per_xcd : 64;  // 2 workgroups per CU, 32 CUs per XCD
  hipLaunchKernelGGL((k_conv3x3_deep16<CIN, UP>), dim3(8 * nslots), dim3(256), DC16_LDS_BYTES, stream, a);
  KERNEL_CHECK();
  return ALIBY_OK;
}

// weights for k_conv3x3_deep16: out[(((cb * 9 + tap) * (CIN / 32) + k32) * 2 + a) * 64 + lane][j] =
// W[32 cb + 8 (r >> 2) + 4 a + (r & 3)][64 (k32 >> 1) + 8 oct + j][tap], r = lane & 15, kg = lane >> 4,
// oct = 2 (k32 & 1) + (kg >> 1) + 4 (kg & 1)
__global__ void k_pack_deep16(const float* __restrict__ w, int cout, int cin, unsigned short* __restrict__ out, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
  size_t rest = i >> 9;
  const int aa = (int)(rest & 1);
  rest >>= 1;
  const int k32n = cin / 32;
  const int k32 = (int)(rest % k32n);
  rest /= k32n;
  const int tap = (int)(rest % 9), cb = (int)(rest / 9);
  const int r = lane & 15, kg = lane >> 4;
  const int co = cb * 32 + 8 * (r >> 2) + 4 * aa + (r & 3);
  const int ci = 64 * (k32 >> 1) + 8 * (2 * (k32 & 1) + (kg >> 1) + 4 * (kg & 1)) + j;
  const float v = w[((size_t)co * cin + ci) * 9 + tap];
  out[i] = (unsigned short)(dc_pack2(v, 0.f) & 0xffffu);
}

// ------------------------------------------------------------------------------------------------
// Loader-specialised form of the same unit (round 3).  Phase elimination on k_conv3x3_deep (scripts/deep_phases.sh, timing only):
// without the window loads 256 -> 256 runs 24 % faster, without residual loads 9 %, without stores 9 %, without all three
// 31 % (1458 TFLOP/s) — although the loads are issued a k-step ahead, and moving the request anywhere in the slice changes
// nothing.  The reason is the in-order vmcnt counter: a wave's weight-fragment waits (L2 hits, a few hundred cycles) cannot pass
// the HBM window loads the same wave issued before them, so every slice stalls for one loaded HBM latency in the middle of its
// MFMA stream.  Here the two kinds of traffic sit in different waves:
//   waves 0-3 (MFMA): stream weight fragments and issue MFMAs, nothing else inside a slice (residual loads and stores once per tile);
//   waves 4-5 (loaders): bring the next slice's window in by LDS-DMA (each wave owns its units of the raw image: no loader-to-loader
//     hand-off), run it through the prologue into the OTHER of two plane images, request the slice after that, and meet the MFMA
//     waves at ONE workgroup barrier per slice.
// One workgroup (4 MFMA + DL_NLW loader waves, ~132 KB of LDS) per CU.  Same MFMA order as k_conv3x3_deep: same bits.
#ifndef DL_HACK
#define DL_HACK 0  // diagnostics like DC_HACK: 1 loaders idle, 2 no MFMA loop, 4 no residual, 8 no stores, 16 no prologue math, 32 no DMA
#endif
#ifndef DL_NLW
#define DL_NLW 4                                          // loader waves
#endif
constexpr int DL_LT = DL_NLW * 64;                        // loader threads
constexpr int DL_UNITS = DC_WIN_MAX * 8;                  // 16-byte units of a slice's window (position-major, 8 octets each)
constexpr int DL_ROUNDS = (DL_UNITS + DL_LT - 1) / DL_LT; // DMA instructions per loader wave and slice
constexpr int DL_LDS_BYTES = 2 * 8 * DC_PLANE * 16 + DL_ROUNDS * DL_LT * 16;

template <int CIN, bool UP>
__global__ __launch_bounds__(256 + DL_LT, 1) void k_conv3x3_deep_ls(DeepArgs a) {
  constexpr int S = CIN / 64, KCT = CIN / 16;
  extern __shared__ uint4 lds[];
  uint4* const planes0 = lds;
  uint4* const rawbuf = lds + 2 * 8 * DC_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int IH = UP ? a.H >> 1 : a.H, IW = UP ? a.W >> 1 : a.W;
  const int LW = a.LW, HP = a.HP;
  const int WIN = DC_RUN + 2 * LW + 2;
  const int cs = CIN / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int per_xcd = (a.ntiles + 7) >> 3;
  const int t_begin = xcd * per_xcd + slot, t_end = min(a.ntiles, (xcd + 1) * per_xcd);
  const int my_tiles = t_begin < t_end ? (t_end - t_begin + nslots - 1) / nslots : 0;
  const int G = my_tiles * S;  // slices this workgroup goes through: one barrier each, for every wave

  if (wave >= 4) {
    // =========================================================================================== loaders
    const int lt = tid - 256;                 // 0..DL_LT-1
    const int oct = lt & 7;                   // this lane's channel octet of the slice (units lt + DL_LT k: positions (lt >> 3) + DL_LT/8 k)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave - 4);
    int goff[DL_ROUNDS];                      // input offsets (16-byte units) of this lane's units, slice 0
    unsigned inside = 0, later = 0;
    int n0 = 0, n1 = 0;
    auto tile_offsets = [&](int tile) {
      const int run = tile / a.nhalf;
      const int q0 = a.q_begin + run * DC_RUN;
      const int p_first = q0 - LW - 1;
      n0 = max(0, ((p_first / LW) - 1) / HP);
      n1 = min(n0 + 1, a.N - 1);
      inside = 0;
      later = 0;
#pragma unroll
      for (int k = 0; k < DL_ROUNDS; ++k) {
        const int wp_ = (lt >> 3) + (DL_LT / 8) * k;    // window position of unit lt + DL_LT k
        const int P0 = p_first + wp_;         // >= -1
        const int r = P0 >= 0 ? P0 / LW : -1, c = P0 - r * LW;
        const int n = r >= 1 ? (r - 1) / HP : 0, y = r >= 1 ? (r - 1) - n * HP : r - 1;
        const bool ok = wp_ < WIN && y >= 0 && y < a.H && c >= 1 && c <= a.W && n < a.N;
        inside |= (unsigned)ok << k;
        later |= (unsigned)(n > n0) << k;
        const int nn = min(n, a.N - 1), yy = min(max(y, 0), a.H - 1), xx = min(max(c - 1, 0), a.W - 1);
        goff[k] = ((nn * IH + (UP ? yy >> 1 : yy)) * IW + (UP ? xx >> 1 : xx)) * cs + oct;
      }
    };
    auto issue_dma = [&](int s) {  // this wave's units of slice s -> rawbuf (lane-linear: unit u lands at rawbuf[u])
      const uint4* inS = a.in + s * 8;
#pragma unroll
      for (int k = 0; k < DL_ROUNDS; ++k)
        if constexpr (!(DL_HACK & 32)) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(inS + (unsigned)goff[k]),
                                         (__attribute__((address_space(3))) void*)(rawbuf + k * DL_LT + wave_u * 64), 16, 0, 0);
    };
    auto activate = [&](int s, uint4* planes) {  // rawbuf -> prologue -> channel-octet planes
      f32x2_t sc[4], sh[4], sh1[4];
      const int ch = s * 64 + oct * 8;
      const float* sp0 = a.shift + (size_t)n0 * a.shift_stride + ch;
      const float* sp1 = a.shift + (size_t)n1 * a.shift_stride + ch;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sc[k] = f32x2_t{a.scale[ch + 2 * k], a.scale[ch + 2 * k + 1]};
        sh[k] = f32x2_t{sp0[2 * k], sp0[2 * k + 1]};
        sh1[k] = f32x2_t{sp1[2 * k], sp1[2 * k + 1]};
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces (and the constants above) have landed
#pragma unroll
      for (int k = 0; k < DL_ROUNDS; ++k) {
        const int wp_ = (lt >> 3) + (DL_LT / 8) * k;
        if (wp_ >= WIN) break;
        f32x2_t shs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) shs[q] = ((later >> k) & 1u) ? sh1[q] : sh[q];
        if constexpr (DL_HACK & 16) planes[oct * DC_PLANE + wp_] = rawbuf[k * DL_LT + lt];
        else planes[oct * DC_PLANE + wp_] = dc_act8(rawbuf[k * DL_LT + lt], sc, shs, 0u - ((inside >> k) & 1u));
      }
    };
    // slice g of the workgroup = (tile g / S, slice g % S); prepared one barrier ahead of its MFMAs
    if (G > 0) {
      tile_offsets(t_begin);
      issue_dma(0);
      activate(0, planes0);
      if (G > 1) {
        if (S == 1) tile_offsets(t_begin + nslots);
        issue_dma(1 % S);
      }
    }
    for (int g = 0; g < G; ++g) {
      __syncthreads();  // B_g: planes[g & 1] hold slice g; the MFMA waves are done with planes[(g + 1) & 1]
      if (g + 1 < G && !(DL_HACK & 1)) {
        const int s1 = (g + 1) % S;
        activate(s1, planes0 + ((g + 1) & 1) * 8 * DC_PLANE);  // (its offsets / masks are the ones its DMA was issued with)
        if (g + 2 < G) {
          const int s2 = (g + 2) % S;
          if (s2 == 0) tile_offsets(t_begin + ((g + 2) / S) * nslots);
          issue_dma(s2);
        }
      }
    }
    return;
  }

  // ============================================================================================= MFMA waves
  const int cb = wave, px = lane & 31, hh = lane >> 5;
  const int ocs = a.COUT / 8;
  const int row_off[3] = {0, LW, 2 * LW};
  int g = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
    const int tile = t_begin + ti * nslots;
    const int half = tile % a.nhalf, run = tile / a.nhalf;
    const int q0 = a.q_begin + run * DC_RUN;
    const int c0 = half * 128 + cb * 32 + hh * 16;
    f32x16_t acc[DC_NB];
    {
      float b16[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) b16[k] = a.bias ? a.bias[c0 + k] : 0.f;
      if (a.res && !(DL_HACK & 4)) {
        const int RH = a.H >> a.res_up, RW = a.W >> a.res_up;
        uint4 rr[DC_NB][2];
        {
          const int q = q0 + px;
          int r = q / LW, c = q - r * LW;
          int n = (r - 1) / HP, y = (r - 1) - n * HP;
#pragma unroll
          for (int b = 0; b < DC_NB; ++b) {
            const int nn = min(n, a.N - 1), yy = min(y, a.H - 1), xx = min(max(c - 1, 0), a.W - 1);
            const unsigned roff = (unsigned)(((nn * RH + (yy >> a.res_up)) * RW + (xx >> a.res_up)) * ocs + (c0 >> 3));
            rr[b][0] = a.res[roff];
            rr[b][1] = a.res[roff + 1];
            c += 32;
            while (c >= LW) {
              c -= LW;
              if (++y == HP) { y = 0; ++n; }
            }
          }
        }
#pragma unroll
        for (int b = 0; b < DC_NB; ++b) {
          const unsigned rw[8] = {rr[b][0].x, rr[b][0].y, rr[b][0].z, rr[b][0].w, rr[b][1].x, rr[b][1].y, rr[b][1].z, rr[b][1].w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc[b][2 * j] = b16[2 * j] + dc_bf2f(rw[j] & 0xffffu);
            acc[b][2 * j + 1] = b16[2 * j + 1] + dc_bf2f(rw[j] >> 16);
          }
        }
      } else {
#pragma unroll
        for (int b = 0; b < DC_NB; ++b)
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[b][k] = b16[k];
      }
    }
    const bf16x8_t* wbase = reinterpret_cast<const bf16x8_t*>(a.wpk) + (size_t)(half * 4 + cb) * 9 * KCT * 64 + lane;
    for (int s = 0; s < S; ++s, ++g) {
      const bf16x8_t* wp = wbase + (size_t)(4 * s) * 64;
      constexpr int NW = 36;
      bf16x8_t wring[DC_WDEPTH];
      auto wfrag = [&](int i) { if constexpr (DL_HACK & 64) return wp[0]; else return wp[((i % 9) * KCT + (i / 9)) * 64]; };
#pragma unroll
      for (int i = 0; i < DC_WDEPTH - 1; ++i) wring[i] = wfrag(i);
      __syncthreads();  // B_g
      const bf16x8_t* L = reinterpret_cast<const bf16x8_t*>(planes0 + (g & 1) * 8 * DC_PLANE) + hh * DC_PLANE + px;
      constexpr int NF = NW * DC_NB;
      bf16x8_t pring[DC_PDEPTH];
      auto pfrag = [&](int f) {
        const int i = f / DC_NB, b = f % DC_NB, kc = i / 9, tap = i % 9;
        if constexpr (DL_HACK & 128) return L[f & 3]; else return L[2 * kc * DC_PLANE + b * 32 + row_off[tap / 3] + tap % 3];
      };
#pragma unroll
      for (int f = 0; f < DC_PDEPTH - 1; ++f) pring[f] = pfrag(f);
      if constexpr (!(DL_HACK & 2)) sfor<NW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i + DC_WDEPTH - 1 < NW) wring[(i + DC_WDEPTH - 1) % DC_WDEPTH] = wfrag(i + DC_WDEPTH - 1);
        sfor<DC_NB>([&](auto bc) {
          constexpr int b = decltype(bc)::value, f = i * DC_NB + b;
          if constexpr (f + DC_PDEPTH - 1 < NF) pring[(f + DC_PDEPTH - 1) % DC_PDEPTH] = pfrag(f + DC_PDEPTH - 1);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wring[i % DC_WDEPTH], pring[f % DC_PDEPTH], acc[b], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const int q = q0 + px;
      int r = q / LW, c = q - r * LW;
      int n = (r - 1) / HP, y = (r - 1) - n * HP;
#pragma unroll
      for (int b = 0; b < DC_NB; ++b) {
        const int x = c - 1;
        const bool ok = q0 + b * 32 + px < a.q_end && y < a.H && x >= 0 && x < a.W && n < a.N;
        if (ok && (!(DL_HACK & 8) || acc[b][3] == 12345.f)) {
          uint4* op = a.out + (unsigned)(((n * a.H + y) * a.W + x) * ocs + (c0 >> 3));
          op[0] = make_uint4(dc_pack2(acc[b][0], acc[b][1]), dc_pack2(acc[b][2], acc[b][3]), dc_pack2(acc[b][4], acc[b][5]),
                             dc_pack2(acc[b][6], acc[b][7]));
          op[1] = make_uint4(dc_pack2(acc[b][8], acc[b][9]), dc_pack2(acc[b][10], acc[b][11]), dc_pack2(acc[b][12], acc[b][13]),
                             dc_pack2(acc[b][14], acc[b][15]));
        }
        c += 32;
        while (c >= LW) {
          c -= LW;
          if (++y == HP) { y = 0; ++n; }
        }
      }
    }
  }
}

template <int CIN, bool UP>
int launch_deep_ls(DeepArgs& a, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_deep_ls<CIN, UP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                DL_LDS_BYTES));
    attr_done = true;
  }
  const int per_xcd = (a.ntiles + 7) / 8;
  const int nslots = per_xcd < 32 ? per_xcd : 32;  // one workgroup per CU, 32 CUs per XCD
  hipLaunchKernelGGL((k_conv3x3_deep_ls<CIN, UP>), dim3(8 * nslots), dim3(256 + DL_LT), DL_LDS_BYTES, stream, a);
  KERNEL_CHECK();
  return ALIBY_OK;
}

// max_pool2d(x, 2, 2) on bf16 NHWC: one 16-byte channel octet per thread (the comparison runs on the bf16 values as
// floats; max commutes with the rounding)
__global__ void k_maxpool2(const uint4* __restrict__ in, uint4* __restrict__ out, int N, int H, int W, int cs) {
  const int PH = H >> 1, PW = W >> 1;
  const size_t total = (size_t)N * PH * PW * cs;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int o = (int)(i % cs);
    size_t p = i / cs;
    const int x = (int)(p % PW);
    p /= PW;
    const int y = (int)(p % PH), n = (int)(p / PH);
    const uint4* q = in + ((size_t)(n * H + 2 * y) * W + 2 * x) * cs + o;
    const uint4 v[4] = {q[0], q[cs], q[(size_t)W * cs], q[(size_t)W * cs + cs]};
    unsigned r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned w[4] = {k == 0 ? v[0].x : k == 1 ? v[0].y : k == 2 ? v[0].z : v[0].w, k == 0 ? v[1].x : k == 1 ? v[1].y : k == 2 ? v[1].z : v[1].w,
                             k == 0 ? v[2].x : k == 1 ? v[2].y : k == 2 ? v[2].z : v[2].w, k == 0 ? v[3].x : k == 1 ? v[3].y : k == 2 ? v[3].z : v[3].w};
      float lo = dc_bf2f(w[0] & 0xffffu), hi = dc_bf2f(w[0] >> 16);
#pragma unroll
      for (int j = 1; j < 4; ++j) {
        lo = fmaxf(lo, dc_bf2f(w[j] & 0xffffu));
        hi = fmaxf(hi, dc_bf2f(w[j] >> 16));
      }
      r[k] = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
    }
    out[i] = make_uint4(r[0], r[1], r[2], r[3]);
  }
}

unsigned long long* g_deep_trace = nullptr;

}  // namespace

extern "C" int aliby_debug_conv_deep_trace(aliby_ctx* ctx, void* stamps_dev) {
  (void)ctx;
  g_deep_trace = static_cast<unsigned long long*>(stamps_dev);
  return ALIBY_OK;
}

static int deep_entry(aliby_ctx* ctx, const void* in, const void* wpk, void* out, const float* scale, const float* shift,
                      int shift_per_sample, const float* bias, const void* res, int res_up, int N, int H, int W, int CIN, int COUT,
                      int in_up, void* stream_, bool m16) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ARG_CHECK(ctx && in && wpk && out && scale && shift, "conv3x3_deep: null argument");
  ARG_CHECK(N > 0 && H > 0 && W > 0, "conv3x3_deep: empty shape");
  ARG_CHECK(!in_up || ((H & 1) == 0 && (W & 1) == 0), "conv3x3_deep: upsampled input needs even H, W");
  ARG_CHECK(!res || !res_up || ((H & 1) == 0 && (W & 1) == 0), "conv3x3_deep: upsampled residual needs even H, W");
  if (!(CIN == 64 || CIN == 128 || CIN == 256) || COUT % 128 != 0 || COUT <= 0 || W > DC_WMAX) {
    aliby_set_error("conv3x3_deep: unsupported (CIN=%d, COUT=%d, W=%d): CIN in {64,128,256}, COUT a multiple of 128, W <= %d", CIN, COUT, W,
                    DC_WMAX);
    return ALIBY_ERR_UNSUPPORTED;
  }
  if (shift_per_sample != 0 && (long long)(H + 1) * (W + 2) < DC_RUN + 2 * (W + 2) + 2) {
    aliby_set_error("conv3x3_deep: images of %dx%d are smaller than one tile window (a window may span two images at most)", H, W);
    return ALIBY_ERR_UNSUPPORTED;
  }
  DeepArgs a;
  a.in = static_cast<const uint4*>(in);
  a.wpk = static_cast<const uint4*>(wpk);
  a.out = static_cast<uint4*>(out);
  a.scale = scale;
  a.shift = shift;
  a.bias = bias;
  a.res = static_cast<const uint4*>(res);
  a.shift_stride = shift_per_sample == 1 ? CIN : shift_per_sample;
  a.res_up = res_up ? 1 : 0;
  a.N = N; a.H = H; a.W = W; a.COUT = COUT;
  a.LW = W + 2;
  a.HP = H + 1;
  const long long rows = (long long)N * a.HP + 1;  // padded tall image: zero row, then N x (H rows + zero row)
  ARG_CHECK(rows * a.LW < (1ll << 30), "conv3x3_deep: batch too large for 32-bit flat positions");
  ARG_CHECK((long long)N * H * W * (CIN > COUT ? CIN : COUT) / 8 < (1ll << 31), "conv3x3_deep: tensor too large for 32-bit offsets");
  a.q_begin = a.LW;                         // row 1, column 0
  a.q_end = (int)((rows - 1) * a.LW);       // the last row is the closing zero row
  a.nruns = (a.q_end - a.q_begin + DC_RUN - 1) / DC_RUN;
  a.nhalf = COUT / 128;
  a.ntiles = a.nruns * a.nhalf;
  static const int stagger = [] { const char* e = getenv("ALIBY_DEEP_STAGGER"); return e ? atoi(e) : 0; }();
  a.stagger = stagger;
  a.trace = g_deep_trace;
  if (m16) {
    if (CIN == 64) return in_up ? launch_deep16<64, true>(a, stream) : launch_deep16<64, false>(a, stream);
    if (CIN == 128) return in_up ? launch_deep16<128, true>(a, stream) : launch_deep16<128, false>(a, stream);
    return in_up ? launch_deep16<256, true>(a, stream) : launch_deep16<256, false>(a, stream);
  }
  const char* ls = getenv("ALIBY_DEEP_LS");  // loader-specialised form: measured slower, kept for A/B; read per call
  if (ls && atoi(ls)) {
    if (CIN == 64) return in_up ? launch_deep_ls<64, true>(a, stream) : launch_deep_ls<64, false>(a, stream);
    if (CIN == 128) return in_up ? launch_deep_ls<128, true>(a, stream) : launch_deep_ls<128, false>(a, stream);
    return in_up ? launch_deep_ls<256, true>(a, stream) : launch_deep_ls<256, false>(a, stream);
  }
  if (CIN == 64) return in_up ? launch_deep<64, true>(a, stream) : launch_deep<64, false>(a, stream);
  if (CIN == 128) return in_up ? launch_deep<128, true>(a, stream) : launch_deep<128, false>(a, stream);
  return in_up ? launch_deep<256, true>(a, stream) : launch_deep<256, false>(a, stream);
}

extern "C" int aliby_nn_conv3x3_deep_bf16(aliby_ctx* ctx, const void* in, const void* wpk, void* out, const float* scale,
                                          const float* shift, int shift_per_sample, const float* bias, const void* res, int res_up,
                                          int N, int H, int W, int CIN, int COUT, int in_up, void* stream_) {
  return deep_entry(ctx, in, wpk, out, scale, shift, shift_per_sample, bias, res, res_up, N, H, W, CIN, COUT, in_up, stream_, false);
}

extern "C" int aliby_nn_conv3x3_deep16_bf16(aliby_ctx* ctx, const void* in, const void* wpk16, void* out, const float* scale,
                                            const float* shift, int shift_per_sample, const float* bias, const void* res, int res_up,
                                            int N, int H, int W, int CIN, int COUT, int in_up, void* stream_) {
  return deep_entry(ctx, in, wpk16, out, scale, shift, shift_per_sample, bias, res, res_up, N, H, W, CIN, COUT, in_up, stream_, true);
}

extern "C" int aliby_nn_pack_conv3x3_deep16_bf16(aliby_ctx* ctx, const float* w_oihw, int COUT, int CIN, void* wpk16, void* stream_) {
  ARG_CHECK(ctx && w_oihw && wpk16, "pack_conv_deep16: null argument");
  ARG_CHECK(COUT > 0 && COUT % 32 == 0 && CIN > 0 && CIN % 64 == 0, "pack_conv_deep16: COUT must be a multiple of 32 and CIN of 64");
  const size_t total = (size_t)COUT * CIN * 9;
  hipLaunchKernelGGL(k_pack_deep16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), w_oihw, COUT, CIN,
                     static_cast<unsigned short*>(wpk16), total);
  KERNEL_CHECK();
  return ALIBY_OK;
}

extern "C" int aliby_nn_maxpool2_bf16(aliby_ctx* ctx, const void* in, void* out, int N, int H, int W, int C, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ARG_CHECK(ctx && in && out, "maxpool2: null argument");
  ARG_CHECK(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (H & 1) == 0 && (W & 1) == 0, "maxpool2: even H, W and C a multiple of 8");
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(k_maxpool2, dim3(grid), dim3(256), 0, stream, static_cast<const uint4*>(in), static_cast<uint4*>(out), N, H, W, C / 8);
  KERNEL_CHECK();
  return ALIBY_OK;
}
