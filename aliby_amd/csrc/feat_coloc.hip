// feat_coloc.hip — cp_measure colocalisation metrics for one channel pair, one workgroup per object.
//
// Reference call site: wrap_cp_corr_features (extraction/core/functions/loaders.py:153-167),
// `fun(pixels1, pixels2, mask)`, with the metric list of the builder's multi tree
// (pipe_builder.py:33-43: pearson, costes, manders_fold, rwc) reached through measure_multi
// (extraction/extract.py:222-226).  cp_measure 0.1.17 is not vendored; restated from CellProfiler's
// MeasureColocalization (per-object branch):
//   pearson      : r = Sxy / (sqrt(Sxx) sqrt(Syy)) about the object means; slope = Sxy / Sxx
//   manders_fold : thresholds at thr% (15) of the object's maximum in each channel;
//                  M1 = sum(f | f>=t1 & s>=t2) / sum(f | f>=t1)
//   rwc          : dense ranks (ties share a rank) of both channels inside the object,
//                  weight = (R - |r1-r2|)/R with R = max rank + 1, applied to the Manders sums
//   costes       : orthogonal regression line + bisection on the candidate threshold ("Faster"
//                  mode) until Pearson r of the below-threshold pixels changes sign, then Manders-style
//                  fractions strictly above the thresholds.
//
// Data movement: the object's pixels of both planes are gathered once (order-preserving compaction)
// into LDS (global scratch for objects larger than the LDS budget); every statistic is an fp64 block
// reduction over that list; ranks come from one LDS bitonic sort per channel + binary searches.
#include "common.h"

typedef unsigned short u16;

// next probe of CellProfiler's "faster" Costes search: floor((right - left) / 1.2) + left while the bracket is wider than 6,
// else the midpoint.  The bounds are integers whenever scale_max is (left = 1, +-1 steps): floor(x / (6.0 / 5.0)) == (5 x) / 6
// and floor(x / 2.0) == x / 2 for every integer 0 <= x < 2^20 (checked exhaustively in tests/test_oracle_golden.py), which
// replaces an fp64 division per probe by integer arithmetic.
__device__ __forceinline__ double costes_next_mid(double left, double right, bool int_bounds) {
  if (int_bounds) {
    const int span = (int)(right - left);
    return (double)(span > 6 ? (5 * span) / 6 : span / 2) + left;
  }
  if (right - left > 6) return floor((right - left) / (6.0 / 5.0)) + left;
  return floor((right - left) / 2.0) + left;
}

struct ColocArgs {
  const u16* labels;
  const void* planes;  // [F,C,Y,X]
  int F, C, Y, X, ch0, ch1;
  const aliby_object* tab;
  int n_obj;
  int cap;  // power of two >= max area
  unsigned char* gscratch;
  double* out;
  int ld;
  int col_pearson, col_manders, col_rwc, col_costes;  // -1 = not requested
  double thr;        // percent of the maximum (15)
  double scale_max;  // costes candidate scale (255)
  const unsigned int* ranks;  // [F,C,Y,X] dense per-object ranks (aliby_object_ranks), needed for rwc
  const int* rmax;            // [n_obj, C] largest rank per (object, channel)
};

template <typename T, bool GLOBAL>
__global__ __launch_bounds__(256) void k_coloc(ColocArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ double vec[4 * 8];
  __shared__ double red_d[8];
  __shared__ float red_f[8];
  __shared__ int red_i[8];
  __shared__ int wsum[4];
  unsigned char* ws = GLOBAL ? (a.gscratch + (size_t)blockIdx.x * (size_t)a.cap * 16) : lds_raw;
  float* fv = reinterpret_cast<float*>(ws);
  float* sv = fv + a.cap;
  unsigned int* rk1 = reinterpret_cast<unsigned int*>(sv + a.cap);
  unsigned int* rk2 = rk1 + a.cap;
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;

  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    double* out = a.out + (size_t)oi * a.ld;
    if (o.area <= 0) {
      if (tid == 0) {
        if (a.col_pearson >= 0) { out[a.col_pearson] = NAN; out[a.col_pearson + 1] = NAN; }
        if (a.col_manders >= 0) { out[a.col_manders] = NAN; out[a.col_manders + 1] = NAN; }
        if (a.col_rwc >= 0) { out[a.col_rwc] = NAN; out[a.col_rwc + 1] = NAN; }
        if (a.col_costes >= 0) { out[a.col_costes] = NAN; out[a.col_costes + 1] = NAN; }
      }
      continue;
    }
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const T* p0 = reinterpret_cast<const T*>(a.planes) + ((size_t)o.tile * a.C + a.ch0) * plane;
    const T* p1 = reinterpret_cast<const T*>(a.planes) + ((size_t)o.tile * a.C + a.ch1) * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0, npix = h * w;
    const u16 L = (u16)o.label;

    // ---- gather (raster order) ------------------------------------------------------------
    __syncthreads();
    int base = 0;
    for (int i0 = 0; i0 < npix; i0 += blockDim.x) {
      const int i = i0 + tid;
      bool in = false;
      size_t idx = 0;
      if (i < npix) {
        idx = (size_t)(o.y0 + i / w) * a.X + (o.x0 + i % w);
        in = lab[idx] == L;
      }
      const int pos = block_compact_slot(in, base, wsum);
      if (in) {
        fv[pos] = px_load<T>(p0, idx); sv[pos] = px_load<T>(p1, idx);
        if (a.col_rwc >= 0) {
          rk1[pos] = a.ranks[((size_t)o.tile * a.C + a.ch0) * plane + idx];
          rk2[pos] = a.ranks[((size_t)o.tile * a.C + a.ch1) * plane + idx];
        }
      }
    }
    __syncthreads();
    const int N = base;
    const double dN = (double)N;

    // ---- sums, maxima -------------------------------------------------------------------------
    double acc[8];
    float m1 = -INFINITY, m2 = -INFINITY;
    acc[0] = acc[1] = 0;
    for (int j = tid; j < N; j += blockDim.x) {
      acc[0] += (double)fv[j]; acc[1] += (double)sv[j];
      m1 = fmaxf(m1, fv[j]); m2 = fmaxf(m2, sv[j]);
    }
    double s2[2] = {acc[0], acc[1]};
    block_sum_vec_all<2>(s2, vec);
    const double mean1 = s2[0] / dN, mean2 = s2[1] / dN;
    const float MAX1 = block_max_f32(m1, red_f), MAX2 = block_max_f32(m2, red_f);

    if (a.col_pearson >= 0) {
      double q[3] = {0, 0, 0};
      for (int j = tid; j < N; j += blockDim.x) {
        const double x = (double)fv[j] - mean1, y = (double)sv[j] - mean2;
        q[0] += x * x; q[1] += y * y; q[2] += x * y;
      }
      block_sum_vec_all<3>(q, vec);
      if (tid == 0) {
        out[a.col_pearson] = q[2] / (sqrt(q[0]) * sqrt(q[1]));
        out[a.col_pearson + 1] = q[2] / q[0];
      }
    }

    // ---- Manders / RWC share thresholds and denominators ---------------------------------------
    const double tff = (a.thr / 100.0) * (double)MAX1, tss = (a.thr / 100.0) * (double)MAX2;
    double tot1 = 0, tot2 = 0;
    int any_comb = 0;
    if (a.col_manders >= 0 || a.col_rwc >= 0) {
      double q[4] = {0, 0, 0, 0};
      int anyc = 0;
      for (int j = tid; j < N; j += blockDim.x) {
        const double f = fv[j], s = sv[j];
        const bool a1 = f >= tff, a2 = s >= tss;
        if (a1) q[0] += f;
        if (a2) q[1] += s;
        if (a1 && a2) { q[2] += f; q[3] += s; anyc = 1; }
      }
      block_sum_vec_all<4>(q, vec);
      any_comb = block_max_i32(anyc, red_i);
      tot1 = q[0]; tot2 = q[1];
      if (a.col_manders >= 0 && tid == 0) {
        out[a.col_manders] = any_comb ? q[2] / tot1 : 0.0;
        out[a.col_manders + 1] = any_comb ? q[3] / tot2 : 0.0;
      }
    }

    if (a.col_rwc >= 0) {
      // dense ranks come from the per-channel rank planes (one sort per object and channel, shared by all pairs)
      const double R = (double)(max(a.rmax[(size_t)oi * a.C + a.ch0], a.rmax[(size_t)oi * a.C + a.ch1]) + 1);
      double q[2] = {0, 0};
      for (int j = tid; j < N; j += blockDim.x) {
        const double f = fv[j], s = sv[j];
        if (f >= tff && s >= tss) {
          const long long di = llabs((long long)rk1[j] - (long long)rk2[j]);
          const double wgt = (R - (double)di) * 1.0 / R;
          q[0] += f * wgt; q[1] += s * wgt;
        }
      }
      block_sum_vec_all<2>(q, vec);
      if (tid == 0) {
        out[a.col_rwc] = any_comb ? q[0] / tot1 : 0.0;
        out[a.col_rwc + 1] = any_comb ? q[1] / tot2 : 0.0;
      }
    }

    if (a.col_costes >= 0) {
      // regression line through the non-zero pixels
      double q[3] = {0, 0, 0};
      for (int j = tid; j < N; j += blockDim.x) {
        const double f = fv[j], s = sv[j];
        if (f > 0 || s > 0) { q[0] += 1; q[1] += f; q[2] += s; }
      }
      block_sum_vec_all<3>(q, vec);
      const double nnz = q[0], xmean = q[1] / nnz, ymean = q[2] / nnz, zmean = (q[1] + q[2]) / nnz;
      double v3[3] = {0, 0, 0};
      for (int j = tid; j < N; j += blockDim.x) {
        const double f = fv[j], s = sv[j];
        if (f > 0 || s > 0) {
          const double dx = f - xmean, dy = s - ymean, dz = (f + s) - zmean;
          v3[0] += dx * dx; v3[1] += dy * dy; v3[2] += dz * dz;
        }
      }
      block_sum_vec_all<3>(v3, vec);
      const double xvar = v3[0] / (nnz - 1), yvar = v3[1] / (nnz - 1), zvar = v3[2] / (nnz - 1);
      const double covar = 0.5 * (zvar - (xvar + yvar));
      const double denom = 2 * covar;
      const double num = (yvar - xvar) + sqrt((yvar - xvar) * (yvar - xvar) + 4 * (covar * covar));
      const double ca = num / denom, cb = ymean - ca * xmean;
      double left = 1, right = a.scale_max;
      const bool int_bounds = a.scale_max == floor(a.scale_max) && a.scale_max >= 1 && a.scale_max <= 1048576.0;
      double mid = floor((right - left) / (6.0 / 5.0)) + left;
      double lastmid = 0, valid = 1;
      for (int it = 0; it < 200 && lastmid != mid; ++it) {
        const double t1 = mid / a.scale_max, t2 = ca * t1 + cb;
        double c3[3] = {0, 0, 0};
        for (int j = tid; j < N; j += blockDim.x) {
          const double f = fv[j], s = sv[j];
          if (f < t1 || s < t2) { c3[0] += 1; c3[1] += f; c3[2] += s; }
        }
        block_sum_vec_all<3>(c3, vec);
        if (c3[0] <= 2) {
          left = mid - 1;
        } else {
          const double mx = c3[1] / c3[0], my = c3[2] / c3[0];
          double p3[3] = {0, 0, 0};
          for (int j = tid; j < N; j += blockDim.x) {
            const double f = fv[j], s = sv[j];
            if (f < t1 || s < t2) { const double dx = f - mx, dy = s - my; p3[0] += dx * dx; p3[1] += dy * dy; p3[2] += dx * dy; }
          }
          block_sum_vec_all<3>(p3, vec);
          // r = clip(p3[2] / (sqrt(p3[0]) sqrt(p3[1])), -1, 1) is only ever compared with 0: its sign is p3[2]'s, and it is NaN
          // (neither bound moves) exactly when one of the variances is 0 — two fp64 square roots and a division less per
          // probe, executed by every lane (the scalar fp64 arithmetic of a probe cost more than its passes over the pixels)
          if (p3[0] != 0 && p3[1] != 0) {
            if (p3[2] < 0) left = mid - 1;
            else if (p3[2] >= 0) { right = mid + 1; valid = mid; }
          }
        }
        lastmid = mid;
        mid = costes_next_mid(left, right, int_bounds);
      }
      const double t1 = (valid - 1) / a.scale_max, t2 = ca * t1 + cb;
      double c4[4] = {0, 0, 0, 0};
      int f_any = 0, s_any = 0, c_any = 0;
      for (int j = tid; j < N; j += blockDim.x) {
        const double f = fv[j], s = sv[j];
        const bool fa = f > t1, sa = s > t2;
        f_any |= fa; s_any |= sa;
        if (f >= t1) c4[0] += f;
        if (s >= t2) c4[1] += s;
        if (fa && sa) { c4[2] += f; c4[3] += s; c_any = 1; }
      }
      block_sum_vec_all<4>(c4, vec);
      const int FA = block_max_i32(f_any, red_i), SA = block_max_i32(s_any, red_i), CA = block_max_i32(c_any, red_i);
      if (tid == 0) {
        const double d1 = FA ? c4[0] : 0.0, d2 = SA ? c4[1] : 0.0;
        out[a.col_costes] = CA ? c4[2] / d1 : 0.0;
        out[a.col_costes + 1] = CA ? c4[3] / d2 : 0.0;
      }
    }
    (void)red_d;
    __syncthreads();
  }
}


// -----------------------------------------------------------------------------------------------------
// All channel pairs of an object in one workgroup: the pixels of every channel in use are gathered once, then each
// wave takes pairs in turn and runs the statistics above with wave-level reductions (shuffles, no barriers) — the
// per-pair kernel is a chain of ~40 dependent reductions over a few hundred pixels, i.e. latency, and ten of them per
// object ran as ten launches.  Same arithmetic per pair; sums are folded in wave order instead of block order.
// -----------------------------------------------------------------------------------------------------
#define COLOC_MAX_CH 8
#define COLOC_MAX_PAIRS 28
struct PairsArgs {
  const u16* labels;
  const void* planes;
  int F, C, Y, X;
  const aliby_object* tab;
  int n_obj, cap;
  double* out;
  int ld;
  double thr, scale_max;
  const unsigned int* ranks;
  const int* rmax;
  int nch, npairs, any_rwc;
  int chan[COLOC_MAX_CH];
  int pair[COLOC_MAX_PAIRS][6];  // local channel a, local channel b, col_pearson, col_manders, col_rwc, col_costes
};

template <int K>
__device__ __forceinline__ void wave_sum_all(double (&v)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);  // (the total, in every lane)
}

template <typename T>
__global__ __launch_bounds__(256) void k_coloc_pairs(PairsArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ int wsum[4];
  float* vals = reinterpret_cast<float*>(lds_raw);                                // [nch][cap]
  unsigned int* rks = reinterpret_cast<unsigned int*>(vals + (size_t)a.nch * a.cap);  // [nch][cap] when any pair wants rwc
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t plane = (size_t)a.Y * a.X;

  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    double* out = a.out + (size_t)oi * a.ld;
    if (o.area <= 0) {
      for (int p = tid; p < a.npairs; p += blockDim.x)
        for (int k = 2; k < 6; ++k)
          if (a.pair[p][k] >= 0) { out[a.pair[p][k]] = NAN; out[a.pair[p][k] + 1] = NAN; }
      continue;
    }
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const T* img = reinterpret_cast<const T*>(a.planes) + (size_t)o.tile * a.C * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0, npix = h * w;
    const u16 L = (u16)o.label;
    __syncthreads();  // the previous object's pairs are done with the lists
    int base = 0;
    for (int i0 = 0; i0 < npix; i0 += blockDim.x) {
      const int i = i0 + tid;
      bool in = false;
      size_t idx = 0;
      if (i < npix) {
        idx = (size_t)(o.y0 + i / w) * a.X + (o.x0 + i % w);
        in = lab[idx] == L;
      }
      const int pos = block_compact_slot(in, base, wsum);
      if (in) {
        for (int c = 0; c < a.nch; ++c) {
          vals[(size_t)c * a.cap + pos] = px_load<T>(img + (size_t)a.chan[c] * plane, idx);
          if (a.any_rwc) rks[(size_t)c * a.cap + pos] = a.ranks[((size_t)o.tile * a.C + a.chan[c]) * plane + idx];
        }
      }
    }
    __syncthreads();
    const int N = base;
    const double dN = (double)N;

    for (int p = wave; p < a.npairs; p += 4) {
      const int ca = a.pair[p][0], cb = a.pair[p][1];
      const int col_pearson = a.pair[p][2], col_manders = a.pair[p][3], col_rwc = a.pair[p][4], col_costes = a.pair[p][5];
      const float* fv = vals + (size_t)ca * a.cap;
      const float* sv = vals + (size_t)cb * a.cap;
      const unsigned int* rk1 = rks + (size_t)ca * a.cap;
      const unsigned int* rk2 = rks + (size_t)cb * a.cap;

      double s2[2] = {0, 0};
      float m1 = -INFINITY, m2 = -INFINITY;
      for (int j = lane; j < N; j += 64) {
        s2[0] += (double)fv[j]; s2[1] += (double)sv[j];
        m1 = fmaxf(m1, fv[j]); m2 = fmaxf(m2, sv[j]);
      }
      wave_sum_all<2>(s2);
      const double mean1 = s2[0] / dN, mean2 = s2[1] / dN;
      const float MAX1 = wave_max(m1), MAX2 = wave_max(m2);

      if (col_pearson >= 0) {
        double q[3] = {0, 0, 0};
        for (int j = lane; j < N; j += 64) {
          const double x = (double)fv[j] - mean1, y = (double)sv[j] - mean2;
          q[0] += x * x; q[1] += y * y; q[2] += x * y;
        }
        wave_sum_all<3>(q);
        if (lane == 0) {
          out[col_pearson] = q[2] / (sqrt(q[0]) * sqrt(q[1]));
          out[col_pearson + 1] = q[2] / q[0];
        }
      }

      const double tff = (a.thr / 100.0) * (double)MAX1, tss = (a.thr / 100.0) * (double)MAX2;
      double tot1 = 0, tot2 = 0;
      int any_comb = 0;
      if (col_manders >= 0 || col_rwc >= 0) {
        double q[4] = {0, 0, 0, 0};
        int anyc = 0;
        for (int j = lane; j < N; j += 64) {
          const double f = fv[j], s = sv[j];
          const bool a1 = f >= tff, a2 = s >= tss;
          if (a1) q[0] += f;
          if (a2) q[1] += s;
          if (a1 && a2) { q[2] += f; q[3] += s; anyc = 1; }
        }
        wave_sum_all<4>(q);
        any_comb = wave_max(anyc);
        tot1 = q[0]; tot2 = q[1];
        if (col_manders >= 0 && lane == 0) {
          out[col_manders] = any_comb ? q[2] / tot1 : 0.0;
          out[col_manders + 1] = any_comb ? q[3] / tot2 : 0.0;
        }
      }

      if (col_rwc >= 0) {
        const double R = (double)(max(a.rmax[(size_t)oi * a.C + a.chan[ca]], a.rmax[(size_t)oi * a.C + a.chan[cb]]) + 1);
        double q[2] = {0, 0};
        for (int j = lane; j < N; j += 64) {
          const double f = fv[j], s = sv[j];
          if (f >= tff && s >= tss) {
            const long long di = llabs((long long)rk1[j] - (long long)rk2[j]);
            const double wgt = (R - (double)di) * 1.0 / R;
            q[0] += f * wgt; q[1] += s * wgt;
          }
        }
        wave_sum_all<2>(q);
        if (lane == 0) {
          out[col_rwc] = any_comb ? q[0] / tot1 : 0.0;
          out[col_rwc + 1] = any_comb ? q[1] / tot2 : 0.0;
        }
      }

      if (col_costes >= 0) {
        double q[3] = {0, 0, 0};
        for (int j = lane; j < N; j += 64) {
          const double f = fv[j], s = sv[j];
          if (f > 0 || s > 0) { q[0] += 1; q[1] += f; q[2] += s; }
        }
        wave_sum_all<3>(q);
        const double nnz = q[0], xmean = q[1] / nnz, ymean = q[2] / nnz, zmean = (q[1] + q[2]) / nnz;
        double v3[3] = {0, 0, 0};
        for (int j = lane; j < N; j += 64) {
          const double f = fv[j], s = sv[j];
          if (f > 0 || s > 0) {
            const double dx = f - xmean, dy = s - ymean, dz = (f + s) - zmean;
            v3[0] += dx * dx; v3[1] += dy * dy; v3[2] += dz * dz;
          }
        }
        wave_sum_all<3>(v3);
        const double xvar = v3[0] / (nnz - 1), yvar = v3[1] / (nnz - 1), zvar = v3[2] / (nnz - 1);
        const double covar = 0.5 * (zvar - (xvar + yvar));
        const double denom = 2 * covar;
        const double num = (yvar - xvar) + sqrt((yvar - xvar) * (yvar - xvar) + 4 * (covar * covar));
        const double cA = num / denom, cB = ymean - cA * xmean;
        double left = 1, right = a.scale_max;
      const bool int_bounds = a.scale_max == floor(a.scale_max) && a.scale_max >= 1 && a.scale_max <= 1048576.0;
        double mid = floor((right - left) / (6.0 / 5.0)) + left;
        double lastmid = 0, valid = 1;
        for (int it = 0; it < 200 && lastmid != mid; ++it) {
          const double t1 = mid / a.scale_max, t2 = cA * t1 + cB;
          double c3[3] = {0, 0, 0};
          for (int j = lane; j < N; j += 64) {
            const double f = fv[j], s = sv[j];
            if (f < t1 || s < t2) { c3[0] += 1; c3[1] += f; c3[2] += s; }
          }
          wave_sum_all<3>(c3);
          if (c3[0] <= 2) {
            left = mid - 1;
          } else {
            const double mx = c3[1] / c3[0], my = c3[2] / c3[0];
            double p3[3] = {0, 0, 0};
            for (int j = lane; j < N; j += 64) {
              const double f = fv[j], s = sv[j];
              if (f < t1 || s < t2) { const double dx = f - mx, dy = s - my; p3[0] += dx * dx; p3[1] += dy * dy; p3[2] += dx * dy; }
            }
            wave_sum_all<3>(p3);
            if (p3[0] != 0 && p3[1] != 0) {  // (the sign of r is p3[2]'s; see k_coloc)
              if (p3[2] < 0) left = mid - 1;
              else if (p3[2] >= 0) { right = mid + 1; valid = mid; }
            }
          }
          lastmid = mid;
          mid = costes_next_mid(left, right, int_bounds);
        }
        const double t1 = (valid - 1) / a.scale_max, t2 = cA * t1 + cB;
        double c4[4] = {0, 0, 0, 0};
        int f_any = 0, s_any = 0, c_any = 0;
        for (int j = lane; j < N; j += 64) {
          const double f = fv[j], s = sv[j];
          const bool fa = f > t1, sa = s > t2;
          f_any |= fa; s_any |= sa;
          if (f >= t1) c4[0] += f;
          if (s >= t2) c4[1] += s;
          if (fa && sa) { c4[2] += f; c4[3] += s; c_any = 1; }
        }
        wave_sum_all<4>(c4);
        const int FA = wave_max(f_any), SA = wave_max(s_any), CA = wave_max(c_any);
        if (lane == 0) {
          const double d1 = FA ? c4[0] : 0.0, d2 = SA ? c4[1] : 0.0;
          out[col_costes] = CA ? c4[2] / d1 : 0.0;
          out[col_costes + 1] = CA ? c4[3] / d2 : 0.0;
        }
      }
    }
  }
}


// -----------------------------------------------------------------------------------------------------
// dense per-object ranks of one channel: rank(p) = number of distinct values of the object smaller than
// the value at p (CellProfiler's Rank_im: lexsort + cumsum of "value changed").  One sort per
// (object, channel), written into a tile-shaped plane so that every channel pair reuses it.
// -----------------------------------------------------------------------------------------------------
struct RankArgs {
  const u16* labels;
  const void* planes;
  int F, C, Y, X, channel;
  const aliby_object* tab;
  int n_obj, cap;
  unsigned char* gscratch;
  unsigned int* ranks;  // [F,C,Y,X]
  int* rmax;            // [n_obj, C]
};

template <typename T, bool GLOBAL>
__global__ __launch_bounds__(256) void k_ranks(RankArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ int wsum[4];
  __shared__ int part[256];
  unsigned char* ws = GLOBAL ? (a.gscratch + (size_t)blockIdx.x * (size_t)a.cap * 16) : lds_raw;
  float* vals = reinterpret_cast<float*>(ws);
  float* S = vals + a.cap;
  int* P = reinterpret_cast<int*>(S + a.cap);
  int* pix = P + a.cap;
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;
  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    if (o.area <= 0) { if (tid == 0) a.rmax[(size_t)oi * a.C + a.channel] = 0; continue; }
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const T* px = reinterpret_cast<const T*>(a.planes) + ((size_t)o.tile * a.C + a.channel) * plane;
    unsigned int* rk = a.ranks + ((size_t)o.tile * a.C + a.channel) * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0, npix = h * w;
    const u16 L = (u16)o.label;
    __syncthreads();
    int base = 0;
    for (int i0 = 0; i0 < npix; i0 += blockDim.x) {
      const int i = i0 + tid;
      bool in = false;
      size_t idx = 0;
      if (i < npix) { idx = (size_t)(o.y0 + i / w) * a.X + (o.x0 + i % w); in = lab[idx] == L; }
      const int pos = block_compact_slot(in, base, wsum);
      if (in) { vals[pos] = px_load<T>(px, idx); pix[pos] = (int)idx; }
    }
    __syncthreads();
    const int N = base;
    // ---- uint16 fast path: presence bitmap over [vmin, vmax] + prefix popcount instead of a sort.  S and P are
    // reused as the bitmap (32 values per word) and its scan; taken when the value span fits them.
    bool done = false;
    if (sizeof(T) == 2 && !GLOBAL) {
      int lo = INT_MAX, hi = 0;
      for (int j = tid; j < N; j += blockDim.x) { const int v = (int)vals[j]; lo = min(lo, v); hi = max(hi, v); }
      const int vmin = block_min_i32(lo, part), vmax = block_max_i32(hi, part);
      const int w0 = vmin >> 5, nw = (vmax >> 5) - w0 + 1;
      if (nw <= a.cap) {
        unsigned int* bits = reinterpret_cast<unsigned int*>(S);
        __syncthreads();
        for (int i = tid; i < nw; i += blockDim.x) bits[i] = 0u;
        __syncthreads();
        for (int j = tid; j < N; j += blockDim.x) { const int v = (int)vals[j]; atomicOr(&bits[(v >> 5) - w0], 1u << (v & 31)); }
        __syncthreads();
        for (int i = tid; i < nw; i += blockDim.x) P[i] = __popc(bits[i]);
        __syncthreads();
        block_inclusive_scan(P, nw, part);
        for (int j = tid; j < N; j += blockDim.x) {
          const int v = (int)vals[j], wi = (v >> 5) - w0;
          const unsigned int below = bits[wi] & ((1u << (v & 31)) - 1u);
          rk[pix[j]] = (unsigned int)(P[wi] - __popc(bits[wi]) + __popc(below));
        }
        if (tid == 0) a.rmax[(size_t)oi * a.C + a.channel] = P[nw - 1] - 1;
        done = true;
      }
    }
    if (done) { __syncthreads(); continue; }
    const int n2 = next_pow2(N);
    for (int i = tid; i < n2; i += blockDim.x) S[i] = (i < N) ? vals[i] : INFINITY;
    block_bitonic_sort(S, n2);
    for (int i = tid; i < N; i += blockDim.x) P[i] = (i > 0 && S[i] != S[i - 1]) ? 1 : 0;
    __syncthreads();
    block_inclusive_scan(P, N, part);
    for (int j = tid; j < N; j += blockDim.x) {
      const float v = vals[j];
      int lo = 0, hi = N;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (S[mid] < v) lo = mid + 1; else hi = mid; }
      rk[pix[j]] = (unsigned int)P[lo];
    }
    if (tid == 0) a.rmax[(size_t)oi * a.C + a.channel] = P[N - 1];
    __syncthreads();
  }
}

extern "C" int aliby_object_ranks(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C,
                                  int Y, int X, int channel, const aliby_object* table_dev, int n_obj, int max_area,
                                  uint32_t* ranks_dev, int32_t* rmax_dev, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && planes && table_dev && ranks_dev && rmax_dev, "NULL argument");
  ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32, "dtype must be ALIBY_U16 or ALIBY_F32");
  ARG_CHECK(channel >= 0 && channel < C, "channel out of range");
  ARG_CHECK((size_t)Y * X < (size_t)INT_MAX, "plane too large");
  RankArgs a;
  a.labels = labels; a.planes = planes; a.F = F; a.C = C; a.Y = Y; a.X = X; a.channel = channel; a.tab = table_dev;
  a.n_obj = n_obj; a.ranks = ranks_dev; a.rmax = rmax_dev;
  int cap = 64;
  while (cap < max_area) cap <<= 1;
  a.cap = cap;
  hipStream_t s = as_stream(stream);
  const size_t need = (size_t)cap * 16;
  if (need <= 96 * 1024) {
    a.gscratch = nullptr;
    dim3 grid(n_obj), block(aliby_pick_block(max_area));
    if (dtype == ALIBY_U16) {
      if (need > 32 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_ranks<u16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      hipLaunchKernelGGL((k_ranks<u16, false>), grid, block, need, s, a);
    } else {
      if (need > 32 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_ranks<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      hipLaunchKernelGGL((k_ranks<float, false>), grid, block, need, s, a);
    }
  } else {
    const int g = n_obj < 512 ? n_obj : 512;
    int rc = aliby_ensure_scratch(ctx, (size_t)g * need);
    if (rc) return rc;
    a.gscratch = (unsigned char*)ctx->scratch;
    dim3 grid(g), block(256);
    if (dtype == ALIBY_U16) hipLaunchKernelGGL((k_ranks<u16, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_ranks<float, true>), grid, block, 0, s, a);
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}

extern "C" int aliby_features_coloc(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype,
                                    int F, int C, int Y, int X, int ch0, int ch1,
                                    const aliby_object* table_dev, int n_obj, int max_area,
                                    double* out, int ld, int col_pearson, int col_manders, int col_rwc,
                                    int col_costes, double thr_percent, double costes_scale_max,
                                    const uint32_t* ranks_dev, const int32_t* rmax_dev, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && planes && table_dev && out, "NULL argument");
  ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32, "dtype must be ALIBY_U16 or ALIBY_F32");
  ARG_CHECK(ch0 >= 0 && ch0 < C && ch1 >= 0 && ch1 < C, "channel out of range");
  ARG_CHECK(F > 0 && Y > 0 && X > 0 && max_area >= 0, "bad shape");
  const int cols[4] = {col_pearson, col_manders, col_rwc, col_costes};
  for (int k = 0; k < 4; ++k) ARG_CHECK(cols[k] < 0 || cols[k] + 2 <= ld, "columns exceed row stride");
  ColocArgs a;
  a.labels = labels; a.planes = planes; a.F = F; a.C = C; a.Y = Y; a.X = X; a.ch0 = ch0; a.ch1 = ch1;
  a.tab = table_dev; a.n_obj = n_obj; a.out = out; a.ld = ld;
  a.col_pearson = col_pearson; a.col_manders = col_manders; a.col_rwc = col_rwc; a.col_costes = col_costes;
  a.thr = thr_percent; a.scale_max = costes_scale_max;
  a.ranks = ranks_dev; a.rmax = rmax_dev;
  ARG_CHECK(col_rwc < 0 || (ranks_dev && rmax_dev), "rwc needs the rank planes (aliby_object_ranks)");
  int cap = 64;
  while (cap < max_area) cap <<= 1;
  a.cap = cap;
  hipStream_t s = as_stream(stream);
  const size_t need = (size_t)cap * 16;
  if (need <= 96 * 1024) {
    a.gscratch = nullptr;
    dim3 grid(n_obj), block(aliby_pick_block(max_area));
    if (dtype == ALIBY_U16) {
      if (need > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_coloc<u16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      hipLaunchKernelGGL((k_coloc<u16, false>), grid, block, need, s, a);
    } else {
      if (need > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_coloc<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      hipLaunchKernelGGL((k_coloc<float, false>), grid, block, need, s, a);
    }
  } else {
    const int g = n_obj < 512 ? n_obj : 512;
    int rc = aliby_ensure_scratch(ctx, (size_t)g * need);
    if (rc) return rc;
    a.gscratch = (unsigned char*)ctx->scratch;
    dim3 grid(g), block(256);
    if (dtype == ALIBY_U16) hipLaunchKernelGGL((k_coloc<u16, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_coloc<float, true>), grid, block, 0, s, a);
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}

// pairs_host: n_pairs x 6 ints (ch0, ch1, col_pearson, col_manders, col_rwc, col_costes; a column of -1 skips the metric).
// Returns ALIBY_ERR_TOO_LARGE without launching when the objects' pixel lists of all channels do not fit the LDS budget
// (the caller then uses aliby_features_coloc pair by pair).
extern "C" int aliby_features_coloc_pairs(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C,
                                          int Y, int X, const int32_t* pairs_host, int n_pairs,
                                          const aliby_object* table_dev, int n_obj, int max_area, double* out, int ld,
                                          double thr_percent, double costes_scale_max, const uint32_t* ranks_dev,
                                          const int32_t* rmax_dev, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0 || n_pairs == 0) return ALIBY_OK;
  ARG_CHECK(labels && planes && table_dev && out && pairs_host, "NULL argument");
  ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32, "dtype must be ALIBY_U16 or ALIBY_F32");
  ARG_CHECK(F > 0 && Y > 0 && X > 0 && max_area >= 0, "bad shape");
  ARG_CHECK(n_pairs > 0 && n_pairs <= COLOC_MAX_PAIRS, "too many channel pairs for one launch");
  PairsArgs a;
  a.labels = labels; a.planes = planes; a.F = F; a.C = C; a.Y = Y; a.X = X; a.tab = table_dev; a.n_obj = n_obj;
  a.out = out; a.ld = ld; a.thr = thr_percent; a.scale_max = costes_scale_max; a.ranks = ranks_dev; a.rmax = rmax_dev;
  a.nch = 0; a.npairs = n_pairs; a.any_rwc = 0;
  for (int p = 0; p < n_pairs; ++p) {
    const int32_t* q = pairs_host + 6 * p;
    for (int k = 0; k < 2; ++k) {
      ARG_CHECK(q[k] >= 0 && q[k] < C, "channel out of range");
      int loc = -1;
      for (int c = 0; c < a.nch; ++c) if (a.chan[c] == q[k]) loc = c;
      if (loc < 0) {
        ARG_CHECK(a.nch < COLOC_MAX_CH, "too many distinct channels for one launch");
        loc = a.nch;
        a.chan[a.nch++] = q[k];
      }
      a.pair[p][k] = loc;
    }
    for (int k = 2; k < 6; ++k) {
      ARG_CHECK(q[k] < 0 || q[k] + 2 <= ld, "columns exceed row stride");
      a.pair[p][k] = q[k];
    }
    if (q[4] >= 0) a.any_rwc = 1;
  }
  ARG_CHECK(!a.any_rwc || (ranks_dev && rmax_dev), "rwc needs the rank planes (aliby_object_ranks)");
  int cap = 64;
  while (cap < max_area) cap <<= 1;
  a.cap = cap;
  const size_t need = (size_t)a.nch * cap * 4 * (a.any_rwc ? 2 : 1);
  if (need > 144 * 1024) {
    aliby_set_error("coloc_pairs: %zu bytes of pixel lists per object exceed the LDS budget", need);
    return ALIBY_ERR_TOO_LARGE;
  }
  hipStream_t s = as_stream(stream);
  dim3 grid(n_obj), block(256);
  if (dtype == ALIBY_U16) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_coloc_pairs<u16>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    hipLaunchKernelGGL((k_coloc_pairs<u16>), grid, block, need, s, a);
  } else {
    HIP_TRY(hipFuncSetAttribute((const void*)k_coloc_pairs<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    hipLaunchKernelGGL((k_coloc_pairs<float>), grid, block, need, s, a);
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}
