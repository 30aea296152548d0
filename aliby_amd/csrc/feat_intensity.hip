// feat_intensity.hip — cp_measure "intensity" family, one workgroup per object.
//
// Reference call site: wrap_cp_measure_features (extraction/core/functions/loaders.py:135-150)
// with fun = cp_measure.bulk.get_core_measurements()["intensity"] (loaders.py:71-73), evaluated by
// the reference once per (object x instruction) on a full-frame binary mask
// (extraction/extract.py:283-288,351-359).  cp_measure 0.1.17 is not vendored in the reference
// (uv.lock:441-442); the arithmetic below restates CellProfiler's published
// MeasureObjectIntensity definitions that cp_measure ports:
//   Integrated/Mean/Std/Min/Max (+ the same five on inner-boundary pixels,
//   skimage find_boundaries(mode="inner", connectivity=1)), MassDisplacement,
//   Lower/Median/Upper quartile with linear interpolation at index area*q, MAD,
//   CenterMassIntensity_{X,Y,Z}, MaxIntensity_{X,Y,Z}.
//
// Kernel shape: HBM/L2-bound gather of each object's bbox (labels + one pixel plane), values staged
// in LDS, bitonic sort in LDS for the order statistics, fp64 accumulation with a fixed
// thread->pixel assignment so results are run-to-run deterministic.
#include "common.h"

typedef unsigned short u16;

#define INT_NCOL 21

struct IntensityArgs {
  const u16* labels;
  const void* planes;
  int F, C, Y, X, channel;
  const aliby_object* tab;
  int n_obj;
  int cap;          // power of two >= max area
  float* gscratch;  // global fallback (cap floats per workgroup) or NULL -> LDS
  int edge;
  double* out;
  int ld, col0;
};

template <typename T, bool GLOBAL>
__global__ __launch_bounds__(256) void k_intensity(IntensityArgs a) {
  extern __shared__ __align__(16) float lds_vals[];
  __shared__ double red_d[8];
  __shared__ long long red_l[8];
  __shared__ float red_f[8];
  __shared__ int red_i[8];
  __shared__ int s_cnt;
  __shared__ double s_res[5];  // lq, med, uq, mad(k), mad(k+1)

  float* vals = GLOBAL ? (a.gscratch + (size_t)blockIdx.x * (a.cap + a.cap / 32 + 1)) : lds_vals;
  unsigned int* eflag = reinterpret_cast<unsigned int*>(vals + a.cap);  // edge bit of the staged value at the same position
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;

  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    double* out = a.out + (size_t)oi * a.ld + a.col0;
    const int ncol = a.edge ? INT_NCOL : INT_NCOL - 5;
    if (o.area <= 0) {
      for (int k = tid; k < ncol; k += blockDim.x) out[k] = NAN;
      continue;
    }
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const T* px = reinterpret_cast<const T*>(a.planes) + ((size_t)o.tile * a.C + a.channel) * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0;
    const int npix = h * w;
    const u16 L = (u16)o.label;

    if (tid == 0) s_cnt = 0;
    if (a.edge)
      for (int i = tid; i < (a.cap + 31) / 32; i += blockDim.x) eflag[i] = 0u;
    __syncthreads();

    // ---- pass 1: accumulate + stage values --------------------------------
    // The bbox walk is latency-bound (one wave per ~400-pixel object): every load of a batch of GU pixels — the label, the
    // pixel and the four neighbour labels of the edge test — is issued before the first is used, so a lane waits for HBM / L2
    // once per batch instead of up to six times per pixel.  Addresses inside the (clamped) bbox are always valid.
    int n = 0, ne = 0;
    double sv = 0, sxv = 0, syv = 0, sve = 0;
    long long sx = 0, sy = 0;
    float vmin = INFINITY, vmax = -INFINITY, emin = INFINITY, emax = -INFINITY;
    int amax = -1;  // raveled index of the max (ties -> largest index)
    constexpr int GU = 4;
    const bool single = blockDim.x <= WAVE;  // (then the staging order, and with it every sum below, is fixed)
    int wbase = 0;
    for (int i0 = tid; i0 < npix; i0 += GU * blockDim.x) {
      int idx[GU];
      u16 lb[GU], nb[GU][4];
      float vv[GU];
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int i = i0 + u * blockDim.x;
        const bool in = i < npix;
        const int ii = in ? i : 0;
        const int yy = o.y0 + ii / w, xx = o.x0 + ii % w;
        idx[u] = yy * a.X + xx;
        lb[u] = lab[idx[u]];
        if (!in) lb[u] = (u16)(L ^ 1);  // anything but L
        vv[u] = px_load<T>(px, (size_t)idx[u]);
        if (a.edge) {
          // inner boundary, 4-neighbourhood, image border replicated (skimage grey erosion/dilation default mode='reflect')
          const int yu = max(yy - 1, 0), yd = min(yy + 1, a.Y - 1);
          const int xl = max(xx - 1, 0), xr = min(xx + 1, a.X - 1);
          nb[u][0] = lab[(size_t)yu * a.X + xx];
          nb[u][1] = lab[(size_t)yd * a.X + xx];
          nb[u][2] = lab[(size_t)yy * a.X + xl];
          nb[u][3] = lab[(size_t)yy * a.X + xr];
        }
      }
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const bool hit = lb[u] == L;
        // slot of the staged value: a single-wave workgroup compacts by ballot (lane order: deterministic, no atomics)
        int pos = 0;
        if (single) {
          const unsigned long long m = __ballot(hit);
          pos = wbase + __popcll(m & ((1ull << (tid & 63)) - 1ull));
          wbase += __popcll(m);
        }
        if (!hit) continue;
        if (!single) pos = atomicAdd(&s_cnt, 1);
        const float v = vv[u];
        const int yy = idx[u] / a.X, xx = idx[u] - yy * a.X;
        ++n;
        sv += (double)v;
        sx += xx;
        sy += yy;
        sxv += (double)xx * (double)v;
        syv += (double)yy * (double)v;
        vmin = fminf(vmin, v);
        if (v > vmax || (v == vmax && idx[u] > amax)) { vmax = v; amax = idx[u]; }
        vals[pos] = v;
        if (a.edge) {
          const bool e = nb[u][0] != L || nb[u][1] != L || nb[u][2] != L || nb[u][3] != L;
          if (e) {
            ++ne;
            sve += (double)v;
            emin = fminf(emin, v);
            emax = fmaxf(emax, v);
            if (single) atomicOr(&eflag[pos >> 5], 1u << (pos & 31));
          }
        }
      }
    }
    const int N = block_sum_i32(n, red_i);
    const double SV = block_sum_f64(sv, red_d);
    const long long SX = block_sum_i64(sx, red_l);
    const long long SY = block_sum_i64(sy, red_l);
    const double SXV = block_sum_f64(sxv, red_d);
    const double SYV = block_sum_f64(syv, red_d);
    const float VMIN = block_min_f32(vmin, red_f);
    const float VMAX = block_max_f32(vmax, red_f);
    // position of the maximum: largest raveled index among the pixels equal to VMAX
    const int AMAX = block_max_i32((vmax == VMAX) ? amax : -1, red_i);
    const double mean = SV / (double)N;

    int NE = 0;
    double SVE = 0, mean_e = 0;
    float EMIN = 0, EMAX = 0;
    if (a.edge) {
      NE = block_sum_i32(ne, red_i);
      SVE = block_sum_f64(sve, red_d);
      EMIN = block_min_f32(emin, red_f);
      EMAX = block_max_f32(emax, red_f);
      mean_e = SVE / (double)NE;
    }

    // ---- edge second moment -----------------------------------------------------------------------------------
    // single wave: over the staged (still unsorted) values whose edge bit is set; several waves stage in atomic order, which
    // varies from run to run, so they walk the bbox again with a fixed pixel -> thread assignment
    double SSE = 0;
    if (a.edge) {
      double sse = 0;
      if (single) {
        for (int i = tid; i < N; i += blockDim.x)
          if ((eflag[i >> 5] >> (i & 31)) & 1u) {
            const double d = (double)vals[i] - mean_e;
            sse += d * d;
          }
      } else {
        for (int i = tid; i < npix; i += blockDim.x) {
          const int yy = o.y0 + i / w, xx = o.x0 + i % w;
          const size_t idx = (size_t)yy * a.X + xx;
          if (lab[idx] != L) continue;
          const int yu = max(yy - 1, 0), yd = min(yy + 1, a.Y - 1);
          const int xl = max(xx - 1, 0), xr = min(xx + 1, a.X - 1);
          const bool e = lab[(size_t)yu * a.X + xx] != L || lab[(size_t)yd * a.X + xx] != L ||
                         lab[(size_t)yy * a.X + xl] != L || lab[(size_t)yy * a.X + xr] != L;
          if (e) {
            const double d = (double)px_load<T>(px, idx) - mean_e;
            sse += d * d;
          }
        }
      }
      SSE = block_sum_f64(sse, red_d);
    }

    // ---- sort staged values (pad with +inf) --------------------------------
    const int n2 = next_pow2(N);
    for (int i = N + tid; i < n2; i += blockDim.x) vals[i] = INFINITY;
    block_bitonic_sort(vals, n2);

    // ---- pass 2: central second moment over the sorted list ----------------
    double ss = 0;
    for (int i = tid; i < N; i += blockDim.x) {
      const double d = (double)vals[i] - mean;
      ss += d * d;
    }
    const double SS = block_sum_f64(ss, red_d);

    // ---- quartiles ----------------------------------------------------------
    if (tid < 3) {
      const double frac = (tid == 0) ? 0.25 : (tid == 1 ? 0.5 : 0.75);
      const double qidx = (double)N * frac;
      const int qi = (int)qidx;
      const double qf = qidx - floor(qidx);
      double r;
      if (qi < N - 1) r = (double)vals[qi] * (1.0 - qf) + (double)vals[qi + 1] * qf;
      else r = (double)vals[qi];
      s_res[tid] = r;
    }
    __syncthreads();
    const double med = s_res[1];

    // ---- MAD: k-th smallest of |v - med| without a second sort ---------------
    // d_i = |s_i - med| falls and then rises along the sorted list, so the j + 1 values with the smallest deviations are a
    // WINDOW [l, l + j] of it, and the j-th smallest deviation (0-based) is the smallest, over all windows of j + 1
    // consecutive values, of the larger end deviation: every window holds at least one of the values whose deviation is
    // >= the answer, and the values whose deviation is <= the answer are consecutive and at least j + 1.  Two LDS reads per
    // window and one reduction, instead of four binary searches per value.
    const double qidx = (double)N * 0.5;
    const int qi = (int)qidx;
    const double qf = qidx - floor(qidx);
    const bool interp = qi < N - 1;
    double best0 = INFINITY, best1 = INFINITY;
    for (int l = tid; l + qi < N; l += blockDim.x) {
      const double dl = fabs((double)vals[l] - med);
      best0 = fmin(best0, fmax(dl, fabs((double)vals[l + qi] - med)));
      if (interp && l + qi + 1 < N) best1 = fmin(best1, fmax(dl, fabs((double)vals[l + qi + 1] - med)));
    }
    const double D0 = -block_max_f64(-best0, red_d);
    const double D1 = interp ? -block_max_f64(-best1, red_d) : NAN;
    if (tid == 0) { s_res[3] = D0; s_res[4] = D1; }
    __syncthreads();

    if (tid == 0) {
      const double mad = interp ? (s_res[3] * (1.0 - qf) + s_res[4] * qf) : s_res[3];
      const double cm_x = (double)SX / (double)N, cm_y = (double)SY / (double)N;
      const double cmi_x = SXV / SV, cmi_y = SYV / SV;
      const double dx = cm_x - cmi_x, dy = cm_y - cmi_y;
      int k = 0;
      out[k++] = SV;
      out[k++] = mean;
      out[k++] = sqrt(SS / (double)N);
      out[k++] = (double)VMIN;
      out[k++] = (double)VMAX;
      if (a.edge) {
        if (NE > 0) {
          out[k++] = SVE;
          out[k++] = mean_e;
          out[k++] = sqrt(SSE / (double)NE);
          out[k++] = (double)EMIN;
          out[k++] = (double)EMAX;
        } else {
          for (int z = 0; z < 5; ++z) out[k++] = 0.0;
        }
      }
      out[k++] = sqrt(dx * dx + dy * dy);
      out[k++] = s_res[0];
      out[k++] = med;
      out[k++] = mad;
      out[k++] = s_res[2];
      out[k++] = cmi_x;
      out[k++] = cmi_y;
      out[k++] = 0.0;
      out[k++] = (double)(AMAX % a.X);
      out[k++] = (double)(AMAX / a.X);
      out[k++] = 0.0;
    }
    __syncthreads();
  }
}

extern "C" int aliby_features_intensity(aliby_ctx* ctx, const uint16_t* labels, const void* planes,
                                        int dtype, int F, int C, int Y, int X, int channel,
                                        const aliby_object* table_dev, int n_obj, int max_area,
                                        int edge_measurements, double* out, int ld, int col0,
                                        void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && planes && table_dev && out, "NULL argument");
  ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32, "dtype must be ALIBY_U16 or ALIBY_F32");
  ARG_CHECK(channel >= 0 && channel < C, "channel out of range");
  ARG_CHECK(F > 0 && Y > 0 && X > 0 && n_obj > 0 && max_area >= 0, "bad shape");
  ARG_CHECK((size_t)Y * X < (size_t)INT_MAX, "plane too large for 32-bit raveled index");
  const int ncol = edge_measurements ? INT_NCOL : INT_NCOL - 5;
  ARG_CHECK(col0 >= 0 && col0 + ncol <= ld, "columns exceed row stride");

  IntensityArgs a;
  a.labels = labels; a.planes = planes; a.F = F; a.C = C; a.Y = Y; a.X = X; a.channel = channel;
  a.tab = table_dev; a.n_obj = n_obj; a.edge = edge_measurements ? 1 : 0;
  a.out = out; a.ld = ld; a.col0 = col0;
  int cap = 64;
  while (cap < max_area) cap <<= 1;
  a.cap = cap;
  hipStream_t s = as_stream(stream);
  const size_t lds_need = ((size_t)cap + cap / 32 + 1) * sizeof(float);  // values + one edge bit each
  const size_t lds_cap = 128 * 1024;
  if (lds_need <= lds_cap) {
    a.gscratch = nullptr;
    dim3 grid(n_obj), block(aliby_pick_block(max_area));
    if (dtype == ALIBY_U16) {
      if (lds_need > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_intensity<u16, false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
      hipLaunchKernelGGL((k_intensity<u16, false>), grid, block, lds_need, s, a);
    } else {
      if (lds_need > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_intensity<float, false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
      hipLaunchKernelGGL((k_intensity<float, false>), grid, block, lds_need, s, a);
    }
  } else {
    int g = n_obj < 512 ? n_obj : 512;
    int rc = aliby_ensure_scratch(ctx, (size_t)g * (cap + cap / 32 + 1) * sizeof(float));
    if (rc) return rc;
    a.gscratch = (float*)ctx->scratch;
    dim3 grid(g), block(256);
    if (dtype == ALIBY_U16) hipLaunchKernelGGL((k_intensity<u16, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_intensity<float, true>), grid, block, 0, s, a);
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}
