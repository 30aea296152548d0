// feat_cell.hip — the reference's in-repo per-cell metrics, one workgroup per object.
//
// Reference: src/extraction/core/functions/cell.py:18-303 (loaded into CELL_FUNS by
// load_cellfuns_core, loaders.py:19-25; one-argument functions wrapped by ignore_pixels, 170-171).
// The only part of the path with live numeric tests in the reference (tests/extraction/test_volume.py);
// the CPU oracle for them is pinned against the reference module itself (tests/golden).
//
// Columns written (CELL_NCOL = 17):
//   0 area                 np.sum(mask)                                         cell.py:18-27
//   1 centroid_x, 2 centroid_y   1-based mean column / row                      cell.py:282-303
//   3 conical_volume       4 * sum(EDT of the 1-padded mask)                    cell.py:175-186
//   4 eccentricity, 6 volume, 7 min_ax, 8 maj_ax   min_maj_approximation       cell.py:30-40,159-172,207-229
//   5 spherical_volume                                                          cell.py:189-204
//   9 mean, 10 median, 11 std, 12 total, 13 total_squared                       cell.py:43-99,148-156
//  14 max2p5pc, 15 max5px_median                                                cell.py:102-145
//  16 moment_of_inertia                                                         cell.py:232-265
// uint16 pixels follow NumPy integer semantics: `total` is exact, `total_squared` squares in uint16
// (wraps modulo 65536, as `trap_image[cell_mask] ** 2` does) before an exact sum.
#include "common.h"

typedef unsigned short u16;

#define CELL_NCOL 17

struct CellArgs {
  const u16* labels;
  const void* planes;  // may be NULL: only the mask-only metrics are written then
  int F, C, Y, X, channel;
  const aliby_object* tab;
  int n_obj;
  size_t cap_cells;  // >= (max_h+2)*(max_w+2)
  int cap_vals;      // power of two >= max area
  unsigned char* gscratch;
  double* out;
  int ld, col0;
};

template <typename T, bool GLOBAL>
__global__ __launch_bounds__(256) void k_cell(CellArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ double red_d[8];
  __shared__ long long red_l[8];
  __shared__ int red_i[8];
  __shared__ double vec[4 * 6];
  __shared__ int s_cnt;
  const size_t slab = a.cap_cells * 8 + (size_t)a.cap_vals * 4;
  unsigned char* ws = GLOBAL ? (a.gscratch + (size_t)blockIdx.x * slab) : lds_raw;
  int* g = reinterpret_cast<int*>(ws);                       // vertical distances, then top flags
  int* d2 = reinterpret_cast<int*>(ws + a.cap_cells * 4);    // squared EDT (padded grid), -1 = not object
  float* vals = reinterpret_cast<float*>(ws + a.cap_cells * 8);
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;
  const int BIG = 1 << 28;

  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    double* out = a.out + (size_t)oi * a.ld + a.col0;
    if (o.area <= 0) {
      // an all-False mask: area 0, the rest NaN (0/0) like the reference's arithmetic
      for (int k = tid; k < CELL_NCOL; k += blockDim.x) out[k] = (k == 0 || k == 3 || k == 5 || k == 12 || k == 13) ? 0.0 : NAN;
      continue;
    }
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const T* px = a.planes ? reinterpret_cast<const T*>(a.planes) + ((size_t)o.tile * a.C + a.channel) * plane : nullptr;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0, ph = h + 2, pw = w + 2;
    const u16 L = (u16)o.label;
    auto is_obj = [&](int r, int c) -> bool {  // padded coordinates
      return r >= 1 && r <= h && c >= 1 && c <= w && lab[(size_t)(o.y0 + r - 1) * a.X + o.x0 + c - 1] == L;
    };
    __syncthreads();
    // ---- EDT of the 1-padded mask (the pad ring and every other pixel are background) ----------------
    for (int c = tid; c < pw; c += blockDim.x) {
      int run = BIG;
      for (int r = 0; r < ph; ++r) { run = is_obj(r, c) ? (run >= BIG ? BIG : run + 1) : 0; g[r * pw + c] = run; }
      run = BIG;
      for (int r = ph - 1; r >= 0; --r) {
        const int cur = g[r * pw + c];
        run = (cur == 0) ? 0 : (run >= BIG ? BIG : run + 1);
        if (run < cur) g[r * pw + c] = run;
      }
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int dmax = 0;
    long long sxi = 0, syi = 0;
    double sdist = 0;
    for (int i = tid; i < ph * pw; i += blockDim.x) {
      const int r = i / pw, c = i % pw;
      int val = -1;
      const int g0 = g[i];
      if (g0 > 0) {
        long long best = (long long)g0 * g0;
        for (int dx = 1; (long long)dx * dx < best; ++dx) {
          if (c - dx >= 0) { const long long gg = g[r * pw + c - dx]; const long long v = gg * gg + (long long)dx * dx; if (v < best) best = v; }
          if (c + dx < pw) { const long long gg = g[r * pw + c + dx]; const long long v = gg * gg + (long long)dx * dx; if (v < best) best = v; }
        }
        val = (int)best;
        dmax = max(dmax, val);
        sdist += sqrt((double)val);
        sxi += (o.x0 + c - 1) + 1;  // 1-based full-frame coordinates
        syi += (o.y0 + r - 1) + 1;
      }
      d2[i] = val;
    }
    const int DMAX = block_max_i32(dmax, red_i);
    const double SD = block_sum_f64(sdist, red_d);
    const long long SX = block_sum_i64(sxi, red_l), SY = block_sum_i64(syi, red_l);
    __syncthreads();
    // ---- cone top = pixels at the maximal distance; dn = straight-line distance to the nearest top ------
    // g is reused as the list of top cells
    for (int i = tid; i < ph * pw; i += blockDim.x)
      if (d2[i] == DMAX) { const int k = atomicAdd(&s_cnt, 1); g[k] = i; }
    __syncthreads();
    const int NT = s_cnt;
    double dnmax = 0.0;
    for (int i = tid; i < ph * pw; i += blockDim.x) {
      if (d2[i] < 0) continue;
      const int r = i / pw, c = i % pw;
      long long best = LLONG_MAX;
      for (int k = 0; k < NT; ++k) {
        const int t = g[k];
        const long long dr = r - t / pw, dc = c - t % pw;
        const long long v = dr * dr + dc * dc;
        if (v < best) best = v;
      }
      dnmax = fmax(dnmax, sqrt((double)best));
    }
    const double DNMAX = block_max_f64(dnmax, red_d);
    // cone_top(t) = distance from a top pixel to the nearest object pixel that is not a top pixel
    double ctsum = 0.0;
    for (int k = tid; k < NT; k += blockDim.x) {
      const int t = g[k];
      const int tr = t / pw, tc = t % pw;
      long long best = LLONG_MAX;
      for (int i = 0; i < ph * pw; ++i) {
        if (d2[i] < 0 || d2[i] == DMAX) continue;
        const long long dr = tr - i / pw, dc = tc - i % pw;
        const long long v = dr * dr + dc * dc;
        if (v < best) best = v;
      }
      if (best != LLONG_MAX) {
        ctsum += sqrt((double)best);
      } else {
        // every pixel of the object is a top pixel (an object one or two pixels thick: all at the same distance from the edge).
        // The reference then asks scipy for distance_transform_edt(dn == 0) of a frame WITHOUT background (cell.py:223); scipy
        // answers with the distance to the virtual point one row above the first column of the (padded) frame,
        // sqrt((row + 1)^2 + col^2).  Not a geometric quantity, but it is what the reference's volume / eccentricity /
        // min_maj_approximation return for such objects, so it is what is returned here (oracle/cell_metrics.py calls scipy).
        const double yp = (double)(o.y0 + tr - 1) + 1.0, xp = (double)(o.x0 + tc - 1) + 1.0;  // padded full-frame coordinates
        ctsum += sqrt((yp + 1.0) * (yp + 1.0) + xp * xp);
      }
    }
    const double CTS = block_sum_f64(ctsum, red_d);
    const double area = (double)o.area;
    const double min_ax = rint(sqrt((double)DMAX));
    const double maj_ax = rint(DNMAX + CTS / 2.0);
    if (tid == 0) {
      out[0] = area;
      out[1] = (double)SX / area;
      out[2] = (double)SY / area;
      out[3] = 4.0 * SD;
      out[4] = sqrt(maj_ax * maj_ax - min_ax * min_ax) / maj_ax;
      const double rr = sqrt(area / M_PI);
      out[5] = (4.0 * M_PI * rr * rr * rr) / 3.0;
      out[6] = (4.0 * M_PI * min_ax * min_ax * maj_ax) / 3.0;
      out[7] = min_ax;
      out[8] = maj_ax;
    }
    if (!px) { __syncthreads(); continue; }

    // ---- pixel statistics ---------------------------------------------------------------------------------
    __syncthreads();
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    double sv = 0, swx = 0, swy = 0;
    long long sq_wrap = 0;
    double sq_f = 0;
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int r = i / w, c = i % w;
      const size_t idx = (size_t)(o.y0 + r) * a.X + o.x0 + c;
      if (lab[idx] != L) continue;
      const float v = px_load<T>(px, idx);
      sv += (double)v;
      swx += (double)v * (double)(c + 1);
      swy += (double)v * (double)(r + 1);
      if (sizeof(T) == 2) { const unsigned q = (unsigned)v; sq_wrap += (long long)((q * q) & 0xFFFFu); }
      else sq_f += (double)(v * v);
      vals[atomicAdd(&s_cnt, 1)] = v;
    }
    double s3[3] = {sv, swx, swy};
    block_sum_vec_all<3>(s3, vec);
    const long long SQW = block_sum_i64(sq_wrap, red_l);
    const double SQF = block_sum_f64(sq_f, red_d);
    const int N = o.area;
    const int n2 = next_pow2(N);
    for (int i = N + tid; i < n2; i += blockDim.x) vals[i] = INFINITY;
    block_bitonic_sort(vals, n2);
    const double mean = s3[0] / area;
    double q[3] = {0, 0, 0};  // sum (v-mean)^2, mu20, mu02 about the intensity centroid
    const double Xm = s3[1] / s3[0], Ym = s3[2] / s3[0];
    for (int i = tid; i < N; i += blockDim.x) { const double d = (double)vals[i] - mean; q[0] += d * d; }
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int r = i / w, c = i % w;
      const size_t idx = (size_t)(o.y0 + r) * a.X + o.x0 + c;
      if (lab[idx] != L) continue;
      const double v = (double)px_load<T>(px, idx);
      q[1] += v * ((double)(c + 1) - Xm) * ((double)(c + 1) - Xm);
      q[2] += v * ((double)(r + 1) - Ym) * ((double)(r + 1) - Ym);
    }
    block_sum_vec_all<3>(q, vec);
    // top-k means
    const int n_top = (int)ceil(area * 0.025);
    double t2[2] = {0, 0};
    for (int i = tid; i < N; i += blockDim.x) {
      if (i >= N - n_top) t2[0] += (double)vals[i];
      if (i >= N - 5) t2[1] += (double)vals[i];
    }
    block_sum_vec_all<2>(t2, vec);
    if (tid == 0) {
      const double med = (N & 1) ? (double)vals[N / 2] : 0.5 * ((double)vals[N / 2 - 1] + (double)vals[N / 2]);
      out[9] = mean;
      out[10] = med;
      out[11] = sqrt(q[0] / area);
      out[12] = s3[0];
      out[13] = (sizeof(T) == 2) ? (double)SQW : SQF;
      out[14] = t2[0] / (double)n_top;
      out[15] = (N > 5 && med != 0.0) ? (t2[1] / 5.0) / med : NAN;
      out[16] = (s3[0] != 0.0) ? q[1] / (s3[0] * s3[0]) + q[2] / (s3[0] * s3[0]) : NAN;
    }
    __syncthreads();
  }
}

extern "C" int aliby_features_cell(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F,
                                   int C, int Y, int X, int channel, const aliby_object* table_dev, int n_obj,
                                   int max_h, int max_w, int max_area, double* out, int ld, int col0,
                                   void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && table_dev && out, "NULL argument");
  ARG_CHECK(F > 0 && Y > 0 && X > 0 && max_h >= 0 && max_w >= 0 && max_area >= 0, "bad shape");
  ARG_CHECK(col0 >= 0 && col0 + CELL_NCOL <= ld, "columns exceed row stride");
  if (planes) {
    ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32, "dtype must be ALIBY_U16 or ALIBY_F32");
    ARG_CHECK(channel >= 0 && channel < C, "channel out of range");
  }
  CellArgs a;
  a.labels = labels; a.planes = planes; a.F = F; a.C = C; a.Y = Y; a.X = X; a.channel = channel;
  a.tab = table_dev; a.n_obj = n_obj; a.out = out; a.ld = ld; a.col0 = col0;
  a.cap_cells = ((size_t)(max_h + 2) * (max_w + 2) + 3) & ~(size_t)3;
  int cv = 64;
  while (cv < max_area) cv <<= 1;
  a.cap_vals = cv;
  const size_t need = a.cap_cells * 8 + (size_t)cv * 4;
  hipStream_t s = as_stream(stream);
  const bool f32 = planes && dtype == ALIBY_F32;
  if (need <= 128 * 1024) {
    a.gscratch = nullptr;
    if (f32) {
      if (need > 32 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_cell<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      hipLaunchKernelGGL((k_cell<float, false>), dim3(n_obj), dim3(aliby_pick_block((long long)max_h * max_w)), need, s, a);
    } else {
      if (need > 32 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_cell<u16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      hipLaunchKernelGGL((k_cell<u16, false>), dim3(n_obj), dim3(aliby_pick_block((long long)max_h * max_w)), need, s, a);
    }
  } else {
    const int gsz = n_obj < 512 ? n_obj : 512;
    int rc = aliby_ensure_scratch(ctx, (size_t)gsz * need);
    if (rc) return rc;
    a.gscratch = (unsigned char*)ctx->scratch;
    if (f32) hipLaunchKernelGGL((k_cell<float, true>), dim3(gsz), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_cell<u16, true>), dim3(gsz), dim3(256), 0, s, a);
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}
