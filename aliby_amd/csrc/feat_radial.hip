// feat_radial.hip — cp_measure "radial_distribution" (FracAtD / MeanFrac / RadialCV per ring).
//
// Reference call site: wrap_cp_measure_features (extraction/core/functions/loaders.py:135-150) with
// fun = get_core_measurements()["radial_distribution"] (default feature list, pipe_builder.py:49-56).
// cp_measure 0.1.17 / centrosome 1.3.3 are not vendored; restated from CellProfiler's
// MeasureObjectIntensityDistribution (centre = the object itself, scaled rings):
//   d_to_edge     exact Euclidean distance to the nearest non-object pixel (image border is not
//                 background) — centrosome.cpmorphology.distance_to_edge;
//   centre        object pixel of maximal d_to_edge (ties: last in raster order);
//   d_from_centre centrosome.propagate.propagate(zeros, centre, mask, 1): shortest 8-connected path
//                 inside the object, step cost sqrt(1/2) (edge) or 1 (diagonal);
//   ring          int(bin_count * d_from / (d_from + d_to_edge + 0.001));
//   wedge         (i>ic) + 2 (j>jc) + 4 (|i-ic|>|j-jc|).
//
// Two kernels: the geometry (EDT + centre + geodesic relaxation, all in LDS, one workgroup per object)
// is channel independent and writes one code byte per object pixel into a tile-shaped map
// (0x80 | wedge<<4 | ring); the statistics kernel then streams labels + map + one pixel plane per
// (object, channel).  Path lengths are kept as exact (edge steps, diagonal steps) integer pairs packed
// in one 32-bit LDS word, so the relaxation is race-free and bit-reproducible.
#include "common.h"

typedef unsigned short u16;

#define RD_MAXBINS 16
#define RD_INF 0xFFFFu

struct RadialGeoArgs {
  const u16* labels;
  int F, Y, X;
  const aliby_object* tab;
  int n_obj;
  size_t cap_cells;  // >= (max_h+2)*(max_w+2)
  unsigned char* gscratch;
  int bin_count;
  double max_radius;  // > 0: unscaled rings of max_radius / bin_count pixels (+ overflow ring); <= 0: scaled rings
  unsigned char* binmap;  // [F,Y,X]
};

__device__ __forceinline__ double path_cost(unsigned int w) {
  return (double)(w & 0xFFFFu) * 0.70710678118654752440 + (double)(w >> 16) * 1.0;
}

template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_radial_geometry(RadialGeoArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ int red_i[8];
  __shared__ int s_changed;
  unsigned char* ws = GLOBAL ? (a.gscratch + (size_t)blockIdx.x * a.cap_cells * 8) : lds_raw;
  int* d2 = reinterpret_cast<int*>(ws);                                   // squared distance to edge (padded grid)
  unsigned int* aux = reinterpret_cast<unsigned int*>(ws + a.cap_cells * 4);  // g (phase 1) then packed path
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;
  const int BIG = 1 << 28;

  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    if (o.area <= 0) continue;
    const u16* lab = a.labels + (size_t)o.tile * plane;
    unsigned char* bm = a.binmap + (size_t)o.tile * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0;
    const int ph = h + 2, pw = w + 2;
    const u16 L = (u16)o.label;
    // cell state in padded coordinates (r,c) <-> image (o.y0+r-1, o.x0+c-1)
    auto state = [&](int r, int c) -> int {  // 0 = background, 1 = object, 2 = outside the image
      const int yy = o.y0 + r - 1, xx = o.x0 + c - 1;
      if (yy < 0 || yy >= a.Y || xx < 0 || xx >= a.X) return 2;
      return lab[(size_t)yy * a.X + xx] == L ? 1 : 0;
    };
    __syncthreads();
    for (int i = tid; i < ph * pw; i += blockDim.x) d2[i] = -1;  // ring + non-object cells
    // ---- EDT phase 1: vertical distance to the nearest background cell of the column -------------
    int* g = reinterpret_cast<int*>(aux);
    for (int c = tid; c < pw; c += blockDim.x) {
      int run = BIG;
      for (int r = 0; r < ph; ++r) {
        const int st = state(r, c);
        run = (st == 0) ? 0 : (run >= BIG ? BIG : run + 1);
        g[r * pw + c] = run;
      }
      run = BIG;
      for (int r = ph - 1; r >= 0; --r) {
        const int cur = g[r * pw + c];
        run = (cur == 0) ? 0 : (run >= BIG ? BIG : run + 1);
        if (run < cur) g[r * pw + c] = run;
      }
    }
    __syncthreads();
    // ---- EDT phase 2 on object cells; track the maximum ------------------------------------------
    int best_d2 = -1, best_idx = -1;
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int r = i / w + 1, c = i % w + 1;
      int val = -1;
      if (state(r, c) == 1) {
        long long best = (long long)g[r * pw + c] * g[r * pw + c];
        for (int dx = 1; (long long)dx * dx < best && (c - dx >= 0 || c + dx < pw); ++dx) {
          if (c - dx >= 0) { const long long gg = g[r * pw + c - dx]; const long long v = gg * gg + (long long)dx * dx; if (v < best) best = v; }
          if (c + dx < pw) { const long long gg = g[r * pw + c + dx]; const long long v = gg * gg + (long long)dx * dx; if (v < best) best = v; }
        }
        val = best > 0x7fffffffLL ? 0x7fffffff : (int)best;
        if (val > best_d2 || (val == best_d2 && i > best_idx)) { best_d2 = val; best_idx = i; }
      }
      d2[r * pw + c] = val;  // -1 marks non-object cells
    }
    const int D2MAX = block_max_i32(best_d2, red_i);
    const int CIDX = block_max_i32(best_d2 == D2MAX ? best_idx : -1, red_i);  // last raster occurrence
    const int ci = CIDX / w, cj = CIDX % w;  // bbox-local centre
    __syncthreads();
    // ---- geodesic relaxation from the centre -------------------------------------------------------
    for (int i = tid; i < ph * pw; i += blockDim.x) aux[i] = 0xFFFFFFFFu;
    __syncthreads();
    if (tid == 0) aux[(ci + 1) * pw + cj + 1] = 0u;
    __syncthreads();
    for (int sweep = 0; sweep < 4 * (h + w) + 8; ++sweep) {
      if (tid == 0) s_changed = 0;
      __syncthreads();
      int changed = 0;
      for (int i = tid; i < h * w; i += blockDim.x) {
        const int r = i / w + 1, c = i % w + 1;
        if (d2[r * pw + c] < 0) continue;
        unsigned int cur = aux[r * pw + c];
        double cc = (cur == 0xFFFFFFFFu) ? 1e300 : path_cost(cur);
#pragma unroll
        for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
          for (int dc = -1; dc <= 1; ++dc) {
            if (dr == 0 && dc == 0) continue;
            const int rr = r + dr, c2 = c + dc;
            if (d2[rr * pw + c2] < 0) continue;
            const unsigned int nb = aux[rr * pw + c2];
            if (nb == 0xFFFFFFFFu) continue;
            const unsigned int cand = (dr != 0 && dc != 0) ? nb + 0x10000u : nb + 1u;
            const double cv = path_cost(cand);
            if (cv < cc) { cc = cv; cur = cand; }
          }
        if (cur != aux[r * pw + c]) { aux[r * pw + c] = cur; changed = 1; }
      }
      if (changed) s_changed = 1;
      __syncthreads();
      const int any = s_changed;
      __syncthreads();
      if (!any) break;
    }
    // ---- ring / wedge code per object pixel -------------------------------------------------------
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int r = i / w, c = i % w;
      const int dd = d2[(r + 1) * pw + c + 1];
      if (dd < 0) continue;
      const unsigned int pth = aux[(r + 1) * pw + c + 1];
      unsigned char code = 0;
      if (pth != 0xFFFFFFFFu) {
        const double dfrom = path_cost(pth), dedge = sqrt((double)dd);
        // scaled rings (the default): distance from the centre as a fraction of centre-to-edge; unscaled: in units of
        // maximum_radius pixels, everything beyond it in the overflow ring `bin_count`
        const double nrm = a.max_radius > 0 ? dfrom / a.max_radius : dfrom / (dfrom + dedge + 0.001);
        int bin = (int)(nrm * (double)a.bin_count);
        if (bin > a.bin_count) bin = a.bin_count;
        const int wedge = (r > ci ? 1 : 0) + (c > cj ? 2 : 0) + (abs(r - ci) > abs(c - cj) ? 4 : 0);
        code = (unsigned char)(0x80 | (wedge << 4) | bin);
      }
      bm[(size_t)(o.y0 + r) * a.X + o.x0 + c] = code;
    }
    __syncthreads();
  }
}

struct RadialStatArgs {
  const u16* labels;
  const unsigned char* binmap;
  const void* planes;
  int F, C, Y, X, channel;
  const aliby_object* tab;
  int n_obj, bin_count;
  int nout;  // rings reported: bin_count (scaled) or bin_count + 1 (unscaled: the overflow ring too)
  double* out;
  int ld, col0;
};

template <typename T>
__global__ __launch_bounds__(256) void k_radial_stats(RadialStatArgs a) {
  __shared__ double tot[RD_MAXBINS + 1];
  __shared__ int cnt[RD_MAXBINS + 1];
  __shared__ double wsum[RD_MAXBINS + 1][8];
  __shared__ int wcnt[RD_MAXBINS + 1][8];
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;
  const int nb = a.bin_count;
  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    double* out = a.out + (size_t)oi * a.ld + a.col0;
    if (o.area <= 0) {
      for (int k = tid; k < 3 * a.nout; k += blockDim.x) out[k] = NAN;
      continue;
    }
    __syncthreads();
    for (int k = tid; k < (RD_MAXBINS + 1) * 8; k += blockDim.x) { (&wsum[0][0])[k] = 0; (&wcnt[0][0])[k] = 0; }
    for (int k = tid; k <= RD_MAXBINS; k += blockDim.x) { tot[k] = 0; cnt[k] = 0; }
    __syncthreads();
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const unsigned char* bm = a.binmap + (size_t)o.tile * plane;
    const T* px = reinterpret_cast<const T*>(a.planes) + ((size_t)o.tile * a.C + a.channel) * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0;
    const u16 L = (u16)o.label;
    for (int i = tid; i < h * w; i += blockDim.x) {
      const size_t idx = (size_t)(o.y0 + i / w) * a.X + (o.x0 + i % w);
      if (lab[idx] != L) continue;
      const int code = bm[idx];
      if (!(code & 0x80)) continue;
      const int bin = code & 15, wedge = (code >> 4) & 7;
      const double v = (double)px_load<T>(px, idx);
      atomicAdd(&tot[bin], v);
      atomicAdd(&cnt[bin], 1);
      atomicAdd(&wsum[bin][wedge], v);
      atomicAdd(&wcnt[bin][wedge], 1);
    }
    __syncthreads();
    if (tid < a.nout) {
      double T_ = 0, N_ = 0;
      for (int b = 0; b <= nb; ++b) { T_ += tot[b]; N_ += (double)cnt[b]; }
      const double fd = tot[tid] / T_;
      const double fb = (double)cnt[tid] / N_;
      double mean = 0;
      int nw = 0;
      double m8[8];
      for (int k = 0; k < 8; ++k)
        if (wcnt[tid][k] > 0) { m8[nw] = wsum[tid][k] / (double)wcnt[tid][k]; mean += m8[nw]; ++nw; }
      double cv = 0.0;
      if (nw > 0) {
        mean /= (double)nw;
        double var = 0;
        for (int k = 0; k < nw; ++k) var += (m8[k] - mean) * (m8[k] - mean);
        cv = sqrt(var / (double)nw) / mean;
      }
      out[tid] = fd;
      out[a.nout + tid] = fd / (fb + 2.220446049250313e-16);
      out[2 * a.nout + tid] = cv;
    }
    __syncthreads();
  }
}

extern "C" {

int aliby_radial_geometry(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                          const aliby_object* table_dev, int n_obj, int max_h, int max_w, int bin_count,
                          uint8_t* binmap_dev, void* stream) {
  return aliby_radial_geometry_unscaled(ctx, labels, F, Y, X, table_dev, n_obj, max_h, max_w, bin_count, 0.0, binmap_dev, stream);
}

int aliby_radial_geometry_unscaled(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                                   const aliby_object* table_dev, int n_obj, int max_h, int max_w, int bin_count,
                                   double maximum_radius, uint8_t* binmap_dev, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && table_dev && binmap_dev, "NULL argument");
  ARG_CHECK(F > 0 && Y > 0 && X > 0 && max_h >= 0 && max_w >= 0, "bad shape");
  ARG_CHECK(bin_count >= 1 && bin_count < RD_MAXBINS, "1 <= bin_count < 16");
  ARG_CHECK(max_h + max_w < 16000, "object too large for 16-bit path counters");
  RadialGeoArgs a;
  a.labels = labels; a.F = F; a.Y = Y; a.X = X; a.tab = table_dev; a.n_obj = n_obj; a.bin_count = bin_count;
  a.max_radius = maximum_radius;
  a.binmap = binmap_dev;
  a.cap_cells = ((size_t)(max_h + 2) * (max_w + 2) + 3) & ~(size_t)3;
  const size_t need = a.cap_cells * 8;
  hipStream_t s = as_stream(stream);
  if (need <= 128 * 1024) {
    a.gscratch = nullptr;
    if (need > 32 * 1024)
      HIP_TRY(hipFuncSetAttribute((const void*)k_radial_geometry<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    hipLaunchKernelGGL((k_radial_geometry<false>), dim3(n_obj), dim3(aliby_pick_block((long long)max_h * max_w)), need, s, a);
  } else {
    const int g = n_obj < 512 ? n_obj : 512;
    int rc = aliby_ensure_scratch(ctx, (size_t)g * need);
    if (rc) return rc;
    a.gscratch = (unsigned char*)ctx->scratch;
    hipLaunchKernelGGL((k_radial_geometry<true>), dim3(g), dim3(256), 0, s, a);
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}

int aliby_features_radial_distribution(aliby_ctx* ctx, const uint16_t* labels, const uint8_t* binmap_dev,
                                       const void* planes, int dtype, int F, int C, int Y, int X,
                                       int channel, const aliby_object* table_dev, int n_obj,
                                       int bin_count, double* out, int ld, int col0, void* stream) {
  return aliby_features_radial_distribution_rings(ctx, labels, binmap_dev, planes, dtype, F, C, Y, X, channel, table_dev, n_obj,
                                                  bin_count, bin_count, out, ld, col0, stream);
}

int aliby_features_radial_distribution_rings(aliby_ctx* ctx, const uint16_t* labels, const uint8_t* binmap_dev,
                                             const void* planes, int dtype, int F, int C, int Y, int X,
                                             int channel, const aliby_object* table_dev, int n_obj,
                                             int bin_count, int rings_out, double* out, int ld, int col0, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  ARG_CHECK(rings_out == bin_count || rings_out == bin_count + 1, "rings_out is bin_count (scaled) or bin_count + 1 (with the overflow ring)");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && binmap_dev && planes && table_dev && out, "NULL argument");
  ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32, "dtype must be ALIBY_U16 or ALIBY_F32");
  ARG_CHECK(channel >= 0 && channel < C, "channel out of range");
  ARG_CHECK(bin_count >= 1 && bin_count < RD_MAXBINS, "1 <= bin_count < 16");
  ARG_CHECK(col0 >= 0 && col0 + 3 * rings_out <= ld, "columns exceed row stride");
  RadialStatArgs a;
  a.labels = labels; a.binmap = binmap_dev; a.planes = planes; a.F = F; a.C = C; a.Y = Y; a.X = X;
  a.channel = channel; a.tab = table_dev; a.n_obj = n_obj; a.bin_count = bin_count;
  a.nout = rings_out;
  a.out = out; a.ld = ld; a.col0 = col0;
  hipStream_t s = as_stream(stream);
  if (dtype == ALIBY_U16) hipLaunchKernelGGL((k_radial_stats<u16>), dim3(n_obj), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_radial_stats<float>), dim3(n_obj), dim3(256), 0, s, a);
  KERNEL_CHECK();
  return ALIBY_OK;
}

}  // extern "C"
