// feat_shape.hip — cp_measure "sizeshape" family (2-D), one workgroup per object.
//
// Reference call site: wrap_cp_measure_features (extraction/core/functions/loaders.py:135-150) with
// fun = get_core_measurements()["sizeshape"], reached on the tree branch
// {"None": {"None": ("sizeshape",)}} that build_pipeline_steps always adds (pipe_builder.py:112-116).
// cp_measure 0.1.17 is not vendored (uv.lock:441-442); it ports CellProfiler's
// MeasureObjectSizeShape, which evaluates skimage.measure.regionprops_table on the label image plus
// scipy's EDT and centrosome's convex hull.  This file restates those published definitions
// (scikit-image 0.26 pinned at uv.lock:1972-1973):
//   core kernel : Area, BoundingBox*, Center_*, Extent, EquivalentDiameter, Perimeter (4-neighbour
//                 border + 3x3 weight table), EulerNumber (2x2 bit-quads, 8-connectivity), FormFactor,
//                 Compactness, raw/central/normalised moments to order 3, Hu moments, inertia tensor
//                 and eigenvalues, Major/MinorAxisLength, Eccentricity, Orientation;
//   edt kernel  : Maximum/Mean/MedianRadius from the exact Euclidean distance transform of the
//                 1-padded bbox crop (two-pass separable scheme in LDS);
//   hull kernel : ConvexArea/Solidity (hull of the diamond-offset pixel coordinates, as
//                 skimage.morphology.convex_hull_image) and Min/MaxFeretDiameter (hull of pixel
//                 centres, as centrosome.cpmorphology.feret_diameter).
//
// Column layout (78, alphabetical inside each block) is mirrored by aliby_amd/extraction/features.py.
#include "common.h"
#include "hull.h"

typedef unsigned short u16;

// column indices --------------------------------------------------------------------------------
enum {
  SS_Area = 0, SS_BBoxArea, SS_BBoxMaxX, SS_BBoxMaxY, SS_BBoxMinX, SS_BBoxMinY, SS_CenterX, SS_CenterY,
  SS_Compactness, SS_ConvexArea, SS_Eccentricity, SS_EquivalentDiameter, SS_EulerNumber, SS_Extent,
  SS_FormFactor, SS_MajorAxisLength, SS_MaxFeretDiameter, SS_MaximumRadius, SS_MeanRadius,
  SS_MedianRadius, SS_MinFeretDiameter, SS_MinorAxisLength, SS_Orientation, SS_Perimeter, SS_Solidity,
  SS_Spatial = 25,      // 12: p in 0..2, q in 0..3
  SS_Central = 37,      // 12
  SS_Normalized = 49,   // 16: p,q in 0..3
  SS_Hu = 65,           // 7
  SS_Inertia = 72,      // 4
  SS_InertiaEig = 76,   // 2
  SS_NCOL = 78
};

struct ShapeArgs {
  const u16* labels;
  int F, Y, X;
  const aliby_object* tab;
  int n_obj;
  int cap;                 // bytes/elements of workspace per workgroup
  unsigned char* gscratch; // global fallback or NULL -> LDS
  double* out;
  int ld, col0;
};

// sum a K-vector of doubles over the block; result valid in threads 0..K-1 (returned by value)
template <int K>
__device__ __forceinline__ double block_sum_vec(const double (&v)[K], double* lds /* >= 4*K */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = wave_sum(v[k]);
    if (lane == 0) lds[wid * K + k] = s;
  }
  __syncthreads();
  double r = 0;
  if (threadIdx.x < K) {
    for (int i = 0; i < nw; ++i) r += lds[i * K + threadIdx.x];
  }
  return r;
}

template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_shape_core(ShapeArgs a) {
  extern __shared__ __align__(16) unsigned char lds_mask[];
  __shared__ double vec[4 * 16];
  __shared__ double M[16], MU[16];
  __shared__ int red_i[8];

  unsigned char* m = GLOBAL ? (a.gscratch + (size_t)blockIdx.x * a.cap) : lds_mask;
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;

  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    double* out = a.out + (size_t)oi * a.ld + a.col0;
    if (o.area <= 0) {
      // regionprops has no entry for an absent label
      for (int k = tid; k < SS_NCOL; k += blockDim.x)
        if (k != SS_ConvexArea && k != SS_Solidity && k != SS_MaxFeretDiameter && k != SS_MinFeretDiameter &&
            k != SS_MaximumRadius && k != SS_MeanRadius && k != SS_MedianRadius)
          out[k] = NAN;
      continue;
    }
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0;
    const int ph = h + 4, pw = w + 4;  // 2-pixel halo
    const u16 L = (u16)o.label;

    __syncthreads();
    for (int i = tid; i < ph * pw; i += blockDim.x) {
      const int r = i / pw - 2, c = i % pw - 2;
      const int yy = o.y0 + r, xx = o.x0 + c;
      unsigned char v = 0;
      if (r >= 0 && r < h && c >= 0 && c < w && lab[(size_t)yy * a.X + xx] == L) v = 1;
      m[i] = v;
    }
    __syncthreads();
#define MK(r, c) m[((r) + 2) * pw + (c) + 2]
    // border pixels (object minus 4-neighbour erosion): bit 1
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int r = i / w, c = i % w;
      if ((MK(r, c) & 1) && !((MK(r - 1, c) & 1) && (MK(r + 1, c) & 1) && (MK(r, c - 1) & 1) && (MK(r, c + 1) & 1)))
        MK(r, c) |= 2;
    }
    __syncthreads();

    // ---- raw moments, perimeter classes, euler quads --------------------------
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0;
    int p1 = 0, p2 = 0, p3 = 0;  // perimeter weight classes 1, sqrt2, (1+sqrt2)/2
    int eu = 0;
    for (int i = tid; i < (h + 1) * (w + 1); i += blockDim.x) {
      const int r = i / (w + 1), c = i % (w + 1);
      // bit-quad with bottom-right corner (r,c): codes 1=BR, 4=BL(c-1), 2=TR(r-1), 8=TL
      const int code = (MK(r, c) & 1) | ((MK(r - 1, c) & 1) << 1) | ((MK(r, c - 1) & 1) << 2) |
                       ((MK(r - 1, c - 1) & 1) << 3);
      eu += (code == 8) - (code == 6) - (code == 14);
      if (r < h && c < w && (MK(r, c) & 1)) {
        const double rr = r, cc = c;
        const double r2 = rr * rr, r3 = r2 * rr, c2 = cc * cc, c3 = c2 * cc;
        acc[0] += 1;  acc[1] += cc;      acc[2] += c2;      acc[3] += c3;
        acc[4] += rr; acc[5] += rr * cc; acc[6] += rr * c2; acc[7] += rr * c3;
        acc[8] += r2; acc[9] += r2 * cc; acc[10] += r2 * c2; acc[11] += r2 * c3;
        acc[12] += r3; acc[13] += r3 * cc; acc[14] += r3 * c2; acc[15] += r3 * c3;
        if (MK(r, c) & 2) {
          const int code9 = 1 +
              2 * (((MK(r - 1, c) >> 1) & 1) + ((MK(r + 1, c) >> 1) & 1) + ((MK(r, c - 1) >> 1) & 1) + ((MK(r, c + 1) >> 1) & 1)) +
              10 * (((MK(r - 1, c - 1) >> 1) & 1) + ((MK(r - 1, c + 1) >> 1) & 1) + ((MK(r + 1, c - 1) >> 1) & 1) + ((MK(r + 1, c + 1) >> 1) & 1));
          if (code9 == 5 || code9 == 7 || code9 == 15 || code9 == 17 || code9 == 25 || code9 == 27) ++p1;
          else if (code9 == 21 || code9 == 33) ++p2;
          else if (code9 == 13 || code9 == 23) ++p3;
        }
      }
    }
    {
      const double r = block_sum_vec<16>(acc, vec);
      if (tid < 16) M[tid] = r;
    }
    const int P1 = block_sum_i32(p1, red_i);
    const int P2 = block_sum_i32(p2, red_i);
    const int P3 = block_sum_i32(p3, red_i);
    const int EU = block_sum_i32(eu, red_i);
    __syncthreads();
    const double m00 = M[0];
    const double rbar = M[4] / m00, cbar = M[1] / m00;

    // ---- central moments (direct, about the local centroid) --------------------
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0;
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int r = i / w, c = i % w;
      if (MK(r, c) & 1) {
        const double dr = (double)r - rbar, dc = (double)c - cbar;
        const double r2 = dr * dr, r3 = r2 * dr, c2 = dc * dc, c3 = c2 * dc;
        acc[0] += 1;  acc[1] += dc;      acc[2] += c2;      acc[3] += c3;
        acc[4] += dr; acc[5] += dr * dc; acc[6] += dr * c2; acc[7] += dr * c3;
        acc[8] += r2; acc[9] += r2 * dc; acc[10] += r2 * c2; acc[11] += r2 * c3;
        acc[12] += r3; acc[13] += r3 * dc; acc[14] += r3 * c2; acc[15] += r3 * c3;
      }
    }
    {
      const double r = block_sum_vec<16>(acc, vec);
      if (tid < 16) MU[tid] = r;
    }
    __syncthreads();
#undef MK

    if (tid == 0) {
      const double area = m00;
      const double bbox_area = (double)h * (double)w;
      const double perim = (double)P1 * 1.0 + (double)P3 * ((1.0 + M_SQRT2) / 2.0) + (double)P2 * M_SQRT2;
      out[SS_Area] = area;
      out[SS_BBoxArea] = bbox_area;
      out[SS_BBoxMaxX] = (double)o.x1;
      out[SS_BBoxMaxY] = (double)o.y1;
      out[SS_BBoxMinX] = (double)o.x0;
      out[SS_BBoxMinY] = (double)o.y0;
      out[SS_CenterX] = (double)o.x0 + cbar;
      out[SS_CenterY] = (double)o.y0 + rbar;
      const double fpa = 4.0 * M_PI * area;
      out[SS_Compactness] = perim * perim / (fpa > 1.0 ? fpa : 1.0);
      out[SS_EquivalentDiameter] = sqrt(4.0 * area / M_PI);
      out[SS_EulerNumber] = (double)EU;
      out[SS_Extent] = area / bbox_area;
      out[SS_FormFactor] = fpa / (perim * perim);
      out[SS_Perimeter] = perim;
      // inertia tensor (skimage.measure.inertia_tensor): rows/cols = (r, c) axes
      // second-order central moments from exact integer sums (n*S2 - S1*S1): the isotropic case
      // (Ta == Tc, Tb == 0) that skimage's orientation branches on is then decided exactly
      const double mu00 = MU[0];
      const __int128 n_ = (__int128)llrint(M[0]);
      const __int128 Sr = (__int128)llrint(M[4]), Sc = (__int128)llrint(M[1]);
      const __int128 Srr = (__int128)llrint(M[8]), Scc = (__int128)llrint(M[2]), Src = (__int128)llrint(M[5]);
      const __int128 I20 = n_ * Srr - Sr * Sr, I02 = n_ * Scc - Sc * Sc, I11 = n_ * Src - Sr * Sc;
      const double nn = (double)n_ * (double)n_;
      const double Ta = (double)I02 / nn, Tb = -(double)I11 / nn, Tc = (double)I20 / nn;
      const bool iso = (I02 == I20);
      out[SS_Inertia + 0] = Ta;
      out[SS_Inertia + 1] = Tb;
      out[SS_Inertia + 2] = Tb;
      out[SS_Inertia + 3] = Tc;
      const double hs = 0.5 * (Ta + Tc), hd = 0.5 * (Ta - Tc);
      const double rad = sqrt(hd * hd + Tb * Tb);
      double l1 = hs + rad, l2 = hs - rad;
      if (l1 < 0) l1 = 0;
      if (l2 < 0) l2 = 0;
      out[SS_InertiaEig + 0] = l1;
      out[SS_InertiaEig + 1] = l2;
      out[SS_MajorAxisLength] = 4.0 * sqrt(l1);
      out[SS_MinorAxisLength] = 4.0 * sqrt(l2);
      out[SS_Eccentricity] = (l1 == 0.0) ? 0.0 : sqrt(1.0 - l2 / l1);
      double orient;
      if (iso) orient = (I11 > 0) ? -M_PI / 4.0 : M_PI / 4.0;  // Tb < 0  <=>  I11 > 0
      else orient = 0.5 * atan2(-2.0 * Tb, Tc - Ta);
      out[SS_Orientation] = orient * (180.0 / M_PI);
      for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 4; ++q) {
          out[SS_Spatial + p * 4 + q] = M[p * 4 + q];
          out[SS_Central + p * 4 + q] = MU[p * 4 + q];
        }
      double nu[16];
      for (int p = 0; p < 4; ++p)
        for (int q = 0; q < 4; ++q) {
          if (p + q < 2) nu[p * 4 + q] = NAN;
          else nu[p * 4 + q] = MU[p * 4 + q] / pow(mu00, (double)(p + q) / 2.0 + 1.0);
          out[SS_Normalized + p * 4 + q] = nu[p * 4 + q];
        }
#define NU(p, q) nu[(p) * 4 + (q)]
      {
        double t0 = NU(3, 0) + NU(1, 2), t1 = NU(2, 1) + NU(0, 3);
        double q0 = t0 * t0, q1 = t1 * t1;
        const double n4 = 4.0 * NU(1, 1);
        const double s = NU(2, 0) + NU(0, 2), d = NU(2, 0) - NU(0, 2);
        double hu[7];
        hu[0] = s;
        hu[1] = d * d + n4 * NU(1, 1);
        hu[3] = q0 + q1;
        hu[5] = d * (q0 - q1) + n4 * t0 * t1;
        t0 *= q0 - 3.0 * q1;
        t1 *= 3.0 * q0 - q1;
        q0 = NU(3, 0) - 3.0 * NU(1, 2);
        q1 = 3.0 * NU(2, 1) - NU(0, 3);
        hu[2] = q0 * q0 + q1 * q1;
        hu[4] = q0 * t0 + q1 * t1;
        hu[6] = q1 * t0 - q0 * t1;
        for (int k = 0; k < 7; ++k) out[SS_Hu + k] = hu[k];
      }
#undef NU
    }
    __syncthreads();
  }
}

// -----------------------------------------------------------------------------------------------
// EDT radii: exact Euclidean distance transform of the bbox crop padded by one ring of zeros
// (CellProfiler: numpy.pad(mini_image, 1); scipy.ndimage.distance_transform_edt), then
// max / mean / median of the distances over the object's pixels.
// Workspace per workgroup: g int32[(h+2)*(w+2)] | d float[(h+2)*(w+2)] (d reused as sort buffer).
// -----------------------------------------------------------------------------------------------
struct EdtArgs {
  const u16* labels;
  int F, Y, X;
  const aliby_object* tab;
  int n_obj;
  size_t cap_bytes;
  unsigned char* gscratch;
  double* out;
  int ld, col0;
};

template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_shape_edt(EdtArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ double red_d[8];
  __shared__ int s_cnt;
  unsigned char* ws = GLOBAL ? (a.gscratch + (size_t)blockIdx.x * a.cap_bytes) : lds_raw;
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;

  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    double* out = a.out + (size_t)oi * a.ld + a.col0;
    if (o.area <= 0) {
      if (tid == 0) { out[SS_MaximumRadius] = NAN; out[SS_MeanRadius] = NAN; out[SS_MedianRadius] = NAN; }
      continue;
    }
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0;
    const int ph = h + 2, pw = w + 2;
    const u16 L = (u16)o.label;
    int* g = reinterpret_cast<int*>(ws);
    float* d = reinterpret_cast<float*>(ws + sizeof(int) * (size_t)ph * pw);
    const int BIG = 1 << 20;

    __syncthreads();
    // phase 1 (columns): g = vertical distance to the nearest background pixel
    for (int c = tid; c < pw; c += blockDim.x) {
      int run = BIG;  // distance to the last background seen above
      for (int r = 0; r < ph; ++r) {
        const int yy = o.y0 + r - 1, xx = o.x0 + c - 1;
        const bool fg = (r >= 1 && r <= h && c >= 1 && c <= w) && lab[(size_t)yy * a.X + xx] == L;
        run = fg ? (run >= BIG ? BIG : run + 1) : 0;
        g[r * pw + c] = run;
      }
      run = BIG;
      for (int r = ph - 1; r >= 0; --r) {
        const int cur = g[r * pw + c];
        run = (cur == 0) ? 0 : (run >= BIG ? BIG : run + 1);
        if (run < cur) g[r * pw + c] = run;
      }
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    // phase 2 (rows): d^2(r,c) = min_c' (c-c')^2 + g(r,c')^2 ; only object pixels matter.
    // Squared distances are exact integers; they are what gets sorted.
    int* d2 = reinterpret_cast<int*>(d);
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int r = i / w + 1, c = i % w + 1;
      const int g0 = g[r * pw + c];
      if (g0 == 0) continue;
      long long best = (long long)g0 * g0;
      // search outwards; stop once the horizontal offset alone reaches the best
      for (int dx = 1; (long long)dx * dx < best; ++dx) {
        const int cl = c - dx, cr = c + dx;
        if (cl >= 0) {
          const long long gg = g[r * pw + cl];
          const long long v = gg * gg + (long long)dx * dx;
          if (v < best) best = v;
        }
        if (cr < pw) {
          const long long gg = g[r * pw + cr];
          const long long v = gg * gg + (long long)dx * dx;
          if (v < best) best = v;
        }
      }
      const int pos = atomicAdd(&s_cnt, 1);
      d2[pos] = (int)best;
    }
    __syncthreads();
    const int N = s_cnt;
    const int n2 = next_pow2(N);
    for (int i = N + tid; i < n2; i += blockDim.x) d2[i] = INT_MAX;
    block_bitonic_sort(d2, n2);
    // mean over the sorted list: deterministic order
    double s2 = 0;
    for (int i = tid; i < N; i += blockDim.x) s2 += sqrt((double)d2[i]);
    const double S = block_sum_f64(s2, red_d);
    if (tid == 0) {
      out[SS_MaximumRadius] = sqrt((double)d2[N - 1]);
      out[SS_MeanRadius] = S / (double)N;
      out[SS_MedianRadius] = (N & 1) ? sqrt((double)d2[N / 2])
                                     : 0.5 * (sqrt((double)d2[N / 2 - 1]) + sqrt((double)d2[N / 2]));
    }
    __syncthreads();
  }
}


// -----------------------------------------------------------------------------------------------
// Convex hulls.  Two hulls per object, both built from per-row extremes with Andrew's monotone
// chain (exact integer cross products), four chains run concurrently on lane 0 of four waves:
//   * "diamond" hull of the doubled coordinates (2r±1,2c),(2r,2c±1) of the object's pixels ->
//     ConvexArea = number of bbox pixels whose centre is inside or on the hull
//     (skimage.morphology.convex_hull_image, offset_coordinates=True, include_borders=True);
//   * hull of the pixel centres -> Min/MaxFeretDiameter (centrosome.cpmorphology.feret_diameter:
//     max = hull diameter, min = minimum width over hull edges) and, optionally, the hull itself
//     for the minimum enclosing circle used by the Zernike family.
// Workspace per workgroup (ints): rmin[2h+1] rmax[2h+1] cmin[h] cmax[h] | chains 4 x 2*(2h+2) points
// -----------------------------------------------------------------------------------------------
struct HullArgs {
  const u16* labels;
  int F, Y, X;
  const aliby_object* tab;
  int n_obj;
  int max_h;
  size_t cap_bytes;
  unsigned char* gscratch;
  double* out;      // sizeshape block (may be NULL)
  int ld, col0;
  double* feret_out;  // separate "feret" family block (may be NULL): [min, max]
  int feret_ld, feret_col0;
};

template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_shape_hull(HullArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ int s_n[4];
  __shared__ int red_i[8];
  __shared__ double red_d[8];
  unsigned char* ws = GLOBAL ? (a.gscratch + (size_t)blockIdx.x * a.cap_bytes) : lds_raw;
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;
  const int H = a.max_h;
  int* rmin = reinterpret_cast<int*>(ws);           // doubled rows: 2H+1
  int* rmax = rmin + (2 * H + 1);
  int* cmin = rmax + (2 * H + 1);                   // pixel rows: H
  int* cmax = cmin + H;
  P2* chains = reinterpret_cast<P2*>(cmax + H);     // 4 chains
  const int chain_cap = 2 * (2 * H + 1) + 2;

  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    double* out = a.out ? a.out + (size_t)oi * a.ld + a.col0 : nullptr;
    double* fout = a.feret_out ? a.feret_out + (size_t)oi * a.feret_ld + a.feret_col0 : nullptr;
    if (o.area <= 0) {
      if (tid == 0) {
        if (out) { out[SS_ConvexArea] = NAN; out[SS_Solidity] = NAN; out[SS_MinFeretDiameter] = NAN; out[SS_MaxFeretDiameter] = NAN; }
        if (fout) { fout[0] = NAN; fout[1] = NAN; }
      }
      continue;
    }
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0;
    const u16 L = (u16)o.label;
    __syncthreads();
    for (int r = tid; r < h; r += blockDim.x) { cmin[r] = INT_MAX; cmax[r] = -1; }
    for (int r = tid; r < 2 * h + 1; r += blockDim.x) { rmin[r] = INT_MAX; rmax[r] = INT_MIN; }
    __syncthreads();
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int r = i / w, c = i % w;
      if (lab[(size_t)(o.y0 + r) * a.X + o.x0 + c] == L) { atomicMin(&cmin[r], c); atomicMax(&cmax[r], c); }
    }
    __syncthreads();
    // doubled rows R = 2r+1 +/- 1 shifted by +1 so that R in [0, 2h]: pixel row r -> R = 2r, 2r+1, 2r+2
    for (int r = tid; r < h; r += blockDim.x) {
      if (cmin[r] > cmax[r]) continue;
      const int lo = 2 * cmin[r], hi = 2 * cmax[r];
      atomicMin(&rmin[2 * r], lo);     atomicMax(&rmax[2 * r], hi);
      atomicMin(&rmin[2 * r + 1], lo - 1); atomicMax(&rmax[2 * r + 1], hi + 1);
      atomicMin(&rmin[2 * r + 2], lo); atomicMax(&rmax[2 * r + 2], hi);
    }
    __syncthreads();
    if ((tid & 63) == 0) {
      // four chains spread over however many waves the workgroup has
      for (int wv = tid >> 6; wv < 4; wv += (int)(blockDim.x >> 6)) {
        int n;
        if (wv == 0) n = chain_build(rmin, rmax, 2 * h + 1, true, chains + 0 * chain_cap);
        else if (wv == 1) n = chain_build(rmin, rmax, 2 * h + 1, false, chains + 1 * chain_cap);
        else if (wv == 2) n = chain_build(cmin, cmax, h, true, chains + 2 * chain_cap);
        else n = chain_build(cmin, cmax, h, false, chains + 3 * chain_cap);
        s_n[wv] = n;
      }
    }
    __syncthreads();
    // ---- convex area: bbox pixel centres (doubled: row 2r+1, col 2c) inside or on the diamond hull
    const P2* lowD = chains;                 const int nl = s_n[0];
    const P2* upD = chains + chain_cap;      const int nu = s_n[1];
    int cnt = 0;
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int r = i / w, c = i % w;
      const P2 p{2 * r + 1, 2 * c};
      bool in = true;
      for (int k = 0; k + 1 < nl && in; ++k) in = cross3(lowD[k], lowD[k + 1], p) >= 0;
      for (int k = 0; k + 1 < nu && in; ++k) in = cross3(upD[k], upD[k + 1], p) >= 0;
      cnt += in ? 1 : 0;
    }
    const int CONVEX = block_sum_i32(cnt, red_i);

    // ---- Feret diameters on the pixel-centre hull: vertices = low[0..nl2-2] ++ up[0..nu2-2]
    const P2* lowC = chains + 2 * chain_cap; const int nl2 = s_n[2];
    const P2* upC = chains + 3 * chain_cap;  const int nu2 = s_n[3];
    const int K = (nl2 <= 1) ? 1 : (nl2 - 1) + (nu2 - 1);
    auto vert = [&](int i) -> P2 { return (i < nl2 - 1 || nl2 <= 1) ? lowC[i] : upC[i - (nl2 - 1)]; };
    double dmax = 0.0, wmin = INFINITY;
    if (K >= 2) {
      for (int i = tid; i < K; i += blockDim.x) {
        const P2 pa = vert(i), pb = vert((i + 1) % K);
        double far2 = 0.0;
        long long wmax = 0;
        for (int j = 0; j < K; ++j) {
          const P2 q = vert(j);
          const double dr = q.r - pa.r, dc = q.c - pa.c;
          far2 = fmax(far2, dr * dr + dc * dc);
          long long cr = cross3(pa, pb, q);
          if (cr < 0) cr = -cr;
          if (cr > wmax) wmax = cr;
        }
        dmax = fmax(dmax, sqrt(far2));
        const double er = pb.r - pa.r, ec = pb.c - pa.c;
        wmin = fmin(wmin, (double)wmax / sqrt(er * er + ec * ec));
      }
    }
    const double DMAX = block_max_f64(dmax, red_d);
    const double WMIN = -block_max_f64(-wmin, red_d);
    if (tid == 0) {
      const double fmin_ = (K <= 2) ? 0.0 : WMIN;
      const double fmax_ = (K <= 1) ? 0.0 : DMAX;
      if (out) {
        out[SS_ConvexArea] = (double)CONVEX;
        out[SS_Solidity] = (double)o.area / (double)CONVEX;
        out[SS_MinFeretDiameter] = fmin_;
        out[SS_MaxFeretDiameter] = fmax_;
      }
      if (fout) { fout[0] = fmin_; fout[1] = fmax_; }
    }
    __syncthreads();
  }
}

static int launch_hull(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                       const aliby_object* table_dev, int n_obj, int max_h, double* out, int ld, int col0,
                       double* fout, int fld, int fcol0, hipStream_t s) {
  HullArgs a;
  a.labels = labels; a.F = F; a.Y = Y; a.X = X; a.tab = table_dev; a.n_obj = n_obj; a.max_h = max_h;
  a.out = out; a.ld = ld; a.col0 = col0; a.feret_out = fout; a.feret_ld = fld; a.feret_col0 = fcol0;
  const size_t ints = (size_t)2 * (2 * max_h + 1) + 2 * (size_t)max_h;
  const size_t chain_cap = 2 * (size_t)(2 * max_h + 1) + 2;
  const size_t need = ints * sizeof(int) + 4 * chain_cap * sizeof(P2);
  a.cap_bytes = (need + 15) & ~(size_t)15;
  if (a.cap_bytes <= 96 * 1024) {
    a.gscratch = nullptr;
    if (a.cap_bytes > 48 * 1024)
      HIP_TRY(hipFuncSetAttribute((const void*)k_shape_hull<false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.cap_bytes));
    hipLaunchKernelGGL((k_shape_hull<false>), dim3(n_obj), dim3(aliby_pick_block((long long)max_h * max_h)), a.cap_bytes, s, a);
  } else {
    const int g = n_obj < 512 ? n_obj : 512;
    int rc = aliby_ensure_scratch(ctx, (size_t)g * a.cap_bytes);
    if (rc) return rc;
    a.gscratch = (unsigned char*)ctx->scratch;
    hipLaunchKernelGGL((k_shape_hull<true>), dim3(g), dim3(256), 0, s, a);
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}

extern "C" int aliby_features_feret(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                                    const aliby_object* table_dev, int n_obj, int max_h, double* out,
                                    int ld, int col0, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && table_dev && out, "NULL argument");
  ARG_CHECK(F > 0 && Y > 0 && X > 0 && max_h >= 0, "bad shape");
  ARG_CHECK(col0 >= 0 && col0 + 2 <= ld, "columns exceed row stride");
  return launch_hull(ctx, labels, F, Y, X, table_dev, n_obj, max_h, nullptr, 0, 0, out, ld, col0,
                     as_stream(stream));
}

static int pow2_at_least(int n, int lo) {
  int p = lo;
  while (p < n) p <<= 1;
  return p;
}

extern "C" int aliby_features_sizeshape(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                                        const aliby_object* table_dev, int n_obj, int max_h, int max_w,
                                        int max_area, double* out, int ld, int col0, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && table_dev && out, "NULL argument");
  ARG_CHECK(F > 0 && Y > 0 && X > 0 && max_h >= 0 && max_w >= 0, "bad shape");
  ARG_CHECK(col0 >= 0 && col0 + SS_NCOL <= ld, "columns exceed row stride");
  hipStream_t s = as_stream(stream);
  const size_t lds_cap = 96 * 1024;

  {  // core
    ShapeArgs a;
    a.labels = labels; a.F = F; a.Y = Y; a.X = X; a.tab = table_dev; a.n_obj = n_obj;
    a.out = out; a.ld = ld; a.col0 = col0;
    const size_t need = (size_t)(max_h + 4) * (max_w + 4);
    a.cap = (int)((need + 15) & ~(size_t)15);
    if (need <= lds_cap) {
      a.gscratch = nullptr;
      if (need > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_shape_core<false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.cap));
      hipLaunchKernelGGL((k_shape_core<false>), dim3(n_obj), dim3(aliby_pick_block((long long)max_h * max_w)), a.cap, s, a);
    } else {
      const int g = n_obj < 512 ? n_obj : 512;
      int rc = aliby_ensure_scratch(ctx, (size_t)g * a.cap);
      if (rc) return rc;
      a.gscratch = (unsigned char*)ctx->scratch;
      hipLaunchKernelGGL((k_shape_core<true>), dim3(g), dim3(256), 0, s, a);
    }
    KERNEL_CHECK();
  }
  {  // edt radii
    EdtArgs a;
    a.labels = labels; a.F = F; a.Y = Y; a.X = X; a.tab = table_dev; a.n_obj = n_obj;
    a.out = out; a.ld = ld; a.col0 = col0;
    const size_t cells = (size_t)(max_h + 2) * (max_w + 2);
    const size_t sortn = (size_t)pow2_at_least(max_area, 64);
    const size_t need = cells * sizeof(int) + (cells > sortn ? cells : sortn) * sizeof(float);
    a.cap_bytes = (need + 15) & ~(size_t)15;
    if (a.cap_bytes <= 128 * 1024) {
      a.gscratch = nullptr;
      if (a.cap_bytes > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_shape_edt<false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.cap_bytes));
      hipLaunchKernelGGL((k_shape_edt<false>), dim3(n_obj), dim3(aliby_pick_block((long long)max_h * max_w)), a.cap_bytes, s, a);
    } else {
      const int g = n_obj < 512 ? n_obj : 512;
      int rc = aliby_ensure_scratch(ctx, (size_t)g * a.cap_bytes);
      if (rc) return rc;
      a.gscratch = (unsigned char*)ctx->scratch;
      hipLaunchKernelGGL((k_shape_edt<true>), dim3(g), dim3(256), 0, s, a);
    }
    KERNEL_CHECK();
  }
  return launch_hull(ctx, labels, F, Y, X, table_dev, n_obj, max_h, out, ld, col0, nullptr, 0, 0, s);
}
