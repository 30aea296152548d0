// feat_zernike.hip — minimum enclosing circles and Zernike moments, one workgroup per object.
//
// Reference call sites: wrap_cp_measure_features (extraction/core/functions/loaders.py:135-150) with
// fun = get_core_measurements()["zernike"] and ["radial_zernikes"], both in the builder's default
// feature list (pipe_builder.py:49-56).  cp_measure 0.1.17 / centrosome 1.3.3 are not vendored;
// restated from centrosome.zernike (construct_zernike_polynomials, score_zernike,
// get_zernike_indexes) and CellProfiler's MeasureObjectIntensityDistribution.calculate_zernikes:
//   - unit disc = minimum enclosing circle of the object's pixel centres (centre (ci,cj), radius r);
//   - y=(i-ci)/r, x=(j-cj)/r, Z_nm = R_nm(x^2+y^2) * (y + i x)^m, zero outside the unit disc,
//     R by Horner over the factorial look-up table;
//   - "zernike"         : |sum Z_nm| / (pi r^2)               for n<=9, m=n%2..n step 2 (30 values)
//   - "radial_zernikes" : v = sum I*Z_nm; magnitude |v|/n_pixels, phase atan2(Re v, Im v) (30 + 30)
//
// The minimum enclosing circle is computed exactly on the convex-hull vertices with the
// Elzinga-Hearn iteration; every step's "farthest vertex" search is a block-wide reduction.
#include "common.h"
#include "hull.h"

typedef unsigned short u16;

#define ZK 30      // (n,m) pairs for n<=9
#define ZW 5       // max Horner terms

// column of (n,m) in centrosome's get_zernike_indexes order (n ascending, m = n%2, n%2+2, .., n)
__host__ __device__ constexpr int zidx(int n, int m) {
  // sum_{q<n} (q/2 + 1) in closed form, so the unrolled loops index acc[] with constants
  return n + ((n - 1) / 2) * (n / 2) + (m - n % 2) / 2;
}
__host__ __device__ constexpr double zfact(int n) { double f = 1; for (int i = 2; i <= n; ++i) f *= i; return f; }
// (-1)^t (n-t)! / (t! ((n+m)/2-t)! ((n-m)/2-t)!)  — construct_zernike_lookuptable
__host__ __device__ constexpr double zlut(int n, int m, int t) {
  return ((t & 1) ? -1.0 : 1.0) * zfact(n - t) / (zfact(t) * zfact((n + m) / 2 - t) * zfact((n - m) / 2 - t));
}

// -----------------------------------------------------------------------------------------------
// minimum enclosing circle
// -----------------------------------------------------------------------------------------------
struct MecArgs {
  const u16* labels;
  int F, Y, X;
  const aliby_object* tab;
  int n_obj, max_h;
  size_t cap_bytes;
  unsigned char* gscratch;
  double* mec;  // [n_obj][4]: ci, cj, r, n_hull
};

struct Circ { double ci, cj, r2; };

__device__ __forceinline__ Circ circ2(double ai, double aj, double bi, double bj) {
  Circ c;
  c.ci = 0.5 * (ai + bi); c.cj = 0.5 * (aj + bj);
  const double di = ai - c.ci, dj = aj - c.cj;
  c.r2 = di * di + dj * dj;
  return c;
}

__device__ __forceinline__ Circ circ3(double ax, double ay, double bx, double by, double cx, double cy) {
  const double d = 2.0 * (ax * (by - cy) + bx * (cy - ay) + cx * (ay - by));
  const double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
  Circ c;
  c.ci = (a2 * (by - cy) + b2 * (cy - ay) + c2 * (ay - by)) / d;
  c.cj = (a2 * (cx - bx) + b2 * (ax - cx) + c2 * (bx - ax)) / d;
  const double di = ax - c.ci, dj = ay - c.cj;
  c.r2 = di * di + dj * dj;
  return c;
}

template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_mec(MecArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ int s_n[2];
  __shared__ double red_d[8];
  __shared__ int red_i[8];
  unsigned char* ws = GLOBAL ? (a.gscratch + (size_t)blockIdx.x * a.cap_bytes) : lds_raw;
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;
  const int H = a.max_h;
  int* cmin = reinterpret_cast<int*>(ws);
  int* cmax = cmin + H;
  const int chain_cap = 2 * H + 2;
  P2* low = reinterpret_cast<P2*>(cmax + H);
  P2* up = low + chain_cap;

  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    double* m = a.mec + (size_t)oi * 4;
    if (o.area <= 0) {
      if (tid == 0) { m[0] = NAN; m[1] = NAN; m[2] = NAN; m[3] = 0; }
      continue;
    }
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0;
    const u16 L = (u16)o.label;
    __syncthreads();
    for (int r = tid; r < h; r += blockDim.x) { cmin[r] = INT_MAX; cmax[r] = -1; }
    __syncthreads();
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int r = i / w, c = i % w;
      if (lab[(size_t)(o.y0 + r) * a.X + o.x0 + c] == L) { atomicMin(&cmin[r], c); atomicMax(&cmax[r], c); }
    }
    __syncthreads();
    if ((tid & 63) == 0) {
      for (int wv = tid >> 6; wv < 2; wv += (int)(blockDim.x >> 6))
        s_n[wv] = chain_build(cmin, cmax, h, wv == 0, wv == 0 ? low : up);
    }
    __syncthreads();
    const int nl = s_n[0], nu = s_n[1];
    const int K = (nl <= 1) ? 1 : (nl - 1) + (nu - 1);
    auto vert = [&](int i) -> P2 { return (i < nl - 1 || nl <= 1) ? low[i] : up[i - (nl - 1)]; };

    // farthest vertex from (ci,cj): returns squared distance and index (smallest index on ties)
    auto farthest = [&](double ci, double cj, int& arg) -> double {
      double best = -1.0;
      int bi = INT_MAX;
      for (int j = tid; j < K; j += blockDim.x) {
        const P2 q = vert(j);
        const double di = q.r - ci, dj = q.c - cj, d2 = di * di + dj * dj;
        if (d2 > best) { best = d2; bi = j; }
      }
      const double B = block_max_f64(best, red_d);
      arg = block_min_i32(best == B ? bi : INT_MAX, red_i);
      return B;
    };

    Circ c;
    if (K == 1) {
      const P2 p = vert(0);
      c.ci = p.r; c.cj = p.c; c.r2 = 0;
    } else {
      // start from vertex 0 and the vertex farthest from it
      int s0 = 0, s1, s2 = -1;
      const P2 p0 = vert(0);
      farthest(p0.r, p0.c, s1);
      int ns = 2;
      for (int it = 0; it < 4 * K + 32; ++it) {
        const P2 A = vert(s0), B = vert(s1);
        if (ns == 2) c = circ2(A.r, A.c, B.r, B.c);
        else { const P2 C = vert(s2); c = circ3(A.r, A.c, B.r, B.c, C.r, C.c); }
        int pi;
        const double d2 = farthest(c.ci, c.cj, pi);
        if (d2 <= c.r2 * (1.0 + 1e-12) + 1e-12) break;
        const P2 Pp = vert(pi);
        int t0, t1, t2;
        if (ns == 2) { t0 = s0; t1 = s1; t2 = pi; }
        else {
          // keep the vertex farthest from P and the one across the line (that vertex, centre) from P
          const P2 C = vert(s2);
          const double dA = (A.r - Pp.r) * (double)(A.r - Pp.r) + (A.c - Pp.c) * (double)(A.c - Pp.c);
          const double dB = (B.r - Pp.r) * (double)(B.r - Pp.r) + (B.c - Pp.c) * (double)(B.c - Pp.c);
          const double dC = (C.r - Pp.r) * (double)(C.r - Pp.r) + (C.c - Pp.c) * (double)(C.c - Pp.c);
          int q, u, v;
          if (dA >= dB && dA >= dC) { q = s0; u = s1; v = s2; }
          else if (dB >= dC) { q = s1; u = s0; v = s2; }
          else { q = s2; u = s0; v = s1; }
          const P2 Q = vert(q), U = vert(u), V = vert(v);
          // side of the line Q -> centre
          const double lx = c.ci - Q.r, ly = c.cj - Q.c;
          const double sp = lx * (Pp.c - Q.c) - ly * (Pp.r - Q.r);
          const double su = lx * (U.c - Q.c) - ly * (U.r - Q.r);
          const double sv_ = lx * (V.c - Q.c) - ly * (V.r - Q.r);
          int keep;
          if (sp * su < 0 && !(sp * sv_ < 0)) keep = u;
          else if (sp * sv_ < 0 && !(sp * su < 0)) keep = v;
          else keep = (fabs(su) >= fabs(sv_)) ? ((sp * su <= 0) ? u : v) : ((sp * sv_ <= 0) ? v : u);
          t0 = q; t1 = keep; t2 = pi;
        }
        // right/obtuse triangle -> the two ends of the longest side define the circle
        const P2 T0 = vert(t0), T1 = vert(t1), T2 = vert(t2);
        const double e01 = (double)(T0.r - T1.r) * (T0.r - T1.r) + (double)(T0.c - T1.c) * (T0.c - T1.c);
        const double e02 = (double)(T0.r - T2.r) * (T0.r - T2.r) + (double)(T0.c - T2.c) * (T0.c - T2.c);
        const double e12 = (double)(T1.r - T2.r) * (T1.r - T2.r) + (double)(T1.c - T2.c) * (T1.c - T2.c);
        if (e01 >= e02 + e12) { s0 = t0; s1 = t1; ns = 2; }        // angle at T2 >= 90
        else if (e02 >= e01 + e12) { s0 = t0; s1 = t2; ns = 2; }   // angle at T1
        else if (e12 >= e01 + e02) { s0 = t1; s1 = t2; ns = 2; }   // angle at T0
        else { s0 = t0; s1 = t1; s2 = t2; ns = 3; }
      }
    }
    if (tid == 0) { m[0] = c.ci + o.y0; m[1] = c.cj + o.x0; m[2] = sqrt(c.r2); m[3] = (double)K; }
    __syncthreads();
  }
}

// -----------------------------------------------------------------------------------------------
// Zernike moments
// -----------------------------------------------------------------------------------------------
struct ZernikeArgs {
  const u16* labels;
  const void* planes;  // [F,C,Y,X] or NULL (unweighted)
  int F, C, Y, X, channel;
  const aliby_object* tab;
  int n_obj;
  const double* mec;
  double* out;
  int ld, col0;
};

// local (pass-relative) slot -> global (n,m) column; passes split at m <= 2 | m >= 3 so that a lane never
// carries more than 32 fp64 accumulators (the single-pass version spilled to scratch)
__host__ __device__ constexpr int zpass_count(int mlo, int mhi) {
  int c = 0;
  for (int m = mlo; m <= mhi; ++m) for (int n = m; n < 10; n += 2) ++c;
  return c;
}
struct ZMap { int g[2][32]; };
constexpr ZMap make_zmap() {
  ZMap z{};
  const int lo[2] = {0, 3}, hi[2] = {2, 9};
  for (int p = 0; p < 2; ++p) {
    for (int j = 0; j < 32; ++j) z.g[p][j] = -1;
    int j = 0;
    for (int m = lo[p]; m <= hi[p]; ++m)
      for (int n = m; n < 10; n += 2) { z.g[p][j++] = 2 * zidx(n, m); z.g[p][j++] = 2 * zidx(n, m) + 1; }
  }
  return z;
}
__device__ __constant__ const ZMap d_zmap = make_zmap();

#define ZROW 17  // padded row of the per-wave transpose buffer (doubles)

// One pass (MLO <= m <= MHI) of one wave over one object's bbox, then an LDS-transposed reduction:
// 16 values per chunk: every lane writes its 16 partials, lane l sums 16 rows of column l&15, two xor-shuffles
// combine the 4 row groups.  ~64 LDS ops + 8 shuffles per lane instead of 60 six-step shuffle trees.
template <typename T, bool WEIGHTED, int PASS, int MLO, int MHI>
__device__ __forceinline__ void zern_pass(const ZernikeArgs& a, const aliby_object& o, bool valid, int lane, double ci,
                                          double cj, double rad, const u16* lab, const T* px, double* wred, double* res) {
  constexpr int KP = zpass_count(MLO, MHI);
  double acc[2 * KP];
#pragma unroll
  for (int k = 0; k < 2 * KP; ++k) acc[k] = 0;
  if (valid) {
    const int h = o.y1 - o.y0, w = o.x1 - o.x0, npix = h * w;
    const u16 L = (u16)o.label;
    int r = lane / w, c = lane - r * w;
    for (int i = lane; i < npix; i += 64) {
      const int yy = o.y0 + r, xx = o.x0 + c;
      const size_t idx = (size_t)yy * a.X + xx;
      c += 64;
      while (c >= w) { c -= w; ++r; }
      if (lab[idx] != L) continue;
      const double y = ((double)yy - ci) / rad, x = ((double)xx - cj) / rad;
      const double r2 = x * x + y * y;
      // zero outside the unit disc.  The 2-3 pixels that DEFINE the enclosing circle sit at r2 = 1 +- 1ulp;
      // a 1e-9 guard band makes their membership independent of rounding (same rule in the oracle).
      if (r2 > 1.0 + 1e-9) continue;
      double wgt = 1.0;
      if (WEIGHTED) wgt = (double)px_load<T>(px, idx);
      double zr = 1.0, zi = 0.0;
#pragma unroll
      for (int m = 1; m <= MLO; ++m) { const double nr = zr * y - zi * x; zi = zr * x + zi * y; zr = nr; }
      int j = 0;
#pragma unroll
      for (int m = MLO; m <= MHI; ++m) {
        if (m > MLO) { const double nr = zr * y - zi * x; zi = zr * x + zi * y; zr = nr; }
#pragma unroll
        for (int n = m; n < 10; n += 2) {
          double s = 0;
#pragma unroll
          for (int t = 0; t < ZW; ++t) if (t <= (n - m) / 2) s = s * r2 + zlut(n, m, t);
          s *= wgt;
          acc[2 * j] += s * zr;
          acc[2 * j + 1] += s * zi;
          ++j;
        }
      }
    }
  }
  constexpr int NCH = (2 * KP + 15) / 16;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) wred[lane * ZROW + t] = (ch * 16 + t < 2 * KP) ? acc[(ch * 16 + t < 2 * KP) ? ch * 16 + t : 0] : 0.0;
    __syncthreads();
    const int col = lane & 15, part = lane >> 4;
    double s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += wred[(part * 16 + q) * ZROW + col];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const int slot = ch * 16 + col;
    if (part == 0 && slot < 2 * KP) res[d_zmap.g[PASS][slot]] = s;
  }
}

// one WAVE per object (4 objects per 256-thread workgroup)
template <typename T, bool WEIGHTED>
__global__ __launch_bounds__(256) void k_zernike(ZernikeArgs a) {
  __shared__ double s_wred[4][64 * ZROW];
  __shared__ double s_res[4][2 * ZK];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t plane = (size_t)a.Y * a.X;
  const int ncol = WEIGHTED ? 2 * ZK : ZK;
  for (int base = blockIdx.x * 4; base < a.n_obj; base += gridDim.x * 4) {
    const int oi = base + wv;
    const bool inrange = oi < a.n_obj;
    aliby_object o;
    o.area = 0;
    if (inrange) o = a.tab[oi];
    const bool valid = inrange && o.area > 0;
    double ci = 0, cj = 0, rad = 1;
    const u16* lab = nullptr;
    const T* px = nullptr;
    if (valid) {
      ci = a.mec[(size_t)oi * 4 + 0]; cj = a.mec[(size_t)oi * 4 + 1]; rad = a.mec[(size_t)oi * 4 + 2];
      lab = a.labels + (size_t)o.tile * plane;
      if (WEIGHTED) px = reinterpret_cast<const T*>(a.planes) + ((size_t)o.tile * a.C + a.channel) * plane;
    }
    zern_pass<T, WEIGHTED, 0, 0, 2>(a, o, valid, lane, ci, cj, rad, lab, px, s_wred[wv], s_res[wv]);
    zern_pass<T, WEIGHTED, 1, 3, 9>(a, o, valid, lane, ci, cj, rad, lab, px, s_wred[wv], s_res[wv]);
    __syncthreads();
    if (inrange) {
      double* out = a.out + (size_t)oi * a.ld + a.col0;
      if (!valid) {
        for (int k = lane; k < ncol; k += 64) out[k] = NAN;
      } else if (lane < ZK) {
        const double re = s_res[wv][2 * lane], im = s_res[wv][2 * lane + 1];
        const double mag = sqrt(re * re + im * im);
        if (WEIGHTED) {
          out[lane] = mag / (double)o.area;
          out[ZK + lane] = atan2(re, im);
        } else {
          out[lane] = mag / (M_PI * rad * rad);
        }
      }
    }
    __syncthreads();
  }
}

// -----------------------------------------------------------------------------------------------
// radial_zernikes of several channels in one launch: the channel-independent part of a pixel's work — unit-disc
// coordinates, the powers (y + i x)^m and the 30 radial polynomials (80 of the ~170 fp64 operations per pixel and channel) —
// is evaluated once and applied to every channel's weight.  NCH channels x 2 x KP accumulators stay below 64 doubles per
// lane, so the 30 terms (m-major order) are cut into passes of KP = 15 / 10 / 8 / 6 terms for 2 / 3 / 4 / 5 channels.
// Same operation order per accumulator as k_zernike: the numbers are the same bits.
// -----------------------------------------------------------------------------------------------
struct ZTerms { int n[ZK], m[ZK], col[ZK]; };
__host__ __device__ constexpr ZTerms make_zterms() {
  ZTerms z{};
  int j = 0;
  for (int m = 0; m <= 9; ++m)
    for (int n = m; n < 10; n += 2) { z.n[j] = n; z.m[j] = m; z.col[j] = zidx(n, m); ++j; }
  return z;
}
constexpr ZTerms ZT = make_zterms();
__device__ __constant__ const ZTerms d_zterms = make_zterms();

struct ZernikeMultiArgs {
  const u16* labels;
  const void* planes;  // [F,C,Y,X]
  int F, C, Y, X;
  int nch, channel[8], col0[8];
  const aliby_object* tab;
  int n_obj;
  const double* mec;
  double* out;
  int ld;
};

template <typename T, int NCH, int T0, int T1>
__device__ __forceinline__ void zern_pass_multi(const ZernikeMultiArgs& a, const aliby_object& o, bool valid, int lane, double ci, double cj,
                                                double rad, const u16* lab, const T* const (&px)[NCH], double* wred, double* res) {
  constexpr int KP = T1 - T0, V = 2 * KP * NCH;
  double acc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0;
  if (valid) {
    const int h = o.y1 - o.y0, w = o.x1 - o.x0, npix = h * w;
    const u16 L = (u16)o.label;
    int r = lane / w, c = lane - r * w;
    for (int i = lane; i < npix; i += 64) {
      const int yy = o.y0 + r, xx = o.x0 + c;
      const size_t idx = (size_t)yy * a.X + xx;
      c += 64;
      while (c >= w) { c -= w; ++r; }
      if (lab[idx] != L) continue;
      const double y = ((double)yy - ci) / rad, x = ((double)xx - cj) / rad;
      const double r2 = x * x + y * y;
      if (r2 > 1.0 + 1e-9) continue;  // (the guard band of k_zernike)
      double wgt[NCH];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) wgt[ch] = (double)px_load<T>(px[ch], idx);
      double zr = 1.0, zi = 0.0;
#pragma unroll
      for (int m = 1; m <= ZT.m[T0]; ++m) { const double nr = zr * y - zi * x; zi = zr * x + zi * y; zr = nr; }
#pragma unroll
      for (int t = T0; t < T1; ++t) {
        if (t > T0 && ZT.m[t] != ZT.m[t - 1]) { const double nr = zr * y - zi * x; zi = zr * x + zi * y; zr = nr; }
        const int n = ZT.n[t], m = ZT.m[t];
        double s = 0;
#pragma unroll
        for (int q = 0; q < ZW; ++q) if (q <= (n - m) / 2) s = s * r2 + zlut(n, m, q);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          const double sw = s * wgt[ch];
          acc[((t - T0) * NCH + ch) * 2] += sw * zr;
          acc[((t - T0) * NCH + ch) * 2 + 1] += sw * zi;
        }
      }
    }
  }
  constexpr int NCHUNK = (V + 15) / 16;
#pragma unroll
  for (int ck = 0; ck < NCHUNK; ++ck) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) wred[lane * ZROW + t] = (ck * 16 + t < V) ? acc[(ck * 16 + t < V) ? ck * 16 + t : 0] : 0.0;
    __syncthreads();
    const int col = lane & 15, part = lane >> 4;
    double s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += wred[(part * 16 + q) * ZROW + col];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const int slot = ck * 16 + col;
    if (part == 0 && slot < V) res[2 * NCH * T0 + slot] = s;  // res[(term * NCH + channel) * 2 + {re, im}], terms m-major
  }
}

template <int NCH>
struct ZMultiCfg {
  static constexpr int KP = NCH <= 2 ? 15 : NCH == 3 ? 10 : NCH == 4 ? 8 : 6;
  static constexpr int NPASS = (ZK + KP - 1) / KP;
};

template <typename T, int NCH, int P>
__device__ __forceinline__ void zern_passes_multi(const ZernikeMultiArgs& a, const aliby_object& o, bool valid, int lane, double ci, double cj,
                                                  double rad, const u16* lab, const T* const (&px)[NCH], double* wred, double* res) {
  constexpr int KP = ZMultiCfg<NCH>::KP;
  if constexpr (P < ZMultiCfg<NCH>::NPASS) {
    constexpr int T0 = P * KP, T1 = (T0 + KP < ZK) ? T0 + KP : ZK;
    zern_pass_multi<T, NCH, T0, T1>(a, o, valid, lane, ci, cj, rad, lab, px, wred, res);
    zern_passes_multi<T, NCH, P + 1>(a, o, valid, lane, ci, cj, rad, lab, px, wred, res);
  }
}

// one WAVE per object (4 objects per 256-thread workgroup), NCH channels
template <typename T, int NCH>
__global__ __launch_bounds__(256) void k_zernike_multi(ZernikeMultiArgs a) {
  __shared__ double s_wred[4][64 * ZROW];
  __shared__ double s_res[4][2 * ZK * NCH];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t plane = (size_t)a.Y * a.X;
  for (int base = blockIdx.x * 4; base < a.n_obj; base += gridDim.x * 4) {
    const int oi = base + wv;
    const bool inrange = oi < a.n_obj;
    aliby_object o;
    o.area = 0;
    o.tile = 0;
    if (inrange) o = a.tab[oi];
    const bool valid = inrange && o.area > 0;
    double ci = 0, cj = 0, rad = 1;
    const u16* lab = nullptr;
    const T* px[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) px[ch] = nullptr;
    if (valid) {
      ci = a.mec[(size_t)oi * 4 + 0]; cj = a.mec[(size_t)oi * 4 + 1]; rad = a.mec[(size_t)oi * 4 + 2];
      lab = a.labels + (size_t)o.tile * plane;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) px[ch] = reinterpret_cast<const T*>(a.planes) + ((size_t)o.tile * a.C + a.channel[ch]) * plane;
    }
    zern_passes_multi<T, NCH, 0>(a, o, valid, lane, ci, cj, rad, lab, px, s_wred[wv], s_res[wv]);
    __syncthreads();
    if (inrange) {
      for (int k = lane; k < ZK * NCH; k += 64) {
        const int t = k / NCH, ch = k - t * NCH;
        double* out = a.out + (size_t)oi * a.ld + a.col0[ch];
        const int col = d_zterms.col[t];
        if (!valid) {
          out[col] = NAN;
          out[ZK + col] = NAN;
        } else {
          const double re = s_res[wv][2 * k], im = s_res[wv][2 * k + 1];
          out[col] = sqrt(re * re + im * im) / (double)o.area;
          out[ZK + col] = atan2(re, im);
        }
      }
    }
    __syncthreads();
  }
}

template <typename T>
static int launch_zernike_multi(const ZernikeMultiArgs& a, hipStream_t s) {
  dim3 grid((a.n_obj + 3) / 4), block(256);
  switch (a.nch) {
    case 2: hipLaunchKernelGGL((k_zernike_multi<T, 2>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_zernike_multi<T, 3>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((k_zernike_multi<T, 4>), grid, block, 0, s, a); break;
    case 5: hipLaunchKernelGGL((k_zernike_multi<T, 5>), grid, block, 0, s, a); break;
    default: aliby_set_error("radial_zernikes: 2 to 5 channels per launch, got %d", a.nch); return ALIBY_ERR_INVALID;
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}

extern "C" {

int aliby_object_mec(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X, const aliby_object* table_dev,
                     int n_obj, int max_h, double* mec_dev, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && table_dev && mec_dev, "NULL argument");
  ARG_CHECK(F > 0 && Y > 0 && X > 0 && max_h >= 0, "bad shape");
  MecArgs a;
  a.labels = labels; a.F = F; a.Y = Y; a.X = X; a.tab = table_dev; a.n_obj = n_obj; a.max_h = max_h; a.mec = mec_dev;
  const size_t need = 2 * (size_t)max_h * sizeof(int) + 2 * (size_t)(2 * max_h + 2) * sizeof(P2);
  a.cap_bytes = (need + 15) & ~(size_t)15;
  hipStream_t s = as_stream(stream);
  if (a.cap_bytes <= 96 * 1024) {
    a.gscratch = nullptr;
    if (a.cap_bytes > 48 * 1024)
      HIP_TRY(hipFuncSetAttribute((const void*)k_mec<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.cap_bytes));
    hipLaunchKernelGGL((k_mec<false>), dim3(n_obj), dim3(aliby_pick_block((long long)max_h * max_h)), a.cap_bytes, s, a);
  } else {
    const int g = n_obj < 512 ? n_obj : 512;
    int rc = aliby_ensure_scratch(ctx, (size_t)g * a.cap_bytes);
    if (rc) return rc;
    a.gscratch = (unsigned char*)ctx->scratch;
    hipLaunchKernelGGL((k_mec<true>), dim3(g), dim3(256), 0, s, a);
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}

int aliby_features_zernike(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C,
                           int Y, int X, int channel, const aliby_object* table_dev, int n_obj,
                           const double* mec_dev, int weighted, double* out, int ld, int col0, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && table_dev && mec_dev && out, "NULL argument");
  ARG_CHECK(F > 0 && Y > 0 && X > 0, "bad shape");
  const int ncol = weighted ? 2 * ZK : ZK;
  ARG_CHECK(col0 >= 0 && col0 + ncol <= ld, "columns exceed row stride");
  if (weighted) {
    ARG_CHECK(planes != nullptr, "planes is NULL");
    ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32, "dtype must be ALIBY_U16 or ALIBY_F32");
    ARG_CHECK(channel >= 0 && channel < C, "channel out of range");
  }
  ZernikeArgs a;
  a.labels = labels; a.planes = planes; a.F = F; a.C = C; a.Y = Y; a.X = X; a.channel = channel;
  a.tab = table_dev; a.n_obj = n_obj; a.mec = mec_dev; a.out = out; a.ld = ld; a.col0 = col0;
  hipStream_t s = as_stream(stream);
  dim3 grid((n_obj + 3) / 4), block(256);
  if (!weighted) hipLaunchKernelGGL((k_zernike<u16, false>), grid, block, 0, s, a);
  else if (dtype == ALIBY_U16) hipLaunchKernelGGL((k_zernike<u16, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((k_zernike<float, true>), grid, block, 0, s, a);
  KERNEL_CHECK();
  return ALIBY_OK;
}

/* radial_zernikes of 2..5 channels of the same planes in one launch (k_zernike_multi): channels[i] -> out[:, col0s[i] .. +60). */
int aliby_features_radial_zernikes_multi(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C, int Y,
                                         int X, const int* channels, const int* col0s, int n_channels, const aliby_object* table_dev,
                                         int n_obj, const double* mec_dev, double* out, int ld, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && planes && table_dev && mec_dev && out && channels && col0s, "NULL argument");
  ARG_CHECK(F > 0 && Y > 0 && X > 0, "bad shape");
  ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32, "dtype must be ALIBY_U16 or ALIBY_F32");
  ARG_CHECK(n_channels >= 2 && n_channels <= 5, "2 <= n_channels <= 5");
  ZernikeMultiArgs a;
  a.labels = labels; a.planes = planes; a.F = F; a.C = C; a.Y = Y; a.X = X; a.nch = n_channels;
  for (int i = 0; i < 8; ++i) { a.channel[i] = 0; a.col0[i] = 0; }
  for (int i = 0; i < n_channels; ++i) {
    ARG_CHECK(channels[i] >= 0 && channels[i] < C, "channel out of range");
    ARG_CHECK(col0s[i] >= 0 && col0s[i] + 2 * ZK <= ld, "columns exceed row stride");
    a.channel[i] = channels[i];
    a.col0[i] = col0s[i];
  }
  a.tab = table_dev; a.n_obj = n_obj; a.mec = mec_dev; a.out = out; a.ld = ld;
  hipStream_t s = as_stream(stream);
  return dtype == ALIBY_U16 ? launch_zernike_multi<u16>(a, s) : launch_zernike_multi<float>(a, s);
}

}  // extern "C"
