// feat_texture.hip — cp_measure "texture" (Haralick), one workgroup per (object, channel).
//
// Reference call site: wrap_cp_measure_features (extraction/core/functions/loaders.py:135-150) with
// fun = get_core_measurements()["texture"] (default feature list, pipe_builder.py:49-56).
// cp_measure 0.1.17 / mahotas 1.4.18 are not vendored; restated from CellProfiler's MeasureTexture
// and mahotas.features.haralick(distance=scale, ignore_zeros=True):
//   grey level = uint16 >> 8 (skimage img_as_ubyte) or rint(255 f) for [0,1] floats;
//   per object: bbox crop, non-object pixels = 0; symmetric co-occurrence at distance `scale` for the
//   directions (0,1),(1,1),(1,0),(1,-1); pairs touching grey level 0 are dropped; 13 statistics.
//
// MI355X shape: a 256x256 co-occurrence matrix does not fit LDS in any useful way and is >99% empty for
// a ~1k-pixel object, so the matrix is kept SPARSE: the (lo,hi) grey-level pair of every pixel pair is
// a 16-bit key, keys are bitonic-sorted in LDS, equal-key runs are the non-zero cells.  Objects with few
// distinct grey levels K (the common case: ~30 for a nucleus) skip the sort: levels are renumbered 0..K-1
// through a 256-bit presence mask and the K(K+1)/2 cells of the symmetric matrix are counted directly with
// integer LDS atomics (16-bit counters, two per word) in the memory the keys would have used.  Marginals
// (p_x, p_{x+y}, p_{x-y}) are integer LDS histograms, so every statistic is exact-integer counts
// divided once by the total: no atomics on floats, run-to-run deterministic.
#include "common.h"
#include <atomic>

typedef unsigned short u16;

#define TX_NSTAT 13

struct TextureArgs {
  const u16* labels;
  const void* planes;
  int F, C, Y, X, channel;
  const aliby_object* tab;
  int n_obj;
  int cap_pix;   // >= max bbox area (bytes for the grey crop), multiple of 16
  int cap_keys;  // power of two >= max area
  int cap_cells; // 16-bit cell counters that fit the key area (0: dense path off, an area could overflow 16 bits)
  unsigned char* gscratch;
  int scale, gray_levels;
  int grey_shift;  // 8: uint16 pixels; 0: uint8 pixels in uint16 storage (ALIBY_U8W)
  double* out;
  int ld, col0;
};

__device__ __forceinline__ int grey_of(unsigned short v, int gl, int shift) {
  int q = v >> shift;
  if (gl != 256) q = (int)((double)q / 255.0 * (double)(gl - 1));
  return q;
}
__device__ __forceinline__ int grey_of(float v, int gl, int) {
  double x = rint((double)v * 255.0);
  x = fmin(fmax(x, 0.0), 255.0);
  int q = (int)x;
  if (gl != 256) q = (int)((double)q / 255.0 * (double)(gl - 1));
  return q;
}

// Every probability of the co-occurrence statistics is an integer count over the total T, so p log2(p) is
// (c / T) (log2 c - log2 T): log2 of the integers below 2^16 comes from a table in device memory (512 KB, L2-resident,
// filled once per device with the same log2() it replaces); the fp64 log2 sequence was most of the kernel's instructions.
#define TX_LOGTAB 65536
__device__ double g_log2_int[TX_LOGTAB];
__global__ void k_init_log2_table() {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < TX_LOGTAB) g_log2_int[i] = i > 0 ? log2((double)i) : 0.0;
}
__device__ __forceinline__ double log2_int(int n) { return n < TX_LOGTAB ? g_log2_int[n] : log2((double)n); }
// (c / T) log2(c / T), logT = log2(T)
__device__ __forceinline__ double plog2p_count(int c, double Tt, double logT) {
  return c > 0 ? ((double)c / Tt) * (log2_int(c) - logT) : 0.0;
}

template <typename T, bool GLOBAL>
__global__ __launch_bounds__(256) void k_texture(TextureArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ int hx[256];      // p_x counts
  __shared__ int hplus[512];   // p_{x+y} counts
  __shared__ int hminus[256];  // p_{x-y} counts
  __shared__ double vec[4 * 8];
  __shared__ int red_i[8];
  __shared__ int wsum[4];
  __shared__ unsigned int bits[8];     // presence of the grey levels 1..255 in the object
  __shared__ unsigned char rk[256];    // grey level -> rank among the present levels
  __shared__ unsigned char lev[256];   // rank -> grey level

  unsigned char* ws = GLOBAL ? (a.gscratch + (size_t)blockIdx.x * ((size_t)a.cap_pix + (size_t)a.cap_keys * 4)) : lds_raw;
  unsigned char* g = ws;
  unsigned int* keys = reinterpret_cast<unsigned int*>(ws + a.cap_pix);
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;
  const int DY[4] = {0, 1, 1, 1}, DX[4] = {1, 1, 0, -1};

  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    double* out = a.out + (size_t)oi * a.ld + a.col0;
    if (o.area <= 0) {
      for (int k = tid; k < 4 * TX_NSTAT; k += blockDim.x) out[k] = NAN;
      continue;
    }
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const T* px = reinterpret_cast<const T*>(a.planes) + ((size_t)o.tile * a.C + a.channel) * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0, npix = h * w;
    const u16 L = (u16)o.label;

    __syncthreads();
    int gmax = 0;
    // (the loads of four pixels are issued together: one memory latency per batch, see feat_intensity.hip)
    for (int i0 = tid; i0 < npix; i0 += 4 * blockDim.x) {
      u16 lb[4];
      T pv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(i0 + u * (int)blockDim.x, npix - 1);
        const size_t idx = (size_t)(o.y0 + i / w) * a.X + (o.x0 + i % w);
        lb[u] = lab[idx];
        pv[u] = px[idx];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * (int)blockDim.x;
        if (i >= npix) break;
        const int q = lb[u] == L ? grey_of(pv[u], a.gray_levels, a.grey_shift) : 0;
        g[i] = (unsigned char)q;
        gmax = max(gmax, q);
      }
    }
    const int maxv = block_max_i32(gmax, red_i) + 1;  // side of mahotas' matrix: max grey level + 1
    __syncthreads();
    // ---- present grey levels -> ranks
    if (tid < 8) bits[tid] = 0;
    __syncthreads();
    for (int i = tid; i < npix; i += blockDim.x) {
      const int q = g[i];
      if (q > 0) atomicOr(&bits[q >> 5], 1u << (q & 31));
    }
    __syncthreads();
    int K = 0;
    for (int w8 = 0; w8 < 8; ++w8) K += __popc(bits[w8]);
    for (int q = tid; q < 256; q += blockDim.x) {
      if ((bits[q >> 5] >> (q & 31)) & 1u) {
        int r = __popc(bits[q >> 5] & ((1u << (q & 31)) - 1u));
        for (int w8 = 0; w8 < (q >> 5); ++w8) r += __popc(bits[w8]);
        rk[q] = (unsigned char)r;
        lev[r] = (unsigned char)q;
      }
    }
    __syncthreads();
    const int minlev = K ? lev[0] : 0, maxlev = K ? lev[K - 1] : 0;
    const int ncell = K * (K + 1) / 2;
    const bool dense = !GLOBAL && ncell > 0 && ncell <= a.cap_cells;
    unsigned int* cells = keys;  // two 16-bit counters per word

    for (int d = 0; d < 4; ++d) {
      const int dy = DY[d] * a.scale, dx = DX[d] * a.scale;
      double* fo = out + d * TX_NSTAT;
      // only the occupied ranges of the marginals are ever touched: [minlev, maxlev], [0, maxlev-minlev], [2 minlev, 2 maxlev]
      for (int k = tid; k <= maxlev - minlev; k += blockDim.x) { hx[minlev + k] = 0; hminus[k] = 0; }
      for (int k = 2 * minlev + tid; k <= 2 * maxlev; k += blockDim.x) hplus[k] = 0;
      int NP;
      if (dense) {
        // ---- direct counting of the K(K+1)/2 cells --------------------------------------------------
        for (int k = tid; k < (ncell + 1) / 2; k += blockDim.x) cells[k] = 0;
        __syncthreads();
        int mine = 0;
        for (int i = tid; i < npix; i += blockDim.x) {
          const int r = i / w, c = i % w, r2 = r + dy, c2 = c + dx;
          if (r2 >= 0 && r2 < h && c2 >= 0 && c2 < w) {
            const int va = g[i], vb = g[r2 * w + c2];
            if (va > 0 && vb > 0) {
              const int ra = rk[va], rb = rk[vb];
              const int hi = max(ra, rb), idx = hi * (hi + 1) / 2 + min(ra, rb);
              atomicAdd(&cells[idx >> 1], (idx & 1) ? 65536u : 1u);
              ++mine;
            }
          }
        }
        NP = block_sum_i32(mine, red_i);
        __syncthreads();
      } else {
        // ---- pair keys (order-preserving compaction is unnecessary: keys get sorted) ----------
        int base = 0;
        for (int i0 = 0; i0 < npix; i0 += blockDim.x) {
          const int i = i0 + tid;
          bool ok = false;
          unsigned int key = 0;
          if (i < npix) {
            const int r = i / w, c = i % w, r2 = r + dy, c2 = c + dx;
            if (r2 >= 0 && r2 < h && c2 >= 0 && c2 < w) {
              const int va = g[i], vb = g[r2 * w + c2];
              if (va > 0 && vb > 0) {
                ok = true;
                key = (unsigned)(min(va, vb) << 8) | (unsigned)max(va, vb);
              }
            }
          }
          const int pos = block_compact_slot(ok, base, wsum);
          if (ok) keys[pos] = key;
        }
        __syncthreads();
        NP = base;  // ordered pixel pairs; T = 2*NP entries in the symmetric matrix
      }
      if (NP == 0) {
        // mahotas raises ValueError on an empty matrix; CellProfiler records NaN
        for (int k = tid; k < TX_NSTAT; k += blockDim.x) fo[k] = NAN;
        __syncthreads();
        continue;
      }
      const double Tt = 2.0 * (double)NP;
      const double logT = log2_int(2 * NP);
      if (!dense) {
        const int n2 = next_pow2(NP);
        for (int i = NP + tid; i < n2; i += blockDim.x) keys[i] = 0xFFFFFFFFu;
        block_bitonic_sort(keys, n2);
      }

      // ---- non-zero cells -> integer marginals + cell sums ------------------------------------------
      double acc[3] = {0, 0, 0};  // sum p^2 (as counts^2), sum i*j*count, sum p*log2(p)
      auto cell1 = [&](int c, int lo, int hi) {  // c pixel pairs with the unordered grey-level pair (lo, hi)
        if (lo == hi) {
          atomicAdd(&hx[lo], 2 * c);
          acc[0] += 4.0 * (double)c * (double)c;
          acc[2] += plog2p_count(2 * c, Tt, logT);
        } else {
          atomicAdd(&hx[lo], c);
          atomicAdd(&hx[hi], c);
          acc[0] += 2.0 * (double)c * (double)c;
          acc[2] += 2.0 * plog2p_count(c, Tt, logT);
        }
        atomicAdd(&hplus[lo + hi], 2 * c);
        atomicAdd(&hminus[hi - lo], 2 * c);
        acc[1] += 2.0 * (double)c * (double)lo * (double)hi;
      };
      auto dense_cell = [&](int idx, int& c, int& lo, int& hi) {  // triangular index -> (count, levels)
        c = (int)((cells[idx >> 1] >> (16 * (idx & 1))) & 0xffffu);
        int rh = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
        while (rh * (rh + 1) / 2 > idx) --rh;
        while ((rh + 1) * (rh + 2) / 2 <= idx) ++rh;
        lo = lev[idx - rh * (rh + 1) / 2];
        hi = lev[rh];
      };
      if (dense) {
        for (int idx = tid; idx < ncell; idx += blockDim.x) {
          int c, lo, hi;
          dense_cell(idx, c, lo, hi);
          if (c) cell1(c, lo, hi);
        }
      } else {
        for (int i = tid; i < NP; i += blockDim.x) {
          const unsigned int key = keys[i];
          if (i > 0 && keys[i - 1] == key) continue;
          int lo_ = i + 1, hi_ = NP;
          while (lo_ < hi_) { const int mid = (lo_ + hi_) >> 1; if (keys[mid] == key) lo_ = mid + 1; else hi_ = mid; }
          cell1(lo_ - i, (int)(key >> 8), (int)(key & 255u));
        }
      }
      block_sum_vec_all<3>(acc, vec);
      __syncthreads();
      const double f_asm = acc[0] / (Tt * Tt);
      const double sum_ij = acc[1] / Tt;
      const double f_entropy = -acc[2];

      // ---- marginal statistics --------------------------------------------------------------------
      double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      // m0 = ux, m1 = sum k^2 px, m2 = HX(sum p log p), m3 = contrast, m4 = IDM, m5 = sum p_minus,
      // m6 = sum p_minus^2 (for the vector variance), m7 = difference entropy (sum p log p)
      for (int k = minlev + tid; k <= maxlev; k += blockDim.x) {  // p_x is zero outside the present levels
        const double pxk = (double)hx[k] / Tt;
        m[0] += (double)k * pxk;
        m[1] += (double)k * (double)k * pxk;
        m[2] += plog2p_count(hx[k], Tt, logT);
      }
      for (int k = tid; k <= maxlev - minlev; k += blockDim.x) {  // |i - j| never exceeds the level range
        const double pm = (double)hminus[k] / Tt;
        m[3] += (double)k * (double)k * pm;
        m[4] += pm / (1.0 + (double)k * (double)k);
        if (k < maxv) { m[5] += pm; m[6] += pm * pm; }
        m[7] += plog2p_count(hminus[k], Tt, logT);
      }
      block_sum_vec_all<8>(m, vec);
      double s[3] = {0, 0, 0};  // sum average, sum k^2 p_plus, sum entropy (sum p log p)
      for (int k = 2 * minlev + tid; k <= 2 * maxlev; k += blockDim.x) {
        const double pp = (double)hplus[k] / Tt;
        s[0] += (double)k * pp;
        s[1] += (double)k * (double)k * pp;
        s[2] += plog2p_count(hplus[k], Tt, logT);
      }
      block_sum_vec_all<3>(s, vec);

      // HXY1 = -sum_ij p_ij log2(px_i py_j): second pass over the cells now that p_x is complete
      double hxy = 0;
      if (dense) {
        for (int idx = tid; idx < ncell; idx += blockDim.x) {
          int c, lo, hi;
          dense_cell(idx, c, lo, hi);
          if (c) hxy += (2.0 * (double)c / Tt) * (log2_int(hx[lo]) + log2_int(hx[hi]) - 2.0 * logT);
        }
      } else {
        for (int i = tid; i < NP; i += blockDim.x) {
          const unsigned int key = keys[i];
          if (i > 0 && keys[i - 1] == key) continue;
          int lo_ = i + 1, hi_ = NP;
          while (lo_ < hi_) { const int mid = (lo_ + hi_) >> 1; if (keys[mid] == key) lo_ = mid + 1; else hi_ = mid; }
          const int c = lo_ - i;
          const int lo = (int)(key >> 8), hi = (int)(key & 255u);
          hxy += (2.0 * (double)c / Tt) * (log2_int(hx[lo]) + log2_int(hx[hi]) - 2.0 * logT);
        }
      }
      double hv[1] = {hxy};
      block_sum_vec_all<1>(hv, vec);

      if (tid == 0) {
        const double ux = m[0], vx = m[1] - ux * ux, sx = sqrt(vx);
        const double HX = -m[2];
        const double HXY1 = -hv[0];
        const double HXY2 = 2.0 * HX;  // -sum (px_i py_j) log2(px_i py_j) with p symmetric
        fo[0] = f_asm;
        fo[1] = m[3];
        fo[2] = (sx == 0.0) ? 1.0 : (1.0 / sx / sx) * (sum_ij - ux * ux);
        fo[3] = vx;
        fo[4] = m[4];
        fo[5] = s[0];
        fo[6] = s[1] - s[0] * s[0];
        fo[7] = -s[2];
        fo[8] = f_entropy;
        {
          // numpy var of the length-maxv vector p_{x-y}: mean(|x - mean|^2)
          const double mean = m[5] / (double)maxv;
          fo[9] = m[6] / (double)maxv - mean * mean;
        }
        fo[10] = -m[7];
        fo[11] = (HX == 0.0) ? (f_entropy - HXY1) : (f_entropy - HXY1) / HX;
        fo[12] = sqrt(fmax(0.0, 1.0 - exp(-2.0 * (HXY2 - f_entropy))));
      }
      __syncthreads();
    }
  }
}

extern "C" int aliby_features_texture(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype,
                                      int F, int C, int Y, int X, int channel,
                                      const aliby_object* table_dev, int n_obj, int max_h, int max_w,
                                      int max_area, int scale, int gray_levels, double* out, int ld,
                                      int col0, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && planes && table_dev && out, "NULL argument");
  ARG_CHECK(dtype == ALIBY_U16 || dtype == ALIBY_F32 || dtype == ALIBY_U8W, "dtype must be ALIBY_U16, ALIBY_U8W or ALIBY_F32");
  ARG_CHECK(channel >= 0 && channel < C, "channel out of range");
  ARG_CHECK(F > 0 && Y > 0 && X > 0 && max_h >= 0 && max_w >= 0 && max_area >= 0, "bad shape");
  ARG_CHECK(scale >= 1 && gray_levels >= 2 && gray_levels <= 256, "scale >= 1 and 2 <= gray_levels <= 256");
  ARG_CHECK(col0 >= 0 && col0 + 4 * TX_NSTAT <= ld, "columns exceed row stride");
  TextureArgs a;
  a.labels = labels; a.planes = planes; a.F = F; a.C = C; a.Y = Y; a.X = X; a.channel = channel;
  a.tab = table_dev; a.n_obj = n_obj; a.scale = scale; a.gray_levels = gray_levels;
  a.out = out; a.ld = ld; a.col0 = col0;
  a.grey_shift = dtype == ALIBY_U8W ? 0 : 8;
  if (dtype == ALIBY_U8W) dtype = ALIBY_U16;
  a.cap_pix = (int)(((size_t)max_h * max_w + 15) & ~(size_t)15);
  int ck = 2048;  // >= 4096 16-bit cells for the direct-counting path (K <= 90 distinct grey levels)
  while (ck < max_area) ck <<= 1;
  a.cap_keys = ck;
  a.cap_cells = max_area < 65536 ? 2 * ck : 0;
  const size_t need = (size_t)a.cap_pix + (size_t)ck * 4;
  hipStream_t s = as_stream(stream);
  {
    static std::atomic<unsigned long long> ready{0};  // one bit per device
    const unsigned long long bit = 1ull << (ctx->device & 63);
    if (!(ready.load(std::memory_order_acquire) & bit)) {
      hipLaunchKernelGGL(k_init_log2_table, dim3(TX_LOGTAB / 256), dim3(256), 0, s);
      KERNEL_CHECK();
      HIP_TRY(hipStreamSynchronize(s));  // other streams may run this kernel next
      ready.fetch_or(bit, std::memory_order_release);
    }
  }
  if (need <= 96 * 1024) {
    a.gscratch = nullptr;
    dim3 grid(n_obj), block(aliby_pick_block((long long)max_h * max_w));
    if (dtype == ALIBY_U16) {
      if (need > 32 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_texture<u16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      hipLaunchKernelGGL((k_texture<u16, false>), grid, block, need, s, a);
    } else {
      if (need > 32 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_texture<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      hipLaunchKernelGGL((k_texture<float, false>), grid, block, need, s, a);
    }
  } else {
    const int g = n_obj < 512 ? n_obj : 512;
    int rc = aliby_ensure_scratch(ctx, (size_t)g * need);
    if (rc) return rc;
    a.gscratch = (unsigned char*)ctx->scratch;
    dim3 grid(g), block(256);
    if (dtype == ALIBY_U16) hipLaunchKernelGGL((k_texture<u16, false ? false : true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_texture<float, true>), grid, block, 0, s, a);
  }
  KERNEL_CHECK();
  return ALIBY_OK;
}
