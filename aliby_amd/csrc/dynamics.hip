// dynamics.hip — Cellpose post-network dynamics: (dY, dX, cellprob) -> label image.
//
// Reference call site: `model.eval(...)` at src/aliby/segment/dispatch.py:208-215 (cellpose 4.0.6,
// uv.lock:130-131 — not vendored, weights not obtainable; PARITY UNPINNED against Cellpose itself).
// Restated from the published algorithm (cellpose.dynamics.compute_masks):
//   follow_flows            200 Euler steps p += bilinear(dP/5)(p), torch grid_sample index mapping
//                           (align_corners=False, zero padding), float32, positions clamped;
//   get_masks               end-point histogram padded by 20, seeds = 5x5 maxima with > 10 points,
//                           seed masks grown 5 x (3x3 dilation AND bin > 2) inside an 11x11 window,
//                           overlaps resolved by (points, raster position) priority, pixel label =
//                           seed owning its end-point bin, masks > max_size_fraction of the image
//                           dropped, labels renumbered in order of first raster appearance;
//   remove_bad_flow_masks   heat diffusion from each mask's centre (2 x max extent iterations, 9-point
//                           mean restricted to the mask), flows = normalised central differences,
//                           masks with mean squared error vs dP/5 above flow_threshold dropped;
//   fill_holes_and_remove_small_masks  masks < min_size dropped, holes filled, labels 1..n.
// The CPU restatement (oracle/cellpose_restated.py) writes the same float32/float64 operations in the
// same order, so label images are compared bit-for-bit.
//
// Kernel shapes: per-pixel kernels are HBM/L2-bound streaming passes (flow following is an L2-resident
// gather loop); everything per mask runs as one workgroup per object with its bbox staged in LDS.
#include "common.h"
#include <vector>

typedef unsigned short u16;
typedef unsigned long long u64;

#define RPAD 20

struct DynShape {
  int F, Y, X, YP, XP;  // YP = Y + 2*RPAD
  size_t P, PP;         // pixels per tile, padded cells per tile
};

// ---------------------------------------------------------------------------------------------
// 0. one pass over (dP, cellprob): the normalised, masked flow field im = ((mask ? dP : 0) / 5) * (2 / (size-1)) and the
//    compacted list of foreground pixels (cellprob > thr).  Only ~10-35 % of the pixels are foreground and each follows 200
//    dependent steps, so everything after this pass that is per followed pixel (end point, temporary label) lives in arrays
//    indexed by the list position j, not by the pixel: the full-frame int32 passes of rounds 1-2 (pt, M0 and their memsets,
//    the three-phase first-appearance scan) are gone.  Wave-aggregated append, one global atomic per 4096 pixels; the order
//    of the list inside a chunk is raster order, the order of the chunks is irrelevant (every consumer is order-free).
// ---------------------------------------------------------------------------------------------
#define FG_CHUNK 4096  // pixels per workgroup pass
template <bool COPY>
__global__ __launch_bounds__(256) void k_prep_compact(const float* __restrict__ dP, const float* __restrict__ prob, float thr,
                                                      DynShape s, float cx, float cy, float* __restrict__ im,
                                                      int* __restrict__ list, int* __restrict__ count, int reverse) {
  __shared__ int red_i[8];
  __shared__ int wsum[4];
  __shared__ int s_base;
  const size_t total = (size_t)s.F * s.P;
  const size_t nchunks = (total + FG_CHUNK - 1) / FG_CHUNK;
  // (reverse: a test hook, ALIBY_DEBUG_FG_REVERSE=1 — chunks reserve their list space in descending raster order, the order the
  // scheduler only produces now and then; everything downstream must not care)
  for (size_t ci = blockIdx.x; ci < nchunks; ci += gridDim.x) {
    const size_t c0 = (reverse ? nchunks - 1 - ci : ci) * FG_CHUNK;
    unsigned fgmask = 0;
#pragma unroll
    for (int k = 0; k < FG_CHUNK / 256; ++k) {
      const size_t i = c0 + (size_t)k * 256 + threadIdx.x;
      const bool m = i < total && prob[i] > thr;
      fgmask |= (unsigned)m << k;
      if (COPY && i < total) {
        const size_t f = i / s.P, p = i % s.P;
        float vy = dP[(f * 2 + 0) * s.P + p], vx = dP[(f * 2 + 1) * s.P + p];
        vy = m ? vy : 0.0f;
        vx = m ? vx : 0.0f;
        vy = vy / 5.0f;
        vx = vx / 5.0f;
        im[(f * 2 + 0) * s.P + p] = vy * cy;
        im[(f * 2 + 1) * s.P + p] = vx * cx;
      }
    }
    const int tot = block_sum_i32(__popc(fgmask), red_i);
    if (tot == 0) continue;  // block-uniform
    if (threadIdx.x == 0) s_base = atomicAdd(count, tot);
    __syncthreads();
    int base = s_base;
#pragma unroll
    for (int k = 0; k < FG_CHUNK / 256; ++k) {
      const bool fg = (fgmask >> k) & 1u;
      const int pos = block_compact_slot(fg, base, wsum);
      if (fg) list[pos] = (int)(c0 + (size_t)k * 256 + threadIdx.x);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ float tap(const float* __restrict__ f, int yy, int xx, int H, int W) {
  return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? f[(size_t)yy * W + xx] : 0.0f;
}

// the same value straight from the network's output: (cellprob > thr ? dP : 0) / 5 * c, the arithmetic of k_prep_compact<true>
__device__ __forceinline__ void tap_direct(const float* __restrict__ dy, const float* __restrict__ dx, const float* __restrict__ pr,
                                           float thr, float cy, float cx, int yy, int xx, int H, int W, float& oy, float& ox) {
  oy = 0.0f;
  ox = 0.0f;
  if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
    const size_t q = (size_t)yy * W + xx;
    const bool m = pr[q] > thr;
    float vy = dy[q], vx = dx[q];
    vy = m ? vy : 0.0f;
    vx = m ? vx : 0.0f;
    vy = vy / 5.0f;
    vx = vx / 5.0f;
    oy = vy * cy;
    ox = vx * cx;
  }
}

// ---------------------------------------------------------------------------------------------
// 1. flow following + end-point histogram
// ---------------------------------------------------------------------------------------------
// DIRECT: the taps come from (dP, cellprob) themselves — twelve loads instead of eight where a point enters another pixel cell
// (rare: see below), and no normalised copy of the flow field is written or read (8 bytes per pixel each way).
template <bool DIRECT>
__global__ __launch_bounds__(256) void k_follow(const float* __restrict__ im, const float* __restrict__ prob, float thr, float cx,
                                                float cy, const int* __restrict__ list,
                                                const int* __restrict__ count, DynShape s, int niter,
                                                int* __restrict__ ptc, int* __restrict__ h1, u64* __restrict__ M1,
                                                float* __restrict__ pfinal) {
  const int total = *count;
  const int H = s.Y, W = s.X;
  const float sx = (float)(W - 1), sy = (float)(H - 1), Wf = (float)W, Hf = (float)H;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
    const size_t i = (size_t)list[j];
    const size_t f = i / s.P, p = i % s.P;
    const float* imy = im + (f * 2 + 0) * s.P;
    const float* imx = im + (f * 2 + 1) * s.P;
    const int y = (int)(p / W), x = (int)(p % W);
    float px = (float)x / sx * 2.0f - 1.0f;
    float py = (float)y / sy * 2.0f - 1.0f;
    // The eight taps of the bilinear sample are kept in registers and re-read only when the point enters another
    // pixel cell: points reach their sink within a few dozen steps and then jitter inside one cell, so most of the
    // 200 steps issue no loads at all (the kernel is bound by the gather address rate otherwise).
    int cell_x = INT_MIN, cell_y = INT_MIN;
    float xnw = 0.f, xne = 0.f, xsw = 0.f, xse = 0.f, ynw = 0.f, yne = 0.f, ysw = 0.f, yse = 0.f;
    for (int t = 0; t < niter; ++t) {
      const float ix = ((px + 1.0f) * Wf - 1.0f) / 2.0f;
      const float iy = ((py + 1.0f) * Hf - 1.0f) / 2.0f;
      const float x0 = floorf(ix), y0 = floorf(iy);
      const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
      const float wnw = (x1 - ix) * (y1 - iy);
      const float wne = (ix - x0) * (y1 - iy);
      const float wsw = (x1 - ix) * (iy - y0);
      const float wse = (ix - x0) * (iy - y0);
      const int x0i = (int)x0, y0i = (int)y0;
      if (x0i != cell_x || y0i != cell_y) {
        const int x1i = (int)x1, y1i = (int)y1;
        if constexpr (DIRECT) {  // (im = dP here)
          const float* pr = prob + f * s.P;
          tap_direct(imy, imx, pr, thr, cy, cx, y0i, x0i, H, W, ynw, xnw);
          tap_direct(imy, imx, pr, thr, cy, cx, y0i, x1i, H, W, yne, xne);
          tap_direct(imy, imx, pr, thr, cy, cx, y1i, x0i, H, W, ysw, xsw);
          tap_direct(imy, imx, pr, thr, cy, cx, y1i, x1i, H, W, yse, xse);
        } else {
        xnw = tap(imx, y0i, x0i, H, W); xne = tap(imx, y0i, x1i, H, W); xsw = tap(imx, y1i, x0i, H, W); xse = tap(imx, y1i, x1i, H, W);
        ynw = tap(imy, y0i, x0i, H, W); yne = tap(imy, y0i, x1i, H, W); ysw = tap(imy, y1i, x0i, H, W); yse = tap(imy, y1i, x1i, H, W);
        }
        cell_x = x0i; cell_y = y0i;
      }
      float dx = 0.0f + xnw * wnw;
      dx = dx + xne * wne;
      dx = dx + xsw * wsw;
      dx = dx + xse * wse;
      float dy = 0.0f + ynw * wnw;
      dy = dy + yne * wne;
      dy = dy + ysw * wsw;
      dy = dy + yse * wse;
      const float npx = fminf(fmaxf(px + dx, -1.0f), 1.0f);
      const float npy = fminf(fmaxf(py + dy, -1.0f), 1.0f);
      // A step is a function of the position alone: once a step leaves (px, py) as it was, every later step does too.  Points
      // reach their sink within a few dozen steps (the increments fall below half an ulp of the position), so when no lane of
      // the wave — neighbouring pixels, mostly of one mask — moved, the remaining iterations are skipped: same bits, fewer steps.
      const bool moved = npx != px || npy != py;
      px = npx;
      py = npy;
      if (__ballot(moved) == 0ull) break;
    }
    const float fx = (px + 1.0f) * 0.5f * sx;
    const float fy = (py + 1.0f) * 0.5f * sy;
    if (pfinal) { pfinal[(f * 2 + 0) * s.P + p] = fy; pfinal[(f * 2 + 1) * s.P + p] = fx; }
    float qy = fmaxf(fy + (float)RPAD, 0.0f), qx = fmaxf(fx + (float)RPAD, 0.0f);
    qy = fminf(qy, (float)(H + RPAD - 1));
    qx = fminf(qx, (float)(W + RPAD - 1));
    const int cell = (int)qy * s.XP + (int)qx;
    ptc[j] = cell;  // (indexed by the list position)
    atomicAdd(&h1[f * s.PP + cell], 1);
    // the seed-ownership map is only ever looked up at end-point cells: they are cleared here, by whoever ends there, instead of
    // by a memset of 8 bytes per padded pixel (k_grow's atomicMax runs in a later launch; cells never looked up may hold anything)
    M1[f * s.PP + cell] = 0ull;
  }
}

// ---------------------------------------------------------------------------------------------
// 2. seeds (5x5 maxima with > 10 points) and their grown masks
// ---------------------------------------------------------------------------------------------
// Per-tile seed lists (65536 slots each; a tile with more seeds than uint16 labels overflows later anyway and is reported).
// (Round 3 tried finding the seeds from the end points of the foreground list instead of this scan of the padded histogram:
// 7 M scattered 4-byte gathers fetched more bytes than the scan streams, 0.52 ms against 0.22 ms.)
#define H_MASK 0x7fffffff
#define SEEDS_PER_TILE 65536
__global__ void k_seeds(const int* __restrict__ h1, DynShape s, int* __restrict__ seed_list, int* __restrict__ seed_count) {
  const size_t total = (size_t)s.F * s.PP;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int h = h1[i];
    if (h <= 10) continue;
    const size_t f = i / s.PP;
    const int cell = (int)(i % s.PP), r = cell / s.XP, c = cell % s.XP;
    const int* hf = h1 + f * s.PP;
    bool ismax = true;
    for (int dr = -2; dr <= 2 && ismax; ++dr)
      for (int dc = -2; dc <= 2; ++dc) {
        const int rr = r + dr, cc = c + dc;
        if (rr < 0 || rr >= s.YP || cc < 0 || cc >= s.XP) continue;
        if (hf[rr * s.XP + cc] > h) { ismax = false; break; }
      }
    if (!ismax) continue;
    const int k = atomicAdd(&seed_count[f], 1);
    if (k < SEEDS_PER_TILE) seed_list[f * SEEDS_PER_TILE + k] = cell;
  }
}

// one wave per seed: 11x11 window, 5 x (3x3 dilation AND h>2); owner = max (points, cell) priority
__global__ __launch_bounds__(64) void k_grow(const int* __restrict__ h1, DynShape s, const int* __restrict__ seed_list,
                                             const int* __restrict__ seed_count, u64* __restrict__ M1, int* __restrict__ cnt,
                                             int* __restrict__ firstpos, int* __restrict__ newid) {
  __shared__ unsigned char ok[121], cur[121], nxt[121];
  const int f = blockIdx.y;
  const int n = min(seed_count[f], SEEDS_PER_TILE);
  for (int k = blockIdx.x; k < n; k += gridDim.x) {
    const int cell = seed_list[(size_t)f * SEEDS_PER_TILE + k];
    const int r0 = cell / s.XP, c0 = cell % s.XP;
    const int* hf = h1 + (size_t)f * s.PP;
    // the words of this seed's temporary label (its cell + 1): pixel count, first raster position, final id — touched only
    // at seed cells, so they are initialised here instead of by three memsets of 4 bytes per padded pixel
    if (threadIdx.x == 0) {
      cnt[(size_t)f * s.PP + cell] = 0;
      firstpos[(size_t)f * s.PP + cell] = INT_MAX;
      newid[(size_t)f * s.PP + cell] = 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 121; i += 64) {
      const int rr = r0 - 5 + i / 11, cc = c0 - 5 + i % 11;
      const bool in = rr >= 0 && rr < s.YP && cc >= 0 && cc < s.XP;
      ok[i] = (in && (hf[rr * s.XP + cc] & H_MASK) > 2) ? 1 : 0;
      cur[i] = (i == 60) ? 1 : 0;
    }
    __syncthreads();
    for (int it = 0; it < 5; ++it) {
      for (int i = threadIdx.x; i < 121; i += 64) {
        const int r = i / 11, c = i % 11;
        unsigned char v = 0;
        for (int dr = -1; dr <= 1; ++dr)
          for (int dc = -1; dc <= 1; ++dc) {
            const int rr = r + dr, cc = c + dc;
            if (rr >= 0 && rr < 11 && cc >= 0 && cc < 11) v |= cur[rr * 11 + cc];
          }
        nxt[i] = v & ok[i];
      }
      __syncthreads();
      for (int i = threadIdx.x; i < 121; i += 64) cur[i] = nxt[i];
      __syncthreads();
    }
    const u64 prio = ((u64)(unsigned)(hf[cell] & H_MASK) << 32) | (u64)(unsigned)cell;
    for (int i = threadIdx.x; i < 121; i += 64) {
      if (!cur[i]) continue;
      const int rr = r0 - 5 + i / 11, cc = c0 - 5 + i % 11;
      atomicMax(&M1[(size_t)f * s.PP + rr * s.XP + cc], prio + 1ull);  // +1: 0 means "no seed"
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 3. pixel labels (temporary id = owning seed's cell + 1), sizes, first raster positions, first-appearance renumbering
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assign(const int* __restrict__ list, const int* __restrict__ count, const int* __restrict__ ptc,
                                                const u64* __restrict__ M1, DynShape s, unsigned int* __restrict__ labc,
                                                int* __restrict__ cnt, int* __restrict__ firstpos) {
  const int total = *count;
  const int rounds = (total + (int)(gridDim.x * blockDim.x) - 1) / (int)(gridDim.x * blockDim.x);
  for (int it = 0; it < rounds; ++it) {  // (every lane takes part in the ballots below)
    const int j = (it * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    const bool live = j < total;
    size_t f = 0;
    int p = 0;
    unsigned int lab = 0;
    if (live) {
      const size_t i = (size_t)list[j];
      f = i / s.P;
      p = (int)(i % s.P);
      const u64 m = M1[f * s.PP + ptc[j]];
      if (m) lab = (unsigned int)((m - 1ull) & 0xFFFFFFFFull) + 1u;
      labc[j] = lab;
    }
    // Neighbouring lanes are neighbouring foreground pixels of a row, mostly of the same mask: one atomic pair per RUN of equal
    // labels inside the wave instead of one per pixel.  Integer atomics: same result in any order.
    const int lane = threadIdx.x & (WAVE - 1);
    const unsigned long long key = ((unsigned long long)f << 32) | lab;
    const unsigned long long prev = __shfl_up(key, 1, WAVE);
    const int prev_p = __shfl_up(p, 1, WAVE);
    // (the list is in raster order inside a 4096-pixel chunk of k_prep_compact, and the chunks land in it in the order their
    // workgroups reserved space: where a wave straddles two chunks the position can step BACK inside a run of one label — a
    // mask that spans both — and the run's first lane no longer holds its smallest position: such a step starts a new run)
#ifdef DYN_OLD_HEADS
    const bool head = lab && (lane == 0 || prev != key);  // (the round-3 bug, kept to show that the test below catches it)
    (void)prev_p;
#else
    const bool head = lab && (lane == 0 || prev != key || prev_p > p);
#endif
    const unsigned long long heads = __ballot(head || !lab) | ~__ballot(1);
    if (head) {
      const unsigned long long after = lane == 63 ? 0ull : (heads >> (lane + 1));
      const int run = after ? __ffsll((long long)after) : 64 - lane;
      atomicAdd(&cnt[f * s.PP + lab - 1], run);
      atomicMin(&firstpos[f * s.PP + lab - 1], p);  // (positions ascend inside a run: the head holds the run's first)
    }
  }
}

// New id of every kept label = 1 + the number of kept labels of its tile that appear earlier in raster order (cellpose renumbers
// in order of first appearance).  The labels are the tile's seeds (a few hundred): one thread per seed counts the others.
__global__ __launch_bounds__(256) void k_rank_ids(const int* __restrict__ seed_list, const int* __restrict__ seed_count, DynShape s,
                                                  const int* __restrict__ cnt, const int* __restrict__ firstpos, float big,
                                                  int* __restrict__ newid, int* __restrict__ ntot) {
  __shared__ int tile_pos[1024];
  __shared__ int red_i[8];
  const int f = blockIdx.x;
  const int n = min(seed_count[f], SEEDS_PER_TILE);
  const int* sl = seed_list + (size_t)f * SEEDS_PER_TILE;
  const size_t base = (size_t)f * s.PP;
  int kept_total = 0;
  for (int k0 = 0; k0 < n; k0 += blockDim.x) {
    const int k = k0 + threadIdx.x;
    int mine = INT_MAX, cell = 0;
    bool kept = false;
    if (k < n) {
      cell = sl[k];
      const int c = cnt[base + cell];
      kept = c > 0 && !((float)c > big);
      mine = kept ? firstpos[base + cell] : INT_MAX;
    }
    int before = 0;
    for (int q0 = 0; q0 < n; q0 += 1024) {  // the other seeds' first positions, 1024 at a time through LDS
      __syncthreads();
      for (int q = threadIdx.x; q < 1024; q += blockDim.x) {
        int v = INT_MAX;
        if (q0 + q < n) {
          const int cq = sl[q0 + q];
          const int c = cnt[base + cq];
          if (c > 0 && !((float)c > big)) v = firstpos[base + cq];
        }
        tile_pos[q] = v;
      }
      __syncthreads();
      const int m = min(1024, n - q0);
      if (kept)
        for (int q = 0; q < m; ++q) before += tile_pos[q] < mine ? 1 : 0;
    }
    if (kept) newid[base + cell] = before + 1;
    kept_total += kept ? 1 : 0;
  }
  kept_total = block_sum_i32(kept_total, red_i);
  if (threadIdx.x == 0) ntot[f] = kept_total;
}

__global__ void k_apply_ids(const int* __restrict__ list, const int* __restrict__ count, const unsigned int* __restrict__ labc,
                            const int* __restrict__ newid, DynShape s, u16* __restrict__ labels) {
  const int total = *count;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
    const unsigned int lab = labc[j];
    if (!lab) continue;  // (labels is zero-filled)
    const size_t i = (size_t)list[j];
    const int id = newid[(i / s.P) * s.PP + lab - 1];
    if (id) labels[i] = (u16)(id > 65535 ? 65535 : id);
  }
}

// ---------------------------------------------------------------------------------------------
// 4. flow QC: heat diffusion per mask, flow error, removal
// ---------------------------------------------------------------------------------------------
struct QcArgs {
  const u16* labels;
  const float* dP;  // [F,2,Y,X] network-scale flows
  int F, Y, X;
  const aliby_object* tab;
  int n_obj;
  const int* niter_tile;  // [F]
  size_t cap_cells;       // >= (max_h+2)*(max_w+2)
  unsigned char* gscratch;
  double* Tg;             // [F,Y,X]
};

template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_diffuse(QcArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ double red_d[8];
  __shared__ int red_i[8];
  __shared__ long long red_l[8];
  unsigned char* ws = GLOBAL ? (a.gscratch + (size_t)blockIdx.x * a.cap_cells * 17) : lds_raw;
  double* T0 = reinterpret_cast<double*>(ws);
  double* T1 = T0 + a.cap_cells;
  unsigned char* mk = reinterpret_cast<unsigned char*>(T1 + a.cap_cells);
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;
  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    if (o.area <= 0) continue;
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0, ph = h + 2, pw = w + 2;
    const u16 L = (u16)o.label;
    __syncthreads();
    long long sy = 0, sx = 0;
    for (int i = tid; i < ph * pw; i += blockDim.x) {
      const int r = i / pw - 1, c = i % pw - 1;
      unsigned char m = 0;
      if (r >= 0 && r < h && c >= 0 && c < w && lab[(size_t)(o.y0 + r) * a.X + o.x0 + c] == L) { m = 1; sy += r; sx += c; }
      mk[i] = m;
      T0[i] = 0.0;
      T1[i] = 0.0;
    }
    const long long SY = block_sum_i64(sy, red_l), SX = block_sum_i64(sx, red_l);
    const double ymed = (double)SY / (double)o.area, xmed = (double)SX / (double)o.area;
    // mask pixel closest to the centre of mass; first in raster order on ties
    double best = INFINITY;
    int bi = INT_MAX;
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int r = i / w, c = i % w;
      if (!mk[(r + 1) * pw + c + 1]) continue;
      const double dx = (double)c - xmed, dy = (double)r - ymed;
      const double d = dx * dx + dy * dy;
      if (d < best) { best = d; bi = i; }
    }
    const double BEST = -block_max_f64(-best, red_d);
    const int CI = block_min_i32(best == BEST ? bi : INT_MAX, red_i);
    const int cidx = (CI / w + 1) * pw + (CI % w) + 1;
    const int niter = a.niter_tile[o.tile];
    double* src = T0;
    double* dst = T1;
    // One heat source at the centre, niter sweeps of the 9-point mean.  T is zero outside the mask for the whole
    // run (only mask pixels are ever written), so the reference's "neighbour * mask" products add exactly the
    // neighbour's value or +0.0 and are dropped; the order of the nine additions is the reference's.  A lane
    // walks down one column strip with the 3x3 window rolling through registers: 3 LDS reads per pixel, not 9
    // (+9 mask bytes).  The "+1 at the centre" of sweep it+1 is applied by the lane that writes the centre in
    // sweep it, which leaves one barrier per sweep.
    const int ngrp = max(1, (int)blockDim.x / max(w, 1));      // row groups working side by side
    const int rows_per = (h + ngrp - 1) / ngrp;
    const int grp = tid / max(w, 1), col0 = tid - grp * w;
    __syncthreads();
    if (tid == 0 && niter > 0) src[cidx] += 1.0;
    for (int it = 0; it < niter; ++it) {
      __syncthreads();
      const bool more = it + 1 < niter;
      if (grp < ngrp) {
        const int r0 = grp * rows_per, r1 = min(h, r0 + rows_per);
        for (int c = col0; c < w; c += (ngrp == 1 ? (int)blockDim.x : w)) {
          if (r0 >= r1) break;
          int q = (r0 + 1) * pw + c + 1;
          double a0 = src[q - pw - 1], a1 = src[q - pw], a2 = src[q - pw + 1];
          double b0 = src[q - 1], b1 = src[q], b2 = src[q + 1];
          for (int r = r0; r < r1; ++r, q += pw) {
            const double c0 = src[q + pw - 1], c1 = src[q + pw], c2 = src[q + pw + 1];
            if (mk[q]) {
              double acc = 0.0;
              acc = acc + b1;
              acc = acc + a1;
              acc = acc + c1;
              acc = acc + b0;
              acc = acc + b2;
              acc = acc + a0;
              acc = acc + a2;
              acc = acc + c0;
              acc = acc + c2;
              // acc / 9.0, correctly rounded, as multiply + two FMAs instead of the ~15-instruction fp64 division
              // sequence (Markstein: q0 = RN(a*c), r = a - 9*q0 exact, RN(q0 + r*c) = RN(a/9) for c = RN(1/9);
              // checked against the division on 2e9 random doubles)
              const double q0 = acc * (1.0 / 9.0);
              double v = fma(fma(-9.0, q0, acc), 1.0 / 9.0, q0);
              if (more && q == cidx) v += 1.0;
              dst[q] = v;
            }
            a0 = b0; a1 = b1; a2 = b2;
            b0 = c0; b1 = c1; b2 = c2;
          }
          if (ngrp > 1) break;  // one column per lane when the box is narrower than the workgroup
        }
      }
      double* t = src; src = dst; dst = t;
    }
    __syncthreads();
    double* Tt = a.Tg + (size_t)o.tile * plane;
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int q = (i / w + 1) * pw + (i % w) + 1;
      if (mk[q]) Tt[(size_t)(o.y0 + i / w) * a.X + o.x0 + i % w] = src[q];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_flow_error(QcArgs a, float flow_thr, int* __restrict__ bad) {
  __shared__ double vec[4 * 2];
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;
  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const aliby_object o = a.tab[oi];
    if (o.area <= 0) { if (tid == 0) bad[oi] = 1; continue; }
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const double* T = a.Tg + (size_t)o.tile * plane;
    const float* dy_net = a.dP + ((size_t)o.tile * 2 + 0) * plane;
    const float* dx_net = a.dP + ((size_t)o.tile * 2 + 1) * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0;
    const u16 L = (u16)o.label;
    double e[2] = {0, 0};
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int y = o.y0 + i / w, x = o.x0 + i % w;
      const size_t idx = (size_t)y * a.X + x;
      if (lab[idx] != L) continue;
      const double tu = (y > 0) ? T[idx - a.X] : 0.0, td = (y + 1 < a.Y) ? T[idx + a.X] : 0.0;
      const double tl = (x > 0) ? T[idx - 1] : 0.0, tr = (x + 1 < a.X) ? T[idx + 1] : 0.0;
      const double dy = td - tu, dx = tr - tl;
      const double nrm = 1e-60 + sqrt(dy * dy + dx * dx);
      const double my = dy / nrm, mx = dx / nrm;
      const double ey = my - (double)dy_net[idx] / 5.0, ex = mx - (double)dx_net[idx] / 5.0;
      e[0] += ey * ey;
      e[1] += ex * ex;
    }
    block_sum_vec_all<2>(e, vec);
    if (tid == 0) {
      const double err = e[0] / (double)o.area + e[1] / (double)o.area;
      bad[oi] = (err > (double)flow_thr) ? 1 : 0;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// 5. survivors -> final ids; hole filling; small-mask removal
// ---------------------------------------------------------------------------------------------
// one workgroup per tile: keep = !bad && area >= min_size; newlabel = rank among kept (1-based)
__global__ __launch_bounds__(1024) void k_final_ids(const aliby_object* __restrict__ tab, const int* __restrict__ offsets,
                                                    const int* __restrict__ bad, int min_size,
                                                    int* __restrict__ newlabel, int* __restrict__ nfinal) {
  __shared__ int part[1024];
  const int f = blockIdx.x, t = threadIdx.x;
  const int lo0 = offsets[f], n = offsets[f + 1] - lo0;
  const int per = (n + 1023) / 1024;
  const int lo = min(t * per, n), hi = min(lo + per, n);
  int c = 0;
  for (int i = lo; i < hi; ++i) {
    const bool keep = !bad[lo0 + i] && tab[lo0 + i].area > 0 && tab[lo0 + i].area >= min_size;
    c += keep ? 1 : 0;
  }
  part[t] = c;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = (t >= o) ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - c;
  for (int i = lo; i < hi; ++i) {
    const bool keep = !bad[lo0 + i] && tab[lo0 + i].area > 0 && tab[lo0 + i].area >= min_size;
    newlabel[lo0 + i] = keep ? ++run : 0;
  }
  if (t == 1023) nfinal[f] = part[1023];
}

struct FillArgs {
  const u16* labels;
  int F, Y, X;
  const aliby_object* tab;
  int n_obj;
  const int* newlabel;
  size_t cap_cells;
  unsigned char* gscratch;
  u16* out;  // [F,Y,X], zeroed; a 16-bit atomic max resolves nested holes (highest label wins)
};

// max into one half of an aligned 32-bit word (there are no 16-bit atomics): compare-and-swap until our half is >= v
__device__ __forceinline__ void atomic_max_u16(u16* addr, unsigned v) {
  unsigned int* word = reinterpret_cast<unsigned int*>(reinterpret_cast<size_t>(addr) & ~(size_t)3);
  const unsigned shift = (reinterpret_cast<size_t>(addr) & 2) ? 16u : 0u;
  unsigned int old = *word;
  while (((old >> shift) & 0xffffu) < v) {
    const unsigned int want = (old & ~(0xffffu << shift)) | (v << shift);
    const unsigned int seen = atomicCAS(word, old, want);
    if (seen == old) break;
    old = seen;
  }
}

template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_fill(FillArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ int s_changed;
  unsigned char* st = GLOBAL ? (a.gscratch + (size_t)blockIdx.x * a.cap_cells) : lds_raw;  // 0 unknown,1 object,2 outside
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;
  for (int oi = blockIdx.x; oi < a.n_obj; oi += gridDim.x) {
    const int nl = a.newlabel[oi];
    if (nl == 0) continue;
    const aliby_object o = a.tab[oi];
    const u16* lab = a.labels + (size_t)o.tile * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0, ph = h + 2, pw = w + 2;
    const u16 L = (u16)o.label;
    __syncthreads();
    for (int i = tid; i < ph * pw; i += blockDim.x) {
      const int r = i / pw - 1, c = i % pw - 1;
      unsigned char v;
      if (r < 0 || r >= h || c < 0 || c >= w) v = 2;  // ring: outside the bbox is background reachable from outside
      else v = (lab[(size_t)(o.y0 + r) * a.X + o.x0 + c] == L) ? 1 : 0;
      st[i] = v;
    }
    __syncthreads();
    for (int sweep = 0; sweep < ph * pw; ++sweep) {
      if (tid == 0) s_changed = 0;
      __syncthreads();
      int ch = 0;
      for (int i = tid; i < h * w; i += blockDim.x) {
        const int q = (i / w + 1) * pw + (i % w) + 1;
        if (st[q] != 0) continue;
        if (st[q - 1] == 2 || st[q + 1] == 2 || st[q - pw] == 2 || st[q + pw] == 2) { st[q] = 2; ch = 1; }
      }
      if (ch) s_changed = 1;
      __syncthreads();
      const int any = s_changed;
      __syncthreads();
      if (!any) break;
    }
    u16* out = a.out + (size_t)o.tile * plane;
    for (int i = tid; i < h * w; i += blockDim.x) {
      const int q = (i / w + 1) * pw + (i % w) + 1;
      if (st[q] != 2) atomic_max_u16(&out[(size_t)(o.y0 + i / w) * a.X + o.x0 + i % w], (unsigned)nl);
    }
    __syncthreads();
  }
}

// The heat map Tg is written at mask pixels and read at mask pixels and their four neighbours: zero is needed on every object's
// box grown by one pixel, not on the whole frame (8 bytes per pixel).  All boxes are cleared before any mask is written.
__global__ __launch_bounds__(256) void k_zero_boxes(const aliby_object* __restrict__ tab, int n_obj, int Y, int X, double* __restrict__ Tg) {
  const size_t plane = (size_t)Y * X;
  for (int oi = blockIdx.x; oi < n_obj; oi += gridDim.x) {
    const aliby_object o = tab[oi];
    if (o.area <= 0) continue;
    const int y0 = max(o.y0 - 1, 0), y1 = min(o.y1 + 1, Y), x0 = max(o.x0 - 1, 0), x1 = min(o.x1 + 1, X);
    const int w = x1 - x0, n = (y1 - y0) * w;
    double* T = Tg + (size_t)o.tile * plane;
    for (int i = threadIdx.x; i < n; i += blockDim.x) T[(size_t)(y0 + i / w) * X + x0 + i % w] = 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" {

size_t aliby_masks_workspace_bytes(int F, int Y, int X) {
  const size_t P = (size_t)Y * X, PP = (size_t)(Y + 2 * RPAD) * (X + 2 * RPAD);
  size_t b = 0;
  b += align256(sizeof(float) * 2 * P * F);   // im
  b += align256(sizeof(int) * P * F) * 3;     // foreground list, end-point cells, temporary labels (sized for an all-foreground frame)
  b += align256(sizeof(int) * PP * F) * 4;    // h1, cnt, firstpos, newid
  b += align256(sizeof(u64) * PP * F);        // M1
  b += align256(sizeof(u16) * P * F);         // first-appearance labels (before QC)
  b += align256(sizeof(double) * P * F);      // Tg
  b += align256(sizeof(int) * SEEDS_PER_TILE * (size_t)F);  // seed lists
  b += align256(sizeof(aliby_object) * 65536 * (size_t)F);  // object table
  b += align256(sizeof(int) * 65536 * (size_t)F) * 2;  // bad, newlabel
  b += align256(sizeof(int) * (size_t)(5 * F + 8));    // counters
  return b;
}

int aliby_object_table(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X, const int32_t* offsets_host,
                       aliby_object* table_dev, aliby_object* table_host, void* stream);

int aliby_masks_from_flows(aliby_ctx* ctx, const float* dP, const float* cellprob, int F, int Y, int X, int niter,
                           float cellprob_threshold, float flow_threshold, int min_size, float max_size_fraction,
                           void* workspace, size_t workspace_bytes, uint16_t* labels_out, int32_t* n_labels_host,
                           float* p_final_out, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  ARG_CHECK(F >= 0 && Y > 1 && X > 1, "bad shape");
  if (F == 0) return ALIBY_OK;
  ARG_CHECK(dP && cellprob && workspace && labels_out && n_labels_host, "NULL argument");
  ARG_CHECK(workspace_bytes >= aliby_masks_workspace_bytes(F, Y, X), "workspace too small (aliby_masks_workspace_bytes)");
  ARG_CHECK(niter >= 0, "niter must be >= 0");
  hipStream_t s = as_stream(stream);
  DynShape sh;
  sh.F = F; sh.Y = Y; sh.X = X; sh.YP = Y + 2 * RPAD; sh.XP = X + 2 * RPAD;
  sh.P = (size_t)Y * X; sh.PP = (size_t)sh.YP * sh.XP;
  ARG_CHECK(sh.PP < (size_t)INT_MAX && F <= 65535, "image too large");

  unsigned char* w = (unsigned char*)workspace;
  auto take = [&](size_t bytes) { unsigned char* p = w; w += align256(bytes); return p; };
  float* im = (float*)take(sizeof(float) * 2 * sh.P * F);
  int* fg_list = (int*)take(sizeof(int) * sh.P * F);
  int* ptc = (int*)take(sizeof(int) * sh.P * F);
  unsigned* labc = (unsigned*)take(sizeof(int) * sh.P * F);
  int* h1 = (int*)take(sizeof(int) * sh.PP * F);
  int* cnt = (int*)take(sizeof(int) * sh.PP * F);
  int* firstpos = (int*)take(sizeof(int) * sh.PP * F);
  int* newid = (int*)take(sizeof(int) * sh.PP * F);
  u64* M1 = (u64*)take(sizeof(u64) * sh.PP * F);
  u16* labels_tmp = (u16*)take(sizeof(u16) * sh.P * F);
  double* Tg = (double*)take(sizeof(double) * sh.P * F);
  int* seed_list = (int*)take(sizeof(int) * SEEDS_PER_TILE * (size_t)F);
  aliby_object* tab = (aliby_object*)take(sizeof(aliby_object) * 65536 * (size_t)F);
  int* bad = (int*)take(sizeof(int) * 65536 * (size_t)F);
  int* newlabel = (int*)take(sizeof(int) * 65536 * (size_t)F);
  int* counters = (int*)take(sizeof(int) * (size_t)(5 * F + 8));
  int* fg_count = counters + 1;
  int* ntot = counters + 8;          // [F]
  int* niter_tile = ntot + F;        // [F]
  int* nfinal = niter_tile + F;      // [F]
  int* seed_count = nfinal + F;      // [F]

  const size_t totP = sh.P * F, totPP = sh.PP * F;
  ARG_CHECK(totP < (size_t)INT_MAX, "batch too large for 32-bit pixel indices");
  const int gP = (int)((totP + 255) / 256 > 16384 ? 16384 : (totP + 255) / 256);

  // Memsets: the end-point histogram (read in 5x5 / 11x11 neighbourhoods: it must be zero everywhere), the two label images
  // (written at foreground pixels only) and the counters.  Everything else that rounds 1-2 cleared per padded pixel — the seed
  // map M1, the per-label count / first position / new id words — is initialised where it is used (k_follow, k_grow).
  HIP_TRY(hipMemsetAsync(h1, 0, sizeof(int) * totPP, s));
  HIP_TRY(hipMemsetAsync(labels_tmp, 0, sizeof(u16) * totP, s));
  HIP_TRY(hipMemsetAsync(labels_out, 0, sizeof(u16) * totP, s));
  HIP_TRY(hipMemsetAsync(counters, 0, sizeof(int) * (size_t)(5 * F + 8), s));
  const float cx = 2.0f / (float)(X - 1), cy = 2.0f / (float)(Y - 1);
  const char* rev_env = getenv("ALIBY_DEBUG_FG_REVERSE");
  const int rev = rev_env && atoi(rev_env) ? 1 : 0;  // (one workgroup then: it walks the chunks last to first, so that is their order in the list)
  // ALIBY_DYN_DIRECT=1: no normalised copy of the flow field, the flow following gathers from (dP, cellprob) themselves — 1.07 GB
  // per 64 frames of 1024^2 less written and read here, more gathered there; the same labels (tested), 4.7 against 4.5 ms, so the
  // copy stays the default
  const char* direct_env = getenv("ALIBY_DYN_DIRECT");  // (read per call: a test switches it)
  const bool flow_copy = !(direct_env && atoi(direct_env) != 0);
  if (flow_copy) hipLaunchKernelGGL(k_prep_compact<true>, dim3(rev ? 1 : gP), dim3(256), 0, s, dP, cellprob, cellprob_threshold, sh, cx, cy, im, fg_list, fg_count, rev);
  else hipLaunchKernelGGL(k_prep_compact<false>, dim3(rev ? 1 : gP), dim3(256), 0, s, dP, cellprob, cellprob_threshold, sh, cx, cy, im, fg_list, fg_count, rev);
  KERNEL_CHECK();
  if (flow_copy) hipLaunchKernelGGL(k_follow<false>, dim3(gP), dim3(256), 0, s, im, cellprob, cellprob_threshold, cx, cy, fg_list, fg_count, sh, niter, ptc, h1, M1, p_final_out);
  else hipLaunchKernelGGL(k_follow<true>, dim3(gP), dim3(256), 0, s, dP, cellprob, cellprob_threshold, cx, cy, fg_list, fg_count, sh, niter, ptc, h1, M1, p_final_out);
  KERNEL_CHECK();
  const int gPP = (int)((totPP + 255) / 256 > 16384 ? 16384 : (totPP + 255) / 256);
  hipLaunchKernelGGL(k_seeds, dim3(gPP), dim3(256), 0, s, h1, sh, seed_list, seed_count);
  KERNEL_CHECK();
  hipLaunchKernelGGL(k_grow, dim3(256, F), dim3(64), 0, s, h1, sh, seed_list, seed_count, M1, cnt, firstpos, newid);
  KERNEL_CHECK();
  hipLaunchKernelGGL(k_assign, dim3(gP), dim3(256), 0, s, fg_list, fg_count, ptc, M1, sh, labc, cnt, firstpos);
  KERNEL_CHECK();
  const float big = (float)((double)Y * (double)X * (double)max_size_fraction);
  hipLaunchKernelGGL(k_rank_ids, dim3(F), dim3(256), 0, s, seed_list, seed_count, sh, cnt, firstpos, big, newid, ntot);
  KERNEL_CHECK();
  hipLaunchKernelGGL(k_apply_ids, dim3(gP), dim3(256), 0, s, fg_list, fg_count, labc, newid, sh, labels_tmp);
  KERNEL_CHECK();
  HIP_TRY(hipMemcpyAsync(n_labels_host, ntot, sizeof(int) * F, hipMemcpyDeviceToHost, s));
  { const int rcw = aliby_wait_stream(s); if (rcw) return rcw; }
  int n_obj = 0;
  {
    std::vector<int> nseeds((size_t)F);
    HIP_TRY(hipMemcpy(nseeds.data(), seed_count, sizeof(int) * F, hipMemcpyDeviceToHost));  // (the stream is idle: just waited)
    for (int f = 0; f < F; ++f)
      if (nseeds[f] >= SEEDS_PER_TILE - 1) {
        aliby_set_error("Segmentation produced %d seeds in one tile; uint16 cast unsafe.", nseeds[f]);
        return ALIBY_ERR_OVERFLOW;
      }
  }
  for (int f = 0; f < F; ++f) {
    if (n_labels_host[f] >= 65535) {
      aliby_set_error("Segmentation produced %d labels; uint16 cast unsafe.", n_labels_host[f]);
      return ALIBY_ERR_OVERFLOW;
    }
    n_obj += n_labels_host[f];
  }
  if (n_obj == 0) return ALIBY_OK;  // labels_out is all zero

  // ---- per-mask stages: object table, flow QC, hole fill ----------------------------------------
  int* offsets = new int[F + 1];
  offsets[0] = 0;
  for (int f = 0; f < F; ++f) offsets[f + 1] = offsets[f] + n_labels_host[f];
  aliby_object* tab_host = new aliby_object[n_obj];
  int rc = aliby_object_table(ctx, labels_tmp, F, Y, X, offsets, tab, tab_host, stream);
  if (rc) { delete[] offsets; delete[] tab_host; return rc; }
  int max_h = 0, max_w = 0;
  int* nit = new int[F];
  for (int f = 0; f < F; ++f) {
    int me = 0;
    for (int i = offsets[f]; i < offsets[f + 1]; ++i) {
      const int hh = tab_host[i].y1 - tab_host[i].y0, ww = tab_host[i].x1 - tab_host[i].x0;
      if (tab_host[i].area <= 0) continue;
      if (hh > max_h) max_h = hh;
      if (ww > max_w) max_w = ww;
      if (hh + 1 + ww + 1 > me) me = hh + 1 + ww + 1;
    }
    nit[f] = 2 * me;
  }
  HIP_TRY(hipMemcpyAsync(niter_tile, nit, sizeof(int) * F, hipMemcpyHostToDevice, s));
  // offsets for k_final_ids live in ctx scratch (aliby_object_table put them there)
  const int* d_off = (const int*)ctx->scratch;
  const size_t cells = ((size_t)(max_h + 2) * (max_w + 2) + 15) & ~(size_t)15;

  if (flow_threshold > 0.0f) {
    hipLaunchKernelGGL(k_zero_boxes, dim3(n_obj < 8192 ? n_obj : 8192), dim3(256), 0, s, tab, n_obj, Y, X, Tg);
    KERNEL_CHECK();
    QcArgs q;
    q.labels = labels_tmp; q.dP = dP; q.F = F; q.Y = Y; q.X = X; q.tab = tab; q.n_obj = n_obj;
    q.niter_tile = niter_tile; q.cap_cells = cells; q.Tg = Tg;
    const size_t need = cells * 17;
    if (need <= 128 * 1024) {
      q.gscratch = nullptr;
      if (need > 32 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_diffuse<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      // four waves per mask whatever its size: a sweep is a few dozen instructions per pixel behind one barrier, and one wave
      // walking 10+ rows per lane per sweep (the per-object default for small windows) left the SIMDs idle 70 % of the time —
      // 2.66 -> 1.8 ms for 16 k nuclei (ALIBY_DIFFUSE_BLOCK: 64 / 128 / 256 = 5.5 / 4.8 / 4.6 ms for the whole dynamics)
      const char* db = getenv("ALIBY_DIFFUSE_BLOCK");
      const int dblock = db && atoi(db) >= 64 && atoi(db) <= 256 ? (atoi(db) & ~63) : 256;
      hipLaunchKernelGGL((k_diffuse<false>), dim3(n_obj), dim3(dblock), need, s, q);
    } else {
      // ctx scratch holds the offsets in its first bytes: put the slabs after them
      const int g = n_obj < 256 ? n_obj : 256;
      const size_t head = align256(sizeof(int) * (size_t)(F + 1));
      // keep a private copy of the offsets: ensure_scratch may reallocate
      int rc2 = aliby_ensure_scratch(ctx, head + (size_t)g * need);
      if (rc2) { delete[] offsets; delete[] tab_host; delete[] nit; return rc2; }
      HIP_TRY(hipMemcpyAsync(ctx->scratch, offsets, sizeof(int) * (size_t)(F + 1), hipMemcpyHostToDevice, s));
      d_off = (const int*)ctx->scratch;
      q.gscratch = (unsigned char*)ctx->scratch + head;
      hipLaunchKernelGGL((k_diffuse<true>), dim3(g), dim3(256), 0, s, q);
    }
    KERNEL_CHECK();
    hipLaunchKernelGGL(k_flow_error, dim3(n_obj), dim3(aliby_pick_block((long long)max_h * max_w)), 0, s, q, flow_threshold, bad);
    KERNEL_CHECK();
  } else {
    HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int) * (size_t)n_obj, s));
  }
  hipLaunchKernelGGL(k_final_ids, dim3(F), dim3(1024), 0, s, tab, d_off, bad, min_size, newlabel, nfinal);
  KERNEL_CHECK();
  {
    FillArgs fa;
    fa.labels = labels_tmp; fa.F = F; fa.Y = Y; fa.X = X; fa.tab = tab; fa.n_obj = n_obj; fa.newlabel = newlabel;
    fa.cap_cells = cells; fa.out = labels_out;
    if (cells <= 128 * 1024) {
      fa.gscratch = nullptr;
      if (cells > 32 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_fill<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cells));
      hipLaunchKernelGGL((k_fill<false>), dim3(n_obj), dim3(aliby_pick_block((long long)max_h * max_w)), cells, s, fa);
    } else {
      const int g = n_obj < 256 ? n_obj : 256;
      const size_t head = align256(sizeof(int) * (size_t)(F + 1));
      int rc2 = aliby_ensure_scratch(ctx, head + (size_t)g * cells);
      if (rc2) { delete[] offsets; delete[] tab_host; delete[] nit; return rc2; }
      fa.gscratch = (unsigned char*)ctx->scratch + head;
      hipLaunchKernelGGL((k_fill<true>), dim3(g), dim3(256), 0, s, fa);
    }
    KERNEL_CHECK();
  }
  HIP_TRY(hipMemcpyAsync(n_labels_host, nfinal, sizeof(int) * F, hipMemcpyDeviceToHost, s));
  { const int rcw = aliby_wait_stream(s); if (rcw) return rcw; }
  delete[] offsets;
  delete[] tab_host;
  delete[] nit;
  return ALIBY_OK;
}

}  // extern "C"
