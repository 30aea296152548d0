// track.hip — frame-to-frame IoU stitching of label images, batched over tiles.
//
// Stands in for the reference's `stitch` tracker (src/aliby/track/trackers.py:14-90: update_labels +
// cellpose.utils.stitch3D on a (previous, current) pair per tile), which cannot be imported as shipped (it needs
// agora.utils.masks.labels_from_masks).  The stitch3D rule for one pair, restated in oracle/track_restated.py:
//   iou[c, p] = |c ∩ p| / (|c| + |p| - |c ∩ p|) in float64; entries below the threshold are dropped; every previous
//   object keeps only the current object(s) with its column maximum; a current object takes the previous TRACKED label
//   of its row maximum (first maximum = smallest tracked label) or, if nothing is left, a new label.
//
// MI355X shape: the (N_cur x N_prev) overlap matrix is never formed.  One workgroup per current object scans its
// bounding box once and counts the previous labels under its mask in a small LDS hash table (integer atomics), so the
// two label planes are read once (HBM-bound, 2·P·2 bytes per tile); at most floor(1/threshold) previous objects can
// pass the threshold for one current object; candidates go to a fixed 16-slot list (more than 16 previous objects above the
// threshold under one mask — only possible for thresholds below 1/16 — is reported as an error, never dropped silently).  Column maxima are one 64-bit
// integer atomicMax per candidate on the bit pattern of the (positive) double.  New labels are handed out in current
// label order by one workgroup per tile.
#include "common.h"

typedef unsigned short u16;

#define TRK_SLOTS 1024  // LDS hash slots per object (distinct previous labels under one mask)
#define TRK_K 16        // candidates kept per current object (a 17th previous object above the threshold under one mask is reported)

struct TrackArgs {
  const u16* prev;
  const u16* cur;
  int F, Y, X;
  const aliby_object* ctab;
  int n_cur;
  const aliby_object* ptab;
  const int* poff;      // [F+1] row offsets of the previous table
  const int* coff;      // [F+1] row offsets of the current table
  const int* ptracked;  // tracked label per previous row, or NULL (= own label)
  double thr;
  int* ncand;                   // [n_cur]
  int* cand_row;                // [n_cur * TRK_K] previous row
  double* cand_iou;             // [n_cur * TRK_K]
  unsigned long long* colmax;   // [n_prev] bit pattern of the column maximum
  int* overflow;
};

__global__ __launch_bounds__(256) void k_track_candidates(TrackArgs a) {
  __shared__ unsigned int keys[TRK_SLOTS];
  __shared__ unsigned int cnt[TRK_SLOTS];
  __shared__ int nc;
  const int tid = threadIdx.x;
  const size_t plane = (size_t)a.Y * a.X;
  for (int oi = blockIdx.x; oi < a.n_cur; oi += gridDim.x) {
    const aliby_object o = a.ctab[oi];
    __syncthreads();
    if (tid == 0) { nc = 0; a.ncand[oi] = 0; }
    if (o.area <= 0) continue;
    for (int k = tid; k < TRK_SLOTS; k += blockDim.x) { keys[k] = 0u; cnt[k] = 0u; }
    __syncthreads();
    const u16* cur = a.cur + (size_t)o.tile * plane;
    const u16* prev = a.prev + (size_t)o.tile * plane;
    const int h = o.y1 - o.y0, w = o.x1 - o.x0, npix = h * w;
    const u16 L = (u16)o.label;
    for (int i = tid; i < npix; i += blockDim.x) {
      const size_t idx = (size_t)(o.y0 + i / w) * a.X + (o.x0 + i % w);
      if (cur[idx] != L) continue;
      const unsigned int lb = prev[idx];
      if (lb == 0u) continue;
      unsigned int s = (lb * 40503u) & (TRK_SLOTS - 1);
      int probes = 0;
      for (;; s = (s + 1) & (TRK_SLOTS - 1)) {
        const unsigned int old = atomicCAS(&keys[s], 0u, lb);
        if (old == 0u || old == lb) { atomicAdd(&cnt[s], 1u); break; }
        if (++probes >= TRK_SLOTS) { atomicExch(a.overflow, 1); break; }
      }
    }
    __syncthreads();
    const int p0 = a.poff[o.tile], pn = a.poff[o.tile + 1] - p0;
    for (int k = tid; k < TRK_SLOTS; k += blockDim.x) {
      const unsigned int lb = keys[k];
      if (lb == 0u || (int)lb > pn) continue;  // a previous label beyond its table cannot happen; guarded anyway
      const int prow = p0 + (int)lb - 1;
      const double ov = (double)cnt[k];
      const double iou = ov / ((double)o.area + (double)a.ptab[prow].area - ov);
      if (!(iou >= a.thr) || !(iou > 0.0)) continue;
      const int slot = atomicAdd(&nc, 1);
      if (slot >= TRK_K) { atomicExch(a.overflow, 2); continue; }
      a.cand_row[(size_t)oi * TRK_K + slot] = prow;
      a.cand_iou[(size_t)oi * TRK_K + slot] = iou;
      atomicMax(&a.colmax[prow], (unsigned long long)__double_as_longlong(iou));
    }
    __syncthreads();
    if (tid == 0) a.ncand[oi] = min(nc, TRK_K);
  }
}

// matched[oi] = tracked label of the winning previous object, 0 = unmatched (or label absent from the frame)
__global__ void k_track_assign(TrackArgs a, int* __restrict__ matched) {
  const int oi = blockIdx.x * blockDim.x + threadIdx.x;
  if (oi >= a.n_cur) return;
  int best_label = 0;
  double best = 0.0;
  const int n = a.ncand[oi];
  for (int k = 0; k < n; ++k) {
    const int prow = a.cand_row[(size_t)oi * TRK_K + k];
    const double iou = a.cand_iou[(size_t)oi * TRK_K + k];
    if ((unsigned long long)__double_as_longlong(iou) != a.colmax[prow]) continue;  // another current object owns this column
    const int tl = a.ptracked ? a.ptracked[prow] : a.ptab[prow].label;
    if (iou > best || (iou == best && tl < best_label)) { best = iou; best_label = tl; }
  }
  matched[oi] = best_label;
}

// one workgroup per tile: unmatched objects get max_label+1, +2, ... in current label order
__global__ __launch_bounds__(256) void k_track_newids(TrackArgs a, const int* __restrict__ matched,
                                                      const int* __restrict__ max_in, int* __restrict__ tracked,
                                                      int* __restrict__ max_out) {
  __shared__ int red[8];
  __shared__ int part[256];
  __shared__ int flags[256];
  const int f = blockIdx.x, tid = threadIdx.x;
  const int c0 = a.coff[f], c1 = a.coff[f + 1], p0 = a.poff[f], p1 = a.poff[f + 1];
  int m = 0;
  for (int r = p0 + tid; r < p1; r += blockDim.x) m = max(m, a.ptracked ? a.ptracked[r] : a.ptab[r].label);
  int next = max(block_max_i32(m, red), max_in ? max_in[f] : 0);
  for (int base = c0; base < c1; base += blockDim.x) {
    const int r = base + tid;
    const bool live = r < c1 && a.ctab[r].area > 0;
    const int need = (live && matched[r] == 0) ? 1 : 0;
    __syncthreads();
    flags[tid] = need;
    __syncthreads();
    block_inclusive_scan(flags, (int)blockDim.x, part);
    if (r < c1) tracked[r] = live ? (need ? next + flags[tid] : matched[r]) : 0;
    __syncthreads();
    next += flags[blockDim.x - 1];
    __syncthreads();
  }
  if (tid == 0) max_out[f] = next;
}

extern "C" int aliby_track_stitch(aliby_ctx* ctx, const uint16_t* prev, const uint16_t* cur, int F, int Y, int X,
                                  const aliby_object* cur_table_dev, const int32_t* cur_offsets_host,
                                  const aliby_object* prev_table_dev, const int32_t* prev_offsets_host,
                                  const int32_t* prev_tracked_dev, const int32_t* max_label_in_host, double threshold,
                                  int32_t* cur_tracked_dev, int32_t* max_label_out_host, void* stream) {
  ARG_CHECK(ctx && cur_offsets_host && prev_offsets_host && max_label_out_host, "NULL argument");
  ARG_CHECK(F > 0 && Y > 0 && X > 0, "bad shape");
  ARG_CHECK(threshold >= 0.01 && threshold <= 1.0, "stitch threshold must lie in [0.01, 1]");
  const int n_cur = cur_offsets_host[F], n_prev = prev_offsets_host[F];
  ARG_CHECK(n_cur >= 0 && n_prev >= 0 && cur_offsets_host[0] == 0 && prev_offsets_host[0] == 0, "offsets must be exclusive prefix sums");
  ARG_CHECK(n_cur == 0 || (cur && prev && cur_table_dev && cur_tracked_dev), "NULL argument");
  ARG_CHECK(n_prev == 0 || prev_table_dev, "NULL argument");
  hipStream_t s = as_stream(stream);
  // device scratch: offsets (2(F+1)), max_in (F), max_out (F), overflow (1), matched (n_cur), ncand (n_cur), cand rows, ious, colmax
  const size_t ints = (size_t)2 * (F + 1) + 2 * (size_t)F + 1 + 2 * (size_t)n_cur + (size_t)n_cur * TRK_K;
  const size_t off_d = (ints * sizeof(int) + 15) & ~(size_t)15;
  const size_t bytes = off_d + sizeof(double) * (size_t)n_cur * TRK_K + sizeof(unsigned long long) * (size_t)(n_prev > 0 ? n_prev : 1);
  int rc = aliby_ensure_scratch(ctx, bytes);
  if (rc) return rc;
  int* base = (int*)ctx->scratch;
  int* d_coff = base;
  int* d_poff = d_coff + (F + 1);
  int* d_maxin = d_poff + (F + 1);
  int* d_maxout = d_maxin + F;
  int* d_over = d_maxout + F;
  int* d_matched = d_over + 1;
  int* d_ncand = d_matched + n_cur;
  int* d_crow = d_ncand + n_cur;
  double* d_ciou = (double*)((char*)ctx->scratch + off_d);
  unsigned long long* d_colmax = (unsigned long long*)(d_ciou + (size_t)n_cur * TRK_K);
  HIP_TRY(hipMemcpyAsync(d_coff, cur_offsets_host, sizeof(int) * (F + 1), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_poff, prev_offsets_host, sizeof(int) * (F + 1), hipMemcpyHostToDevice, s));
  if (max_label_in_host) HIP_TRY(hipMemcpyAsync(d_maxin, max_label_in_host, sizeof(int) * F, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(d_over, 0, sizeof(int), s));
  HIP_TRY(hipMemsetAsync(d_colmax, 0, sizeof(unsigned long long) * (size_t)(n_prev > 0 ? n_prev : 1), s));
  TrackArgs a;
  a.prev = prev; a.cur = cur; a.F = F; a.Y = Y; a.X = X; a.ctab = cur_table_dev; a.n_cur = n_cur; a.ptab = prev_table_dev;
  a.poff = d_poff; a.coff = d_coff; a.ptracked = prev_tracked_dev; a.thr = threshold; a.ncand = d_ncand; a.cand_row = d_crow;
  a.cand_iou = d_ciou; a.colmax = d_colmax; a.overflow = d_over;
  if (n_cur > 0) {
    hipLaunchKernelGGL(k_track_candidates, dim3(n_cur), dim3(64), 0, s, a);
    KERNEL_CHECK();
    hipLaunchKernelGGL(k_track_assign, dim3((n_cur + 255) / 256), dim3(256), 0, s, a, d_matched);
    KERNEL_CHECK();
  }
  hipLaunchKernelGGL(k_track_newids, dim3(F), dim3(256), 0, s, a, d_matched, max_label_in_host ? d_maxin : nullptr, cur_tracked_dev,
                     d_maxout);
  KERNEL_CHECK();
  int over = 0;
  HIP_TRY(hipMemcpyAsync(max_label_out_host, d_maxout, sizeof(int) * F, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&over, d_over, sizeof(int), hipMemcpyDeviceToHost, s));
  { const int rcw = aliby_wait_stream(s); if (rcw) return rcw; }
  if (over) {
    aliby_set_error("track_stitch: more than %d previous objects under one current object", over == 1 ? TRK_SLOTS : TRK_K);
    return ALIBY_ERR_TOO_LARGE;
  }
  return ALIBY_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Labels through a per-object table: out[t, p] = lut[offsets[t] + in[t, p] - 1] (0 stays 0).  Used to write the stitched
// (tracked) labels of a Z-stack's planes back into the planes (the do_3D branch of the segmenter, segment/dispatch.py).
__global__ void k_apply_lut(const u16* __restrict__ in, size_t plane, int T, const int* __restrict__ offsets,
                            const int* __restrict__ lut, u16* __restrict__ out, int* __restrict__ over) {
  const size_t total = plane * (size_t)T;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned lb = in[i];
    unsigned v = 0;
    if (lb) {
      const int t = (int)(i / plane);
      const int row = offsets[t] + (int)lb - 1;
      const int m = row < offsets[t + 1] ? lut[row] : 0;
      if (m >= 65535) { *over = m; v = 65535; } else v = (unsigned)(m > 0 ? m : 0);
    }
    out[i] = (u16)v;
  }
}

extern "C" int aliby_labels_apply_lut(aliby_ctx* ctx, const uint16_t* labels_in, int T, int Y, int X, const int32_t* offsets_host,
                                      const int32_t* lut_dev, uint16_t* labels_out, void* stream) {
  ARG_CHECK(ctx && labels_in && labels_out && offsets_host, "NULL argument");
  ARG_CHECK(T > 0 && Y > 0 && X > 0 && offsets_host[0] == 0 && offsets_host[T] >= 0, "bad shape / offsets");
  ARG_CHECK(offsets_host[T] == 0 || lut_dev, "NULL table");
  hipStream_t s = as_stream(stream);
  int rc = aliby_ensure_scratch(ctx, sizeof(int) * (size_t)(T + 2));
  if (rc) return rc;
  int* d_off = (int*)ctx->scratch;
  int* d_over = d_off + (T + 1);
  HIP_TRY(hipMemcpyAsync(d_off, offsets_host, sizeof(int) * (size_t)(T + 1), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(d_over, 0, sizeof(int), s));
  const size_t total = (size_t)T * Y * X;
  const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(k_apply_lut, dim3(grid), dim3(256), 0, s, labels_in, (size_t)Y * X, T, d_off, lut_dev, labels_out, d_over);
  KERNEL_CHECK();
  int over = 0;
  HIP_TRY(hipMemcpyAsync(&over, d_over, sizeof(int), hipMemcpyDeviceToHost, s));
  { const int rcw = aliby_wait_stream(s); if (rcw) return rcw; }
  if (over) {
    aliby_set_error("Segmentation produced %d labels; uint16 cast unsafe.", over);
    return ALIBY_ERR_OVERFLOW;
  }
  return ALIBY_OK;
}
