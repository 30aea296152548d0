// objects.hip — label images -> compact per-object table; sequential relabelling.
//
// Replaces the reference's (N,Y,X) bool explosion
// (agora/utils/masks.py:35-37, called at extraction/extract.py:348,425) with a
// table of bboxes/areas keyed by (tile, label); rows are in the order of
// process_tree_masks' ind_masks (extract.py:276-281).
//
// All kernels here are streaming passes over u16 label planes: HBM-bound,
// 16-byte loads per lane, integer atomics only on run boundaries.
#include "common.h"

typedef unsigned short u16;
struct alignas(16) u16x8 { u16 v[8]; };

// ---------------------------------------------------------------------------
// per-tile maximum label
// ---------------------------------------------------------------------------
__global__ void k_label_max(const u16* __restrict__ labels, size_t plane, int* __restrict__ out) {
  __shared__ int red[16];
  const int f = blockIdx.y;
  const u16* base = labels + (size_t)f * plane;
  int m = 0;
  const size_t nvec = plane / 8;
  const u16x8* vb = reinterpret_cast<const u16x8*>(base);
  const bool aligned = ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
  if (aligned) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (size_t)gridDim.x * blockDim.x) {
      u16x8 v = vb[i];
#pragma unroll
      for (int k = 0; k < 8; ++k) m = max(m, (int)v.v[k]);
    }
    for (size_t i = nvec * 8 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane;
         i += (size_t)gridDim.x * blockDim.x)
      m = max(m, (int)base[i]);
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane;
         i += (size_t)gridDim.x * blockDim.x)
      m = max(m, (int)base[i]);
  }
  m = block_max_i32(m, red);
  if (threadIdx.x == 0 && m > 0) atomicMax(&out[f], m);
}

// ---------------------------------------------------------------------------
// object table
// ---------------------------------------------------------------------------
__global__ void k_table_init(aliby_object* __restrict__ tab, const int* __restrict__ offsets, int F) {
  const int f = blockIdx.y;
  const int lo = offsets[f], hi = offsets[f + 1];
  for (int i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += gridDim.x * blockDim.x) {
    aliby_object o;
    o.tile = f;
    o.label = i - lo + 1;
    o.y0 = INT_MAX; o.x0 = INT_MAX; o.y1 = 0; o.x1 = 0; o.area = 0; o.pad_ = 0;
    tab[i] = o;
  }
}

__device__ __forceinline__ void table_push(aliby_object* tab, int base, int nmax, int lab, int y,
                                           int xs, int xe /*exclusive*/) {
  if (lab == 0 || lab > nmax) return;
  aliby_object* o = tab + base + lab - 1;
  atomicMin(&o->y0, y);
  atomicMax(&o->y1, y + 1);
  atomicMin(&o->x0, xs);
  atomicMax(&o->x1, xe);
  atomicAdd(&o->area, xe - xs);
}

// one thread = 8 consecutive pixels of one row; runs of equal labels collapse to one push
__global__ void k_table_fill(const u16* __restrict__ labels, int Y, int X,
                             const int* __restrict__ offsets, aliby_object* __restrict__ tab) {
  const int f = blockIdx.z;
  const int y = blockIdx.y;
  const int base = offsets[f], nmax = offsets[f + 1] - offsets[f];
  if (nmax == 0) return;
  const u16* row = labels + ((size_t)f * Y + y) * X;
  const int chunks = (X + 7) / 8;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += gridDim.x * blockDim.x) {
    const int x0 = c * 8;
    u16 v[8];
    if (x0 + 8 <= X && ((reinterpret_cast<uintptr_t>(row + x0) & 15) == 0)) {
      u16x8 t = *reinterpret_cast<const u16x8*>(row + x0);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = t.v[k];
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (x0 + k < X) ? row[x0 + k] : (u16)0;
    }
    int cur = v[0], start = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      if (v[k] != cur) {
        table_push(tab, base, nmax, cur, y, x0 + start, x0 + k);
        cur = v[k];
        start = k;
      }
    }
    table_push(tab, base, nmax, cur, y, x0 + start, min(x0 + 8, X));
  }
}

// ---------------------------------------------------------------------------
// relabel_sequential
// ---------------------------------------------------------------------------
__global__ void k_mark_present(const u16* __restrict__ labels, size_t plane,
                               unsigned char* __restrict__ present) {
  const int f = blockIdx.y;
  const u16* base = labels + (size_t)f * plane;
  unsigned char* pf = present + (size_t)f * 65536;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane;
       i += (size_t)gridDim.x * blockDim.x) {
    u16 l = base[i];
    if (l) pf[l] = 1;
  }
}

// one block (1024 threads) per tile: exclusive scan of 65536 presence flags -> forward map
__global__ void k_build_map(const unsigned char* __restrict__ present, u16* __restrict__ map,
                            int* __restrict__ count) {
  __shared__ int part[1024];
  const int f = blockIdx.x;
  const unsigned char* pf = present + (size_t)f * 65536;
  u16* mf = map + (size_t)f * 65536;
  const int t = threadIdx.x;  // 64 labels per thread
  int c = 0;
  for (int k = 0; k < 64; ++k) c += pf[t * 64 + k] ? 1 : 0;
  part[t] = c;
  __syncthreads();
  // Hillis-Steele inclusive scan over 1024 partials
  for (int o = 1; o < 1024; o <<= 1) {
    int v = (t >= o) ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - c;  // exclusive prefix
  for (int k = 0; k < 64; ++k) {
    int l = t * 64 + k;
    if (pf[l]) { ++run; mf[l] = (u16)min(run, 65535); } else mf[l] = 0;
  }
  if (t == 1023) count[f] = part[1023];
}

__global__ void k_apply_map(u16* __restrict__ labels, size_t plane, const u16* __restrict__ map) {
  const int f = blockIdx.y;
  u16* base = labels + (size_t)f * plane;
  const u16* mf = map + (size_t)f * 65536;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane;
       i += (size_t)gridDim.x * blockDim.x) {
    u16 l = base[i];
    if (l) base[i] = mf[l];
  }
}

extern "C" {

int aliby_label_max(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X, int32_t* max_host,
                    void* stream) {
  ARG_CHECK(ctx && max_host, "NULL argument");
  ARG_CHECK(F >= 0 && Y >= 0 && X >= 0, "negative shape");
  if (F == 0) return ALIBY_OK;
  const size_t plane = (size_t)Y * X;
  if (plane == 0) { memset(max_host, 0, sizeof(int32_t) * F); return ALIBY_OK; }
  ARG_CHECK(labels != nullptr, "labels is NULL");
  int rc = aliby_ensure_scratch(ctx, sizeof(int) * (size_t)F);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  int* d = (int*)ctx->scratch;
  HIP_TRY(hipMemsetAsync(d, 0, sizeof(int) * (size_t)F, s));
  int bx = (int)((plane / 8 + 255) / 256);
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(k_label_max, dim3(bx, F), dim3(256), 0, s, labels, plane, d);
  KERNEL_CHECK();
  HIP_TRY(hipMemcpyAsync(max_host, d, sizeof(int) * (size_t)F, hipMemcpyDeviceToHost, s));
  { const int rcw = aliby_wait_stream(s); if (rcw) return rcw; }
  return ALIBY_OK;
}

int aliby_object_table(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                       const int32_t* offsets_host, aliby_object* table_dev,
                       aliby_object* table_host, void* stream) {
  ARG_CHECK(ctx && offsets_host, "NULL argument");
  ARG_CHECK(F >= 0 && Y >= 0 && X >= 0, "negative shape");
  if (F == 0) return ALIBY_OK;
  const int n_obj = offsets_host[F];
  ARG_CHECK(offsets_host[0] == 0 && n_obj >= 0, "offsets must be an exclusive prefix sum");
  if (n_obj == 0) return ALIBY_OK;
  ARG_CHECK(labels && table_dev, "NULL argument");
  int rc = aliby_ensure_scratch(ctx, sizeof(int) * (size_t)(F + 1));
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  int* d_off = (int*)ctx->scratch;
  HIP_TRY(hipMemcpyAsync(d_off, offsets_host, sizeof(int) * (size_t)(F + 1), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_table_init, dim3(16, F), dim3(256), 0, s, table_dev, d_off, F);
  KERNEL_CHECK();
  const int chunks = (X + 7) / 8;
  const int bx = (chunks + 127) / 128;
  hipLaunchKernelGGL(k_table_fill, dim3(bx, Y, F), dim3(128), 0, s, labels, Y, X, d_off, table_dev);
  KERNEL_CHECK();
  if (table_host) {
    HIP_TRY(hipMemcpyAsync(table_host, table_dev, sizeof(aliby_object) * (size_t)n_obj,
                           hipMemcpyDeviceToHost, s));
    { const int rcw = aliby_wait_stream(s); if (rcw) return rcw; }
  }
  return ALIBY_OK;
}

int aliby_relabel_sequential(aliby_ctx* ctx, uint16_t* labels, int F, int Y, int X, int32_t* n_host,
                             void* stream) {
  ARG_CHECK(ctx && n_host, "NULL argument");
  ARG_CHECK(F >= 0 && Y >= 0 && X >= 0, "negative shape");
  if (F == 0) return ALIBY_OK;
  const size_t plane = (size_t)Y * X;
  if (plane == 0) { memset(n_host, 0, sizeof(int32_t) * F); return ALIBY_OK; }
  ARG_CHECK(labels != nullptr, "labels is NULL");
  // scratch: present u8[F*65536] | map u16[F*65536] | count int[F]
  const size_t b_present = (size_t)F * 65536, b_map = (size_t)F * 65536 * 2, b_cnt = sizeof(int) * (size_t)F;
  int rc = aliby_ensure_scratch(ctx, b_present + b_map + b_cnt);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  unsigned char* present = (unsigned char*)ctx->scratch;
  u16* map = (u16*)(present + b_present);
  int* cnt = (int*)(present + b_present + b_map);
  HIP_TRY(hipMemsetAsync(present, 0, b_present, s));
  int bx = (int)((plane + 255) / 256);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(k_mark_present, dim3(bx, F), dim3(256), 0, s, labels, plane, present);
  KERNEL_CHECK();
  hipLaunchKernelGGL(k_build_map, dim3(F), dim3(1024), 0, s, present, map, cnt);
  KERNEL_CHECK();
  hipLaunchKernelGGL(k_apply_map, dim3(bx, F), dim3(256), 0, s, labels, plane, map);
  KERNEL_CHECK();
  HIP_TRY(hipMemcpyAsync(n_host, cnt, b_cnt, hipMemcpyDeviceToHost, s));
  { const int rcw = aliby_wait_stream(s); if (rcw) return rcw; }
  for (int f = 0; f < F; ++f) {
    if (n_host[f] >= 65535) {
      aliby_set_error("Segmentation produced %d labels; uint16 cast unsafe.", n_host[f]);
      return ALIBY_ERR_OVERFLOW;
    }
  }
  return ALIBY_OK;
}

}  // extern "C"
