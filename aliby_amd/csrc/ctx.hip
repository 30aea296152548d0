// ctx.hip — context, error reporting and plain memory helpers of the C ABI.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void aliby_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int aliby_ensure_scratch(aliby_ctx* ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return ALIBY_OK;
  if (ctx->scratch) HIP_TRY(hipFree(ctx->scratch));
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
  HIP_TRY(hipMalloc(&ctx->scratch, want));
  ctx->scratch_bytes = want;
  return ALIBY_OK;
}

// Wait for everything queued on `s`.  A blocked hipStreamSynchronize behind a deep queue wakes the host up ~0.7 ms
// late on this stack (interrupt-driven wait; measured as identical 683 us gaps after every mid-step readback), which
// is GPU idle time at each of the path's scalar readbacks; polling an event returns within microseconds.
int aliby_wait_stream(hipStream_t s) {
  static thread_local hipEvent_t ev[16] = {};
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  hipEvent_t& e = ev[dev & 15];
  if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(e, s));
  for (;;) {
    const hipError_t q = hipEventQuery(e);
    if (q == hipSuccess) return ALIBY_OK;
    if (q != hipErrorNotReady) {
      aliby_set_error("hipEventQuery failed: %s", hipGetErrorString(q));
      return ALIBY_ERR_HIP;
    }
    __builtin_ia32_pause();
  }
}

extern "C" {

int aliby_abi_version(void) { return ALIBY_ABI_VERSION; }

const char* aliby_last_error(void) { return g_err; }

int aliby_ctx_create(int device, aliby_ctx** out) {
  ARG_CHECK(out != nullptr, "out is NULL");
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (n <= 0) {
    aliby_set_error("no HIP device visible: libaliby_hip has no CPU fallback");
    return ALIBY_ERR_HIP;
  }
  ARG_CHECK(device >= 0 && device < n, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, device));
  aliby_ctx* c = new aliby_ctx();
  c->device = device;
  c->cu_count = p.multiProcessorCount;
  c->lds_bytes = (int)p.sharedMemPerBlock;
  c->hbm_bytes = p.totalGlobalMem;
  snprintf(c->name, sizeof(c->name), "%s (%s)", p.name, p.gcnArchName);
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  *out = c;
  return ALIBY_OK;
}

int aliby_ctx_destroy(aliby_ctx* ctx) {
  if (!ctx) return ALIBY_OK;
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  delete ctx;
  return ALIBY_OK;
}

int aliby_device_info(aliby_ctx* ctx, int* cu_count, int* lds_bytes, size_t* hbm_bytes, char* name,
                      int name_len) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (cu_count) *cu_count = ctx->cu_count;
  if (lds_bytes) *lds_bytes = ctx->lds_bytes;
  if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
  if (name && name_len > 0) {
    strncpy(name, ctx->name, (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  return ALIBY_OK;
}

int aliby_malloc(aliby_ctx* ctx, size_t bytes, void** dptr) {
  ARG_CHECK(ctx && dptr, "NULL argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
  return ALIBY_OK;
}

int aliby_free(aliby_ctx* ctx, void* dptr) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (dptr) HIP_TRY(hipFree(dptr));
  return ALIBY_OK;
}

int aliby_memcpy_h2d(aliby_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream) {
  ARG_CHECK(ctx && (dst || !bytes) && (src || !bytes), "NULL argument");
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  return ALIBY_OK;
}

int aliby_memcpy_d2h(aliby_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream) {
  ARG_CHECK(ctx && (dst || !bytes) && (src || !bytes), "NULL argument");
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  { const int rcw = aliby_wait_stream(as_stream(stream)); if (rcw) return rcw; }
  return ALIBY_OK;
}

int aliby_memset(aliby_ctx* ctx, void* dst, int value, size_t bytes, void* stream) {
  ARG_CHECK(ctx && (dst || !bytes), "NULL argument");
  HIP_TRY(hipMemsetAsync(dst, value, bytes, as_stream(stream)));
  return ALIBY_OK;
}

int aliby_stream_sync(aliby_ctx* ctx, void* stream) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  { const int rcw = aliby_wait_stream(as_stream(stream)); if (rcw) return rcw; }
  return ALIBY_OK;
}

}  // extern "C"
