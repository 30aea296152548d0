/* A plain-C caller of libaliby_hip.so: no Python, no torch, only include/aliby_hip.h.
 *
 *   gcc -std=c11 -I include examples/c_abi_demo.c -o /tmp/c_abi_demo -L aliby_amd -laliby_hip \
 *       -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/aliby_amd -Wl,-rpath,/opt/rocm/lib
 *   /tmp/c_abi_demo [file.tif]
 *
 * Without a GPU it prints the ABI version, probes the TIFF (host code) and reports the loud failure of
 * aliby_ctx_create; with an MI355X it stages a small TCZYX stack, crops two tiles (one hanging over the corner: median
 * padding) and max-projects Z.  tests/test_cpu_host.py compiles and runs it here; tests/test_gpu_pipeline.py on the GPU box.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aliby_hip.h"

#define CHECK(call)                                                             \
  do {                                                                          \
    int rc__ = (call);                                                          \
    if (rc__ != ALIBY_OK) {                                                     \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc__, aliby_last_error());       \
      return 2;                                                                 \
    }                                                                           \
  } while (0)

int main(int argc, char** argv) {
  printf("abi %d\n", aliby_abi_version());
  if (argc > 1) {
    int64_t info[12];
    char desc[256];
    CHECK(aliby_tiff_probe(argv[1], info, desc, (int)sizeof desc));
    printf("tiff pages=%lld width=%lld height=%lld bits=%lld compression=%lld\n", (long long)info[0], (long long)info[1],
           (long long)info[2], (long long)info[3], (long long)info[6]);
  }
  aliby_ctx* ctx = NULL;
  if (aliby_ctx_create(0, &ctx) != ALIBY_OK) {
    printf("no context: %s\n", aliby_last_error());
    return 0; /* the library has no CPU fallback; on a box without a GPU this is the expected end */
  }
  enum { C = 1, Z = 3, Y = 40, X = 48, F = 2, TH = 16, TW = 16 };
  static uint16_t stack[C * Z * Y * X];
  for (int z = 0; z < Z; ++z)
    for (int y = 0; y < Y; ++y)
      for (int x = 0; x < X; ++x) stack[(z * Y + y) * X + x] = (uint16_t)(100 * z + y + x);
  const int32_t rects[F * 4] = {4, 6, TH, TW, 30, 40, TH, TW}; /* the second tile hangs over the corner: median padding */
  void *d_stack, *d_tiles, *d_rects, *d_planes;
  CHECK(aliby_malloc(ctx, sizeof stack, &d_stack));
  CHECK(aliby_malloc(ctx, (size_t)F * C * Z * TH * TW * 2, &d_tiles));
  CHECK(aliby_malloc(ctx, sizeof rects, &d_rects));
  CHECK(aliby_malloc(ctx, (size_t)F * C * TH * TW * 2, &d_planes));
  CHECK(aliby_memcpy_h2d(ctx, d_stack, stack, sizeof stack, NULL));
  int32_t flags[F] = {0, 0};
  CHECK(aliby_crop_pad_u16(ctx, (const uint16_t*)d_stack, C, Z, Y, X, rects, F, TH, TW, (uint16_t*)d_tiles, flags, NULL));
  CHECK(aliby_reduce_z(ctx, d_tiles, ALIBY_U16, (size_t)F * C, Z, (size_t)TH * TW, ALIBY_RED_MAX, d_planes, ALIBY_U16, NULL));
  static uint16_t planes[F * C * TH * TW];
  CHECK(aliby_memcpy_d2h(ctx, planes, d_planes, sizeof planes, NULL));
  CHECK(aliby_stream_sync(ctx, NULL));
  /* max over z of 100 z + y + x at tile 0's first pixel (4, 6) */
  printf("tile0[0,0]=%u (expect %u) nan_flags=%d,%d\n", planes[0], 100u * (Z - 1) + 4 + 6, flags[0], flags[1]);
  int ok = planes[0] == 100u * (Z - 1) + 4 + 6;
  CHECK(aliby_free(ctx, d_stack));
  CHECK(aliby_free(ctx, d_tiles));
  CHECK(aliby_free(ctx, d_rects));
  CHECK(aliby_free(ctx, d_planes));
  CHECK(aliby_ctx_destroy(ctx));
  printf(ok ? "ok\n" : "MISMATCH\n");
  return ok ? 0 : 1;
}
