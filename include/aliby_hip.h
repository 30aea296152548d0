/*
 * aliby_hip.h — C ABI of libaliby_hip.so (MI355X / gfx950).
 *
 * The reference (afermg/aliby) has no FFI: its seam is the Python step-callable
 * protocol chosen by step-name prefix (src/aliby/pipe.py:47-72).
 * This header is the boundary the Python step callables in `aliby_amd/` sit on;
 * every entry point cites the reference interface whose arithmetic it replaces.
 *
 * Conventions
 *   - every function returns an int status (ALIBY_OK == 0); on failure
 *     aliby_last_error() returns a thread-local message.  The ctypes shim maps
 *     codes to the Python exception types the reference raises
 *     (ValueError / OverflowError / Exception, SURVEY.md §8b "Errors").
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - pointers marked [dev] are device pointers (hipMalloc / torch), [host] host.
 *   - images are row-major; a "tile" is one [Y,X] label image plus C planes.
 *   - pixel planes are ALIBY_U16 (uint16) or ALIBY_F32 (float).
 *   - feature outputs are float64, written at out[row*ld + col].
 *   - nothing here allocates inside a launch path: workspaces are explicit.
 */
#ifndef ALIBY_HIP_H
#define ALIBY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALIBY_ABI_VERSION 1

enum {
  ALIBY_OK = 0,
  ALIBY_ERR_INVALID = 1,      /* bad argument             -> ValueError    */
  ALIBY_ERR_OVERFLOW = 2,     /* labels >= 65535          -> OverflowError */
  ALIBY_ERR_HIP = 3,          /* HIP runtime failure      -> RuntimeError  */
  ALIBY_ERR_TOO_LARGE = 4,    /* object exceeds workspace -> RuntimeError  */
  ALIBY_ERR_UNSUPPORTED = 5   /* e.g. non-ufunc reducer   -> Exception     */
};

enum { ALIBY_U16 = 0, ALIBY_F32 = 1,
       ALIBY_U64 = 2, ALIBY_F64 = 3 /* output dtypes of aliby_reduce_z only: NumPy's own result types */,
       ALIBY_U8W = 4 /* uint16 storage holding uint8 / bool pixels: aliby_features_texture only (their grey level is the value
                        itself, skimage.util.img_as_ubyte leaves uint8 alone); every other entry takes such planes as ALIBY_U16 */ };

/* reduce_z operators — extraction/core/functions/loaders.py:110-127 ("max","add","div";
 * "mean"/"median" are not ufuncs and raise in distributors.py:20-24). */
enum { ALIBY_RED_MAX = 0, ALIBY_RED_ADD = 1, ALIBY_RED_DIV = 2 };

typedef struct aliby_ctx aliby_ctx;

/* One row per object, in (tile, label) order — the row order of
 * process_tree_masks' ind_masks (extraction/extract.py:276-281). */
typedef struct aliby_object {
  int32_t tile;   /* index into the F tiles                               */
  int32_t label;  /* 1..max(labels[tile])                                 */
  int32_t y0, x0; /* bbox, inclusive; y0>y1 when the label is absent      */
  int32_t y1, x1; /* bbox, exclusive                                      */
  int32_t area;   /* pixel count (0 when absent)                          */
  int32_t pad_;
} aliby_object;

/* ---- context / errors ------------------------------------------------- */
int aliby_abi_version(void);
const char* aliby_last_error(void);
int aliby_ctx_create(int device, aliby_ctx** out);
int aliby_ctx_destroy(aliby_ctx* ctx);
int aliby_device_info(aliby_ctx* ctx, int* cu_count, int* lds_bytes, size_t* hbm_bytes,
                      char* name, int name_len);

/* plain memory helpers for callers without torch */
int aliby_malloc(aliby_ctx* ctx, size_t bytes, void** dptr);
int aliby_free(aliby_ctx* ctx, void* dptr);
int aliby_memcpy_h2d(aliby_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes, void* stream);
int aliby_memcpy_d2h(aliby_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes, void* stream);
int aliby_memset(aliby_ctx* ctx, void* dst_dev, int value, size_t bytes, void* stream);
int aliby_stream_sync(aliby_ctx* ctx, void* stream);

/* ---- a4: tile stager --------------------------------------------------- */
/* Tiler.get_fczyx / get_tp_channel / if_out_of_bounds_pad
 * (src/aliby/tile/tiler.py:309-366,601-650; Tile.as_range tiles.py:151-166).
 * stack [dev] is one time point [C,Z,Y,X]; rects [host] is F x (y0,x0,h,w) in
 * image coordinates and may leave the image.  out [dev] is [F,C,Z,h,w].
 * Partly outside -> np.pad(mode="median") semantics (per-line medians, rounded
 * for integers); more than 25% outside along an axis -> the tile is NaN and
 * flags[f] = 1 (out_f32 only).  All tiles share (h,w). */
int aliby_crop_pad_u16(aliby_ctx* ctx, const uint16_t* stack, int C, int Z, int Y, int X,
                       const int32_t* rects, int F, int h, int w, uint16_t* out,
                       int32_t* nan_flags_host, void* stream);

/* ---- a5: reduce_z ------------------------------------------------------- */
/* reduce_z(pixels, ufunc, axis) (extraction/core/functions/distributors.py:6-24):
 * in [dev] is [outer, Z, inner]; out [dev] is [outer, inner]; ufunc.reduce order
 * (left fold over Z).  max keeps the input dtype.  For u16 input NumPy's results are
 * uint64 for add.reduce (small unsigned ints are promoted, no wrap) and float64 for
 * divide.reduce: out_dtype ALIBY_U64 / ALIBY_F64 give exactly those; out_dtype
 * ALIBY_F32 gives the same values narrowed to f32, which is what the feature kernels
 * consume (sums exact while < 2^24, i.e. Z <= 256).  f32 input: f32 left fold. */
int aliby_reduce_z(aliby_ctx* ctx, const void* in, int dtype, size_t outer, int Z, size_t inner,
                   int op, void* out, int out_dtype, void* stream);

/* ---- a7/a8: object table (replaces transform_2d_to_3d) ------------------ */
/* process_tree_masks enumerates labels 1..mask.max() per tile
 * (extract.py:276-281); transform_2d_to_3d (agora/utils/masks.py:35-37) explodes
 * them to (N,Y,X) bool.  Here the label image stays as is and a compact table of
 * per-object bboxes/areas is built instead.
 * Step 1: per-tile maximum label -> max_host[F] (synchronises `stream`). */
int aliby_label_max(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                    int32_t* max_host, void* stream);
/* Step 2: fill table [dev] (n_obj rows = sum(max_host)), offsets_host[F+1] is the
 * exclusive prefix sum of max_host.  Also copies the table to table_host if non-NULL
 * (synchronises). */
int aliby_object_table(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                       const int32_t* offsets_host, aliby_object* table_dev,
                       aliby_object* table_host, void* stream);

/* relabel_sequential (skimage, used at segment/dispatch.py:223 and extract.py:496):
 * labels [dev, in place] renumbered 1..n in increasing original-label order,
 * n_host[F] receives the per-tile count.  Raises ALIBY_ERR_OVERFLOW if n >= 65535
 * (dispatch.py:230-233). */
int aliby_relabel_sequential(aliby_ctx* ctx, uint16_t* labels, int F, int Y, int X,
                             int32_t* n_host, void* stream);

/* ---- a6: segment closure pre-processing ----------------------------------- */
/* pixels[:, channel_to_segment] then max over Z (segment/dispatch.py:192,199-206):
 * pixels [dev] [F,C,Z,Y,X] uint16 -> out [dev] [F,Y,X]. */
int aliby_select_project_u16(aliby_ctx* ctx, const uint16_t* pixels, int F, int C, int Z, int Y, int X,
                             int channel, uint16_t* out, void* stream);
/* cellpose.transforms.normalize99 (normalize=True at dispatch.py:203,212): per image
 * (x - p_lower) / (p_upper - p_lower) as float32, zeros when the range is < 1e-3; percentiles are exact
 * order statistics with linear interpolation (numpy.percentile).  percentiles_dev [F,2] receives them. */
int aliby_normalize99_u16(aliby_ctx* ctx, const uint16_t* img, int F, int Y, int X, double lower,
                          double upper, float* out, double* percentiles_dev, void* stream);
/* cellpose.transforms.make_tiles on the zero-padded image: tiles [dev] [F*ny*nx, nchan, by, bx]; channel 0
 * is the image, the others are zero (grayscale input of the 2-channel U-Net). */
int aliby_make_tiles(aliby_ctx* ctx, const float* img, int F, int Y, int X, int ypad1, int xpad1, int Ly,
                     int Lx, int by, int bx, int ny, int nx, const int32_t* ystart_dev,
                     const int32_t* xstart_dev, int nchan, float* tiles, void* stream);
/* cellpose.transforms.average_tiles + un-padding: network output tiles [F*ny*nx, 3, by, bx] -> dP [F,2,Y,X],
 * cellprob [F,Y,X], weighted by taper_dev [by,bx] and accumulated in tile order in float32. */
int aliby_average_tiles(aliby_ctx* ctx, const float* ytiles, int F, int Y, int X, int ypad1, int xpad1,
                        int Ly, int Lx, int by, int bx, int ny, int nx, const int32_t* ystart_dev,
                        const int32_t* xstart_dev, const float* taper_dev, float* dP, float* cellprob,
                        void* stream);

/* Fused pointwise stage between two convolutions of the U-Net (bf16, NHWC): sum = A (+ B) (+ bias[c]), A/B
 * optionally read through a 2x nearest upsample; act = relu?(scale[c]*sum + shift[n,c] or shift[c]).
 * shift_per_sample: 0 = one shift row for all samples, 1 = contiguous [N, C], >1 = row stride in floats.  SUM
 * and/or ACT are written.  Replaces the eager conv-bias / BatchNorm / ReLU / add / style-add / upsample
 * passes of the network that `model.eval` (segment/dispatch.py:208-215) runs.  Only the A/B fallback path
 * (`FusedUNet(mfma_levels=())`, kept for the tests) uses it: the product forward runs the aliby_nn_conv* units
 * below, which carry these stages in their prologue / epilogue. */
int aliby_nn_fused_act_bf16(aliby_ctx* ctx, const void* A, const void* B, void* SUM, void* ACT,
                            const float* bias, const float* scale, const float* shift, int N, int H, int W,
                            int C, int upA, int upB, int relu, int shift_per_sample, void* stream);
/* One convolution unit of the U-Net on the matrix cores (bf16 NHWC, fp32 accumulate), hand-written MFMA
 * implicit GEMM with the pointwise stages fused around it:
 *   OUT[n,y,x,:] = conv3x3( relu(scale[c]*IN[n, y>>in_up, x>>in_up, c] + shift[n or 0, c]), zero padded )
 *                  + bias[:] + RES[n, y>>res_up, x>>res_up, :]
 * (bias, RES may be NULL; shift_per_sample as for the fused pointwise stage).  This is cellpose's `batchconv` / `batchconvstyle` (BatchNorm -> ReLU -> Conv2d)
 * plus the residual / skip / style adds of `resdown` / `resup`, i.e. the network `model.eval`
 * (segment/dispatch.py:208-215) runs.  wpk comes from the packing call below.  IN may be a channel slice
 * [in_channel0, in_channel0 + CIN) of a tensor with in_channels channels (0 = exactly CIN): a convolution
 * over more input channels than one launch holds is split along K, each launch adding to the previous one
 * through RES.  OUT / RES / pool_out may likewise be the channel slice [out_channel0, out_channel0 + COUT) of
 * tensors with out_channels channels (0 = exactly COUT): wide outputs are produced slice by slice.  pool_out, when not NULL, also receives max_pool2d(OUT, 2, 2) as bf16 NHWC [N,H/2,W/2,COUT]
 * (cellpose's `downsample` maxpool, fused into the epilogue of the block's last convolution).
 * Supported (CIN, COUT, in_up): (32,32,0) (32,64,0) (64,64,0) (64,32,1) (64,64,1) (64,128,0) (64,128,1);
 * anything else returns ALIBY_ERR_UNSUPPORTED (wider layers: K/N slices, see above). */
int aliby_nn_conv3x3_bf16(aliby_ctx* ctx, const void* in, const void* wpk, void* out, const float* scale,
                          const float* shift, int shift_per_sample, const float* bias, const void* res,
                          int res_up, int N, int H, int W, int CIN, int COUT, int in_up, int in_channels,
                          int in_channel0, int out_channels, int out_channel0, void* pool_out, void* stream);
/* Frame-to-frame IoU stitching of label images, batched over tiles: the working stand-in for the reference's `stitch`
 * tracker (src/aliby/track/trackers.py:14-90 = update_labels + cellpose.utils.stitch3D on a (previous, current) pair
 * per tile; not importable as shipped).  prev / cur are uint16 [F,Y,X]; the tables and their host offsets come from
 * aliby_object_table (rows in (tile, label) order); prev_tracked_dev[row] is the tracked label the previous frame's
 * object already carries (NULL = its own label); max_label_in_host[tile] the running maximum (NULL = 0).
 * cur_tracked_dev[row] receives the tracked label of every current object (0 for labels absent from the frame):
 * the previous tracked label of the best IoU >= threshold partner that is also that partner's best, else
 * max_label + 1, + 2, ... in current label order; max_label_out_host[tile] the new running maximum.
 * threshold in [0.01, 1] (cellpose's default is 0.25; the reference's 3-D branch passes 0.01, dispatch.py:195); more than 16
 * previous objects above the threshold under one current mask is an error. */
int aliby_track_stitch(aliby_ctx* ctx, const uint16_t* prev, const uint16_t* cur, int F, int Y, int X,
                       const aliby_object* cur_table_dev, const int32_t* cur_offsets_host,
                       const aliby_object* prev_table_dev, const int32_t* prev_offsets_host,
                       const int32_t* prev_tracked_dev, const int32_t* max_label_in_host, double threshold,
                       int32_t* cur_tracked_dev, int32_t* max_label_out_host, void* stream);

/* The same unit for the deep levels of the network (128 / 256 channels at 56 x 56 and 28 x 28) as ONE launch: the whole
 * K = 9 * CIN reduction is accumulated in fp32 registers (K loop over 64-channel slices inside the kernel, weights streamed
 * from the packed array), instead of K/N-slice launches of aliby_nn_conv3x3_bf16 adding bf16 partial sums through HBM.
 * Arguments as aliby_nn_conv3x3_bf16 (full tensors, no channel slices); wpk = aliby_nn_pack_conv3x3_bf16 over the full CIN.
 * Supported: CIN in {64, 128, 256}, COUT a multiple of 128, W <= 56, (H + 1) * (W + 2) >= 226 + 2 * (W + 2) when the shift
 * is per sample; anything else returns ALIBY_ERR_UNSUPPORTED (the caller then uses the slice launches).
 * Replaces: cellpose `batchconv` / `batchconvstyle` at the deep levels of the network `model.eval` runs
 * (segment/dispatch.py:208-215). */
int aliby_nn_conv3x3_deep_bf16(aliby_ctx* ctx, const void* in, const void* wpk, void* out, const float* scale,
                               const float* shift, int shift_per_sample, const float* bias, const void* res, int res_up,
                               int N, int H, int W, int CIN, int COUT, int in_up, void* stream);

/* The same unit on v_mfma_f32_16x16x32_bf16 (the shape the chip holds a higher clock on: nn_conv_deep.hip).  Same arguments and
 * meaning; `wpk16` is packed by aliby_nn_pack_conv3x3_deep16_bf16 (w_oihw float32 [COUT, CIN, 3, 3], COUT a multiple of 32, CIN
 * of 64 -> COUT * CIN * 9 bf16).  Accumulates 32 input channels per instruction: agrees with aliby_nn_conv3x3_deep_bf16 to fp32
 * rounding, not bit for bit. */
int aliby_nn_conv3x3_deep16_bf16(aliby_ctx* ctx, const void* in, const void* wpk16, void* out, const float* scale,
                               const float* shift, int shift_per_sample, const float* bias, const void* res, int res_up,
                               int N, int H, int W, int CIN, int COUT, int in_up, void* stream);
int aliby_nn_pack_conv3x3_deep16_bf16(aliby_ctx* ctx, const float* w_oihw, int COUT, int CIN, void* wpk16, void* stream);
/* Diagnostics for the deep kernel: stamps_dev [8 tiles][32] uint64 shader-clock stamps of wave 0 of workgroup 0 (0 tile start,
 * 1 set-up done, then per K slice s: 2+4s after the first barrier, 3+4s prologue written, 4+4s after the second barrier,
 * 5+4s MFMA loop done; 2+4S epilogue done).  NULL switches it off (the default). */
int aliby_debug_conv_deep_trace(aliby_ctx* ctx, void* stamps_dev);
/* max_pool2d(IN, 2, 2) on bf16 NHWC [N,H,W,C] -> [N,H/2,W/2,C] (cellpose `downsample.maxpool` between the levels whose
 * last convolution runs on the deep kernel; elsewhere the pooled tensor is an epilogue of aliby_nn_conv3x3_bf16). */
int aliby_nn_maxpool2_bf16(aliby_ctx* ctx, const void* in, void* out, int N, int H, int W, int C, void* stream);

/* Diagnostics: when stamps_dev != NULL, wave 0 of workgroup 0 of every following conv3x3 launch (register-staged
 * variant) writes its shader-clock stamps at the phase boundaries of its first 16 tiles into stamps_dev[16][8]
 * (uint64: 0 tile start, 1 loads issued, 2 after barrier, 3 prologue done, 4 after barrier, 5 MFMA done,
 * 6 stores issued).  NULL switches it off (the default). */
int aliby_debug_conv_trace(aliby_ctx* ctx, void* stamps_dev);
/* The same unit (segment/dispatch.py:208-215) with the residual block's 1x1 projection fused in (cellpose `resdown`: x = proj(x_in) + conv1(conv0(x_in))):
 *   OUT = conv3x3( relu(scale*IN + shift) ) + bias + proj_w . PROJ_IN[n,y,x,:]
 * PROJ_IN is the block's RAW input [N,H,W,proj_channels] bf16 (no activation; the projection's BatchNorm is folded into
 * proj_w, its bias into `bias`), so the projected tensor never exists in HBM.  proj_wpk comes from
 * aliby_nn_pack_conv1x1_bf16 with CIN = 16 (for (CIN,COUT) = (32,32), proj_channels <= 16) or 32 ((64,64), <= 32). */
int aliby_nn_conv3x3_proj_bf16(aliby_ctx* ctx, const void* in, const void* wpk, void* out, const float* scale,
                               const float* shift, int shift_per_sample, const float* bias, int N, int H, int W,
                               int CIN, int COUT, const void* proj_in, const void* proj_wpk, int proj_channels,
                               void* stream);
/* The network's LAST unit with the output head in its epilogue (cellpose CPnet.output: BatchNorm + ReLU + 1x1 convolution
 * to `head_channels` <= 3 maps, segment/unet.py): head_out [N, head_channels, H, W] float32 =
 * head_bias[o] + sum_c head_w[o, c] * bf16(relu(head_scale[c] * OUT[c] + head_shift[c])) — bit-identical to running
 * aliby_nn_out_head_bf16 on the unit's bf16 output.  out_or_null = NULL skips writing that output (nothing else reads
 * it).  Built for the 32 -> 32 unit. */
int aliby_nn_conv3x3_head_bf16(aliby_ctx* ctx, const void* in, const void* wpk, void* out_or_null, const float* scale,
                               const float* shift, int shift_per_sample, const float* bias, const void* res, int res_up,
                               int N, int H, int W, int CIN, int COUT, const float* head_scale,
                               const float* head_shift, const float* head_w, const float* head_bias, int head_channels,
                               float* head_out, void* stream);
/* The network's first two units in ONE launch (round 3): x1 = conv3x3(act1(conv3x3(act0(tiles)))) + proj(tiles) + bias1 — what
 * aliby_nn_first_conv_bf16 followed by aliby_nn_conv3x3_proj_bf16 compute, without the first layer's output (c0, 32 channels)
 * and the raw bf16 copy of the tiles ever reaching HBM: same bits as the two launches.  tiles float32 [N, Cin <= 2, H, W];
 * scale0 / shift0 [Cin] and w_oihw [32][Cin][9] as in aliby_nn_first_conv_bf16; wpk1 the second unit's packed weights
 * (32 -> 32), scale1 / shift1 [32] its prologue (shift1 already carries the first layer's bias), bias1 [32] its bias plus the
 * projection's; proj_wpk the projection packed with aliby_nn_pack_conv1x1_bf16(..., CIN = 16).  out [N, H, W, 32] bf16. */
int aliby_nn_first_pair_bf16(aliby_ctx* ctx, const float* tiles, int N, int Cin, int H, int W, const float* scale0,
                             const float* shift0, const float* w_oihw, const void* wpk1, const float* scale1,
                             const float* shift1, const float* bias1, const void* proj_wpk, void* out, void* stream);
/* Two consecutive 32 -> 32 units of a residual block in ONE launch (cellpose resdown / resup: x + conv3(conv2(x)),
 * segment/unet.py): out = conv_b(act_b(conv_a(act_a(in)) + bias_a)) + bias_b + res, the tensor in between kept in LDS
 * (rounded to bf16 where the two-launch path stores it: same bits as two aliby_nn_conv3x3_bf16 launches).  An 8-wave
 * workgroup per CU: four waves hold unit A's weights and produce a tile's intermediate block while the other four hold
 * unit B's and consume the previous tile's.  pool_out as in aliby_nn_conv3x3_bf16; head_* as in
 * aliby_nn_conv3x3_head_bf16 (head_out != NULL: out_or_null may be NULL). */
int aliby_nn_conv3x3_pair_bf16(aliby_ctx* ctx, const void* in, const void* wpk_a, const void* wpk_b, void* out_or_null,
                               const float* scale_a, const float* shift_a, int shift_a_per_sample, const float* bias_a,
                               const float* scale_b, const float* shift_b, int shift_b_per_sample, const float* bias_b,
                               const void* res, int N, int H, int W, void* pool_out, const float* head_scale,
                               const float* head_shift, const float* head_w, const float* head_bias, int head_channels,
                               float* head_out, void* stream);
/* float32 OIHW [COUT, CIN_src, 3, 3] device weights -> the MFMA fragment order the kernel above reads
 * ([COUT/32][9 taps][CIN/16][64 lanes][8] bf16, COUT*CIN*9*2 bytes; input channels >= CIN_src are zero). */
int aliby_nn_pack_conv3x3_bf16(aliby_ctx* ctx, const float* w_oihw, int COUT, int CIN_src, int CIN, void* wpk,
                               void* stream);
/* Weight preparation for the two entries above (what the reference leaves to torch's module loading,
 * segment/dispatch.py:161-175): float32 [COUT, CIN_src] device weights of a 1x1 convolution -> [COUT/32][CIN/16][64 lanes][8] bf16 (same row order). */
int aliby_nn_pack_conv1x1_bf16(aliby_ctx* ctx, const float* w_oi, int COUT, int CIN_src, int CIN, void* wpk,
                               void* stream);
/* network output bf16 NHWC [N,H,W,Cpad] (+ bias[Cout]) -> float32 NCHW [N,Cout,H,W]. */
int aliby_nn_nhwc_to_nchw_f32(aliby_ctx* ctx, const void* y, int N, int H, int W, int Cpad, int Cout,
                              const float* bias, float* out, void* stream);
/* 1x1 convolution of the network that `model.eval` runs (segment/dispatch.py:208-215; cellpose `resdown.proj` /
 * `resup.proj`: BatchNorm -> Conv2d(1x1), BatchNorm folded into the weights by the caller): OUT[n,y,x,:] = W . IN[n,y,x,:] + bias, bf16 NHWC, hand-written MFMA GEMM over the N*H*W
 * pixels.  wpk from aliby_nn_pack_conv1x1_bf16(COUT, CIN, CIN).  CIN in {32, 64, 128, 256}; COUT 32, 64 or a multiple of
 * 128; bias may be NULL. */
int aliby_nn_conv1x1_bf16(aliby_ctx* ctx, const void* in, const void* wpk, const float* bias, void* out, int N, int H, int W,
                          int CIN, int COUT, void* stream);
/* First layer of the network that `model.eval` runs (segment/dispatch.py:208-215) on float32 NCHW tiles with Cin <= 2
 * channels (the 2-channel input cellpose builds from the selected plane, dispatch.py:192-206):
 * c0 = conv3x3(bf16(relu(scale[c]*x + shift[c])), zero padded) as bf16 NHWC[32] WITHOUT the convolution's bias, and the raw
 * input as bf16 NHWC[8] (channels >= Cin zero) for the block's projection.  w_oihw: float32 [32, Cin, 3, 3] (pass bf16-
 * representable values to reproduce a bf16 convolution). */
int aliby_nn_first_conv_bf16(aliby_ctx* ctx, const float* tiles, int N, int Cin, int H, int W, const float* scale,
                             const float* shift, const float* w_oihw, void* raw8, void* c0, void* stream);
/* Style vector of the network that `model.eval` runs (segment/dispatch.py:208-215; cellpose `make_style`: global average pool of the deepest feature map, L2-normalised) and
 * the per-sample shifts of every styled unit derived from it in one batched product: x bf16 NHWC [N,H,W,C] ->
 * style float32 [N,C]; shifts[n,:] = b + style[n,:] . wt, wt float32 [C,J] (`batchconvstyle.full` of all units, folded with
 * their BatchNorm by the caller), shifts float32 [N,J]. */
int aliby_nn_style_bf16(aliby_ctx* ctx, const void* x, int N, int H, int W, int C, const float* wt, const float* b, int J,
                        float* style, float* shifts, void* stream);
/* Output head of the network (cellpose CPnet.output = BatchNorm -> ReLU -> 1x1 Conv2d, the flows + cellprob
 * that `model.eval` (segment/dispatch.py:208-215) returns): x bf16 NHWC [N,H,W,32] -> float32 NCHW [N,O,H,W],
 * y = bias[o] + sum_c bf16(w[o,c]) * bf16(relu(scale[c]*x + shift[c])); w is float32 [O,32], rounded to bf16 as the bf16
 * convolution it replaces holds it.  Round 3: two k-steps of the 32x32x16 MFMA per 32 pixels (csrc/nn_conv.hip, head_apply)
 * — the same function the fused forms (aliby_nn_conv3x3_head_bf16, aliby_nn_conv3x3_pair_bf16) call: same bits. */
int aliby_nn_out_head_bf16(aliby_ctx* ctx, const void* x, const float* scale, const float* shift, const float* w,
                           const float* bias, int N, int H, int W, int C, int O, float* out, void* stream);
/* float32 NCHW network tiles (Cin <= 8) -> bf16 NHWC padded to 8 channels: raw copy and relu(bn(x)). */
int aliby_nn_tiles_to_nhwc8_bf16(aliby_ctx* ctx, const float* tiles, int N, int Cin, int H, int W,
                                 const float* scale, const float* shift, void* raw, void* act, void* stream);

/* ---- a6': Cellpose post-network dynamics ---------------------------------- */
/* What `model.eval` (segment/dispatch.py:208-215; cellpose.dynamics.compute_masks) does after the
 * network for 2-D images: dP [dev] is [F,2,Y,X] float32 (dY,dX at network scale), cellprob [dev] is
 * [F,Y,X]; labels_out [dev] receives uint16 labels 1..n per tile, n_labels_host[F] the counts.
 * Defaults of the reference's call: niter=200, cellprob_threshold=0, flow_threshold=0.4,
 * min_size=15, max_size_fraction=0.4.  workspace [dev] >= aliby_masks_workspace_bytes(F,Y,X).
 * p_final_out [dev, optional] receives the flow-following end points [F,2,Y,X] (debug/tests).
 * Raises ALIBY_ERR_OVERFLOW when a tile would need >= 65535 labels (dispatch.py:230-233). */
size_t aliby_masks_workspace_bytes(int F, int Y, int X);
int aliby_masks_from_flows(aliby_ctx* ctx, const float* dP, const float* cellprob, int F, int Y, int X,
                           int niter, float cellprob_threshold, float flow_threshold, int min_size,
                           float max_size_fraction, void* workspace, size_t workspace_bytes,
                           uint16_t* labels_out, int32_t* n_labels_host, float* p_final_out, void* stream);

/* ---- a13: cp_measure single-image features ------------------------------ */
/* Call site wrap_cp_measure_features (loaders.py:135-150): fun(mask.astype(uint16), pixels).
 * planes [dev] is [F, C, Y, X] (already z-reduced); channel selects the plane.
 * Rows of `out` follow the object table; columns start at col0 in the order listed in
 * aliby_amd/extraction/features.py (intensity: 21 columns, 16 without the edge block). */
int aliby_features_intensity(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype,
                             int F, int C, int Y, int X, int channel,
                             const aliby_object* table_dev, int n_obj, int max_area,
                             int edge_measurements, double* out, int ld, int col0, void* stream);

int aliby_features_sizeshape(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                             const aliby_object* table_dev, int n_obj, int max_h, int max_w,
                             int max_area, double* out, int ld, int col0, void* stream);

/* cp_measure "feret" (get_core_measurements()["feret"], default feature list pipe_builder.py:49-56):
 * out[:, col0] = MinFeretDiameter, out[:, col0+1] = MaxFeretDiameter. */
int aliby_features_feret(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                         const aliby_object* table_dev, int n_obj, int max_h, double* out, int ld,
                         int col0, void* stream);

/* Minimum enclosing circle of each object's pixel centres (centrosome.cpmorphology.
 * minimum_enclosing_circle, used by the Zernike families): mec_dev[n_obj][4] = (centre_i, centre_j,
 * radius, n_hull_vertices) in tile coordinates. */
int aliby_object_mec(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                     const aliby_object* table_dev, int n_obj, int max_h, double* mec_dev, void* stream);

/* cp_measure "zernike" (weighted=0: 30 columns, |sum Z_nm|/(pi r^2)) and "radial_zernikes"
 * (weighted=1: 30 magnitudes |sum I Z_nm|/n_pixels then 30 phases atan2(Re,Im)); n<=9. */
int aliby_features_zernike(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype,
                           int F, int C, int Y, int X, int channel, const aliby_object* table_dev,
                           int n_obj, const double* mec_dev, int weighted, double* out, int ld, int col0,
                           void* stream);
/* radial_zernikes (the weighted form above) of 2..5 channels of the same planes in ONE launch: the unit-disc coordinates, the
 * powers (y + ix)^m and the radial polynomials of a pixel are evaluated once and applied to every channel's weight.
 * channels[i] -> out[:, col0s[i] .. col0s[i] + 60); same numbers as n_channels calls of aliby_features_zernike(weighted = 1). */
int aliby_features_radial_zernikes_multi(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C, int Y,
                                         int X, const int* channels, const int* col0s, int n_channels, const aliby_object* table_dev,
                                         int n_obj, const double* mec_dev, double* out, int ld, void* stream);

/* cp_measure "texture": 13 Haralick statistics x 4 directions (direction-major, 52 columns) of the
 * object's bbox crop quantised to 8-bit grey levels (uint16 >> 8; ALIBY_U8W: the value; [0,1] floats -> rint(255 f)),
 * co-occurrence distance `scale` (3), zero grey level ignored (mahotas ignore_zeros=True). */
int aliby_features_texture(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype,
                           int F, int C, int Y, int X, int channel, const aliby_object* table_dev,
                           int n_obj, int max_h, int max_w, int max_area, int scale, int gray_levels,
                           double* out, int ld, int col0, void* stream);

/* cp_measure "radial_distribution" (scaled rings, centre = the object itself), two steps:
 *  1. aliby_radial_geometry: channel-independent ring/wedge code of every object pixel, written into
 *     binmap_dev [F,Y,X] (0x80 | wedge<<4 | ring; 0 = not reached from the centre);
 *  2. aliby_features_radial_distribution: per channel, 3*bin_count columns
 *     FracAtD_1..n, MeanFrac_1..n, RadialCV_1..n. */
int aliby_radial_geometry(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                          const aliby_object* table_dev, int n_obj, int max_h, int max_w, int bin_count,
                          uint8_t* binmap_dev, void* stream);
int aliby_features_radial_distribution(aliby_ctx* ctx, const uint16_t* labels, const uint8_t* binmap_dev,
                                       const void* planes, int dtype, int F, int C, int Y, int X,
                                       int channel, const aliby_object* table_dev, int n_obj,
                                       int bin_count, double* out, int ld, int col0, void* stream);
/* The same with `scaled=False` (CellProfiler's unscaled bins): rings of maximum_radius / bin_count pixels of centre distance,
 * everything beyond maximum_radius in an overflow ring.  aliby_radial_geometry_unscaled writes ring `bin_count` for those
 * pixels; aliby_features_radial_distribution_rings with rings_out = bin_count + 1 reports it as a last column of each of
 * FracAtD / MeanFrac / RadialCV (3 * rings_out columns; rings_out = bin_count is the scaled call above). */
int aliby_radial_geometry_unscaled(aliby_ctx* ctx, const uint16_t* labels, int F, int Y, int X,
                                   const aliby_object* table_dev, int n_obj, int max_h, int max_w, int bin_count,
                                   double maximum_radius, uint8_t* binmap_dev, void* stream);
int aliby_features_radial_distribution_rings(aliby_ctx* ctx, const uint16_t* labels, const uint8_t* binmap_dev,
                                             const void* planes, int dtype, int F, int C, int Y, int X,
                                             int channel, const aliby_object* table_dev, int n_obj,
                                             int bin_count, int rings_out, double* out, int ld, int col0, void* stream);

/* ---- a15: the reference's in-repo per-cell metrics ------------------------ */
/* extraction/core/functions/cell.py:18-303.  17 columns: area, centroid_x, centroid_y, conical_volume,
 * eccentricity, spherical_volume, volume, min_ax, maj_ax (min_maj_approximation), mean, median, std, total,
 * total_squared, max2p5pc, max5px_median, moment_of_inertia.  planes may be NULL (mask-only metrics). */
int aliby_features_cell(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C,
                        int Y, int X, int channel, const aliby_object* table_dev, int n_obj, int max_h,
                        int max_w, int max_area, double* out, int ld, int col0, void* stream);

/* cell.ratio (src/extraction/core/functions/cell.py:268-279): out[n_obj] = median over the object of channel0 / channel1 (true
 * division), NaN when any channel1 pixel of the object is 0 or the object is empty. */
int aliby_features_cell_ratio(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C, int Y,
                              int X, int channel0, int channel1, const aliby_object* table_dev, int n_obj, int max_area,
                              double* out, void* stream);
/* trap.imBackground / trap.background_max5 (src/extraction/core/functions/trap.py:6-43): per tile, over the pixels of
 * `channel` under NO mask (labels == 0): out[f*2] = their median (numpy.median), out[f*2+1] = the mean of the five largest
 * (of all of them when fewer); NaN for a tile without background. */
int aliby_features_trap_background(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C,
                                   int Y, int X, int channel, double* out, void* stream);

/* ---- a13 (widening): cp_measure "granularity" ------------------------------ */
/* Bound like every core measurement at src/extraction/core/functions/loaders.py:71-73 (fun(mask, pixels) -> dict of
 * Granularity_1..L); not in the builder's default list (pipe_builder.py:49-56).  CellProfiler's MeasureGranularity: frame
 * subsampled by subsample_size (bilinear), background (erode + dilate with disk(element_size) on a further
 * image_sample_size subsample) removed, then spectrum_length rounds of erode-by-disk(1) + reconstruction-by-dilation;
 * out[obj, col0 + i - 1] = (mean_{i-1} - mean_i) * 100 / max(mean_0, eps) over the object's pixels, mean_0 on the
 * original pixels.  image_mask_objects = 0: the image mask is the frame (CellProfiler's default); 1: labels > 0, sampled
 * bilinearly.  The image-level part runs once per (tile, channel), float64.  `workspace` is device memory of at least
 * aliby_granularity_workspace_bytes(); the call synchronises `stream` while it iterates the reconstruction to its fixed
 * point.  PARITY UNPINNED (oracle/granularity_restated.py; cp_measure is not available offline). */
size_t aliby_granularity_workspace_bytes(int F, int Y, int X, int n_obj, double subsample_size, double image_sample_size);
int aliby_features_granularity(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C, int Y,
                               int X, int channel, const aliby_object* table_dev, int n_obj, double subsample_size,
                               double image_sample_size, int element_size, int spectrum_length, int image_mask_objects,
                               void* workspace, size_t workspace_bytes, double* out, int ld, int col0, void* stream);

/* ---- a14: cp_measure colocalisation -------------------------------------- */
/* Call site wrap_cp_corr_features (loaders.py:153-167): fun(pixels1, pixels2, mask); metric list
 * pipe_builder.py:37.  One launch evaluates any subset of {pearson, manders_fold, rwc, costes} for the
 * channel pair (ch0, ch1); col_* is the first of the metric's two columns or -1 to skip it.
 * thr_percent = 15 and costes_scale_max = 255 are CellProfiler's defaults.  rwc needs ranks_dev / rmax_dev
 * (aliby_object_ranks) for both channels; they may be NULL otherwise. */
int aliby_features_coloc(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype,
                         int F, int C, int Y, int X, int ch0, int ch1,
                         const aliby_object* table_dev, int n_obj, int max_area,
                         double* out, int ld, int col_pearson, int col_manders, int col_rwc,
                         int col_costes, double thr_percent, double costes_scale_max,
                         const uint32_t* ranks_dev, const int32_t* rmax_dev, void* stream);
/* The same metrics for a list of channel pairs in ONE launch: an object's pixels of every channel in use are gathered
 * once and each wave of its workgroup takes pairs in turn (the builder's multi tree is C(5,2) = 10 pairs x 4 metrics,
 * pipe_builder.py:33-43).  pairs_host: n_pairs x 6 ints (ch0, ch1, col_pearson, col_manders, col_rwc, col_costes; -1
 * skips a metric).  Returns ALIBY_ERR_TOO_LARGE, without launching, when the pixel lists exceed the LDS budget: the
 * caller then goes pair by pair through aliby_features_coloc. */
int aliby_features_coloc_pairs(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C,
                               int Y, int X, const int32_t* pairs_host, int n_pairs, const aliby_object* table_dev,
                               int n_obj, int max_area, double* out, int ld, double thr_percent,
                               double costes_scale_max, const uint32_t* ranks_dev, const int32_t* rmax_dev,
                               void* stream);
/* Dense per-object ranks of one channel (CellProfiler's Rank_im of the RWC coefficient): ranks_dev
 * [F,C,Y,X] uint32 receives, at every object pixel, the number of distinct smaller values of that object in
 * `channel`; rmax_dev [n_obj, C] the largest rank.  One sort per (object, channel), shared by all pairs. */
int aliby_object_ranks(aliby_ctx* ctx, const uint16_t* labels, const void* planes, int dtype, int F, int C,
                       int Y, int X, int channel, const aliby_object* table_dev, int n_obj, int max_area,
                       uint32_t* ranks_dev, int32_t* rmax_dev, void* stream);

/* ---- §8f-2: image ingest --------------------------------------------------- */
/* Host-side TIFF decode feeding the stager (aliby_crop_pad_u16).  Replaces the per-file imageio.imread of
 * ImageList.get_data_lazy (src/aliby/io/image.py:395-408), the dask.array.image.imread of ImageDir /
 * ImageMultiTiff (image.py:190, 285) and zarr's chunk decompression behind ImageZarr (image.py:246-259).
 * Baseline TIFF 6.0 and BigTIFF; strips or tiles; compression none / LZW / Deflate / PackBits / Zstandard;
 * horizontal predictor; sample 0 of chunky multi-sample pixels.
 * aliby_tiff_probe: info[12] = pages, width, height, bits, sample_format (1 uint, 2 int, 3 float),
 * samples_per_pixel, compression, predictor, tiled, bigtiff, big_endian, uniform; description receives page 0's
 * ImageDescription (ImageJ hyperstack layout).  Needs no context and no GPU. */
int aliby_tiff_probe(const char* path, int64_t* info, char* description, int description_len);
/* Decode page pages[i] of paths[i] (i < n) into dst + i * plane_stride with a pool of n_threads host threads
 * (<= 0: one per core, at most 16).  dst_is_device = 0: dst is host memory, ctx and stream may be NULL.
 * dst_is_device = 1: every plane is decoded into pinned staging memory and queued for upload on `stream` as soon as
 * its thread finishes, so transfers overlap the remaining decodes; returns after the last upload completed. */
int aliby_ingest_tiff_planes(aliby_ctx* ctx, const char* const* paths, const int32_t* pages, int n, int width,
                             int height, int bytes_per_sample, void* dst, size_t plane_stride, int dst_is_device,
                             int n_threads, void* stream);
/* One compressed zarr chunk -> dst; codec 0 = zlib / gzip, 1 = Zstandard (libzstd.so.1 loaded on first use), 2 = a Blosc
 * version-1 frame (zarr v2's default compressor; blosclz / lz4 / zlib / zstd streams, byte or bit shuffle), decoded by hand. */
int aliby_ingest_inflate(int codec, const void* src, size_t src_bytes, void* dst, size_t dst_bytes,
                         size_t* out_bytes);

/* ---- §8f-3: trap detection -------------------------------------------------- */
/* Building blocks of segment_traps / identify_trap_locations (src/aliby/tile/process_traps.py:24-218), each standing
 * in for one scikit-image call of that file; float64 images, once per position.  aliby_amd/tile/traps.py sequences them.
 * gauss1d: one axis of transform.rescale's anti-aliasing filter (scipy gaussian_filter, mode 'mirror'; truncate = 1
 *   reproduces the integer frame's truncation).  warp: bilinear transform.warp with the 2x3 matrix acting on (col,row),
 *   mode 0 constant / 1 reflect (rescale, rotate).  entropy: filters.rank.entropy with disk(radius).  morph: k x k
 *   max / min (morphology.closing with square(k)).  label: measure.label, 8-connected, label = 1 + raster index of the
 *   first pixel.  region_sums: raw moments per label for regionprops centroid / major_axis_length, and the border flag of
 *   segmentation.clear_border.  match_template: feature.match_template on the median-padded image.  maxfilter1d: the
 *   window maximum of feature.peak_local_max. */
int aliby_trap_gauss1d(aliby_ctx* ctx, const double* in, double* out, int H, int W, int axis,
                       const double* weights_dev, int radius, int truncate, void* stream);
int aliby_trap_warp(aliby_ctx* ctx, const double* in, int H, int W, double* out, int OH, int OW,
                    const double* matrix6_host, int mode, double cval, void* stream);
int aliby_trap_entropy(aliby_ctx* ctx, const uint8_t* in, int H, int W, int radius, double* out, void* stream);
int aliby_trap_morph(aliby_ctx* ctx, const uint8_t* in, uint8_t* out, int H, int W, int lo, int hi, int is_max,
                     void* stream);
int aliby_trap_label(aliby_ctx* ctx, const uint8_t* bw, int H, int W, int32_t* labels, void* stream);
int aliby_trap_region_sums(aliby_ctx* ctx, const int32_t* labels, int H, int W, uint64_t* sums, void* stream);
int aliby_trap_match_template(aliby_ctx* ctx, const double* padded, int PH, int PW, const double* templ, int th,
                              int tw, double* out, int H, int W, double t_mean, double t_ssd, void* stream);
int aliby_trap_maxfilter1d(aliby_ctx* ctx, const double* in, double* out, int H, int W, int axis, int radius,
                           void* stream);

/* ---- round 3 extension: a Z-stack as a volume (BASELINE config 5, "true 3-D"; beyond what the reference wires) -------- */
/* Labels through a per-object table: out[t, p] = lut[offsets[t] + in[t, p] - 1] (0 stays 0): writes the Z-stitched labels
 * (aliby_track_stitch along Z, the reference's stitch_threshold = 0.01 of dispatch.py:193-198) back into the planes.  A
 * label >= 65535 is ALIBY_ERR_OVERFLOW. */
int aliby_labels_apply_lut(aliby_ctx* ctx, const uint16_t* labels_in, int T, int Y, int X, const int32_t* offsets_host,
                           const int32_t* lut_dev, uint16_t* labels_out, void* stream);
/* Per-object intensity statistics over labelled stacks: labels uint16 [F,Z,Y,X] (labels 1..n_f per stack), pixels uint16
 * [F,C,Z,Y,X]; object (f, label) is row offsets_host[f] + label - 1.  12 columns at out[row * ld + col0 ..]: Volume,
 * IntegratedIntensity, MeanIntensity, StdIntensity (population), MinIntensity, MaxIntensity, CenterMassIntensity_X/Y/Z,
 * Center_X/Y/Z.  Exact integer sums: run-to-run deterministic. */
int aliby_features_intensity3d(aliby_ctx* ctx, const uint16_t* labels, const uint16_t* pixels, int F, int C, int Z, int Y,
                               int X, int channel, const int32_t* offsets_host, double* out, int ld, int col0, void* stream);

/* ---- a17: the step API's files, encoded natively (host code, no GPU work) ------------------ */
/* profiles/<name>.parquet — pyarrow.parquet.write_table(profiles, path, compression="zstd")
 * (src/aliby/pipe_core.py:412-413).  One row group, one PLAIN data page per column, fields OPTIONAL with every value
 * present, zstd level `zstd_level` (negative levels are zstd's fast modes; ALIBY_PQ_UNCOMPRESSED: no compression) through
 * libzstd.so.1 (dlopen).  A column's rows come as n_segs
 * segments (a position's table is the concatenation of one slice per object set): values[c * n_segs + s] points at
 * seg_rows[s] items — double / int64 / uint16, or for ALIBY_PQ_STR the Arrow utf8 layout: int32 offsets[seg_rows[s] + 1]
 * with the characters at aux[c * n_segs + s].  Thread-safe (one compression context per calling thread). */
enum { ALIBY_PQ_F64 = 0, ALIBY_PQ_I64 = 1, ALIBY_PQ_U16 = 2, ALIBY_PQ_STR = 3 };
#define ALIBY_PQ_UNCOMPRESSED (-1000000)
typedef struct aliby_pq_column {
  const char* name; /* UTF-8, NUL-terminated */
  int32_t type;     /* ALIBY_PQ_* */
  int32_t reserved;
} aliby_pq_column;
int aliby_parquet_write(const char* path, const aliby_pq_column* cols, int n_cols, const int64_t* seg_rows, int n_segs,
                        const void* const* values, const void* const* aux, int zstd_level);
/* steps/<name>/<step>/<tp:04d>.npz — numpy.savez_compressed(out_file, labels) (src/aliby/io/write.py:25-51): a zip
 * archive of deflated .npy members (version 1.0 headers, C order).  libdeflate.so.0 (dlopen) when present, zlib otherwise. */
typedef struct aliby_npy_member {
  const char* name;     /* "arr_0", "tile_3", ... (".npy" is appended) */
  const char* descr;    /* NumPy dtype string, e.g. "<u2" */
  const int64_t* shape; /* [ndim] */
  const void* data;     /* C-contiguous */
  int32_t ndim;
  int32_t itemsize;     /* bytes per element */
} aliby_npy_member;
int aliby_npz_write(const char* path, const aliby_npy_member* members, int n_members, int level);
/* which of the two optional codec libraries were found on this machine */
int aliby_host_codecs(int* have_zstd, int* have_libdeflate);

/* memcpy of a large host block on `threads` threads (1..16; blocks under 8 MB are copied by the caller's thread): the
 * position-batched runner takes a batch's feature rows (~130 MB) out of its page-locked download arena into memory the returned
 * tables own — one thread pays ~12 ms for that, mostly first-touch page faults, while every writer thread of the batch waits for
 * the table (aliby_amd/runner.py _LazyRows; the reference has no such copy: extract.py:520-599 builds its table from Python
 * lists). */
int aliby_host_copy(void* dst, const void* src, size_t bytes, int threads);

#ifdef __cplusplus
}
#endif
#endif /* ALIBY_HIP_H */
