/*
 * rvdd.h -- C ABI of librvdd_hip.so, the MI355X (gfx950) runtime for the
 * recurrent video denoise+demosaic inference path of centreborelli/RVDD-release.
 *
 * The reference has no FFI on this path: the path sits behind two string-keyed
 * Python plugin registries (models/__init__.py:25-45 `--model recurrent`,
 * networks/__init__.py:121-176 `--netDenoiser ...`).  This header is what a
 * binding for that plugin surface binds (the ctypes stub is in INTEGRATION.md
 * and rvdd-release_amd/runtime.py).  Conventions mirror the reference's only
 * real FFI, library.CPPbridge (library.py:143-175): plain pointers and sizes,
 * caller-allocated buffers, no exceptions across the boundary.
 *
 *  - every function returns 0 on success or a negative rvdd_status;
 *    rvdd_last_error() gives the message (per handle; NULL -> last create error);
 *  - all tensor pointers are DEVICE pointers to dense fp32 tensors in the
 *    reference's own layout (NCHW), owned by the caller (e.g. obtained from
 *    torch.Tensor.data_ptr() on PyTorch-ROCm);
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *    calls are asynchronous and stream-ordered unless stated otherwise;
 *  - a handle owns its weights, workspace and the recurrent state; one handle
 *    = one device; a handle is not thread-safe, distinct handles are.
 */
#ifndef RVDD_H
#define RVDD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rvdd_handle rvdd_t;

enum rvdd_status {
    RVDD_OK = 0,
    RVDD_ERR_ARG = -1,      /* bad argument / shape */
    RVDD_ERR_STATE = -2,    /* call sequence violated (e.g. step before weights) */
    RVDD_ERR_WEIGHT = -3,   /* unknown / missing / mis-shaped state_dict key */
    RVDD_ERR_HIP = -4,      /* HIP runtime error */
    RVDD_ERR_NOMEM = -5
};

/* netDenoiser families (networks/__init__.py:121-176). */
enum rvdd_arch {
    RVDD_ARCH_CONVUNET = 0,       /* convunet-mode=fixedfeatures       networks/unet.py:595-720 */
    RVDD_ARCH_CONVUNET_FEAT = 1,  /* convunet-mode=fixedfeatures+feat  networks/unet.py:725-825 */
    RVDD_ARCH_CONVNEXT = 2,       /* newunet                           networks/new_unet.py:207-362 */
    RVDD_ARCH_CONVNEXT_FEAT = 3   /* newunet-mode=feat                 networks/new_unet.py:365-430 */
};

typedef struct rvdd_cfg {
    int32_t arch;     /* enum rvdd_arch */
    int32_t future;   /* --future_patch_depth: 0 or 1 (options/base_options.py:56) */
    int32_t batch;    /* B sequences advanced in lockstep */
    int32_t height;   /* RGB frame height H (even, >= 16); raw frames are 4 x H/2 x W/2 */
    int32_t width;    /* RGB frame width  W (even, >= 16); H x W x 192 bytes < 2 GiB (11.1 Mpx: 3840 x 2176 fits) */
    int32_t device;   /* HIP device ordinal */
} rvdd_cfg;

/* ---- life cycle ---------------------------------------------------------- */

/* Replaces recurrentModel.__init__ + networks.define_net_arch
 * (models/recurrent_model.py:38-99, networks/__init__.py:121-176). */
int rvdd_create(const rvdd_cfg* cfg, rvdd_t** out);
void rvdd_destroy(rvdd_t* h);
const char* rvdd_last_error(const rvdd_t* h);

/* Replaces BaseModel.load_networks -> net.load_state_dict
 * (models/base_model.py:173-196).  `host` points to HOST fp32 data of one
 * state_dict entry in PyTorch layout (conv weight = OIHW).  Unlike the
 * reference (strict=False) loading is strict: an unknown key fails here, a
 * missing one fails in rvdd_finalize_weights. */
int rvdd_set_weight(rvdd_t* h, const char* key, const float* host, const int64_t* shape, int32_t ndim);
int rvdd_finalize_weights(rvdd_t* h);

/* ---- the recurrent hot path ---------------------------------------------- */

/* FirstOfVideo (validate.py:76-77 -> recurrent_model.py:115,233-245): the next
 * rvdd_step re-initialises lastden from raw_prev and zeroes the features. */
int rvdd_reset(rvdd_t* h);

/* One output frame for each of the B sequences: recurrentModel.set_input +
 * forward, test branch (models/recurrent_model.py:105-135, 161-349):
 * Hamilton-Adams demosaic, flow x2 upsample, bicubic backward warp of the
 * previous output / features / next frame, U-Net forward, state hand-over.
 *   raw_prev, raw_cur, raw_next : [B,4,H/2,W/2] packed GBRG raw in [-1,1]
 *                                 (raw_prev is read only on the first step after
 *                                  create/reset; raw_next only when future=1)
 *   flow_prev : [B,2,H/2,W/2] raw-resolution flow cur->prev (x first)
 *   flow_next : [B,2,H/2,W/2] raw-resolution flow cur->next (future=1)
 *   out_rgb   : [B,3,H,W] denoised linear RGB
 */
int rvdd_step(rvdd_t* h, const float* raw_prev, const float* raw_cur, const float* raw_next,
              const float* flow_prev, const float* flow_next, float* out_rgb, void* stream);

/* rvdd_step on channel slices of the reference's own input tensors, without a copy: the model hands the net
 * `n[:, 0:4]`, `n[:, 4:8]`, `n[:, 8:12]` of one [B,(2+f)*4,h,w] tensor and `flow[:, 0]`, `flow[:, 1]` of one
 * [B,1+f,2,h,w] tensor (models/recurrent_model.py:299-324; data/infer4rec_dataset.py:226-230).  Each slice is dense
 * inside a sequence ([4,h,w] / [2,h,w]) and `raw_batch_stride` / `flow_batch_stride` floats apart from one
 * sequence to the next (0 = dense, i.e. 4hw / 2hw as rvdd_step assumes). */
int rvdd_step_strided(rvdd_t* h, const float* raw_prev, const float* raw_cur, const float* raw_next,
                      const float* flow_prev, const float* flow_next, int64_t raw_batch_stride,
                      int64_t flow_batch_stride, float* out_rgb, void* stream);

/* Recurrent state in the reference's layout, for get_current_features /
 * set_rec_features parity (networks/unet.py:814-818) and for tests.
 *   lastden  [B,3,H,W]; lastfeat [B,48,H,W] (NULL to skip either). */
int rvdd_get_state(rvdd_t* h, float* lastden, float* lastfeat, void* stream);
int rvdd_set_state(rvdd_t* h, const float* lastden, const float* lastfeat, void* stream);

/* compute_losses, test branch (models/recurrent_model.py:512-525;
 * util/util.py:9-20): out2 (HOST, 2 floats) = { 100*mean|den-gt|,
 * 10*log10(4/mean((den-gt)^2)) } over `count` elements.  Synchronises
 * `stream` (the reference reads the losses back with float(), base_model.py:151). */
int rvdd_psnr_l1(rvdd_t* h, const float* den, const float* gt, int64_t count, float* out2, void* stream);

/* ---- the same ops one at a time (plugin-level entry points; also test hooks) */

/* netDenoise(x) (networks/unet.py:544-588 / networks/new_unet.py:332-362).
 *   x [B,Cin,H,W] with Cin = 3*(2+future); feat_in/feat_out [B,48,H,W] for the
 *   *_FEAT archs (NULL otherwise); out [B,3,H,W]. */
int rvdd_unet_forward(rvdd_t* h, const float* x, const float* feat_in, float* out, float* feat_out,
                      void* stream);

/* HamiltonAdam('gbrg').forward (util/Hamilton_Adam_demo.py:249-289):
 *   raw [n,4,h,w] -> rgb [n,3,2h,2w]. */
int rvdd_demosaic_ha(rvdd_t* h, const float* raw, int32_t n, int32_t hh, int32_t ww, float* rgb,
                     void* stream);

/* util.flow_utils.warp(x, flow, "bicubic")[0] (util/flow_utils.py:70-102):
 *   x [n,c,H,W], flow [n,2,H,W] at full resolution -> y [n,c,H,W]. */
int rvdd_warp_bicubic(rvdd_t* h, const float* x, const float* flow, int32_t n, int32_t c, int32_t H,
                      int32_t W, float* y, void* stream);

/* util.flow_utils.upsample_factor_2(t, multiply_by) (util/flow_utils.py:159-174):
 *   t [n,c,hh,ww] -> [n,c,2hh,2ww]. */
int rvdd_upsample_factor_2(rvdd_t* h, const float* t, int32_t n, int32_t c, int32_t hh, int32_t ww,
                           float multiply_by, float* out, void* stream);

/* library.CPPbridge.TVL1_flow -> libBridge `tvl1flow(I0, I1, u, nx, ny)` (libBridge.cpp:44-163): Dual TV-L1
 * optical flow with the reference's hard-wired parameters (3rdparty/tvl1flow/tvl1flow_lib.c:91-278, 343-472).
 *   I0, I1 [ny,nx] gray images; u [2,ny,nx] = x displacement then y displacement (libBridge.cpp:150) such
 *   that I1(x + u) ~ I0(x).  `iterations` (HOST, nullable) receives the total primal-dual iterations run.
 * One cooperative kernel per pyramid scale, all of its blocks resident (csrc/tvl1.hip: since round 4 a block owns a patch of
 * the image and exchanges its perimeter and convergence sum through tagged records that the readers poll -- no grid barrier
 * in the iteration; round 3's barrier kernel is kept, RVDD_TVL1_PATCH=0, and the two are bit-identical, flows and iteration
 * counts).  Synchronous: returns after the flow is complete (the control words are read back to catch an exchange that
 * timed out -- RVDD_ERR_HIP then, never a half-finished flow). */
int rvdd_tvl1flow(rvdd_t* h, const float* I0, const float* I1, float* u, int32_t nx, int32_t ny,
                  int32_t* iterations, void* stream);

/* `n` independent pairs of the same size -- what data/base_dataset.py:134-249 computes one call at a time when it
 * fills the dataset's flow folder.  I0, I1: [n][ny][nx]; u: [n][2][ny][nx]; iterations: HOST [n], nullable.
 * The pairs of a call share launches: the pre-processing and pyramid of all of them, up to eight pairs per scale kernel at
 * the coarse scales, two at 640 x 360 (0.76-0.80 ms per 640 x 360 flow in batches of eight, 1.8-1.9 ms one at a time);
 * every flow is bit-identical to the one rvdd_tvl1flow returns for that pair.  Synchronous. */
int rvdd_tvl1flow_batch(rvdd_t* h, const float* I0, const float* I1, float* u, int32_t n, int32_t nx, int32_t ny,
                        int32_t* iterations, void* stream);

/* dataset/fwd_ppipe.py `ppipe(im, rgb_gain, red_gain, blue_gain, iso)` (:48-77) fused with the range
 * normalisation in front of it (:131-137) and the uint8 conversion behind it (:141): linear camera RGB ->
 * display sRGB (inverse percentile matching per ISO, black level, white-balance gains, inverse CCM,
 * gamma 1/2.2, smoothstep tone curve, x255).
 *   img        n images of 3 channels, element (i,c,y,x) at img[i*stride_n + c*stride_c + y*stride_y + x*stride_x]
 *              (NCHW network output or the HWC image validate.py writes -- both are strides);
 *   bit_depth  fwd_ppipe.py --bit_depth: 0 ([0,1]), 8 ([0,255]), 10, anything else = already [0,4095];
 *              RVDD_PPIPE_FROM_NET (-1) = network output in [-1,1]: util/util.py:40 (tensor2im) then the 8-bit branch;
 *   gains      the three values fwd_ppipe.py:116-118 passes (rgb_gain = 1/n from the white-balance table);
 *   out_u8     [n,H,W,3] uint8 (what the reference writes to *_processed_pipeline.png);
 *   out_f32    [n,H,W,3] float32 = ppipe()'s return value before rounding; may be NULL. */
#define RVDD_PPIPE_FROM_NET (-1)
int rvdd_ppipe(rvdd_t* h, const float* img, int32_t n, int32_t height, int32_t width, int64_t stride_n,
               int64_t stride_c, int64_t stride_y, int64_t stride_x, int32_t bit_depth, double rgb_gain,
               double red_gain, double blue_gain, int32_t iso, uint8_t* out_u8, float* out_f32, void* stream);

/* dataset/fwd_ppipe.py `psnr(img1, img2)` (:79-84) and `ssim` (:86, skimage.metrics.structural_similarity with
 * multichannel=True, data_range=255: 7x7 uniform window, sample covariance, 3-pixel border cropped) on uint8
 * [n,H,W,3] images.  psnr / ssim: HOST arrays of n doubles (either may be NULL).  H, W >= 7.  Synchronous. */
int rvdd_srgb_metrics(rvdd_t* h, const uint8_t* a, const uint8_t* b, int32_t n, int32_t height, int32_t width,
                      double* psnr, double* ssim, void* stream);

/* Options of recurrentModel that change what a step does (models/recurrent_model.py:27-36).  Known names:
 *   "no_warp"  (--no_warp, :137-159): the previous output, the previous features and the next frame enter the net
 *              unwarped; rvdd_step then ignores flow_prev / flow_next (they may be NULL).
 *   "warp_raw" (--warp_raw, :149-152): the previous output is re-mosaicked, warped at RAW resolution with the
 *              raw-resolution flow and demosaicked again (the next frame: warped as packed raw, then demosaicked).
 *              Not defined with feature recurrence (the reference fails on the shapes there): error.
 *   "prev_noisy_frame" (--prev_noisy_frame, :33, :335-337): the frame handed to the next step as "previous" is the
 *              demosaiced NOISY current frame, not the denoised one (the feature recurrence is unaffected).
 *   "conv_kernel": which kernel runs the convunet's 3x3 convs.  0 (default) = EVERY 3x3 conv of the net -- the
 *              16-channel first layer and UpConv's fused upsample included -- on the F16 matrix pipe with each f32
 *              operand split into two f16 halves, three MFMAs per product, f32 accumulation (conv3x3h.hip: as close
 *              to the reference as the f32 kernels).  The f16 exponent range is not a limit
 *              of the path: every map carries its max |x| per sequence and is multiplied by a power of two before
 *              the split (block floating point, exact), so frames of any finite magnitude keep fp32 semantics
 *              (tests/test_gpu_parity.py::test_split_path_any_magnitude).  1 = the direct f32 kernel everywhere;
 *              2 = the Winograd f32 kernel everywhere; 4 = f32 kernels chosen by launch size (1, 2, 4: exact-f32
 *              products, the A/B reference, about 0.7 of the default's frame rate at 720p).
 *   "seq_major": 1 = the full-resolution stages of the convunet run one sequence at a time (measured slower; off).
 *   "fuse_upsample": 0 = UpConv's bilinear x2 upsample runs as its own kernel instead of inside the Winograd patch
 *              load of the conv behind it (default 1; same bits either way).
 *   "graphs":   1 = frame-steps are captured into hipGraphs and replayed (measured slower on ROCm 7.2; off).
 *   "next_split": 0 = ConvNeXtUnet's ConvBlock (networks/new_unet.py:74-103, one fused kernel) multiplies its two 1x1 convs on the f32 matrix pipe (exact-f32 products) instead
 *               of the F16 pipe with split f32 operands (the default, as "conv_kernel" 0; the A/B reference).  The split
 *               operands are bounded by the block's LayerNorm whatever the frames are; a block whose weights would let
 *               them leave the f16 range (checked at rvdd_finalize_weights) runs the f32 form by itself.
 *   "next_pipe": 0 = the fused ConvBlock runs its three phases one after the other in all eight waves of a workgroup
 *               (convblock_kernel) instead of as a pipeline over tiles -- depth-wise conv and LayerNorm of the next tile on
 *               four waves beside the MLP of the current one on the other four (convblock_pipe_kernel, the default with
 *               next_split; same bits either way).
 *   "next_pool": 0 = MaxPool2d(2) in front of a DownConv (new_unet.py:200-204) as its own kernel instead
 *               of the fused block's epilogue (default 1; same bits either way).
 *   "cout_split": 0 = every launch of the split-f16 conv kernel forms all 48 output channels of a tile in one workgroup.
 *               Default 1: a launch with at most a third of a 16x16 tile per compute unit (the coarse levels of one small
 *               sequence) gives a tile to three workgroups of 16 output channels each -- a third of the filter bank's copy and
 *               of the matrix work per workgroup; same sums in the same order, same bits.  Process-wide.
 *   "small_prestage": 0 = the pre-stage of a frame-step (bound of the network input, green plane, network input) always as
 *               its three kernels.  Default 1: a step of at most 1024 tiles of 16x16 pixels without a future frame (a single
 *               small sequence: the launches there are 5-15 us each, back to back) forms them in one kernel; same bits.
 *               Process-wide.
 *   "tvl1_async": 1 = rvdd_tvl1flow_batch called without iteration counts enqueues its launches on the stream and returns
 *               (the flows are ready in stream order; nothing is read back, the stream is not synchronised): the form for a
 *               caller that feeds the flows straight into rvdd_step on the same stream (validate.py's --val_flow_from_denoised loop
 *               on the device).  The batch's control word -- set only if a grid barrier of the TV-L1 kernels gave up -- is then
 *               read by the next call that synchronises anyway: rvdd_psnr_l1, a TV-L1 call that returns iteration counts or runs
 *               with the option off, or this option set back to 0 (each reports RVDD_ERR_HIP then).  Default 0: the reference
 *               bridge's behaviour (library.py:150-175 returns finished host arrays).
 *   "next_projfuse": 0 = the 96 -> 48 projection of the ConvBlock behind a concat (new_unet.py:85-88, 321-329) as its own
 *               kernel, instead of as two 48 -> 48 halves in the epilogues of the blocks that form the two concatenated maps
 *               (default 1 with the pipelined split-f16 block; the A/B reference: the same linear map, summed in another
 *               order).
 *   "fuse_pre": 0 = preprocessing_layer (3x3, no activation, networks/unet.py:742) and the first source of EncoderConvs[0][0]
 *               (3x3, :743) run as the two convs they are, instead of as their composition -- ONE 5x5 conv of the network
 *               input plus a fix of the border ring, where the zero padding between the two layers matters (default 1, the
 *               feature-recurrent convunet on the split-f16 path; the A/B reference: the same linear map, summed in another
 *               order, a few 1e-7 apart).
 *   "block_fp": 0 = the split-f16 convs split their operands without the per-map power of two (the A/B reference of the block
 *               floating point; right only while every activation stays within 2^-14 .. 65504).  Default 1.
 * Unknown names are an error. */
int rvdd_set_option(rvdd_t* h, const char* name, int32_t value);

/* Host-only helper of the TIFF reader that replaces `iio.read` (library.py:75-77): LZW strip / tile decoder
 * (TIFF 6.0 section 13).  in[n] -> out (capacity cap); returns the bytes produced or -1 (corrupt stream, output full). */
int64_t rvdd_tiff_lzw_decode(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap);

/* ---- measurement ----------------------------------------------------------- */

/* When enabled, every launch of the U-Net kernels is bracketed by HIP events
 * on the launch stream.  rvdd_profile_read synchronises, accumulates and
 * returns, for kernel class `idx` (0..rvdd_profile_count()-1): its name, the
 * number of launches, the summed duration (ms) and the summed algorithmic
 * FLOPs and bytes since the last rvdd_profile_enable(h, 1). */
int rvdd_profile_enable(rvdd_t* h, int32_t on);
/* Restrict the bracketing to one kernel class (NULL = all) and to every
 * `stride`-th launch of a class: an event pair costs a few microseconds of
 * queue bubble, so a headline run samples the dominant kernel only. */
int rvdd_profile_select(rvdd_t* h, const char* kernel_class, int32_t stride);
int rvdd_profile_count(const rvdd_t* h);
int rvdd_profile_read(rvdd_t* h, int32_t idx, char* name, int32_t name_cap, int64_t* launches,
                      double* total_ms, double* flops, double* bytes);

/* HIP-event stopwatch on `stream` (bench.py times the whole step loop with it). */
int rvdd_timer_start(rvdd_t* h, void* stream);
int rvdd_timer_stop_ms(rvdd_t* h, void* stream, float* ms);   /* synchronises */

/* Kernel A/B hook: time `iters` back-to-back launches of the 48->48 3x3 conv +
 * ReLU (the dominant kernel) on the handle's own level-`level` maps with code
 * variant `variant` (0..2: code variants of the direct f32-MFMA kernel, 3: the Winograd f32-MFMA kernel, 4: the
 * split-f16 kernel, the default of the convunet); *ms = mean milliseconds per launch.  Synchronises. */
int rvdd_debug_conv_bench(rvdd_t* h, int32_t variant, int32_t level, int32_t iters, float* ms, void* stream);

const char* rvdd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RVDD_H */
