/* pyrapose_hip.h -- C ABI of the MI355X-native PyraPose hot path (gfx950 / CDNA4).
 *
 * Drop-in boundary for the path SURVEY.md §8 scopes: ResNet-50 -> PyraPose feature pyramid ->
 * shared heads (forward + backward), its losses, optimizer, and the anchor / target / decode ops.
 * The reference has no operator ABI for this path (it is a Keras graph); each entry
 * point below names the reference symbol (file:line, relative to the reference repo root) whose
 * arithmetic it replaces.  The reference's only real C ABI (`uncertainty_pnp/src/ext.h:1-9`) sets
 * the house style: caller-owned buffers, plain pointers and sizes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - tensors are NHWC float32 matrices `[rows][ld]` (row = (image, y, x), ld >= channels);
 *   - all launches are asynchronous on the ctx stream; no hidden device synchronisation;
 *   - return: 0 ok; < 0 argument / shape error detected on the host before any launch;
 *     > 0 a hipError_t.  Never throws, never aborts.  `pp_last_error_string` gives detail.
 *   - one ctx per (process, device, stream); calls on one ctx are not thread-safe, calls on
 *     different ctxs are independent.
 */
#ifndef PYRAPOSE_HIP_H
#define PYRAPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_OK 0
#define PP_ERR_ARG (-1)
#define PP_ERR_SHAPE (-2)
#define PP_ERR_ALIGN (-3)
#define PP_ERR_NOCTX (-4)
#define PP_ERR_UNSUPPORTED (-5) /* an optional dependency is missing (librccl.so for the pp_comm_* / pp_allreduce_* entry points) */
#define PP_ERR_COMM (-6)        /* RCCL returned an error (text in pp_last_error) */

#define PP_MAX_SEG 5

typedef struct pp_ctx pp_ctx;

/* ---- context -------------------------------------------------------------------------- */
int pp_ctx_create(pp_ctx** out, int device, void* hip_stream);
void pp_ctx_destroy(pp_ctx* ctx);
int pp_ctx_set_stream(pp_ctx* ctx, void* hip_stream);
/* Optional scratch for the bf16x3 convolutions (any contents).  With it, launches whose output tiles cannot fill the
 * chip split their reduction over S workgroups per tile: each writes its partial sums to slice s of the scratch and a
 * finishing pass adds the slices in a fixed order and applies bias / residual / mask / ReLU (deterministic, no
 * atomics).  A launch uses S * rows * ld_out * 4 bytes.  One buffer per context (= per stream); NULL, 0 removes it. */
int pp_ctx_set_workspace(pp_ctx* ctx, void* device_buffer, size_t bytes);
/* One-shot: the NEXT pp_conv2d_nhwc_fwd_bf16x3 / pp_conv2d_nhwc_bwd_data_bf16x3 call on this context, which must be given
 * its gathered operand (x / dy) as float32, also writes that operand's bf16 (hi, lo) split into these planes -- the
 * output of pp_split_planes_bf16x3 on it ([rows][ld] geometry; columns past the channels the conv reads are left
 * untouched).  The kernel has the converted values in registers anyway; the weight-gradient launch of the same layer
 * (pp_conv2d_nhwc_bwd_weight_bf16x3 with all four planes) then skips its own conversion.  NULL, NULL cancels. */
int pp_ctx_set_split_capture(pp_ctx* ctx, void* hi, void* lo);
/* Row-block skip for sparse gradients.  The 3D-box loss (losses.py:321-408, orthogonal_l1) keeps only the rows with anchor
 * state 1, so the gradient that flows back through the 3D-box head is exactly zero away from the positive anchors (and
 * stays so layer after layer, dilated by one pixel per 3x3 conv).  pp_row_block_list scans a gradient tensor x [rows][ld]
 * (first `cols` columns) once: flags[b] = 1 when the 32-row block b holds a non-zero (or a NaN), list = {count, the flagged
 * block indices ascending}.  With nb = ceil(rows/32): flags holds 2 nb bytes and list 2 (nb + 1) ints of device memory --
 * the first halves are the result, the second halves scratch of the bwd-data launch that takes the hint (after that launch the
 * second nb bytes of `flags` flag every block of dx that may hold a non-zero: the dilated blocks, or all of them when the
 * launch ran dense; without an addend dx is zero in the others -- pp_row_block_list_planes_within takes that as `within`).
 * pp_ctx_set_row_block_skip is one-shot: the NEXT pp_conv2d_nhwc_bwd_weight_bf16x3 call on this context (float32 operands)
 * reduces over the listed blocks of dy only; the NEXT pp_conv2d_nhwc_bwd_data_bf16x3 call (3x3, stride 1, pad 1) computes
 * only the 32-row blocks of dx that a flagged block of dy can reach (dilated by one pixel in 2-D, compacted four to a tile)
 * and writes mask?(addend or 0) to the others.  A
 * block of zero rows contributes exactly 0.0 to every sum: the results are those of the dense launch (up to the order of
 * the float32 atomics between reduction splits; the one exception is an Inf / NaN operand opposite an exact zero, which the
 * dense launch turns into NaN and this one into 0).  Launches that cannot use the hint run dense.  NULL, NULL cancels. */
int pp_row_block_list(pp_ctx* ctx, const float* x, int rows, int ld, int cols, unsigned char* flags, int* list);
/* the same scan of a tensor stored as bf16 (hi, lo) planes ([rows][ld] each) */
int pp_row_block_list_planes(pp_ctx* ctx, const void* x_hi, const void* x_lo, int rows, int ld, int cols, unsigned char* flags, int* list);
/* the same, reading only the blocks flagged in `within` ([n_blocks] bytes; NULL = all): for a tensor the caller knows to be zero
 * elsewhere -- the data gradient a row-block-skip launch has just written is zero outside the blocks that launch listed (the
 * second n_blocks bytes of ITS flags buffer), so the scan for the next layer reads 10-25 % of the tensor instead of all of it. */
int pp_row_block_list_planes_within(pp_ctx* ctx, const void* x_hi, const void* x_lo, int rows, int ld, int cols, const unsigned char* within,
                                    unsigned char* flags, int* list);
/* One-shot: the NEXT pp_conv2d_nhwc_fwd_bf16x3 (residual) / pp_conv2d_nhwc_bwd_data_bf16x3 (addend, relu_src) call on this
 * context reads those epilogue operands from bf16 (hi, lo) planes instead of float32 tensors -- the storage format of every
 * activation and gradient a bf16x3 conv produces when its output is requested as planes only (value = hi + lo, 4 bytes per
 * element like float32, so no conv ever converts inside its loop).  The call's ld_res / ld_add / ld_rs arguments give the
 * row pitch of the planes; its float32 residual / addend / relu_src argument must then be NULL.  Of the ReLU source only the
 * hi plane is read (hi > 0 <=> value > 0).  Any of the three may be NULL (add_hi and add_lo go together). */
int pp_ctx_set_epilogue_planes(pp_ctx* ctx, const void* add_hi, const void* add_lo, const void* mask_hi);
int pp_ctx_set_row_block_skip(pp_ctx* ctx, const unsigned char* flags, const int* list);
/* Lazy sparse gradients (one-shot, for the NEXT bwd-data / bwd-weight call on this context, beside pp_ctx_set_row_block_skip).
 * lazy_out: a bwd-data call that runs the listed-block launch leaves the rows of dx outside the blocks it computes UNTOUCHED instead
 * of writing mask?(addend or 0) to them -- legitimate when every reader of that dx goes by flags: a scan restricted to the computed
 * blocks (pp_row_block_list_planes_within with the second half of this call's flags), a listed-block bwd-weight, a listed-block
 * bwd-data (whose gather never fetches a row outside the flagged blocks of its dy).  lazy_in: the dy of the call is such a tensor;
 * the call FAILS (PP_ERR_ARG) instead of running a launch that would read the unwritten rows.  The 3D-box head's backward uses
 * both: four fill passes of 103 MB per training step are not made. */
int pp_ctx_set_row_block_lazy(pp_ctx* ctx, int lazy_out, int lazy_in);
const char* pp_last_error_string(pp_ctx* ctx);
const char* pp_version(void);
/* number of compute units / name of the device the ctx is bound to (for bench metadata) */
int pp_device_info(pp_ctx* ctx, int* n_cu, char* name_host, int name_len);

/* ---- convolution family ---------------------------------------------------------------
 * Replaces the TensorFlow conv kernels behind every keras.layers.Conv2D of
 *   models/retinanet.py:9-54,57-98,101-131 (heads), :180-214 (__create_sparceFPN) and
 *   keras_resnet.models.ResNet50 called at models/resnet.py:87 (backbone).
 * Implicit-GEMM direct convolution (no im2col buffer) on v_mfma_f32_32x32x2_f32.
 *
 * A "row space" is a list of up to PP_MAX_SEG segments; segment s holds n_img images of
 * h[s] x w[s] cells, stored image-major, and rows are numbered segment after segment.  One
 * segment = an ordinary NHWC tensor; several segments = pyramid levels that share weights
 * (models/retinanet.py:224-225) processed by ONE launch.
 */
typedef struct {
  int n_img;             /* images per segment (batch) */
  int n_seg;             /* 1..PP_MAX_SEG */
  int h[PP_MAX_SEG];     /* cells per image, per segment */
  int w[PP_MAX_SEG];
} pp_rowspace;

typedef struct {
  pp_rowspace in;        /* geometry of x (forward input) */
  pp_rowspace out;       /* geometry of y (forward output); out.n_seg == in.n_seg */
  int cin, cout;         /* logical channels */
  int kh, kw, stride;    /* square stride */
  int pad_t, pad_l;      /* top / left zero padding (TF 'same' with stride 2 on even extents pads
                            bottom/right only -> pad_t = pad_l = 0; ZeroPadding2D(1) -> 1) */
  int ld_x, ld_y, ld_w;  /* leading dimensions (floats): x rows, y rows, weight rows.
                            weights are Keras HWIO flattened: w[(ky*kw+kx)*cin + ci][co], ld_w >= cout,
                            ld_w % 16 == 0, padding columns zero. cin % 16 == 0, or cin == 4 (packed-RGB stem:
                            the weight buffer then holds kh*kw*4 rows rounded up to a multiple of 16, zero rows). */
} pp_conv_desc;

/* y = [relu]( conv(x, w) + bias + residual ).  bias / residual may be NULL.  residual has y's shape
 * with leading dimension ld_res. */
int pp_conv2d_nhwc_fwd(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const float* w,
                       const float* bias, const float* residual, int ld_res, int relu, float* y);

/* dx = mask( conv_transpose(dy, w) + addend ),  mask(v) = relu_src > 0 ? v : 0 (relu_src NULL = no mask).
 * dy rows have ld_y floats with channels >= cout zero up to the next multiple of 16.
 * addend / relu_src have dx's shape (leading dims ld_add / ld_rs). */
int pp_conv2d_nhwc_bwd_data(pp_ctx* ctx, const pp_conv_desc* d, const float* dy, const float* w,
                            const float* addend, int ld_add, const float* relu_src, int ld_rs, float* dx);

/* dw += x^T (*) dy  (atomic accumulation into a caller-zeroed HWIO buffer, ld_w);  if dbias != NULL,
 * dbias[co] += sum_rows dy[row][co]. */
int pp_conv2d_nhwc_bwd_weight(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const float* dy,
                              float* dw, float* dbias);

/* ---- float32-class convolution on the bf16 matrix cores ("bf16x3") ---------------------------------------
 * Same operators as above with every product evaluated as x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on
 * v_mfma_f32_32x32x16_bf16 (f32 accumulation; ~2^-16 relative error per product, 5.3x the f32-MFMA rate).
 * Activations stay float32; weights are split once per optimizer step into bf16 (hi, lo) planes:
 *   forward planes  [tap][cout][cin]              (cin % 32 == 0)
 *   bwd-data planes [tap][cin][cout rounded to 32] (zero padded)
 * Any of the plane pairs may be NULL in pp_conv_split_weights_bf16x3. */
int pp_conv_split_weights_bf16x3(pp_ctx* ctx, const pp_conv_desc* d, const float* w, void* fwd_hi, void* fwd_lo,
                                 void* dgrad_hi, void* dgrad_lo);
/* The same for many tensors in one launch.  jobs_dev: DEVICE array of n_jobs entries; job i owns the tiles
 * [tile_begin, tile_begin + taps * (cin/32) * ceil(cout/32)) of the launch, tile_begin ascending from 0;
 * total_tiles = the sum.  Plane pointers as in pp_conv_split_weights_bf16x3 (either pair may be NULL). */
typedef struct pp_split_job {
  const float* w;      /* f32 HWIO [taps*cin][ld_w] */
  void* fwd_hi; void* fwd_lo; void* dg_hi; void* dg_lo;
  int taps, cin, cout, ld_w;
  int tile_begin, reserved;
} pp_split_job;
int pp_conv_split_weights_bf16x3_batch(pp_ctx* ctx, int n_jobs, const pp_split_job* jobs_dev, int total_tiles);
/* Sparse FORWARD of a head whose loss reads a few rows only (training; the 3D-box head: orthogonal_l1 keeps the rows with anchor
 * state 1, losses.py:332-333 -- every other output row of that head is dead in train_on_batch).
 * pp_positive_row_blocks: flags[b] = 1 when the 32-row block b of the head's row space holds an anchor whose last target column
 * (state, column stride - 1 of y_true [n_img][cells][A][stride]) is 1.  pp_row_block_dilate: the blocks within one pixel of a
 * flagged block (what a 3x3 stride-1 conv d reads to produce the flagged blocks): applied once per layer from the head's output
 * back to its first conv.  pp_ctx_set_row_block_out is one-shot: the NEXT pp_conv2d_nhwc_fwd_bf16x3 call on this context (3x3,
 * stride 1, pad 1, plane-stored input, no residual) computes the flagged 32-row output blocks only (compacted four to a tile;
 * list = scratch of n_blocks + 1 ints) and leaves every other output row as it is.  Exact for the loss, its gradient and the
 * weight gradients (the backward reads those activations only where its own row-block skip goes); the caller opts in. */
int pp_positive_row_blocks(pp_ctx* ctx, const pp_rowspace* rs, int A, int stride, const float* y_true, unsigned char* flags);
int pp_row_block_dilate(pp_ctx* ctx, const pp_conv_desc* d, const unsigned char* in_flags, unsigned char* out_flags);
int pp_ctx_set_row_block_out(pp_ctx* ctx, const unsigned char* flags, int* list);
/* f32 tensor of n elements (n % 8 == 0) -> bf16 (hi, lo) planes with the same [rows][ld] geometry.  Convs that are
 * given planes for their gathered operand skip the conversion inside the kernel (the f32 pointer may then be NULL). */
int pp_split_planes_bf16x3(pp_ctx* ctx, size_t n, const float* src, void* hi, void* lo);
/* Format of every (hi, lo) plane pair the *_bf16x3 / *_v entry points of this context read and write -- and with it the arithmetic
 * of the convolutions on them (csrc/planes_fmt.h).  0 (default): bf16 pairs, value = hi + lo, three bf16 MFMAs per product
 * ("bf16x3", 4.5e-6 per launch against float64).  1: "P16" -- hi = IEEE half, lo = two e5m2 bytes per element (e5m2(x), e5m2((x - hi)
 * * 2^12); swapped for weights), value = hi + lo8 * 2^-12; one f16 MFMA + half a block-scaled e5m2 MFMA per 16-deep step ("f16c8":
 * 2 MFMA units per product instead of 3, 2.1e-5 per launch, |x| clamped to 28672).  Same packed geometry, same entry points; a
 * tensor written under one format must be read under the same one.  pp_convert_planes re-encodes n elements (n % 8 == 0) from one
 * format into the other, multiplied by scale2_dev[scale_index] when scale2_dev != NULL (pp_grad_scale_from_counts), and set to zero
 * where the tensor whose hi plane is relu_src_hi (same geometry, either format; NULL: none) is not positive -- the ReLU a gradient
 * passes when it crosses the boundary backwards. */
int pp_ctx_set_planes_format(pp_ctx* ctx, int fmt);
int pp_convert_planes(pp_ctx* ctx, size_t n, const void* src_hi, const void* src_lo, int src_fmt, void* dst_hi, void* dst_lo, int dst_fmt,
                      const float* scale2_dev, int scale_index, const void* relu_src_hi);
/* The gradient chain travels multiplied by a power of two (the hi plane holds IEEE halves, which stop at 6e-8; loss gradients are
 * ~1e-7): pp_grad_scale_from_counts writes scale2 = {2^G, 2^-G} with 2^G = 2^8 * 2^floor(log2(max(1, min_i counts[i]))) -- every loss
 * gradient of losses.py:22-68 / :321-408 is bounded by ~1 / max(1, positives of its head); pp_split_planes_scaled_bf16x3 is
 * pp_split_planes_bf16x3 of src * scale_dev[0] (the three loss gradients); every gradient a bwd-data launch or a pointwise kernel
 * derives from them carries the same factor; pp_ctx_set_grad_scale (persistent; NULL = none) makes the weight-gradient launches of
 * this context multiply dW / dbias by scale2[1] when their operands are planes. */
int pp_grad_scale_from_counts(pp_ctx* ctx, const int* counts_dev, int n_counts, float* scale2_dev);
/* the same with G shifted by log2_adjust (-16 .. 16): a caller whose loss weights differ from the reference's defaults (losses.py:
 * orthogonal_l1 weight 0.125, focal alpha 0.25) takes the headroom out of / puts it into the scale -- Engine: -ceil(log2(largest ratio)) */
int pp_grad_scale_from_counts_adj(pp_ctx* ctx, const int* counts_dev, int n_counts, float* scale2_dev, int log2_adjust);
int pp_split_planes_scaled_bf16x3(pp_ctx* ctx, size_t n, const float* src, void* hi, void* lo, const float* scale_dev);
int pp_ctx_set_grad_scale(pp_ctx* ctx, const float* scale2_dev);

int pp_conv2d_nhwc_fwd_bf16x3(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const void* x_hi, const void* x_lo,
                              const void* w_fwd_hi, const void* w_fwd_lo, const float* bias, const float* residual,
                              int ld_res, int relu, float* y, void* y_hi, void* y_lo);
/* y_hi / y_lo (may be NULL): the epilogue also writes the output pre-split, so that the consumer convs skip the split.
 * With planes given, y (fwd) / dx (bwd_data) may be NULL: the output then exists only as planes. */
/* dy rows need ld_y >= cout rounded up to 32 with zero padding. */
int pp_conv2d_nhwc_bwd_data_bf16x3(pp_ctx* ctx, const pp_conv_desc* d, const float* dy, const void* dy_hi,
                                   const void* dy_lo, const void* w_dgrad_hi, const void* w_dgrad_lo,
                                   const float* addend, int ld_add, const float* relu_src, int ld_rs, float* dx,
                                   void* dx_hi, void* dx_lo);
/* dw += x^T (*) dy; operands either f32 (split on the fly) or all four planes; same contract as
 * pp_conv2d_nhwc_bwd_weight, cin % 64 == 0. */
int pp_conv2d_nhwc_bwd_weight_bf16x3(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const float* dy,
                                     const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo,
                                     float* dw, float* dbias);

/* ---- data-parallel gradient exchange on RCCL -----------------------------------------------------------
 * The reference trains on one device (bin/train.py:82-89: the multi_gpu_model branch is disabled); what its single-device
 * semantics fix is that the loss normalisers count positives over the WHOLE batch (losses.py:62-66, :402-405) and that Adam clips
 * by the GLOBAL gradient norm (bin/train.py:101).  With the batch sharded per image over one process per GPU that takes a SUM
 * all-reduce of the three positive counts and a SUM all-reduce of the flat gradient buffer -- here as plain entry points on a
 * communicator the library owns, so that the engine decides the stream (its own), the bucket and the moment (pyrapose_amd/
 * parallel.py launches a bucket as soon as the last weight-gradient kernel that writes into it is enqueued).
 * librccl.so is loaded with dlopen on first use (PP_RCCL_LIB overrides the name); without it these calls return
 * PP_ERR_UNSUPPORTED and pp_comm_available() is 0.  id128: the 128 bytes of an ncclUniqueId -- rank 0 calls pp_comm_unique_id and
 * hands the bytes to every rank by any channel; pp_comm_init is collective over the `world` ranks (one communicator per process,
 * bound to the context's device).  All-reduces run in place, asynchronously, on the context's stream. */
typedef struct pp_comm pp_comm;
int pp_comm_available(void);
int pp_comm_unique_id(pp_ctx* ctx, void* id128);
int pp_comm_init(pp_ctx* ctx, int world, int rank, const void* id128, pp_comm** out);
int pp_comm_destroy(pp_comm* comm);
int pp_allreduce_bucket(pp_ctx* ctx, pp_comm* comm, float* buf, size_t count);
int pp_allreduce_counts(pp_ctx* ctx, pp_comm* comm, int* counts, int n);

/* ---- pooling / resampling / pointwise --------------------------------------------------
 * keras_resnet pool1 = MaxPooling2D(3, strides 2, 'same') (called via models/resnet.py:87). */
int pp_maxpool3x3s2_fwd(pp_ctx* ctx, int n_img, int h, int w, int c, const float* x, int oh, int ow, float* y);
/* UpsampleLike: layers/_misc.py:96-109 -> backend/tf_backend.py:28-35 (tf.image.resize NEAREST, TF 2.1:
 * src = floor((dst + 0.5) * in / out)).   out = up(src) + other  (other may be NULL). */
int pp_upsample_nearest_add_fwd(pp_ctx* ctx, int n_img, int sh, int sw, int th, int tw, int c,
                                const float* src, const float* other, float* out);
/* dsrc = base + sum over the target cells that map to each source cell of dtarget (base may be NULL) */
int pp_upsample_nearest_add_bwd(pp_ctx* ctx, int n_img, int sh, int sw, int th, int tw, int c,
                                const float* dtarget, const float* base, float* dsrc);
/* out = a + b (+ c)   (keras.layers.Add, models/retinanet.py:198-211); b, c may be NULL */
int pp_add_n(pp_ctx* ctx, size_t n, const float* a, const float* b, const float* c, float* out);
/* y = max(x, 0): the stand-alone Activation('relu') between P6 and the P7 conv of __create_pyramid_features
 * (models/retinanet.py:154).  Its backward is the relu_src mask of the consumer's pp_conv2d_nhwc_bwd_data*. */
int pp_relu_fwd(pp_ctx* ctx, size_t n, const float* x, float* y);
/* The same pointwise ops on tensors in either storage format: a view is a float32 tensor (f32) or a pair of bf16 (hi, lo)
 * planes with the same [rows][ld] geometry (value = hi + lo); an OUTPUT view may carry both, and both are then written.
 * Inputs that may be NULL in the float32 entry points may be NULL views (all three pointers NULL) here. */
typedef struct pp_tview {
  const float* f32;
  const void* hi;
  const void* lo;
} pp_tview;
int pp_add_n_v(pp_ctx* ctx, size_t n, const pp_tview* a, const pp_tview* b, const pp_tview* c, const pp_tview* out);
int pp_relu_fwd_v(pp_ctx* ctx, size_t n, const pp_tview* x, const pp_tview* y);
int pp_upsample_nearest_add_fwd_v(pp_ctx* ctx, int n_img, int sh, int sw, int th, int tw, int c, const pp_tview* src,
                                  const pp_tview* other, const pp_tview* out);
int pp_upsample_nearest_add_bwd_v(pp_ctx* ctx, int n_img, int sh, int sw, int th, int tw, int c, const pp_tview* dtarget,
                                  const pp_tview* base, const pp_tview* dsrc);
/* planes -> float32 (value = hi + lo), n % 4 == 0: the inverse of pp_split_planes_bf16x3 up to 2^-17 */
int pp_merge_planes_bf16x3(pp_ctx* ctx, size_t n, const void* hi, const void* lo, float* dst);
/* Audit of a P16 tensor [rows][ld] (packed planes, columns < cols; ld, cols % 8 == 0; the context must be in plane format 1): ADDS to
 * stats5_dev[0..3] (uint64, device) the elements looked at, the non-zero halves, the halves AT the encode's clamp (|h| >= 28 672) and
 * the subnormal halves (0 < |h| < 2^-14); stats5_dev[4] = max(stats5_dev[4], largest |half| seen as its 15 bits).  within (may be NULL): uint8 flags of the 32-row blocks to look at.  The P16 encode clamps
 * and a half underflows silently (csrc/p16.h); this is how a caller sees whether a step came near either end: Engine.p16_stats(),
 * asserted in tests/test_gpu_parity.py, printed by bench.py.  No reference counterpart (the reference computes in float32). */
int pp_planes_stats(pp_ctx* ctx, const void* hi, const void* lo, long long rows, int ld, int cols, const unsigned char* within,
                    unsigned long long* stats5_dev);
/* [n_img,h,w,3] -> [n_img,h,w,4] zero-padded channel (feeds conv1 as cin == 4) */
int pp_pack_rgb_to_4(pp_ctx* ctx, size_t n_pixels, const float* x3, float* x4);
/* utils/image.py:35-62 preprocess_image(mode='caffe') + preprocessing/generator.py:319-336 compute_inputs in one pass:
 * images_u8 [n_img,H,W,3] uint8 (BGR as the reference reads them; image b occupies the upper-left sizes_hw[b] = (h, w)
 * corner of the frame) -> x4 [n_img,H,W,4] float32 = pixel - (103.939, 116.779, 123.68) inside the image, 0 in the
 * padding and in channel 3.  sizes_hw is a HOST array of 2*n_img ints (n_img <= 64). */
int pp_preprocess_caffe_u8(pp_ctx* ctx, int n_img, int H, int W, const int* sizes_hw_host, const unsigned char* images_u8,
                           float* x4);
/* ---- geometric augmentation / resize of the input pipeline ------------------------------------------
 * Replace the reference's OpenCV calls: utils/image.py:150-216 apply_transform (cv2.warpAffine of the uint8 image, INTER_LINEAR,
 * border per TransformParameters: 'constant' / 'nearest' = replicate), :219-230 apply_transform2mask (cv2.warpAffine of the
 * id mask, INTER_NEAREST, BORDER_CONSTANT 0), :281-323 compute_resize_scale / resize_image (cv2.resize, fx = fy = scale).
 * OpenCV's fixed-point scheme for 8-bit images, in integer arithmetic (bit-exact against oracle/image_np.py; OpenCV itself is
 * not installed: parity unpinned).  mats_host: n_img (<= 64) row-major 2x3 FORWARD matrices as the reference passes them to
 * cv2.warpAffine (the kernels invert them like OpenCV does).  interpolation 0 = nearest (1 channel), 1 = linear (1 or 3
 * channels); border 0 = constant (cval), 1 = replicate.  src / dst: [n_img][H][W][channels] uint8, distinct buffers. */
int pp_warp_affine_u8(pp_ctx* ctx, int n_img, int H, int W, int channels, const double* mats_host, int interpolation, int border,
                      int cval, const unsigned char* src, unsigned char* dst);
/* compute_resize_scale (host): min side -> min_side unless the max side would exceed max_side */
int pp_resize_scale(int rows, int cols, int min_side, int max_side, double* scale);
/* cv2.resize(img, None, fx = fy = scale) bilinear: dst [n_img][DH][DW][channels] with DH = round(SH * scale), DW = round(SW * scale) */
int pp_resize_linear_u8(pp_ctx* ctx, int n_img, int SH, int SW, int channels, double scale, int DH, int DW, const unsigned char* src,
                        unsigned char* dst);
/* The same two producers writing into a zero frame [n_img][Hp][Wp][4] with the image at (pad, pad): the input layout of
 * pp_stem7x7s2_fwd_bf16x3 (pad = 3). */
int pp_pack_rgb_to_4_padded(pp_ctx* ctx, int n_img, int H, int W, int Hp, int Wp, int pad, const float* x3, float* x4p);
int pp_preprocess_caffe_u8_padded(pp_ctx* ctx, int n_img, int H, int W, int Hp, int Wp, int pad, const int* sizes_hw_host,
                                  const unsigned char* images_u8, float* x4p);
/* keras_resnet conv1 (ZeroPadding2D(3) + Conv2D(64, 7, strides 2), models/resnet.py:87) + folded BN + ReLU on the bf16x3
 * path: x4p as above (Hp >= H + 6, Wp >= W + 8, even), weight planes [7][cout][32] (kernel row ty, column tx * 4 + c),
 * y [n_img * OH * OW][ld_y] with OH = (H - 1) / 2 + 1. */
int pp_stem7x7s2_fwd_bf16x3(pp_ctx* ctx, int n_img, int H, int W, int Hp, int Wp, const float* x4p, const void* w_hi,
                            const void* w_lo, int cout, const float* bias, int relu, float* y, int ld_y);

/* ---- head output export ---------------------------------------------------------------
 * Head convs write level-major matrices [rows][ld]; Keras concatenates the per-level reshapes on
 * axis 1 (models/retinanet.py:224-229) and applies sigmoid to cls / mask (:52, :96).
 * out[b][level_off + cell*A + a][v] = f(src[row(level,b,cell)][a*V + v]),  f = sigmoid or identity. */
int pp_export_head(pp_ctx* ctx, const pp_rowspace* rs, int n_anchor, int n_val, const float* src, int ld,
                   int apply_sigmoid, float* out);

/* ---- losses -----------------------------------------------------------------------------
 * counts[0..2] += #(state == 1) in y_true_3dbox / y_true_cls / y_true_mask (last column).
 * Normalisers of losses.py:62-66 and :402-405 (over the WHOLE batch). */
int pp_count_positives(pp_ctx* ctx, size_t rows_box, const float* y_box, size_t rows_cls, int c_cls,
                       const float* y_cls, size_t rows_mask, int c_mask, const float* y_mask, int* counts);
/* focal(alpha, gamma): losses.py:22-68 (cls and mask heads, bin/train.py:98-99).
 * logits: level-major [rows][ld] pre-sigmoid; y_true: Keras layout (B, N, C+1);
 * loss_sum += sum(focal)/normaliser; dlogits (same layout as logits, padding channels zeroed)
 * = d(loss)/d(logit) * loss_weight.  normaliser = max(1, *count).  dlogits may be NULL. */
int pp_sigmoid_focal_fwd_bwd(pp_ctx* ctx, const pp_rowspace* rs, int n_anchor, int n_class,
                             const float* logits, int ld, const float* y_true, float alpha, float gamma,
                             const int* count, float loss_weight, float* loss_sum, float* dlogits);
/* orthogonal_l1(weight .125, sigma 3): losses.py:321-408 ('3Dbox', bin/train.py:97). */
int pp_orth_smoothl1_fwd_bwd(pp_ctx* ctx, const pp_rowspace* rs, int n_anchor, const float* pred, int ld,
                             const float* y_true, float weight, float sigma, const int* count,
                             float loss_weight, float* loss_sum, float* dpred);

/* ---- optimizer: keras 2.3.1 Adam(lr, clipnorm) as compiled at bin/train.py:101 --------
 * A parameter set is a flat float32 buffer cut into tensors described by pp_param_desc. */
typedef struct {
  long long offset;      /* first element in the flat buffers */
  long long count;       /* elements (rows * ld) */
  int ld;                /* row length; scale index = element % ld */
  int trainable;         /* 0 = frozen (models/resnet.py:100-103) */
  long long scale_off;   /* offset into `scales` of the per-output-channel frozen-BN scale, or -1 */
  float l2;              /* kernel_regularizer l2 coefficient (models/retinanet.py:108), else 0 */
} pp_param_desc;

typedef struct pp_optimizer pp_optimizer;
int pp_optimizer_create(pp_ctx* ctx, pp_optimizer** out, const pp_param_desc* descs_host, int n_desc,
                        long long total);
void pp_optimizer_destroy(pp_optimizer* opt);
/* gnorm_sq[0] = sum over trainable tensors of || scale * g_eff + 2*l2*w ||^2 (fixed reduction order).
 * l2_loss (may be NULL) += sum l2 * w^2. */
int pp_grad_global_norm(pp_ctx* ctx, pp_optimizer* opt, const float* w_master, const float* g_eff,
                        const float* scales, float* gnorm_sq, float* l2_loss);
/* One Adam step with global-norm clipping (scale every gradient by clipnorm/norm when norm >= clipnorm);
 * rewrites w_eff = w_master * scale.  `step` is 1-based: lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t). */
int pp_adam_step_clipnorm(pp_ctx* ctx, pp_optimizer* opt, float* w_master, float* w_eff, const float* g_eff,
                          const float* scales, float* m, float* v, const float* gnorm_sq, float lr,
                          float beta1, float beta2, float eps, float clipnorm, long long step);

/* ---- anchors / targets / decode -----------------------------------------------------------
 * utils/anchors.py:447-478 (generate_anchors) -- host, float64, bit-exact op order. */
int pp_generate_base_anchors_host(int base_size, const float* ratios_host, int n_ratios,
                                  const float* scales_host, int n_scales, double* out_host);
/* utils/anchors.py:415-444 + :372-412: all levels, float64 [N,4].  base_anchors_host: [n_levels][A][4]. */
int pp_anchors_shift_f64(pp_ctx* ctx, int n_levels, const int* feat_h_host, const int* feat_w_host,
                         const int* strides_host, int n_anchor, const double* base_anchors_host,
                         double* anchors_out);
/* layers/_misc.py:60-71 -> backend/common.py:93-116: float32 device-side anchors [N,4]. */
int pp_anchors_shift_f32(pp_ctx* ctx, int n_levels, const int* feat_h_host, const int* feat_w_host,
                         const int* strides_host, int n_anchor, const double* base_anchors_host,
                         float* anchors_out);
/* utils/compute_overlap.pyx:13-53: float64 IoU, '+1' convention, [N,K] row-major. */
int pp_compute_overlap_f64(pp_ctx* ctx, int n, const double* boxes, int k, const double* query, double* overlaps);
/* utils/anchors.py:290-318: first-max argmax + thresholds. state: 1 positive, -1 ignore, 0 background. */
int pp_compute_gt_annotations(pp_ctx* ctx, int n, const double* anchors, int k, const double* gt_boxes,
                              double negative_overlap, double positive_overlap, int* argmax, signed char* state);

/* utils/anchors.py:72-287 anchor_targets_bbox, whole batch in one call.
 * Ground truth is packed: image b owns gt_offset_host[b] .. gt_offset_host[b+1]-1.
 *   gt_boxes  [G,4] f64 (x1,y1,x2,y2);  gt_labels [G] i32;  gt_box3d [G,16] f64 projected corner
 *   pixels (anchors.py:207-215 is evaluated by pp_project_box3d_host);  gt_mask_ids [G] i32;
 *   id_masks: uint8 [B, mask_h, mask_w] object-id images, image b valid in its top-left
 *   mask_hw_host[b] = (h, w) corner (NULL = all full size);  image_hw_host [B,2] unpadded image sizes.
 * Outputs (float32, Keras layout, fully written): regression [B,N,17], labels [B,N,C+1],
 * mask [B,MH*MW,C+1] with (MH,MW) = level-3 shape of image 0. */
int pp_anchor_targets(pp_ctx* ctx, int n_anchor_total, const double* anchors, int batch,
                      const int* gt_offset_host, const double* gt_boxes, const int* gt_labels,
                      const double* gt_box3d, const int* gt_mask_ids, const unsigned char* id_masks,
                      int mask_h, int mask_w, const int* mask_hw_host, const int* image_hw_host, int num_classes,
                      double negative_overlap, double positive_overlap, int out_mh, int out_mw,
                      float* regression, float* labels, float* mask);
/* utils/anchors.py:207-215 + toPix_array :562-567 (host, float64; quat2mat = transforms3d 0.3.1). */
int pp_project_box3d_host(const double* pose7_host, const double* box8x3_host, const double* cam4_host,
                          double* out16_host);
/* PIL NEAREST index map used at anchors.py:158 (host). */
int pp_pil_nearest_index_host(int n_in, int n_out, int* out_host);

/* layers/_misc.py:195-197 -> backend/common.py:25-56: boxes3D = anchors (+) 0.2 * reg * (w|h), float32. */
int pp_box3d_decode(pp_ctx* ctx, int batch, int n, const float* anchors, const float* regression, float* boxes3d);
/* utils/linemod_eval.py:317-319: per (image, class) ascending indices with score > thr.
 * idx_out [B,C,cap] (int32, -1 padded), counts [B,C]. */
int pp_score_threshold_compact(pp_ctx* ctx, int batch, int n, int n_class, const float* scores, float thr,
                               int cap, int* idx_out, int* counts);
/* layers/filter_detections.py:21-118 (class-specific, NMS on).  boxes [N,4], boxes3d [N,16],
 * scores [N,C] for ONE image.  Outputs padded with -1 to max_det.  workspace >= pp_filter_workspace_bytes. */
size_t pp_filter_workspace_bytes(int n, int n_class, int max_det);
int pp_filter_detections(pp_ctx* ctx, int n, int n_class, const float* boxes, const float* boxes3d,
                         const float* scores, float score_thr, float iou_thr, int max_det, void* workspace,
                         float* out_boxes, float* out_boxes3d, float* out_scores, int* out_labels);
/* The same for n_img images in one set of launches (the Keras layer maps filter_detections over the batch,
 * filter_detections.py:182-196): tensors gain a leading image dimension, workspace >= n_img * pp_filter_workspace_bytes. */
int pp_filter_detections_batch(pp_ctx* ctx, int n_img, int n, int n_class, const float* boxes, const float* boxes3d,
                               const float* scores, float score_thr, float iou_thr, int max_det, void* workspace,
                               float* out_boxes, float* out_boxes3d, float* out_scores, int* out_labels);

/* ---- pose-error metrics of the evaluation tail -------------------------------------------------------
 * utils/pose_error.py:210-228 add() and :231-246 adi(), as called at utils/linemod_eval.py:525-531 (decision:
 * error < 0.1 * model diameter).  float64; n_pose (R, t) pairs against ONE model point set pts [n_pts,3];
 * R row-major [n_pose,3,3], t [n_pose,3]; out [n_pose].  workspace >= pp_pose_error_workspace_bytes. */
size_t pp_pose_error_workspace_bytes(int n_pose, int n_pts);
int pp_pose_add_f64(pp_ctx* ctx, int n_pose, int n_pts, const double* pts, const double* R_est, const double* t_est,
                    const double* R_gt, const double* t_gt, void* workspace, double* out);
int pp_pose_adi_f64(pp_ctx* ctx, int n_pose, int n_pts, const double* pts, const double* R_est, const double* t_est,
                    const double* R_gt, const double* t_gt, void* workspace, double* out);

/* ---- RANSAC-PnP of the evaluation tail ----------------------------------------------------------------
 * In place of cv2.solvePnPRansac(obj_points, est_points, K, None, iterationsCount=300, reprojectionError=5.0,
 * confidence=0.99, flags=cv2.SOLVEPNP_ITERATIVE) + cv2.Rodrigues at utils/linemod_eval.py:479-485 (same call in the
 * other *_eval.py): n_problems independent problems (one per detected class and image) in one launch; problem p owns
 * the correspondences offsets[p] .. offsets[p+1] (device int array) of obj [N,3] / img [N,2] (float64, pixels) and the
 * intrinsics K4[p] = (fx, fy, cx, cy).  points_per_vote = 8 for the reference's layout (k votes x the 8 cuboid corners,
 * linemod_eval.py:421-431): a hypothesis then takes six distinct corners, each from a random vote; 0 = unstructured.
 * Out: R [P,3,3] row-major (what cv2.Rodrigues(rvec) returns), t [P,3], n_inliers [P], inlier_mask [N] (1 = squared
 * reprojection error < reproj_error^2), ok [P] (0: fewer than 4 inliers / no valid sample; R = I, t = 0 then).
 * Deterministic for a given seed (counter-based draws, fixed-order reductions).  OpenCV is not in the reference tree:
 * the estimator is this library's own (csrc/pnp.hip, restated in oracle/pnp_np.py) -- parity with cv2 is unpinned.
 * workspace >= pp_pnp_ransac_workspace_bytes.  All iterations are run (no confidence-based early exit). */
size_t pp_pnp_ransac_workspace_bytes(int n_problems, int iterations);
int pp_pnp_ransac_f64(pp_ctx* ctx, int n_problems, const int* offsets_dev, int n_points_total, const double* obj,
                      const double* img, const double* K4, int iterations, double reproj_error, unsigned long long seed,
                      int points_per_vote, void* workspace, double* R_out, double* t_out, int* n_inliers,
                      unsigned char* inlier_mask, int* ok);

#ifdef __cplusplus
}
#endif
#endif /* PYRAPOSE_HIP_H */
