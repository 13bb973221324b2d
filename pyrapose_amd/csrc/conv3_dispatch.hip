// Forwards the plane-format-dependent entry points of the conv3.hip family to the build the context asks for
// (pp_ctx_set_planes_format: 0 = bf16 pairs / bf16x3 arithmetic, 1 = P16 / f16c8 arithmetic; csrc/planes_fmt.h).
#include "pp_internal.h"

extern "C" int pp_row_block_list_fmt0(pp_ctx* ctx, const float* x, int rows, int ld, int cols, unsigned char* flags, int* list);
extern "C" int pp_row_block_list_fmt1(pp_ctx* ctx, const float* x, int rows, int ld, int cols, unsigned char* flags, int* list);
extern "C" int pp_row_block_list_planes_within_fmt0(pp_ctx* ctx, const void* x_hi, const void* x_lo, int rows, int ld, int cols, const unsigned char* within, unsigned char* flags, int* list);
extern "C" int pp_row_block_list_planes_within_fmt1(pp_ctx* ctx, const void* x_hi, const void* x_lo, int rows, int ld, int cols, const unsigned char* within, unsigned char* flags, int* list);
extern "C" int pp_row_block_list_planes_fmt0(pp_ctx* ctx, const void* x_hi, const void* x_lo, int rows, int ld, int cols, unsigned char* flags, int* list);
extern "C" int pp_row_block_list_planes_fmt1(pp_ctx* ctx, const void* x_hi, const void* x_lo, int rows, int ld, int cols, unsigned char* flags, int* list);
extern "C" int pp_split_planes_bf16x3_fmt0(pp_ctx* ctx, size_t n, const float* src, void* hi, void* lo);
extern "C" int pp_split_planes_bf16x3_fmt1(pp_ctx* ctx, size_t n, const float* src, void* hi, void* lo);
extern "C" int pp_split_planes_scaled_bf16x3_fmt0(pp_ctx* ctx, size_t n, const float* src, void* hi, void* lo, const float* scale_dev);
extern "C" int pp_split_planes_scaled_bf16x3_fmt1(pp_ctx* ctx, size_t n, const float* src, void* hi, void* lo, const float* scale_dev);
extern "C" int pp_grad_scale_from_counts_fmt0(pp_ctx* ctx, const int* counts_dev, int n_counts, float* scale2_dev);
extern "C" int pp_grad_scale_from_counts_adj_fmt0(pp_ctx* ctx, const int* counts_dev, int n_counts, float* scale2_dev, int log2_adjust);
extern "C" int pp_grad_scale_from_counts_adj_fmt1(pp_ctx* ctx, const int* counts_dev, int n_counts, float* scale2_dev, int log2_adjust);
extern "C" int pp_grad_scale_from_counts_fmt1(pp_ctx* ctx, const int* counts_dev, int n_counts, float* scale2_dev);
extern "C" int pp_conv_split_weights_bf16x3_batch_fmt0(pp_ctx* ctx, int n_jobs, const pp_split_job* jobs_dev, int total_tiles);
extern "C" int pp_conv_split_weights_bf16x3_batch_fmt1(pp_ctx* ctx, int n_jobs, const pp_split_job* jobs_dev, int total_tiles);
extern "C" int pp_conv_split_weights_bf16x3_fmt0(pp_ctx* ctx, const pp_conv_desc* d, const float* w, void* fwd_hi, void* fwd_lo, void* dgrad_hi, void* dgrad_lo);
extern "C" int pp_conv_split_weights_bf16x3_fmt1(pp_ctx* ctx, const pp_conv_desc* d, const float* w, void* fwd_hi, void* fwd_lo, void* dgrad_hi, void* dgrad_lo);
extern "C" int pp_conv2d_nhwc_fwd_bf16x3_fmt0(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const void* x_hi, const void* x_lo, const void* w_fwd_hi, const void* w_fwd_lo, const float* bias, const float* residual, int ld_res, int relu, float* y, void* y_hi, void* y_lo);
extern "C" int pp_conv2d_nhwc_fwd_bf16x3_fmt1(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const void* x_hi, const void* x_lo, const void* w_fwd_hi, const void* w_fwd_lo, const float* bias, const float* residual, int ld_res, int relu, float* y, void* y_hi, void* y_lo);
extern "C" int pp_row_block_dilate_fmt0(pp_ctx* ctx, const pp_conv_desc* d, const unsigned char* in_flags, unsigned char* out_flags);
extern "C" int pp_row_block_dilate_fmt1(pp_ctx* ctx, const pp_conv_desc* d, const unsigned char* in_flags, unsigned char* out_flags);
extern "C" int pp_stem7x7s2_fwd_bf16x3_fmt0(pp_ctx* ctx, int n_img, int H, int W, int Hp, int Wp, const float* x4p, const void* w_hi, const void* w_lo, int cout, const float* bias, int relu, float* y, int ld_y);
extern "C" int pp_stem7x7s2_fwd_bf16x3_fmt1(pp_ctx* ctx, int n_img, int H, int W, int Hp, int Wp, const float* x4p, const void* w_hi, const void* w_lo, int cout, const float* bias, int relu, float* y, int ld_y);
extern "C" int pp_conv2d_nhwc_bwd_data_bf16x3_fmt0(pp_ctx* ctx, const pp_conv_desc* d, const float* dy, const void* dy_hi, const void* dy_lo, const void* w_dgrad_hi, const void* w_dgrad_lo, const float* addend, int ld_add, const float* relu_src, int ld_rs, float* dx, void* dx_hi, void* dx_lo);
extern "C" int pp_conv2d_nhwc_bwd_data_bf16x3_fmt1(pp_ctx* ctx, const pp_conv_desc* d, const float* dy, const void* dy_hi, const void* dy_lo, const void* w_dgrad_hi, const void* w_dgrad_lo, const float* addend, int ld_add, const float* relu_src, int ld_rs, float* dx, void* dx_hi, void* dx_lo);
extern "C" int pp_conv2d_nhwc_bwd_weight_bf16x3_fmt0(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const float* dy, const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw, float* dbias);
extern "C" int pp_conv2d_nhwc_bwd_weight_bf16x3_fmt1(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const float* dy, const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw, float* dbias);

extern "C" int pp_row_block_list(pp_ctx* ctx, const float* x, int rows, int ld, int cols, unsigned char* flags, int* list) {
  return (ctx && ctx->planes_fmt == 1) ? pp_row_block_list_fmt1(ctx, x, rows, ld, cols, flags, list) : pp_row_block_list_fmt0(ctx, x, rows, ld, cols, flags, list);
}
extern "C" int pp_row_block_list_planes_within(pp_ctx* ctx, const void* x_hi, const void* x_lo, int rows, int ld, int cols, const unsigned char* within, unsigned char* flags, int* list) {
  return (ctx && ctx->planes_fmt == 1) ? pp_row_block_list_planes_within_fmt1(ctx, x_hi, x_lo, rows, ld, cols, within, flags, list) : pp_row_block_list_planes_within_fmt0(ctx, x_hi, x_lo, rows, ld, cols, within, flags, list);
}
extern "C" int pp_row_block_list_planes(pp_ctx* ctx, const void* x_hi, const void* x_lo, int rows, int ld, int cols, unsigned char* flags, int* list) {
  return (ctx && ctx->planes_fmt == 1) ? pp_row_block_list_planes_fmt1(ctx, x_hi, x_lo, rows, ld, cols, flags, list) : pp_row_block_list_planes_fmt0(ctx, x_hi, x_lo, rows, ld, cols, flags, list);
}
extern "C" int pp_split_planes_bf16x3(pp_ctx* ctx, size_t n, const float* src, void* hi, void* lo) {
  return (ctx && ctx->planes_fmt == 1) ? pp_split_planes_bf16x3_fmt1(ctx, n, src, hi, lo) : pp_split_planes_bf16x3_fmt0(ctx, n, src, hi, lo);
}
extern "C" int pp_split_planes_scaled_bf16x3(pp_ctx* ctx, size_t n, const float* src, void* hi, void* lo, const float* scale_dev) {
  return (ctx && ctx->planes_fmt == 1) ? pp_split_planes_scaled_bf16x3_fmt1(ctx, n, src, hi, lo, scale_dev) : pp_split_planes_scaled_bf16x3_fmt0(ctx, n, src, hi, lo, scale_dev);
}
extern "C" int pp_grad_scale_from_counts(pp_ctx* ctx, const int* counts_dev, int n_counts, float* scale2_dev) {
  return (ctx && ctx->planes_fmt == 1) ? pp_grad_scale_from_counts_fmt1(ctx, counts_dev, n_counts, scale2_dev) : pp_grad_scale_from_counts_fmt0(ctx, counts_dev, n_counts, scale2_dev);
}
extern "C" int pp_grad_scale_from_counts_adj(pp_ctx* ctx, const int* counts_dev, int n_counts, float* scale2_dev, int log2_adjust) {
  return (ctx && ctx->planes_fmt == 1) ? pp_grad_scale_from_counts_adj_fmt1(ctx, counts_dev, n_counts, scale2_dev, log2_adjust) : pp_grad_scale_from_counts_adj_fmt0(ctx, counts_dev, n_counts, scale2_dev, log2_adjust);
}
extern "C" int pp_conv_split_weights_bf16x3_batch(pp_ctx* ctx, int n_jobs, const pp_split_job* jobs_dev, int total_tiles) {
  return (ctx && ctx->planes_fmt == 1) ? pp_conv_split_weights_bf16x3_batch_fmt1(ctx, n_jobs, jobs_dev, total_tiles) : pp_conv_split_weights_bf16x3_batch_fmt0(ctx, n_jobs, jobs_dev, total_tiles);
}
extern "C" int pp_conv_split_weights_bf16x3(pp_ctx* ctx, const pp_conv_desc* d, const float* w, void* fwd_hi, void* fwd_lo, void* dgrad_hi, void* dgrad_lo) {
  return (ctx && ctx->planes_fmt == 1) ? pp_conv_split_weights_bf16x3_fmt1(ctx, d, w, fwd_hi, fwd_lo, dgrad_hi, dgrad_lo) : pp_conv_split_weights_bf16x3_fmt0(ctx, d, w, fwd_hi, fwd_lo, dgrad_hi, dgrad_lo);
}
extern "C" int pp_conv2d_nhwc_fwd_bf16x3(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const void* x_hi, const void* x_lo, const void* w_fwd_hi, const void* w_fwd_lo, const float* bias, const float* residual, int ld_res, int relu, float* y, void* y_hi, void* y_lo) {
  return (ctx && ctx->planes_fmt == 1) ? pp_conv2d_nhwc_fwd_bf16x3_fmt1(ctx, d, x, x_hi, x_lo, w_fwd_hi, w_fwd_lo, bias, residual, ld_res, relu, y, y_hi, y_lo) : pp_conv2d_nhwc_fwd_bf16x3_fmt0(ctx, d, x, x_hi, x_lo, w_fwd_hi, w_fwd_lo, bias, residual, ld_res, relu, y, y_hi, y_lo);
}
extern "C" int pp_row_block_dilate(pp_ctx* ctx, const pp_conv_desc* d, const unsigned char* in_flags, unsigned char* out_flags) {
  return (ctx && ctx->planes_fmt == 1) ? pp_row_block_dilate_fmt1(ctx, d, in_flags, out_flags) : pp_row_block_dilate_fmt0(ctx, d, in_flags, out_flags);
}
extern "C" int pp_stem7x7s2_fwd_bf16x3(pp_ctx* ctx, int n_img, int H, int W, int Hp, int Wp, const float* x4p, const void* w_hi, const void* w_lo, int cout, const float* bias, int relu, float* y, int ld_y) {
  return (ctx && ctx->planes_fmt == 1) ? pp_stem7x7s2_fwd_bf16x3_fmt1(ctx, n_img, H, W, Hp, Wp, x4p, w_hi, w_lo, cout, bias, relu, y, ld_y) : pp_stem7x7s2_fwd_bf16x3_fmt0(ctx, n_img, H, W, Hp, Wp, x4p, w_hi, w_lo, cout, bias, relu, y, ld_y);
}
extern "C" int pp_conv2d_nhwc_bwd_data_bf16x3(pp_ctx* ctx, const pp_conv_desc* d, const float* dy, const void* dy_hi, const void* dy_lo, const void* w_dgrad_hi, const void* w_dgrad_lo, const float* addend, int ld_add, const float* relu_src, int ld_rs, float* dx, void* dx_hi, void* dx_lo) {
  return (ctx && ctx->planes_fmt == 1) ? pp_conv2d_nhwc_bwd_data_bf16x3_fmt1(ctx, d, dy, dy_hi, dy_lo, w_dgrad_hi, w_dgrad_lo, addend, ld_add, relu_src, ld_rs, dx, dx_hi, dx_lo) : pp_conv2d_nhwc_bwd_data_bf16x3_fmt0(ctx, d, dy, dy_hi, dy_lo, w_dgrad_hi, w_dgrad_lo, addend, ld_add, relu_src, ld_rs, dx, dx_hi, dx_lo);
}
extern "C" int pp_conv2d_nhwc_bwd_weight_bf16x3(pp_ctx* ctx, const pp_conv_desc* d, const float* x, const float* dy, const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw, float* dbias) {
  return (ctx && ctx->planes_fmt == 1) ? pp_conv2d_nhwc_bwd_weight_bf16x3_fmt1(ctx, d, x, dy, x_hi, x_lo, dy_hi, dy_lo, dw, dbias) : pp_conv2d_nhwc_bwd_weight_bf16x3_fmt0(ctx, d, x, dy, x_hi, x_lo, dy_hi, dy_lo, dw, dbias);
}
