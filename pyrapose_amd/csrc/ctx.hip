// Context management for libpyrapose_hip.so.
#include <stdlib.h>

#include "pp_internal.h"

extern "C" const char* pp_version(void) { return "pyrapose_hip 0.6 (gfx950; convolutions on packed planes in two formats -- bf16 pairs / bf16x3 and P16 / f16c8 (f16 + block-scaled e5m2 MFMA) -- and exact f32 MFMA; sparse 3D-box backward, pose tail, device augmentation)"; }

extern "C" int pp_ctx_create(pp_ctx** out, int device, void* hip_stream) {
  if (!out) return PP_ERR_ARG;
  *out = nullptr;
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess) return (int)e;
  if (device < 0 || device >= n_dev) return PP_ERR_ARG;
  pp_ctx* c = (pp_ctx*)calloc(1, sizeof(pp_ctx));
  if (!c) return PP_ERR_ARG;
  c->device = device;
  c->stream = (hipStream_t)hip_stream;
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    free(c);
    return (int)e;
  }
  c->n_cu = prop.multiProcessorCount;
  snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
  c->err[0] = 0;
  *out = c;
  return PP_OK;
}

extern "C" void pp_ctx_destroy(pp_ctx* ctx) { free(ctx); }

extern "C" int pp_ctx_set_stream(pp_ctx* ctx, void* hip_stream) {
  PP_REQUIRE_CTX(ctx);
  ctx->stream = (hipStream_t)hip_stream;
  return PP_OK;
}

extern "C" int pp_ctx_set_workspace(pp_ctx* ctx, void* zeroed, size_t bytes) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, (zeroed == nullptr) == (bytes == 0) && pp_is_aligned16(zeroed), PP_ERR_ARG, "pp_ctx_set_workspace: bad buffer");
  ctx->ws = (float*)zeroed;
  ctx->ws_bytes = bytes;
  return PP_OK;
}

extern "C" int pp_ctx_set_split_capture(pp_ctx* ctx, void* hi, void* lo) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, (hi == nullptr) == (lo == nullptr) && pp_is_aligned16(hi) && pp_is_aligned16(lo), PP_ERR_ARG,
               "pp_ctx_set_split_capture: hi and lo go together, 16-byte aligned");
  ctx->cap_hi = hi;
  ctx->cap_lo = lo;
  return PP_OK;
}

extern "C" int pp_ctx_set_epilogue_planes(pp_ctx* ctx, const void* add_hi, const void* add_lo, const void* mask_hi) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, (add_hi == nullptr) == (add_lo == nullptr), PP_ERR_ARG, "pp_ctx_set_epilogue_planes: add_hi and add_lo go together");
  PP_CHECK_ARG(ctx, pp_is_packed(add_hi, add_lo) && pp_is_aligned16(mask_hi), PP_ERR_ALIGN,
               "pp_ctx_set_epilogue_planes: planes must be packed (lo = hi + 16 bytes) and 16-byte aligned");
  ctx->ep_add_hi = add_hi;
  ctx->ep_add_lo = add_lo;
  ctx->ep_mask_hi = mask_hi;
  return PP_OK;
}

extern "C" int pp_ctx_set_row_block_skip(pp_ctx* ctx, const unsigned char* flags, const int* list) {
  PP_REQUIRE_CTX(ctx);
  ctx->skip_flags = flags;
  ctx->skip_list = list;
  return PP_OK;
}

extern "C" int pp_ctx_set_row_block_lazy(pp_ctx* ctx, int lazy_out, int lazy_in) {
  PP_REQUIRE_CTX(ctx);
  ctx->lazy_out = lazy_out != 0;
  ctx->lazy_in = lazy_in != 0;
  return PP_OK;
}

extern "C" int pp_ctx_set_row_block_out(pp_ctx* ctx, const unsigned char* flags, int* list) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, (flags == nullptr) == (list == nullptr), PP_ERR_ARG, "pp_ctx_set_row_block_out: flags and list go together");
  ctx->out_flags = flags;
  ctx->out_list = list;
  return PP_OK;
}

extern "C" int pp_ctx_set_planes_format(pp_ctx* ctx, int fmt) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, fmt == 0 || fmt == 1, PP_ERR_ARG, "pp_ctx_set_planes_format: 0 (bf16 pairs) or 1 (P16)");
  ctx->planes_fmt = fmt;
  return PP_OK;
}

extern "C" int pp_ctx_set_grad_scale(pp_ctx* ctx, const float* scale2_dev) {
  PP_REQUIRE_CTX(ctx);
  ctx->grad_scale = scale2_dev;  // persistent (NULL: gradients are unscaled)
  return PP_OK;
}

extern "C" const char* pp_last_error_string(pp_ctx* ctx) { return ctx ? ctx->err : "no context"; }

extern "C" int pp_device_info(pp_ctx* ctx, int* n_cu, char* name, int name_len) {
  PP_REQUIRE_CTX(ctx);
  if (n_cu) *n_cu = ctx->n_cu;
  if (name && name_len > 0) {
    strncpy(name, ctx->name, (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  return PP_OK;
}
