// Internal helpers shared by the HIP translation units of libpyrapose_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "pyrapose_hip.h"

struct pp_ctx {
  int device;
  hipStream_t stream;
  int n_cu;
  char name[128];
  char err[512];
  float* ws;        // caller-provided scratch for split-K partial sums (slices)
  size_t ws_bytes;
  void* cap_hi;     // one-shot: the next bf16x3 fwd / bwd-data launch also writes the split of its gathered operand here
  void* cap_lo;
  const void* ep_add_hi;  // one-shot: the next bf16x3 fwd / bwd-data launch reads its addend (residual) and / or its ReLU source
  const void* ep_add_lo;  // from bf16 (hi, lo) planes (pp_ctx_set_epilogue_planes)
  const void* ep_mask_hi;
  const int* skip_list;              // one-shot: row-block skip of the next bf16x3 bwd-weight (list) / bwd-data (flags) call
  const unsigned char* skip_flags;
  int lazy_out, lazy_in;             // one-shot (pp_ctx_set_row_block_lazy): the next sparse bwd-data leaves the rows of dx outside the
                                     // blocks it computes untouched / the next bwd-data or bwd-weight's dy holds anything outside its flagged blocks
  const unsigned char* out_flags;    // one-shot: the next bf16x3 forward call computes the flagged 32-row output blocks only
  int* out_list;                     //           (pp_ctx_set_row_block_out; list = its scratch)
  int planes_fmt;                    // format of every (hi, lo) plane pair this context sees: 0 = bf16 pairs (bf16x3), 1 = P16 (f16c8)
  const float* grad_scale;           // device {2^G, 2^-G}: the gradient planes this context's weight gradients read carry the factor 2^G
};

static inline int pp_fail(pp_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define PP_REQUIRE_CTX(ctx) \
  do {                      \
    if (!(ctx)) return PP_ERR_NOCTX; \
  } while (0)

#define PP_CHECK_ARG(ctx, cond, code, ...)            \
  do {                                                \
    if (!(cond)) return pp_fail(ctx, code, __VA_ARGS__); \
  } while (0)

// Launch-error check: hipGetLastError is not a synchronisation.
#define PP_CHECK_LAUNCH(ctx, what)                                                   \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess)                                                           \
      return pp_fail(ctx, (int)e__, "%s: launch failed: %s", what, hipGetErrorString(e__)); \
  } while (0)

#define PP_HIP(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e__ = (call);                                                                     \
    if (e__ != hipSuccess) return pp_fail(ctx, (int)e__, "%s: %s", #call, hipGetErrorString(e__)); \
  } while (0)

static inline long long pp_rowspace_rows(const pp_rowspace* rs) {
  long long r = 0;
  for (int s = 0; s < rs->n_seg; ++s) r += (long long)rs->n_img * rs->h[s] * rs->w[s];
  return r;
}

static inline int pp_rowspace_ok(const pp_rowspace* rs) {
  if (rs->n_img <= 0 || rs->n_seg <= 0 || rs->n_seg > PP_MAX_SEG) return 0;
  for (int s = 0; s < rs->n_seg; ++s)
    if (rs->h[s] <= 0 || rs->w[s] <= 0) return 0;
  return pp_rowspace_rows(rs) < (1ll << 31);
}

static inline int pp_is_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
// packed (hi, lo) planes: one buffer of 32-byte groups (8 channels: 16 bytes of hi, 16 bytes of lo); lo = hi + 16 bytes
static inline int pp_is_packed(const void* hi, const void* lo) {
  return (hi == nullptr && lo == nullptr) || (hi != nullptr && (const char*)lo == (const char*)hi + 16 && pp_is_aligned16(hi));
}
