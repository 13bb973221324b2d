// HBM-bound pointwise / resampling kernels of the PyraPose graph (float32, NHWC, 16 B per lane).
#include "pp_internal.h"
#include "p16.h"

static inline unsigned grid_for(size_t n_items, int block, pp_ctx* ctx) {
  size_t blocks = (n_items + block - 1) / block;
  size_t cap = (size_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 8;  // grid-stride beyond 8 blocks/CU
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// ---- keras_resnet pool1: MaxPooling2D(3, strides=2, padding='same') --------------------------
__global__ void maxpool3x3s2_kernel(int n_img, int h, int w, int c4, const float4* __restrict__ x, int oh, int ow,
                                    int pad_t, int pad_l, float4* __restrict__ y) {
  const size_t total = (size_t)n_img * oh * ow * c4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % c4);
    size_t t = i / c4;
    int ox = (int)(t % ow);
    t /= ow;
    int oy = (int)(t % oh);
    int n = (int)(t / oh);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dy = 0; dy < 3; ++dy) {
      int iy = oy * 2 + dy - pad_t;
      if ((unsigned)iy >= (unsigned)h) continue;
      for (int dx = 0; dx < 3; ++dx) {
        int ix = ox * 2 + dx - pad_l;
        if ((unsigned)ix >= (unsigned)w) continue;
        float4 v = x[((size_t)(n * h + iy) * w + ix) * c4 + c];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    y[i] = m;
  }
}

extern "C" int pp_maxpool3x3s2_fwd(pp_ctx* ctx, int n_img, int h, int w, int c, const float* x, int oh, int ow, float* y) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, x && y && n_img > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, PP_ERR_SHAPE, "pp_maxpool3x3s2_fwd: bad shape");
  PP_CHECK_ARG(ctx, oh == (h + 1) / 2 && ow == (w + 1) / 2, PP_ERR_SHAPE, "pp_maxpool3x3s2_fwd: 'same' output must be ceil(in/2)");
  // TF 'same': pad_total = max((out-1)*2 + 3 - in, 0), pad_before = pad_total / 2
  int pt = ((oh - 1) * 2 + 3 - h); pt = pt > 0 ? pt / 2 : 0;
  int pl = ((ow - 1) * 2 + 3 - w); pl = pl > 0 ? pl / 2 : 0;
  size_t total = (size_t)n_img * oh * ow * (c / 4);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total, 256, ctx)), dim3(256), 0, ctx->stream, n_img, h, w, c / 4,
                     (const float4*)x, oh, ow, pt, pl, (float4*)y);
  PP_CHECK_LAUNCH(ctx, "pp_maxpool3x3s2_fwd");
  return PP_OK;
}

// ---- UpsampleLike (TF 2.1 tf.image.resize NEAREST, half-pixel centres) ------------------------
__device__ __forceinline__ int nn_src(int dst, float scale, int n_in) {
  int s = (int)floorf(((float)dst + 0.5f) * scale);
  return s < n_in - 1 ? s : n_in - 1;
}

__global__ void upsample_add_fwd_kernel(int n_img, int sh, int sw, int th, int tw, int c4, float scale_y, float scale_x,
                                        const float4* __restrict__ src, const float4* __restrict__ other,
                                        float4* __restrict__ out) {
  const size_t total = (size_t)n_img * th * tw * c4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % c4);
    size_t t = i / c4;
    int x = (int)(t % tw);
    t /= tw;
    int y = (int)(t % th);
    int n = (int)(t / th);
    int sy = nn_src(y, scale_y, sh), sx = nn_src(x, scale_x, sw);
    float4 v = src[((size_t)(n * sh + sy) * sw + sx) * c4 + c];
    if (other) {
      float4 o = other[i];
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    out[i] = v;
  }
}

__global__ void upsample_add_bwd_kernel(int n_img, int sh, int sw, int th, int tw, int c4, float scale_y, float scale_x,
                                        float inv_y, float inv_x, const float4* __restrict__ dtarget,
                                        const float4* __restrict__ base, float4* __restrict__ dsrc) {
  const size_t total = (size_t)n_img * sh * sw * c4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % c4);
    size_t t = i / c4;
    int sx = (int)(t % sw);
    t /= sw;
    int sy = (int)(t % sh);
    int n = (int)(t / sh);
    float4 acc = base ? base[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    int y_lo = (int)(sy * inv_y) - 2, y_hi = (int)((sy + 1) * inv_y) + 2;
    int x_lo = (int)(sx * inv_x) - 2, x_hi = (int)((sx + 1) * inv_x) + 2;
    y_lo = y_lo < 0 ? 0 : y_lo; x_lo = x_lo < 0 ? 0 : x_lo;
    y_hi = y_hi > th - 1 ? th - 1 : y_hi; x_hi = x_hi > tw - 1 ? tw - 1 : x_hi;
    for (int y = y_lo; y <= y_hi; ++y) {
      if (nn_src(y, scale_y, sh) != sy) continue;
      for (int x = x_lo; x <= x_hi; ++x) {
        if (nn_src(x, scale_x, sw) != sx) continue;
        float4 g = dtarget[((size_t)(n * th + y) * tw + x) * c4 + c];
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
      }
    }
    dsrc[i] = acc;
  }
}

extern "C" int pp_upsample_nearest_add_fwd(pp_ctx* ctx, int n_img, int sh, int sw, int th, int tw, int c, const float* src,
                                           const float* other, float* out) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, src && out && n_img > 0 && sh > 0 && sw > 0 && th > 0 && tw > 0 && c > 0 && c % 4 == 0, PP_ERR_SHAPE,
               "pp_upsample_nearest_add_fwd: bad shape");
  size_t total = (size_t)n_img * th * tw * (c / 4);
  hipLaunchKernelGGL(upsample_add_fwd_kernel, dim3(grid_for(total, 256, ctx)), dim3(256), 0, ctx->stream, n_img, sh, sw, th, tw,
                     c / 4, (float)sh / (float)th, (float)sw / (float)tw, (const float4*)src, (const float4*)other, (float4*)out);
  PP_CHECK_LAUNCH(ctx, "pp_upsample_nearest_add_fwd");
  return PP_OK;
}

extern "C" int pp_upsample_nearest_add_bwd(pp_ctx* ctx, int n_img, int sh, int sw, int th, int tw, int c, const float* dtarget,
                                           const float* base, float* dsrc) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, dtarget && dsrc && n_img > 0 && sh > 0 && sw > 0 && th > 0 && tw > 0 && c > 0 && c % 4 == 0, PP_ERR_SHAPE,
               "pp_upsample_nearest_add_bwd: bad shape");
  size_t total = (size_t)n_img * sh * sw * (c / 4);
  hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3(grid_for(total, 256, ctx)), dim3(256), 0, ctx->stream, n_img, sh, sw, th, tw,
                     c / 4, (float)sh / (float)th, (float)sw / (float)tw, (float)th / (float)sh, (float)tw / (float)sw,
                     (const float4*)dtarget, (const float4*)base, (float4*)dsrc);
  PP_CHECK_LAUNCH(ctx, "pp_upsample_nearest_add_bwd");
  return PP_OK;
}

// ---- keras.layers.Add ---------------------------------------------------------------------------
__global__ void add_n_kernel(size_t n4, const float4* __restrict__ a, const float4* __restrict__ b,
                             const float4* __restrict__ c, float4* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = a[i];
    if (b) { float4 w = b[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    if (c) { float4 w = c[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    out[i] = v;
  }
}

// keras.layers.Activation('relu') as its own op (models/retinanet.py:154: ReLU between P6 and the P7 conv); its gradient
// mask is applied by the consumer conv's bwd-data epilogue (relu_src), like every other ReLU of the graph
__global__ void relu_kernel(size_t n4, const float4* __restrict__ x, float4* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = x[i];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    y[i] = v;
  }
}

extern "C" int pp_relu_fwd(pp_ctx* ctx, size_t n, const float* x, float* y) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, x && y && n % 4 == 0, PP_ERR_SHAPE, "pp_relu_fwd: n must be a multiple of 4");
  if (n == 0) return PP_OK;
  hipLaunchKernelGGL(relu_kernel, dim3(grid_for(n / 4, 256, ctx)), dim3(256), 0, ctx->stream, n / 4, (const float4*)x, (float4*)y);
  PP_CHECK_LAUNCH(ctx, "pp_relu_fwd");
  return PP_OK;
}

extern "C" int pp_add_n(pp_ctx* ctx, size_t n, const float* a, const float* b, const float* c, float* out) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, a && out && n % 4 == 0, PP_ERR_SHAPE, "pp_add_n: n must be a multiple of 4");
  if (n == 0) return PP_OK;
  hipLaunchKernelGGL(add_n_kernel, dim3(grid_for(n / 4, 256, ctx)), dim3(256), 0, ctx->stream, n / 4, (const float4*)a,
                     (const float4*)b, (const float4*)c, (float4*)out);
  PP_CHECK_LAUNCH(ctx, "pp_add_n");
  return PP_OK;
}

// ---- the same pointwise ops on tensor views: float32 or bf16 (hi, lo) planes (see pp_tview) ----
// Every activation / gradient that a bf16x3 conv produces as planes only stays in that format through the FPN's adds and
// resampling: 4 bytes per element either way, and the convs downstream never convert in their loops.
struct TV {
  const float* f;
  const void* hi;
  const void* lo;
  int fmt;  // plane format of hi / lo: 0 = bf16 pairs, 1 = P16 (csrc/planes_fmt.h); taken from the context
};
static inline TV tv_of(const pp_tview* v, const pp_ctx* ctx = nullptr) {
  TV t = {nullptr, nullptr, nullptr, ctx ? ctx->planes_fmt : 0};
  if (v) { t.f = v->f32; t.hi = v->hi; t.lo = v->lo; }
  return t;
}
// two elements <-> their dword in each plane, in either format
__device__ __forceinline__ void tv_value2(int fmt, unsigned h2, unsigned l2, float* e0, float* e1) {
  if (fmt == 1) {
    p16_value2(h2, l2, e0, e1);
  } else {
    *e0 = __uint_as_float(h2 << 16) + __uint_as_float(l2 << 16);
    *e1 = __uint_as_float(h2 & 0xffff0000u) + __uint_as_float(l2 & 0xffff0000u);
  }
}
__device__ __forceinline__ void tv_encode2(int fmt, float v0, float v1, unsigned* h2, unsigned* l2) {
  if (fmt == 1) {
    p16_encode2<false>(v0, v1, h2, l2);
  } else {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 h, l;
    h[0] = (__bf16)v0;
    h[1] = (__bf16)v1;
    l[0] = (__bf16)(v0 - (float)h[0]);
    l[1] = (__bf16)(v1 - (float)h[1]);
    *h2 = __builtin_bit_cast(unsigned, h);
    *l2 = __builtin_bit_cast(unsigned, l);
  }
}
static inline bool tv_null(const TV& t) { return !t.f && !t.hi; }
static inline bool tv_ok_in(const TV& t) { return (t.f != nullptr) != (t.hi != nullptr) && pp_is_packed(t.hi, t.lo) && pp_is_aligned16(t.f); }
static inline bool tv_ok_out(const TV& t) { return (t.f || t.hi) && pp_is_packed(t.hi, t.lo) && pp_is_aligned16(t.f); }

// eight consecutive elements (one 32-byte group of a packed-plane tensor, two float4 of a float32 tensor)
struct F8 { float4 a, b; };
__device__ __forceinline__ F8 tv_ld8(const TV& t, size_t i8) {
  F8 v;
  if (t.hi) {
    const uint4 h = reinterpret_cast<const uint4*>(t.hi)[2 * i8], l = reinterpret_cast<const uint4*>(t.hi)[2 * i8 + 1];
    tv_value2(t.fmt, h.x, l.x, &v.a.x, &v.a.y);
    tv_value2(t.fmt, h.y, l.y, &v.a.z, &v.a.w);
    tv_value2(t.fmt, h.z, l.z, &v.b.x, &v.b.y);
    tv_value2(t.fmt, h.w, l.w, &v.b.z, &v.b.w);
    return v;
  }
  v.a = reinterpret_cast<const float4*>(t.f)[2 * i8];
  v.b = reinterpret_cast<const float4*>(t.f)[2 * i8 + 1];
  return v;
}
__device__ __forceinline__ void tv_st8(const TV& t, size_t i8, const F8& v) {
  if (t.f) {
    reinterpret_cast<float4*>(const_cast<float*>(t.f))[2 * i8] = v.a;
    reinterpret_cast<float4*>(const_cast<float*>(t.f))[2 * i8 + 1] = v.b;
  }
  if (t.hi) {
    uint4 h, l;
    tv_encode2(t.fmt, v.a.x, v.a.y, &h.x, &l.x);
    tv_encode2(t.fmt, v.a.z, v.a.w, &h.y, &l.y);
    tv_encode2(t.fmt, v.b.x, v.b.y, &h.z, &l.z);
    tv_encode2(t.fmt, v.b.z, v.b.w, &h.w, &l.w);
    uint4* dst = reinterpret_cast<uint4*>(const_cast<void*>(t.hi));
    dst[2 * i8] = h;
    dst[2 * i8 + 1] = l;
  }
}
__device__ __forceinline__ void f8_add(F8& v, const F8& w) {
  v.a.x += w.a.x; v.a.y += w.a.y; v.a.z += w.a.z; v.a.w += w.a.w;
  v.b.x += w.b.x; v.b.y += w.b.y; v.b.z += w.b.z; v.b.w += w.b.w;
}

__global__ void add_n_v_kernel(size_t n8, const TV a, const TV b, const TV c, const TV out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    F8 v = tv_ld8(a, i);
    if (b.f || b.hi) f8_add(v, tv_ld8(b, i));
    if (c.f || c.hi) f8_add(v, tv_ld8(c, i));
    tv_st8(out, i, v);
  }
}

__global__ void relu_v_kernel(size_t n8, const TV x, const TV y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    F8 v = tv_ld8(x, i);
    v.a.x = fmaxf(v.a.x, 0.f); v.a.y = fmaxf(v.a.y, 0.f); v.a.z = fmaxf(v.a.z, 0.f); v.a.w = fmaxf(v.a.w, 0.f);
    v.b.x = fmaxf(v.b.x, 0.f); v.b.y = fmaxf(v.b.y, 0.f); v.b.z = fmaxf(v.b.z, 0.f); v.b.w = fmaxf(v.b.w, 0.f);
    tv_st8(y, i, v);
  }
}

__global__ void upsample_add_fwd_v_kernel(int n_img, int sh, int sw, int th, int tw, int c8, float scale_y, float scale_x, const TV src,
                                          const TV other, const TV out) {
  const size_t total = (size_t)n_img * th * tw * c8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % c8);
    size_t t = i / c8;
    int x = (int)(t % tw);
    t /= tw;
    int y = (int)(t % th);
    int n = (int)(t / th);
    int sy = nn_src(y, scale_y, sh), sx = nn_src(x, scale_x, sw);
    F8 v = tv_ld8(src, ((size_t)(n * sh + sy) * sw + sx) * c8 + c);
    if (other.f || other.hi) f8_add(v, tv_ld8(other, i));
    tv_st8(out, i, v);
  }
}

__global__ void upsample_add_bwd_v_kernel(int n_img, int sh, int sw, int th, int tw, int c8, float scale_y, float scale_x, float inv_y,
                                          float inv_x, const TV dtarget, const TV base, const TV dsrc) {
  const size_t total = (size_t)n_img * sh * sw * c8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % c8);
    size_t t = i / c8;
    int sx = (int)(t % sw);
    t /= sw;
    int sy = (int)(t % sh);
    int n = (int)(t / sh);
    F8 acc;
    acc.a = acc.b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (base.f || base.hi) acc = tv_ld8(base, i);
    int y_lo = (int)(sy * inv_y) - 2, y_hi = (int)((sy + 1) * inv_y) + 2;
    int x_lo = (int)(sx * inv_x) - 2, x_hi = (int)((sx + 1) * inv_x) + 2;
    y_lo = y_lo < 0 ? 0 : y_lo; x_lo = x_lo < 0 ? 0 : x_lo;
    y_hi = y_hi > th - 1 ? th - 1 : y_hi; x_hi = x_hi > tw - 1 ? tw - 1 : x_hi;
    for (int y = y_lo; y <= y_hi; ++y) {
      if (nn_src(y, scale_y, sh) != sy) continue;
      for (int x = x_lo; x <= x_hi; ++x) {
        if (nn_src(x, scale_x, sw) != sx) continue;
        f8_add(acc, tv_ld8(dtarget, ((size_t)(n * th + y) * tw + x) * c8 + c));
      }
    }
    tv_st8(dsrc, i, acc);
  }
}

__global__ void merge_planes_kernel(size_t n8, const TV src, float4* __restrict__ dst) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const F8 v = tv_ld8(src, i);
    dst[2 * i] = v.a;
    dst[2 * i + 1] = v.b;
  }
}

extern "C" int pp_add_n_v(pp_ctx* ctx, size_t n, const pp_tview* a, const pp_tview* b, const pp_tview* c, const pp_tview* out) {
  PP_REQUIRE_CTX(ctx);
  const TV ta = tv_of(a, ctx), tb = tv_of(b, ctx), tc = tv_of(c, ctx), to = tv_of(out, ctx);
  PP_CHECK_ARG(ctx, n % 8 == 0 && tv_ok_in(ta) && (tv_null(tb) || tv_ok_in(tb)) && (tv_null(tc) || tv_ok_in(tc)) && tv_ok_out(to), PP_ERR_ARG,
               "pp_add_n_v: bad views (n %% 8 == 0; an input is f32 or packed planes, 16-byte aligned)");
  if (n == 0) return PP_OK;
  hipLaunchKernelGGL(add_n_v_kernel, dim3(grid_for(n / 8, 256, ctx)), dim3(256), 0, ctx->stream, n / 8, ta, tb, tc, to);
  PP_CHECK_LAUNCH(ctx, "pp_add_n_v");
  return PP_OK;
}

extern "C" int pp_relu_fwd_v(pp_ctx* ctx, size_t n, const pp_tview* x, const pp_tview* y) {
  PP_REQUIRE_CTX(ctx);
  const TV tx = tv_of(x, ctx), ty = tv_of(y, ctx);
  PP_CHECK_ARG(ctx, n % 8 == 0 && tv_ok_in(tx) && tv_ok_out(ty), PP_ERR_ARG, "pp_relu_fwd_v: bad views");
  if (n == 0) return PP_OK;
  hipLaunchKernelGGL(relu_v_kernel, dim3(grid_for(n / 8, 256, ctx)), dim3(256), 0, ctx->stream, n / 8, tx, ty);
  PP_CHECK_LAUNCH(ctx, "pp_relu_fwd_v");
  return PP_OK;
}

extern "C" int pp_upsample_nearest_add_fwd_v(pp_ctx* ctx, int n_img, int sh, int sw, int th, int tw, int c, const pp_tview* src,
                                             const pp_tview* other, const pp_tview* out) {
  PP_REQUIRE_CTX(ctx);
  const TV ts = tv_of(src, ctx), to = tv_of(other, ctx), tout = tv_of(out, ctx);
  PP_CHECK_ARG(ctx, n_img > 0 && sh > 0 && sw > 0 && th > 0 && tw > 0 && c > 0 && c % 8 == 0, PP_ERR_SHAPE, "pp_upsample_nearest_add_fwd_v: bad shape (c %% 8 == 0)");
  PP_CHECK_ARG(ctx, tv_ok_in(ts) && (tv_null(to) || tv_ok_in(to)) && tv_ok_out(tout), PP_ERR_ARG, "pp_upsample_nearest_add_fwd_v: bad views");
  size_t total = (size_t)n_img * th * tw * (c / 8);
  hipLaunchKernelGGL(upsample_add_fwd_v_kernel, dim3(grid_for(total, 256, ctx)), dim3(256), 0, ctx->stream, n_img, sh, sw, th, tw, c / 8,
                     (float)sh / (float)th, (float)sw / (float)tw, ts, to, tout);
  PP_CHECK_LAUNCH(ctx, "pp_upsample_nearest_add_fwd_v");
  return PP_OK;
}

extern "C" int pp_upsample_nearest_add_bwd_v(pp_ctx* ctx, int n_img, int sh, int sw, int th, int tw, int c, const pp_tview* dtarget,
                                             const pp_tview* base, const pp_tview* dsrc) {
  PP_REQUIRE_CTX(ctx);
  const TV td = tv_of(dtarget, ctx), tb = tv_of(base, ctx), ts = tv_of(dsrc, ctx);
  PP_CHECK_ARG(ctx, n_img > 0 && sh > 0 && sw > 0 && th > 0 && tw > 0 && c > 0 && c % 8 == 0, PP_ERR_SHAPE, "pp_upsample_nearest_add_bwd_v: bad shape (c %% 8 == 0)");
  PP_CHECK_ARG(ctx, tv_ok_in(td) && (tv_null(tb) || tv_ok_in(tb)) && tv_ok_out(ts), PP_ERR_ARG, "pp_upsample_nearest_add_bwd_v: bad views");
  size_t total = (size_t)n_img * sh * sw * (c / 8);
  hipLaunchKernelGGL(upsample_add_bwd_v_kernel, dim3(grid_for(total, 256, ctx)), dim3(256), 0, ctx->stream, n_img, sh, sw, th, tw, c / 8,
                     (float)sh / (float)th, (float)sw / (float)tw, (float)th / (float)sh, (float)tw / (float)sw, td, tb, ts);
  PP_CHECK_LAUNCH(ctx, "pp_upsample_nearest_add_bwd_v");
  return PP_OK;
}

// planes of one format -> planes of the other (the boundary between the backbone's bf16 pairs and the P16 tensors of FPN + heads):
// dst = encode_dst(decode_src(src) * scale[scale_index]) (scale NULL: 1; the gradient chain of the P16 side travels scaled);
// relu_hi: hi plane of the tensor whose ReLU this gradient passes on the way (zero where that tensor is not positive; the sign
// test of a hi half is the same in both formats)
__global__ void convert_planes_kernel(size_t n8, const TV src, const TV dst, const float* __restrict__ scale, int scale_index,
                                      const uint4* __restrict__ relu_hi) {
  const float sc = scale ? scale[scale_index] : 1.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    F8 v = tv_ld8(src, i);
    v.a.x *= sc; v.a.y *= sc; v.a.z *= sc; v.a.w *= sc;
    v.b.x *= sc; v.b.y *= sc; v.b.z *= sc; v.b.w *= sc;
    if (relu_hi) {
      const uint4 m = relu_hi[2 * i];
      if ((short)(m.x & 0xffffu) <= 0) v.a.x = 0.f;
      if ((short)(m.x >> 16) <= 0) v.a.y = 0.f;
      if ((short)(m.y & 0xffffu) <= 0) v.a.z = 0.f;
      if ((short)(m.y >> 16) <= 0) v.a.w = 0.f;
      if ((short)(m.z & 0xffffu) <= 0) v.b.x = 0.f;
      if ((short)(m.z >> 16) <= 0) v.b.y = 0.f;
      if ((short)(m.w & 0xffffu) <= 0) v.b.z = 0.f;
      if ((short)(m.w >> 16) <= 0) v.b.w = 0.f;
    }
    tv_st8(dst, i, v);
  }
}

extern "C" int pp_convert_planes(pp_ctx* ctx, size_t n, const void* src_hi, const void* src_lo, int src_fmt, void* dst_hi, void* dst_lo,
                                 int dst_fmt, const float* scale2_dev, int scale_index, const void* relu_src_hi) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, src_hi && dst_hi && n % 8 == 0 && pp_is_packed(src_hi, src_lo) && pp_is_packed(dst_hi, dst_lo) &&
                        pp_is_aligned16(relu_src_hi), PP_ERR_ARG,
               "pp_convert_planes: null tensor, planes not packed, or n %% 8 != 0");
  PP_CHECK_ARG(ctx, (src_fmt == 0 || src_fmt == 1) && (dst_fmt == 0 || dst_fmt == 1) && (scale_index == 0 || scale_index == 1), PP_ERR_ARG,
               "pp_convert_planes: formats are 0 (bf16 pairs) or 1 (P16), scale_index 0 or 1");
  if (n == 0) return PP_OK;
  TV s = {nullptr, src_hi, src_lo, src_fmt}, d = {nullptr, dst_hi, dst_lo, dst_fmt};
  hipLaunchKernelGGL(convert_planes_kernel, dim3(grid_for(n / 8, 256, ctx)), dim3(256), 0, ctx->stream, n / 8, s, d, scale2_dev, scale_index,
                     (const uint4*)relu_src_hi);
  PP_CHECK_LAUNCH(ctx, "pp_convert_planes");
  return PP_OK;
}

extern "C" int pp_merge_planes_bf16x3(pp_ctx* ctx, size_t n, const void* hi, const void* lo, float* dst) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, hi && lo && dst && n % 8 == 0 && pp_is_packed(hi, lo) && pp_is_aligned16(dst), PP_ERR_ARG,
               "pp_merge_planes_bf16x3: null / unaligned tensor, planes not packed, or n %% 8 != 0");
  if (n == 0) return PP_OK;
  TV t = {nullptr, hi, lo, ctx->planes_fmt};
  hipLaunchKernelGGL(merge_planes_kernel, dim3(grid_for(n / 8, 256, ctx)), dim3(256), 0, ctx->stream, n / 8, t, (float4*)dst);
  PP_CHECK_LAUNCH(ctx, "pp_merge_planes_bf16x3");
  return PP_OK;
}

// ---- audit of a P16 tensor: where do its halves sit in the format's range? ----
// P16 clamps |x| to 28 672 when it encodes and a half stops at 6e-8 (planes_fmt.h): neither leaves a trace in the tensor's consumers.
// This pass reads the hi plane of a tensor [rows][ld] (packed planes; columns < cols) and ADDS to stats[0..3]: elements looked at,
// (stats[4] = the largest |half| seen, as its 15 bits, by atomic max: how much of the range the tensor uses)
// non-zero halves, halves AT the clamp (|h| >= 28 672: a saturated encode), SUBNORMAL halves (0 < |h| < 2^-14: the element lost
// significand bits -- with the gradient scale 2^G of pp_grad_scale_from_counts that is where a too-small G shows).  within (may be
// NULL): uint8 flags of the 32-row blocks to look at (a lazily filled sparse gradient is undefined outside its flagged blocks).
__global__ void planes_stats_kernel(const uint4* __restrict__ hi, long long rows, int ld8, int cols8, const unsigned char* __restrict__ within,
                                    unsigned long long* __restrict__ stats) {
  unsigned long long n = 0, nz = 0, clamp = 0, sub = 0, amax = 0;
  const long long total = rows * cols8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols8;
    const int g = (int)(i - r * cols8);
    if (within && !within[r >> 5]) continue;
    const uint4 q = hi[(r * ld8 + g) * 2];  // 32-byte groups: 16 bytes of hi, 16 bytes of lo
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned a = (w[k] >> (16 * e)) & 0x7fffu;
        n += 1;
        nz += a != 0u;
        clamp += a >= 0x7700u;           // 28 672 = 0x7700 as a half (NaN / infinity patterns land here too)
        sub += a != 0u && a < 0x0400u;   // below the smallest normal half 2^-14
        amax = a > amax ? a : amax;
      }
  }
  // wave reduction, then one atomic per wave and counter
  for (int o = 32; o > 0; o >>= 1) {
    n += __shfl_down(n, o, 64); nz += __shfl_down(nz, o, 64); clamp += __shfl_down(clamp, o, 64); sub += __shfl_down(sub, o, 64);
    const unsigned long long om = __shfl_down(amax, o, 64);
    amax = om > amax ? om : amax;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(stats + 0, n); atomicAdd(stats + 1, nz); atomicAdd(stats + 2, clamp); atomicAdd(stats + 3, sub);
    atomicMax(stats + 4, amax);
  }
}

extern "C" int pp_planes_stats(pp_ctx* ctx, const void* hi, const void* lo, long long rows, int ld, int cols, const unsigned char* within,
                               unsigned long long* stats5_dev) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, hi && lo && stats5_dev && rows >= 0 && ld > 0 && ld % 8 == 0 && cols > 0 && cols % 8 == 0 && cols <= ld && pp_is_packed(hi, lo),
               PP_ERR_ARG, "pp_planes_stats: packed planes, ld and cols multiples of 8");
  PP_CHECK_ARG(ctx, ctx->planes_fmt == 1, PP_ERR_ARG, "pp_planes_stats: a P16 context (pp_ctx_set_planes_format(ctx, 1)): bf16 pairs keep the f32 range");
  if (rows == 0) return PP_OK;
  hipLaunchKernelGGL(planes_stats_kernel, dim3(grid_for((size_t)rows * (cols / 8), 256, ctx)), dim3(256), 0, ctx->stream, (const uint4*)hi, rows,
                     ld / 8, cols / 8, within, stats5_dev);
  PP_CHECK_LAUNCH(ctx, "pp_planes_stats");
  return PP_OK;
}

// ---- packed-RGB stem input ----------------------------------------------------------------------
__global__ void pack_rgb4_kernel(size_t n_pix, const float* __restrict__ x3, float4* __restrict__ x4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_pix; i += (size_t)gridDim.x * blockDim.x)
    x4[i] = make_float4(x3[3 * i], x3[3 * i + 1], x3[3 * i + 2], 0.f);
}

// padded frame for the bf16x3 stem: output [n_img][Hp][Wp][4], image at (pad, pad), zeros elsewhere and in channel 3
__global__ void pack_rgb4_padded_kernel(int n_img, int H, int W, int Hp, int Wp, int pad, const float* __restrict__ x3,
                                        float4* __restrict__ x4) {
  const size_t n_pix = (size_t)n_img * Hp * Wp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_pix; i += (size_t)gridDim.x * blockDim.x) {
    const int X = (int)(i % Wp), Y = (int)((i / Wp) % Hp), b = (int)(i / ((size_t)Wp * Hp));
    const int x = X - pad, y = Y - pad;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
      const float* s = x3 + (((size_t)b * H + y) * W + x) * 3;
      v = make_float4(s[0], s[1], s[2], 0.f);
    }
    x4[i] = v;
  }
}

extern "C" int pp_pack_rgb_to_4_padded(pp_ctx* ctx, int n_img, int H, int W, int Hp, int Wp, int pad, const float* x3, float* x4p) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, x3 && x4p && pp_is_aligned16(x4p), PP_ERR_ARG, "pp_pack_rgb_to_4_padded: null / unaligned tensor");
  PP_CHECK_ARG(ctx, n_img > 0 && H > 0 && W > 0 && pad >= 0 && Hp >= H + pad && Wp >= W + pad, PP_ERR_SHAPE, "pp_pack_rgb_to_4_padded: bad frame");
  hipLaunchKernelGGL(pack_rgb4_padded_kernel, dim3(grid_for((size_t)n_img * Hp * Wp, 256, ctx)), dim3(256), 0, ctx->stream, n_img, H, W, Hp, Wp,
                     pad, x3, (float4*)x4p);
  PP_CHECK_LAUNCH(ctx, "pp_pack_rgb_to_4_padded");
  return PP_OK;
}

extern "C" int pp_pack_rgb_to_4(pp_ctx* ctx, size_t n_pixels, const float* x3, float* x4) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, x3 && x4, PP_ERR_ARG, "pp_pack_rgb_to_4: null tensor");
  if (n_pixels == 0) return PP_OK;
  hipLaunchKernelGGL(pack_rgb4_kernel, dim3(grid_for(n_pixels, 256, ctx)), dim3(256), 0, ctx->stream, n_pixels, x3, (float4*)x4);
  PP_CHECK_LAUNCH(ctx, "pp_pack_rgb_to_4");
  return PP_OK;
}

// ---- uint8 BGR batch -> caffe mean-subtracted, zero-padded, packed stem input (first piece) ----------
// utils/image.py:35-62 preprocess_image(mode='caffe'): x.astype(float32); x[..., c] -= (103.939, 116.779, 123.68)[c]
// preprocessing/generator.py:319-336 compute_inputs: images copied into the upper-left corner of a ZERO batch, i.e.
// the padding stays 0.0 (it is not mean-subtracted).  sizes[b] = (h_b, w_b) of image b inside the [H, W] batch frame.
struct U8Sizes { int hw[2 * 64]; };
__global__ void preprocess_u8_kernel(int n_img, int H, int W, int Hp, int Wp, int pad, U8Sizes sz, const unsigned char* __restrict__ u8,
                                     float4* __restrict__ x4) {
  const size_t n_pix = (size_t)n_img * Hp * Wp;  // output frame [Hp][Wp], image frame [H][W] at (pad, pad)
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_pix; i += (size_t)gridDim.x * blockDim.x) {
    const int X = (int)(i % Wp), Y = (int)((i / Wp) % Hp), b = (int)(i / ((size_t)Wp * Hp));
    const int x = X - pad, y = Y - pad;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x >= 0 && y >= 0 && y < sz.hw[2 * b] && x < sz.hw[2 * b + 1]) {
      const unsigned char* s = u8 + (((size_t)b * H + y) * W + x) * 3;
      v.x = (float)s[0] - 103.939f;
      v.y = (float)s[1] - 116.779f;
      v.z = (float)s[2] - 123.68f;
    }
    x4[i] = v;
  }
}

static int preprocess_u8(pp_ctx* ctx, int n_img, int H, int W, int Hp, int Wp, int pad, const int* sizes_hw_host, const unsigned char* images_u8,
                         float* x4);

extern "C" int pp_preprocess_caffe_u8(pp_ctx* ctx, int n_img, int H, int W, const int* sizes_hw_host, const unsigned char* images_u8,
                                      float* x4) {
  return preprocess_u8(ctx, n_img, H, W, H, W, 0, sizes_hw_host, images_u8, x4);
}

extern "C" int pp_preprocess_caffe_u8_padded(pp_ctx* ctx, int n_img, int H, int W, int Hp, int Wp, int pad, const int* sizes_hw_host,
                                             const unsigned char* images_u8, float* x4p) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, pad >= 0 && Hp >= H + pad && Wp >= W + pad, PP_ERR_SHAPE, "pp_preprocess_caffe_u8_padded: bad frame");
  return preprocess_u8(ctx, n_img, H, W, Hp, Wp, pad, sizes_hw_host, images_u8, x4p);
}

static int preprocess_u8(pp_ctx* ctx, int n_img, int H, int W, int Hp, int Wp, int pad, const int* sizes_hw_host, const unsigned char* images_u8,
                         float* x4) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, n_img > 0 && n_img <= 64 && H > 0 && W > 0, PP_ERR_SHAPE, "pp_preprocess_caffe_u8: 1..64 images per call");
  PP_CHECK_ARG(ctx, images_u8 && x4 && sizes_hw_host && pp_is_aligned16(x4), PP_ERR_ARG, "pp_preprocess_caffe_u8: null / unaligned tensor");
  U8Sizes sz;
  for (int b = 0; b < n_img; ++b) {
    PP_CHECK_ARG(ctx, sizes_hw_host[2 * b] >= 0 && sizes_hw_host[2 * b] <= H && sizes_hw_host[2 * b + 1] >= 0 && sizes_hw_host[2 * b + 1] <= W,
                 PP_ERR_SHAPE, "pp_preprocess_caffe_u8: image %d (%d x %d) does not fit the %d x %d batch frame", b, sizes_hw_host[2 * b],
                 sizes_hw_host[2 * b + 1], H, W);
    sz.hw[2 * b] = sizes_hw_host[2 * b];
    sz.hw[2 * b + 1] = sizes_hw_host[2 * b + 1];
  }
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid_for((size_t)n_img * Hp * Wp, 256, ctx)), dim3(256), 0, ctx->stream, n_img, H, W, Hp, Wp,
                     pad, sz, images_u8, (float4*)x4);
  PP_CHECK_LAUNCH(ctx, "pp_preprocess_caffe_u8");
  return PP_OK;
}

// ---- head export: level-major [rows][ld] -> Keras (B, sum_l HW_l*A, V) ---------------------------
struct ExportGeo {
  int n_seg, n_img;
  int row_begin[PP_MAX_SEG + 1];
  int hw[PP_MAX_SEG];
  int cell_off[PP_MAX_SEG];
  int cells_total;
};

__global__ void export_head_kernel(ExportGeo g, int av, const float* __restrict__ src, int ld, int sig,
                                   float* __restrict__ out) {
  const size_t total = (size_t)g.row_begin[g.n_seg] * av;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int ch = (int)(i % av);
    int m = (int)(i / av);
    int s = 0;
    for (int k = 1; k < g.n_seg; ++k)
      if (m >= g.row_begin[k]) s = k;
    int local = m - g.row_begin[s];
    int b = local / g.hw[s];
    int p = local - b * g.hw[s];
    float v = src[(size_t)m * ld + ch];
    if (sig) v = 1.f / (1.f + expf(-v));
    out[((size_t)b * g.cells_total + g.cell_off[s] + p) * av + ch] = v;
  }
}

static void fill_export_geo(const pp_rowspace* rs, ExportGeo* g) {
  g->n_seg = rs->n_seg;
  g->n_img = rs->n_img;
  int rb = 0, co = 0;
  for (int s = 0; s < rs->n_seg; ++s) {
    g->row_begin[s] = rb;
    g->hw[s] = rs->h[s] * rs->w[s];
    g->cell_off[s] = co;
    rb += rs->n_img * g->hw[s];
    co += g->hw[s];
  }
  g->row_begin[rs->n_seg] = rb;
  g->cells_total = co;
}

extern "C" int pp_export_head(pp_ctx* ctx, const pp_rowspace* rs, int n_anchor, int n_val, const float* src, int ld,
                              int apply_sigmoid, float* out) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, rs && pp_rowspace_ok(rs) && src && out && n_anchor > 0 && n_val > 0 && ld >= n_anchor * n_val, PP_ERR_SHAPE,
               "pp_export_head: bad shape");
  ExportGeo g;
  fill_export_geo(rs, &g);
  size_t total = (size_t)g.row_begin[g.n_seg] * n_anchor * n_val;
  hipLaunchKernelGGL(export_head_kernel, dim3(grid_for(total, 256, ctx)), dim3(256), 0, ctx->stream, g, n_anchor * n_val, src, ld,
                     apply_sigmoid, out);
  PP_CHECK_LAUNCH(ctx, "pp_export_head");
  return PP_OK;
}
