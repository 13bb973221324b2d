// Multi-tensor optimizer kernels: keras 2.3.1 Adam with global-norm `clipnorm`
// (bin/train.py:101: Adam(lr=1e-5, clipnorm=0.001); keras/optimizers.py get_gradients + clip_norm
// -- third-party semantics).  One launch covers every tensor: a device
// block map assigns 1024-element chunks to workgroups; the norm is reduced in a fixed order
// (per-chunk partials in double, then one workgroup sums them) so the clip factor is reproducible.
//
// Frozen-BN folding: backbone convs compute with w_eff = w * scale[co]; their gradient w.r.t. the
// master weight is scale[co] * dL/dw_eff.  L2 kernel regularisation (models/retinanet.py:108) enters
// as 2*l2*w (and l2*w^2 in the reported loss).
#include <math.h>
#include <stdlib.h>

#include "pp_internal.h"

#define PP_OPT_CHUNK 1024

struct pp_optimizer {
  pp_param_desc* d_descs;
  int2* d_map;  // (desc index, chunk index)
  double* d_partial;
  double* d_partial_l2;
  int n_desc;
  int n_blocks;
  long long total;
};

extern "C" int pp_optimizer_create(pp_ctx* ctx, pp_optimizer** out, const pp_param_desc* descs, int n_desc, long long total) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, out && descs && n_desc > 0 && total > 0, PP_ERR_ARG, "pp_optimizer_create: bad argument");
  long long n_blocks = 0;
  for (int i = 0; i < n_desc; ++i) {
    PP_CHECK_ARG(ctx, descs[i].offset >= 0 && descs[i].count > 0 && descs[i].offset + descs[i].count <= total && descs[i].ld > 0 &&
                          descs[i].offset % 4 == 0,
                 PP_ERR_SHAPE, "pp_optimizer_create: tensor %d out of range", i);
    n_blocks += (descs[i].count + PP_OPT_CHUNK - 1) / PP_OPT_CHUNK;
  }
  PP_CHECK_ARG(ctx, n_blocks < (1ll << 30), PP_ERR_SHAPE, "pp_optimizer_create: too many chunks");
  pp_optimizer* o = (pp_optimizer*)calloc(1, sizeof(pp_optimizer));
  if (!o) return pp_fail(ctx, PP_ERR_ARG, "pp_optimizer_create: out of host memory");
  o->n_desc = n_desc;
  o->n_blocks = (int)n_blocks;
  o->total = total;
  int2* map = (int2*)malloc(sizeof(int2) * (size_t)n_blocks);
  long long b = 0;
  for (int i = 0; i < n_desc; ++i) {
    long long nc = (descs[i].count + PP_OPT_CHUNK - 1) / PP_OPT_CHUNK;
    for (long long c = 0; c < nc; ++c) map[b++] = make_int2(i, (int)c);
  }
  hipError_t e = hipMalloc((void**)&o->d_descs, sizeof(pp_param_desc) * n_desc);
  if (e == hipSuccess) e = hipMalloc((void**)&o->d_map, sizeof(int2) * (size_t)n_blocks);
  if (e == hipSuccess) e = hipMalloc((void**)&o->d_partial, sizeof(double) * (size_t)n_blocks);
  if (e == hipSuccess) e = hipMalloc((void**)&o->d_partial_l2, sizeof(double) * (size_t)n_blocks);
  if (e == hipSuccess) e = hipMemcpy(o->d_descs, descs, sizeof(pp_param_desc) * n_desc, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(o->d_map, map, sizeof(int2) * (size_t)n_blocks, hipMemcpyHostToDevice);
  free(map);
  if (e != hipSuccess) {
    pp_optimizer_destroy(o);
    return pp_fail(ctx, (int)e, "pp_optimizer_create: %s", hipGetErrorString(e));
  }
  *out = o;
  return PP_OK;
}

extern "C" void pp_optimizer_destroy(pp_optimizer* o) {
  if (!o) return;
  if (o->d_descs) (void)hipFree(o->d_descs);
  if (o->d_map) (void)hipFree(o->d_map);
  if (o->d_partial) (void)hipFree(o->d_partial);
  if (o->d_partial_l2) (void)hipFree(o->d_partial_l2);
  free(o);
}

__device__ __forceinline__ float eff_grad(const pp_param_desc& d, long long e, const float* g_eff, const float* w_master,
                                          const float* scales) {
  float g = g_eff[d.offset + e];
  if (d.scale_off >= 0) g *= scales[d.scale_off + (e % d.ld)];
  if (d.l2 != 0.f) g += 2.0f * d.l2 * w_master[d.offset + e];
  return g;
}

__global__ void grad_norm_partial_kernel(const pp_param_desc* __restrict__ descs, const int2* __restrict__ map,
                                         const float* __restrict__ w_master, const float* __restrict__ g_eff,
                                         const float* __restrict__ scales, double* __restrict__ partial,
                                         double* __restrict__ partial_l2) {
  const int2 mp = map[blockIdx.x];
  const pp_param_desc d = descs[mp.x];
  const long long base = (long long)mp.y * PP_OPT_CHUNK;
  double s = 0.0, s2 = 0.0;
  if (d.trainable) {
#pragma unroll
    for (int k = 0; k < PP_OPT_CHUNK / 256; ++k) {
      const long long e = base + threadIdx.x + 256 * k;
      if (e < d.count) {
        const float g = eff_grad(d, e, g_eff, w_master, scales);
        s += (double)g * (double)g;
        if (d.l2 != 0.f) {
          const float w = w_master[d.offset + e];
          s2 += (double)d.l2 * (double)w * (double)w;
        }
      }
    }
  }
  __shared__ double sh[256], sh2[256];
  sh[threadIdx.x] = s;
  sh2[threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[threadIdx.x] += sh[threadIdx.x + o];
      sh2[threadIdx.x] += sh2[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = sh[0];
    partial_l2[blockIdx.x] = sh2[0];
  }
}

__global__ void grad_norm_final_kernel(int n, const double* __restrict__ partial, const double* __restrict__ partial_l2,
                                       float* __restrict__ gnorm_sq, float* __restrict__ l2_loss) {
  __shared__ double sh[1024], sh2[1024];
  double s = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    s += partial[i];
    s2 += partial_l2[i];
  }
  sh[threadIdx.x] = s;
  sh2[threadIdx.x] = s2;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[threadIdx.x] += sh[threadIdx.x + o];
      sh2[threadIdx.x] += sh2[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    gnorm_sq[0] = (float)sh[0];
    if (l2_loss) l2_loss[0] += (float)sh2[0];
  }
}

extern "C" int pp_grad_global_norm(pp_ctx* ctx, pp_optimizer* opt, const float* w_master, const float* g_eff, const float* scales,
                                   float* gnorm_sq, float* l2_loss) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, opt && w_master && g_eff && gnorm_sq, PP_ERR_ARG, "pp_grad_global_norm: null argument");
  hipLaunchKernelGGL(grad_norm_partial_kernel, dim3((unsigned)opt->n_blocks), dim3(256), 0, ctx->stream, opt->d_descs, opt->d_map,
                     w_master, g_eff, scales, opt->d_partial, opt->d_partial_l2);
  hipLaunchKernelGGL(grad_norm_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, opt->n_blocks, opt->d_partial, opt->d_partial_l2,
                     gnorm_sq, l2_loss);
  PP_CHECK_LAUNCH(ctx, "pp_grad_global_norm");
  return PP_OK;
}

__global__ void adam_kernel(const pp_param_desc* __restrict__ descs, const int2* __restrict__ map, float* __restrict__ w_master,
                            float* __restrict__ w_eff, const float* __restrict__ g_eff, const float* __restrict__ scales,
                            float* __restrict__ mom, float* __restrict__ vel, const float* __restrict__ gnorm_sq, float lr_t,
                            float beta1, float beta2, float eps, float clipnorm) {
  const int2 mp = map[blockIdx.x];
  const pp_param_desc d = descs[mp.x];
  const long long base = (long long)mp.y * PP_OPT_CHUNK;
  float clip = 1.0f;
  if (clipnorm > 0.f) {
    const float norm = sqrtf(gnorm_sq[0]);
    if (norm >= clipnorm) clip = clipnorm / norm;  // keras clip_norm: switch(n >= c, g*c/n, g)
  }
#pragma unroll
  for (int k = 0; k < PP_OPT_CHUNK / 256; ++k) {
    const long long e = base + threadIdx.x + 256 * k;
    if (e >= d.count) continue;
    const long long i = d.offset + e;
    float w = w_master[i];
    if (d.trainable) {
      const float g = eff_grad(d, e, g_eff, w_master, scales) * clip;
      const float m = beta1 * mom[i] + (1.0f - beta1) * g;
      const float v = beta2 * vel[i] + (1.0f - beta2) * g * g;
      mom[i] = m;
      vel[i] = v;
      w = w - lr_t * m / (sqrtf(v) + eps);
      w_master[i] = w;
    }
    w_eff[i] = d.scale_off >= 0 ? w * scales[d.scale_off + (e % d.ld)] : w;
  }
}

extern "C" int pp_adam_step_clipnorm(pp_ctx* ctx, pp_optimizer* opt, float* w_master, float* w_eff, const float* g_eff,
                                     const float* scales, float* m, float* v, const float* gnorm_sq, float lr, float beta1,
                                     float beta2, float eps, float clipnorm, long long step) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, opt && w_master && w_eff && g_eff && m && v && gnorm_sq && step >= 1, PP_ERR_ARG,
               "pp_adam_step_clipnorm: bad argument");
  // keras Adam: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
  const double t = (double)step;
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)opt->n_blocks), dim3(256), 0, ctx->stream, opt->d_descs, opt->d_map, w_master,
                     w_eff, g_eff, scales, m, v, gnorm_sq, lr_t, beta1, beta2, eps, clipnorm);
  PP_CHECK_LAUNCH(ctx, "pp_adam_step_clipnorm");
  return PP_OK;
}
