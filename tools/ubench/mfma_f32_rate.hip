// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 on gfx950, alone and beside LDS reads.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_rate.hip -o tools/ubench/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 15) * 0.001f;
  __syncthreads();
  floatx16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float a0 = lane * 0.01f, a1 = a0 + 1.f, b0 = 0.5f, b1 = 0.25f;
  const float2* L = reinterpret_cast<const float2*>(lds) + lane;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {  // one ds_read_b64 pair per 4 MFMAs, consumed one step later
      float2 na = L[(it & 31) * 64], nb = L[2048 + (it & 31) * 64];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
      a0 = na.x; a1 = na.y; b0 = nb.x; b1 = nb.y;
    } else {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
      if (MODE == 2) { a0 += 1e-9f; b0 += 1e-9f; }
    }
  }
  float s = 0;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu) {
  int iters = 20000;
  float* out;
  int nblk = 256 * blocks_per_cu;
  hipMalloc(&out, nblk * 256 * 4);
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, out, iters);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  double flops = (double)nblk * 4 /*waves*/ * iters * 4 /*mfma*/ * 4096.0;
  printf("%-28s blocks/CU=%d  %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", name, blocks_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
  hipFree(out);
}

int main() {
  for (int b = 1; b <= 4; b *= 2) {
    run<0>("mfma only", b);
    run<2>("mfma + 2 valu", b);
    run<1>("mfma + ds_read_b64 x2", b);
  }
  return 0;
}
