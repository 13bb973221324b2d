// Measured denominators (SURVEY.md 8d "Peaks to divide by: measure on the box"): sustained issue rate of
// v_mfma_f32_32x32x16_bf16 on RANDOM operands (the chip lowers its clock under MFMA load, MI355X_MICROARCH.md
// 'DVFS give-back'), with and without the LDS fragment reads of a real tile loop, and an HBM read+write triad.
// Build on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/peaks.hip -o /tmp/peaks && /tmp/peaks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: operands stay in registers.  MODE 1: operands re-read from LDS by ds_read_b128 every step (2 reads per 3 MFMAs,
// the ratio of the 128x128 bf16x3 conv loop).
template <int MODE>
__global__ __launch_bounds__(256) void mfma_kernel(const uint4* __restrict__ seed, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = seed[(blockIdx.x * 2048 + i) & 0xffff];
  __syncthreads();
  floatx16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4 ra[2] = {lds[lane], lds[64 + lane]}, rb[2] = {lds[128 + lane], lds[192 + lane]};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {
      const int o = ((it & 3) * 512 + wave * 64 + lane) & 2047;
      ra[0] = lds[o]; ra[1] = lds[(o + 256) & 2047]; rb[0] = lds[(o + 1024) & 2047]; rb[1] = lds[(o + 1280) & 2047];
    }
    const bf16x8 a0 = *reinterpret_cast<bf16x8*>(&ra[0]), a1 = *reinterpret_cast<bf16x8*>(&ra[1]);
    const bf16x8 b0 = *reinterpret_cast<bf16x8*>(&rb[0]), b1 = *reinterpret_cast<bf16x8*>(&rb[1]);
#pragma unroll
    for (int rep = 0; rep < 3; ++rep) {  // 12 MFMAs per step (as one 16-deep sub-step of the 128x128 tile)
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
    }
  }
  float s = 0;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void triad_kernel(size_t n4, const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x = a[i], y = b[i];
    c[i] = make_float4(x.x + 2.f * y.x, x.y + 2.f * y.y, x.z + 2.f * y.z, x.w + 2.f * y.w);
  }
}

template <int MODE>
static double run_mfma(const char* name, int blocks_per_cu, const uint4* seed, float ms_target) {
  float* out;
  const int nblk = 256 * blocks_per_cu;
  hipMalloc(&out, (size_t)nblk * 256 * 4);
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  int iters = 2000;
  double tf = 0;
  for (int pass = 0; pass < 2; ++pass) {  // pass 0 calibrates the length, pass 1 is sustained for >= ms_target
    hipEventRecord(s);
    hipLaunchKernelGGL(mfma_kernel<MODE>, dim3(nblk), dim3(256), 0, 0, seed, out, iters);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    tf = (double)nblk * 4 * iters * 12 * 32768.0 / ms / 1e9;
    if (pass == 0) iters = (int)(iters * ms_target / ms) + 1;
    else printf("{\"probe\": \"%s\", \"workgroups_per_cu\": %d, \"ms\": %.2f, \"tflops\": %.1f, \"frac_of_2500\": %.3f}\n", name, blocks_per_cu, ms, tf, tf / 2500.0);
  }
  hipFree(out);
  return tf;
}

int main() {
  uint4* seed;
  hipMalloc(&seed, 65536 * 16);
  {  // random bf16 pairs in (-2, 2): sign/exponent/mantissa all vary
    unsigned* h = (unsigned*)malloc(65536 * 16);
    srand(1);
    for (int i = 0; i < 65536 * 4; ++i) {
      unsigned lo = (rand() & 0x807f) | (0x3f00 + ((rand() & 1) << 7)), hi = (rand() & 0x807f) | (0x3f00 + ((rand() & 1) << 7));
      h[i] = lo | (hi << 16);
    }
    hipMemcpy(seed, h, 65536 * 16, hipMemcpyHostToDevice);
    free(h);
  }
  for (int b = 1; b <= 3; ++b) run_mfma<0>("mfma_bf16_32x32x16 registers, random data", b, seed, 20.f);
  for (int b = 1; b <= 3; ++b) run_mfma<1>("mfma_bf16_32x32x16 + ds_read_b128 (2 per 3 MFMA), random data", b, seed, 20.f);
  // HBM triad: 3 x 2 GiB streams (beyond the 256 MiB Infinity Cache)
  const size_t n4 = (size_t)1 << 27;  // 2 GiB per array
  float4 *a, *b, *c;
  hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16); hipMalloc(&c, n4 * 16);
  hipMemset(a, 0, n4 * 16); hipMemset(b, 0, n4 * 16);
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(s);
    hipLaunchKernelGGL(triad_kernel, dim3(256 * 16), dim3(256), 0, 0, n4, a, b, c);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    if (rep == 2) printf("{\"probe\": \"hbm triad c = a + 2b, 3 x 2 GiB\", \"ms\": %.2f, \"GBps\": %.0f, \"frac_of_8000\": %.3f}\n", ms, 3.0 * n4 * 16 / ms / 1e6, 3.0 * n4 * 16 / ms / 1e6 / 8000.0);
  }
  return 0;
}
