// How close may two v_mfma_f32_32x32x16_bf16 on the SAME accumulator be issued?  (bf16x3 adds three products into one
// accumulator; the order of those MFMAs decides the dependency distance.)  Random operands, 20 ms launches, 3 workgroups/CU.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_dep.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define M(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)

template <int DIST>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ seed, float* out, int iters) {
  floatx16 A, B, C, D;
  for (int r = 0; r < 16; ++r) A[r] = B[r] = C[r] = D[r] = 0.f;
  const int lane = threadIdx.x & 63;
  uint4 q0 = seed[lane], q1 = seed[64 + lane], q2 = seed[128 + lane], q3 = seed[192 + lane];
  const bf16x8 a0 = *reinterpret_cast<bf16x8*>(&q0), a1 = *reinterpret_cast<bf16x8*>(&q1);
  const bf16x8 b0 = *reinterpret_cast<bf16x8*>(&q2), b1 = *reinterpret_cast<bf16x8*>(&q3);
  for (int it = 0; it < iters; ++it) {
    if (DIST == 4) { M(A, a0, b0); M(B, a0, b1); M(C, a1, b0); M(D, a1, b1); M(A, a1, b0); M(B, a1, b1); M(C, a0, b0); M(D, a0, b1); M(A, a0, b1); M(B, a0, b0); M(C, a1, b1); M(D, a1, b0); }
    if (DIST == 2) { M(A, a0, b0); M(B, a0, b1); M(A, a1, b0); M(B, a1, b1); M(A, a0, b1); M(B, a0, b0); M(C, a1, b0); M(D, a1, b1); M(C, a0, b0); M(D, a0, b1); M(C, a1, b1); M(D, a1, b0); }
    if (DIST == 1) { M(A, a0, b0); M(A, a1, b0); M(A, a0, b1); M(B, a0, b1); M(B, a1, b1); M(B, a0, b0); M(C, a1, b0); M(C, a0, b0); M(C, a1, b1); M(D, a1, b1); M(D, a0, b1); M(D, a1, b0); }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += A[r] + B[r] + C[r] + D[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int DIST>
static void run(int wg_per_cu, const uint4* seed) {
  float* out;
  const int nblk = 256 * wg_per_cu;
  hipMalloc(&out, (size_t)nblk * 256 * 4);
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  int iters = 2000;
  for (int pass = 0; pass < 2; ++pass) {
    hipEventRecord(s);
    hipLaunchKernelGGL(k<DIST>, dim3(nblk), dim3(256), 0, 0, seed, out, iters);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    const double tf = (double)nblk * 4 * iters * 12 * 32768.0 / ms / 1e9;
    if (pass == 0) iters = (int)(iters * 20.f / ms) + 1;
    else printf("{\"probe\": \"same-accumulator distance %d\", \"workgroups_per_cu\": %d, \"ms\": %.2f, \"tflops\": %.1f}\n", DIST, wg_per_cu, ms, tf);
  }
  hipFree(out);
}

int main() {
  uint4* seed;
  hipMalloc(&seed, 256 * 16);
  unsigned h[1024];
  srand(1);
  for (int i = 0; i < 1024; ++i) {
    unsigned lo = (rand() & 0x807f) | (0x3f00 + ((rand() & 1) << 7)), hi = (rand() & 0x807f) | (0x3f00 + ((rand() & 1) << 7));
    h[i] = lo | (hi << 16);
  }
  hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  for (int w = 1; w <= 3; ++w) { run<4>(w, seed); run<2>(w, seed); run<1>(w, seed); }
  return 0;
}
