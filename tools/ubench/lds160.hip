// can a workgroup own all 160 KB of a CU's LDS (static allocation)?  prints the launch status and a checksum
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 1) void k(float* out) {
  __shared__ __attribute__((aligned(16))) uint4 a[4096];   // 64 KB
  __shared__ __attribute__((aligned(16))) uint4 b[4096];   // 64 KB
  __shared__ __attribute__((aligned(16))) uint4 c[2048];   // 32 KB  -> 163840 bytes
  for (int i = threadIdx.x; i < 4096; i += 512) { a[i] = make_uint4(i, 0, 0, 0); b[i] = make_uint4(2 * i, 0, 0, 0); }
  for (int i = threadIdx.x; i < 2048; i += 512) c[i] = make_uint4(3 * i, 0, 0, 0);
  __syncthreads();
  out[threadIdx.x] = (float)(a[(threadIdx.x * 7) & 4095].x + b[(threadIdx.x * 5) & 4095].x + c[(threadIdx.x * 3) & 2047].x);
}
int main() {
  float* d; hipMalloc(&d, 512 * 4);
  hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, d);
  hipError_t e = hipDeviceSynchronize();
  float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 512; ++i) s += h[i];
  printf("launch: %s  last error: %s  checksum %.0f\n", hipGetErrorString(e), hipGetErrorString(hipGetLastError()), s);
  return 0;
}
