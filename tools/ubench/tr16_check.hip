// Checks the lane mapping of ds_read_b64_tr_b16 (gfx950) that conv3.hip's weight-gradient kernel relies on:
// per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a 4x16 block of 16-bit
// elements; lane i receives column i of the 4 rows (row q in element q).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short shortx4 __attribute__((ext_vector_type(4)));
#define PITCH 160  // elements per LDS row
__global__ void k(unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[32 * PITCH];
  for (int i = threadIdx.x; i < 32 * PITCH; i += 64) lds[i] = (unsigned short)((i / PITCH) * 256 + (i % PITCH));  // row*256+col
  __syncthreads();
  const int lane = threadIdx.x;
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int cbase = 16 * (g & 1), h = g >> 1;
  __attribute__((address_space(3))) shortx4* ptr =
      (__attribute__((address_space(3))) shortx4*)(lds + (8 * h + q) * PITCH + cbase + 4 * p);
  shortx4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
  out[lane] = *reinterpret_cast<unsigned long long*>(&v);
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 64 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned long long h[64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int g = lane >> 4, i = lane & 15, cbase = 16 * (g & 1), hh = g >> 1;
    for (int q = 0; q < 4; ++q) {
      unsigned short got = (unsigned short)(h[lane] >> (16 * q));
      unsigned short want = (unsigned short)((8 * hh + q) * 256 + cbase + i);
      if (got != want) {
        if (bad < 8) printf("lane %d elem %d: got row %d col %d, want row %d col %d\n", lane, q, got >> 8, got & 255, want >> 8, want & 255);
        ++bad;
      }
    }
  }
  printf(bad ? "tr16 mapping MISMATCH (%d)\n" : "tr16 mapping OK\n", bad);
  return bad != 0;
}
