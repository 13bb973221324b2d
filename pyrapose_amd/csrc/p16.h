// The "P16" plane format of the f16c8 arithmetic (csrc/conv3.hip): element = IEEE half (hi plane) + two e5m2 bytes (lo plane):
// [e5m2(x) | e5m2((x - hi) * 2^12) << 8] for gathered operands, swapped for weights.  value = hi + lo8 * 2^-12 (within 2^-15).
// Conversions on the hardware's own instructions: v_cvt_pk_f16_f32 (round to nearest even), v_cvt_pk_bf8_f32 (two f32 -> two
// e5m2 bytes into one word of the destination: exactly one lo unit), v_cvt_f32_bf8 with a byte select -- 5 VALU per encoded
// element, 3 per decoded one.  |x| is clamped to 28672 = 2^14 * 1.75: up to there the scaled remainder stays below the largest
// finite e5m2 (57344), so no conversion can produce an infinity.
#pragma once
#include <hip/hip_runtime.h>

typedef _Float16 p16_half2 __attribute__((ext_vector_type(2)));

// two consecutive elements -> their dword of the hi plane and their dword of the lo plane (WGT: the weights' byte order)
template <bool WGT>
__device__ __forceinline__ void p16_encode2(float v0, float v1, unsigned* hi2, unsigned* lo2) {
  v0 = fminf(fmaxf(v0, -28672.f), 28672.f);
  v1 = fminf(fmaxf(v1, -28672.f), 28672.f);
  p16_half2 h;
  h[0] = (_Float16)v0;
  h[1] = (_Float16)v1;
  // a value whose half is zero (|x| < 2^-25) is stored as exactly zero: "hi > 0 <=> value > 0" then holds without exception, which
  // is what the ReLU masks (read from the hi plane alone) and the forward's own sign test must agree on
  const float f0 = (float)h[0], f1 = (float)h[1];
  const float r0 = f0 != 0.f ? (v0 - f0) * 4096.f : 0.f, r1 = f1 != 0.f ? (v1 - f1) * 4096.f : 0.f;
  int w = 0;
  if (WGT) {
    w = __builtin_amdgcn_cvt_pk_bf8_f32(r0, v0, w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(r1, v1, w, true);
  } else {
    w = __builtin_amdgcn_cvt_pk_bf8_f32(v0, r0, w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(v1, r1, w, true);
  }
  *hi2 = __builtin_bit_cast(unsigned, h);
  *lo2 = (unsigned)w;
}
// two elements packed in one dword of each plane (gathered-operand byte order: the remainders are bytes 1 and 3 of the lo dword)
__device__ __forceinline__ void p16_value2(unsigned hi2, unsigned lo2, float* e0, float* e1) {
  const p16_half2 h = __builtin_bit_cast(p16_half2, hi2);
  *e0 = fmaf(__builtin_amdgcn_cvt_f32_bf8((int)lo2, 1), 1.f / 4096.f, (float)h[0]);
  *e1 = fmaf(__builtin_amdgcn_cvt_f32_bf8((int)lo2, 3), 1.f / 4096.f, (float)h[1]);
}
// hi > 0 <=> value > 0 (the remainder never changes the sign of a non-zero half; a value that rounds to a zero half is stored as 0)
__device__ __forceinline__ bool p16_pos(unsigned hi16) { return (short)(unsigned short)hi16 > 0; }
