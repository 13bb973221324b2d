// The "P16" plane format of the f16c8 arithmetic (csrc/conv3.hip): element = IEEE half (hi plane) + two e5m2 bytes (lo plane):
// [e5m2(hi) | e5m2((x - hi) * 2^12) << 8] for gathered operands, swapped for weights.  value = hi + lo8 * 2^-12.
#pragma once
#include <hip/hip_runtime.h>

// f32 -> half bits, e5m2(hi), e5m2(remainder * 2^12); e5m2 roundings to nearest even, clamped below the infinity encoding
__device__ __forceinline__ void p16_encode(float v, unsigned* hi, unsigned* hi8, unsigned* lo8) {
  v = fminf(fmaxf(v, -65504.f), 65504.f);
  const _Float16 h = (_Float16)v;
  const unsigned hb = __builtin_bit_cast(unsigned short, h);
  const float r = fminf(fmaxf((v - (float)h) * 4096.f, -57344.f), 57344.f);
  const unsigned lb = __builtin_bit_cast(unsigned short, (_Float16)r);
  unsigned l8 = (lb + 0x7fu + ((lb >> 8) & 1u)) >> 8;
  unsigned h8 = (hb + 0x7fu + ((hb >> 8) & 1u)) >> 8;
  h8 = ((h8 & 0x7fu) >= 0x7cu) ? ((h8 & 0x80u) | 0x7bu) : h8;
  l8 = ((l8 & 0x7fu) >= 0x7cu) ? ((l8 & 0x80u) | 0x7bu) : l8;
  *hi = hb;
  *hi8 = h8;
  *lo8 = l8;
}
// value of one element: its half and the HIGH byte of its lo unit (gathered-operand byte order)
__device__ __forceinline__ float p16_value(unsigned hi16, unsigned lo_unit) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)hi16) +
         (float)__builtin_bit_cast(_Float16, (unsigned short)(lo_unit & 0xff00u)) * (1.f / 4096.f);
}
// two elements packed in one dword of each plane
__device__ __forceinline__ void p16_value2(unsigned hi2, unsigned lo2, float* e0, float* e1) {
  *e0 = p16_value(hi2 & 0xffffu, lo2 & 0xffffu);
  *e1 = p16_value(hi2 >> 16, lo2 >> 16);
}
// hi > 0 <=> value > 0 (the remainder never changes the sign of a non-zero half; a value that rounds to a zero half is < 2^-25)
__device__ __forceinline__ bool p16_pos(unsigned hi16) { return (short)(unsigned short)hi16 > 0; }
