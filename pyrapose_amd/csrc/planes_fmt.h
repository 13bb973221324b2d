// The two (hi, lo) plane formats of the conv3.hip kernel family and their matrix-core arithmetic, behind one interface.
// conv3.hip is compiled ONCE PER FORMAT (PP_FMT = 0 / 1, csrc/Makefile) into its own namespace; conv3_dispatch.hip picks the
// build a context asks for (pp_ctx_set_planes_format).  Same packed geometry in both: a tensor [rows][ld] (ld % 8 == 0) is
// rows * ld * 4 bytes in 32-byte groups of 8 channels, 16 bytes of hi then 16 bytes of lo.
//
// PP_FMT == 0  "bf16 pairs" / bf16x3: hi = bf16(x), lo = bf16(x - hi); value = hi + lo (2^-17);
//              x*w ~= x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on three v_mfma_f32_32x32x16_bf16 (~2^-17 per product, 4.5e-6 per launch
//              against float64).  6 MFMA issue units per 32-deep step of a 32x32 block.
// PP_FMT == 1  "P16" / f16c8: hi = IEEE half (|x| clamped to 28672), lo = two e5m2 bytes per element,
//              [e5m2(x) | e5m2((x - hi) * 2^12) << 8] for gathered operands and swapped for weights; value = hi + lo8 * 2^-12 (2^-15);
//              x*w ~= x_hi*w_hi (v_mfma_f32_32x32x16_f16) + (x_hi8*w_lo8 + x_lo8*w_hi8) * 2^-12: the two lo-plane fragments a lane
//              reads per 32-deep step ARE the 32-byte operands of ONE v_mfma_scale_f32_32x32x64_f8f6f4 (E8M0 scale 2^-12 on the
//              gathered side) -- 4 issue units per step, 2.1e-5 per launch against float64, 1.15-1.19x faster on the MFMA-bound
//              launches (profiles/r03_p16_*).  Gradients travel multiplied by a power of two (halves stop at 6e-8).
// The engine runs the HBM- / latency-bound backbone on format 0 (its ~50 layers are where rounding accumulates and where MFMA
// work is not what costs) and the MFMA-bound FPN + heads on format 1.
#pragma once
#include <hip/hip_runtime.h>

#ifndef PP_FMT
#error "planes_fmt.h: PP_FMT (0 = bf16 pairs, 1 = P16) must be defined"
#endif

typedef float pf_floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));  // (also the 16-byte fragment container of the transposing LDS reads)
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef short pf_shortx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_frag(const unsigned short* lds, int elem_off0, int elem_off1) {
  typedef __attribute__((address_space(3))) pf_shortx4 lds_s4;
  pf_shortx4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + elem_off0));
  pf_shortx4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + elem_off1));
  typedef short shortx8 __attribute__((ext_vector_type(8)));
  shortx8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return *reinterpret_cast<bf16x8*>(&v);
}

// LDS slot (in 16-byte units) of k-octet o of tile row `row`.  LAY 0: the rotated plane image the register-staged kernels write
// (one plane per pointer: slot o BM + ((row + 2 o) mod BM)).  LAY 1 (igemm4x, filled by LDS-DMA in full 128-byte lines): the
// gathered tile row-major, eight pieces per row [hi0 lo0 hi1 lo1 ...] with the piece index XOR-ed by (row / 2) mod 8 (hi and lo
// of an octet are slots s and s ^ 1 of ONE region); the weight tile row-major per plane, four pieces per row XOR-ed by
// (row / 4) mod 4.  Either way 16 consecutive rows of one octet fall into 16 different 16-byte bank groups.
template <int BM_, int LAY>
__device__ __forceinline__ int pf_slot_a(int row, int o) {
  if constexpr (LAY == 0) {
    return o * BM_ + ((row + 2 * o) & (BM_ - 1));
  } else {
    const int r = row & (BM_ - 1);
    return 8 * r + ((2 * o) ^ ((r >> 1) & 7));
  }
}
template <int BN_, int LAY>
__device__ __forceinline__ int pf_slot_b(int n, int o) {
  if constexpr (LAY == 0) {
    return o * BN_ + ((n + 2 * o) & (BN_ - 1));
  } else {
    const int r = n & (BN_ - 1);
    return 4 * r + (o ^ ((r >> 2) & 3));
  }
}

#if PP_FMT == 1
#include "p16.h"
// E8M0 scales of the scaled MFMA in ONE register: byte 0 (opsel 0) = 115 = 2^-12 for the gathered operand, byte 1 (opsel 1) = 127 = 1
#define P16_SCALES 0x7f73

// two consecutive elements -> their dword in each plane (gathered-operand byte order)
__device__ __forceinline__ void fmt_encode2(float v0, float v1, unsigned* hi2, unsigned* lo2) { p16_encode2<false>(v0, v1, hi2, lo2); }
// one weight -> its 16-bit unit in each plane (the weights' byte order)
__device__ __forceinline__ void fmt_encode_weight(float v, unsigned short* hi, unsigned short* lo) {
  unsigned h2, l2;
  p16_encode2<true>(v, 0.f, &h2, &l2);
  *hi = (unsigned short)h2;
  *lo = (unsigned short)l2;
}
__device__ __forceinline__ void fmt_value2(unsigned hi2, unsigned lo2, float* e0, float* e1) { p16_value2(hi2, lo2, e0, e1); }
// value > 0, from the hi unit alone
__device__ __forceinline__ bool fmt_pos(unsigned hi16) { return p16_pos(hi16); }
// any non-zero among the elements of these dwords (sign bits aside)
__device__ __forceinline__ bool fmt_any_nonzero(unsigned hi_or, unsigned lo_or) { return ((hi_or & 0x7fff7fffu) | (lo_or & 0x7f7f7f7fu)) != 0u; }

// One 32-deep k-step of a (32 TM) x (32 TN) block set from the rotated LDS images (row_a / row_b: this lane's first tile row of
// each operand; bit a of a_ok clear = row block a reads the all-zero slot `a_zero`): cross terms first (their fragments die
// with them), one row block at a time, then the two f16 half-steps.  HINT: 1 = igemm3x's scheduling hint, 2 = igemm3f's.
template <int TM, int TN, int BM, int BN, int HINT, int LAY = 0>
__device__ __forceinline__ void mma_step(pf_floatx16 (&acc)[TM][TN], const uint4* Ahi, const uint4* Alo, const uint4* Bhi, const uint4* Blo,
                                         int row_a, int row_b, int h, unsigned a_ok = ~0u, int a_zero = 0) {
  {
    intx8 bq[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int o = 2 * s + h;
        const uint4 t = Blo[pf_slot_b<BN, LAY>(row_b + b * 32, o)];
        bq[b][4 * s] = (int)t.x; bq[b][4 * s + 1] = (int)t.y; bq[b][4 * s + 2] = (int)t.z; bq[b][4 * s + 3] = (int)t.w;
      }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      intx8 aq;
      const bool ok = (a_ok >> a) & 1u;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int o = 2 * s + h;
        const uint4 t = Alo[(ok ? pf_slot_a<BM, LAY>(row_a + a * 32, o) : a_zero) ^ (LAY ? 1 : 0)];
        aq[4 * s] = (int)t.x; aq[4 * s + 1] = (int)t.y; aq[4 * s + 2] = (int)t.z; aq[4 * s + 3] = (int)t.w;
      }
#pragma unroll
      for (int b = 0; b < TN; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq, bq[b], acc[a][b], 1, 1, 0, P16_SCALES, 1, P16_SCALES);
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int o = 2 * s + h;
    halfx8 bh[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      uint4 t = Bhi[pf_slot_b<BN, LAY>(row_b + b * 32, o)];
      bh[b] = *reinterpret_cast<halfx8*>(&t);
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const bool ok = (a_ok >> a) & 1u;
      uint4 t = Ahi[ok ? pf_slot_a<BM, LAY>(row_a + a * 32, o) : a_zero];
      const halfx8 ah = *reinterpret_cast<halfx8*>(&t);
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[b], acc[a][b], 0, 0, 0);
    }
  }
  if (HINT == 1) __builtin_amdgcn_iglp_opt(1);
}

// One 32-pixel step of the weight gradient's (32 TM) x (32 TN) blocks from the pixel-major LDS images through the transposing
// reads: the lo fragments of the two 16-pixel halves are the e5m2 operands of ONE scaled MFMA (a (hi8, lo8) pair travels
// through ds_read_b64_tr_b16 as one 16-bit unit); both operands are gathered-type tensors, so the units of dy swap their bytes
// (8 v_perm per column block), then the two f16 half-steps.
template <int TM, int TN>
__device__ __forceinline__ void wgrad_step(pf_floatx16 (&acc)[TM][TN], const unsigned short* Xhi, const unsigned short* Xlo,
                                           const unsigned short* Ghi, const unsigned short* Glo, int PA, int PB, int xcol0, int gcol0, int hh,
                                           int gq) {
  {
    intx8 gq8[TN];
#pragma unroll
    for (int c = 0; c < TN; ++c)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int row0 = 16 * s + 8 * hh + gq, col = gcol0 + c * 32;
        const bf16x8 t = tr_frag(Glo, row0 * PB + col, (row0 + 4) * PB + col);
        const uint4 u = *reinterpret_cast<const uint4*>(&t);
        gq8[c][4 * s] = (int)__builtin_amdgcn_perm(u.x, u.x, 0x02030001u);
        gq8[c][4 * s + 1] = (int)__builtin_amdgcn_perm(u.y, u.y, 0x02030001u);
        gq8[c][4 * s + 2] = (int)__builtin_amdgcn_perm(u.z, u.z, 0x02030001u);
        gq8[c][4 * s + 3] = (int)__builtin_amdgcn_perm(u.w, u.w, 0x02030001u);
      }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      intx8 xq8;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int row0 = 16 * s + 8 * hh + gq, col = xcol0 + a * 32;
        const bf16x8 t = tr_frag(Xlo, row0 * PA + col, (row0 + 4) * PA + col);
        const uint4 u = *reinterpret_cast<const uint4*>(&t);
        xq8[4 * s] = (int)u.x; xq8[4 * s + 1] = (int)u.y; xq8[4 * s + 2] = (int)u.z; xq8[4 * s + 3] = (int)u.w;
      }
#pragma unroll
      for (int c = 0; c < TN; ++c)
        acc[a][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xq8, gq8[c], acc[a][c], 1, 1, 0, P16_SCALES, 1, P16_SCALES);
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int row0 = 16 * s + 8 * hh + gq;
    halfx8 gh[TN];
#pragma unroll
    for (int c = 0; c < TN; ++c) {
      const int col = gcol0 + c * 32;
      const bf16x8 t = tr_frag(Ghi, row0 * PB + col, (row0 + 4) * PB + col);
      gh[c] = *reinterpret_cast<const halfx8*>(&t);
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int col = xcol0 + a * 32;
      const bf16x8 t = tr_frag(Xhi, row0 * PA + col, (row0 + 4) * PA + col);
      const halfx8 xh = *reinterpret_cast<const halfx8*>(&t);
#pragma unroll
      for (int c = 0; c < TN; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, gh[c], acc[a][c], 0, 0, 0);
    }
    if (s == 0) __builtin_amdgcn_sched_barrier(0);
  }
}

#else  // ---------------------------------------------------------------- PP_FMT == 0: bf16 pairs, bf16x3

__device__ __forceinline__ void fmt_encode2(float v0, float v1, unsigned* hi2, unsigned* lo2) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 h, l;
  h[0] = (__bf16)v0;
  h[1] = (__bf16)v1;
  l[0] = (__bf16)(v0 - (float)h[0]);
  l[1] = (__bf16)(v1 - (float)h[1]);
  *hi2 = __builtin_bit_cast(unsigned, h);
  *lo2 = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void fmt_encode_weight(float v, unsigned short* hi, unsigned short* lo) {
  const __bf16 hh = (__bf16)v;
  const __bf16 ll = (__bf16)(v - (float)hh);
  *hi = __builtin_bit_cast(unsigned short, hh);
  *lo = __builtin_bit_cast(unsigned short, ll);
}
__device__ __forceinline__ void fmt_value2(unsigned hi2, unsigned lo2, float* e0, float* e1) {
  *e0 = __uint_as_float(hi2 << 16) + __uint_as_float(lo2 << 16);
  *e1 = __uint_as_float(hi2 & 0xffff0000u) + __uint_as_float(lo2 & 0xffff0000u);
}
__device__ __forceinline__ bool fmt_pos(unsigned hi16) { return (short)(unsigned short)hi16 > 0; }  // (bf16 keeps the f32 exponent range)
__device__ __forceinline__ bool fmt_any_nonzero(unsigned hi_or, unsigned lo_or) { return ((hi_or | lo_or) & 0x7fff7fffu) != 0u; }

template <int TM, int TN, int BM, int BN, int HINT, int LAY = 0>
__device__ __forceinline__ void mma_step(pf_floatx16 (&acc)[TM][TN], const uint4* Ahi, const uint4* Alo, const uint4* Bhi, const uint4* Blo,
                                         int row_a, int row_b, int h, unsigned a_ok = ~0u, int a_zero = 0) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int o = 2 * s + h;
    // (TM == 4: the gathered fragments are fetched two row blocks at a time, against the same weight fragments -- 16 fewer live
    // registers, which is what keeps the 256-row tile from spilling)
    constexpr int AG = TM > 2 ? 2 : TM;
    bf16x8 bh[TN], bl[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int slot = pf_slot_b<BN, LAY>(row_b + b * 32, o);
      uint4 t = Bhi[slot];
      bh[b] = *reinterpret_cast<bf16x8*>(&t);
      t = Blo[slot];
      bl[b] = *reinterpret_cast<bf16x8*>(&t);
    }
#pragma unroll
    for (int a0 = 0; a0 < TM; a0 += AG) {
      bf16x8 ah[AG], al[AG];
#pragma unroll
      for (int aa = 0; aa < AG; ++aa) {
        const int a = a0 + aa;
        const bool ok = (a_ok >> a) & 1u;
        // padded taps read the all-zero slot: one address select per fragment
        const int slot = ok ? pf_slot_a<BM, LAY>(row_a + a * 32, o) : a_zero;
        uint4 t = Ahi[slot];
        ah[aa] = *reinterpret_cast<bf16x8*>(&t);
        t = Alo[slot ^ (LAY ? 1 : 0)];
        al[aa] = *reinterpret_cast<bf16x8*>(&t);
      }
#pragma unroll
      for (int aa = 0; aa < AG; ++aa)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a0 + aa][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[aa], bh[b], acc[a0 + aa][b], 0, 0, 0);
          acc[a0 + aa][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[aa], bl[b], acc[a0 + aa][b], 0, 0, 0);
          acc[a0 + aa][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[aa], bh[b], acc[a0 + aa][b], 0, 0, 0);
        }
    }
    // scheduling hints (measured on the head shapes): igemm3x interleaves the fragment reads of a half-step with its MFMAs
    // (+3 % over a plain order); igemm3f pins "first half of the MFMAs | rest + conversion"
    if (HINT == 1) __builtin_amdgcn_iglp_opt(1);
    if (HINT == 2 && s == 0) __builtin_amdgcn_sched_barrier(0);
  }
}

template <int TM, int TN>
__device__ __forceinline__ void wgrad_step(pf_floatx16 (&acc)[TM][TN], const unsigned short* Xhi, const unsigned short* Xlo,
                                           const unsigned short* Ghi, const unsigned short* Glo, int PA, int PB, int xcol0, int gcol0, int hh,
                                           int gq) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int row0 = 16 * s + 8 * hh + gq;  // pixel row this lane addresses in the first transposed read
    bf16x8 xh[TM], xl[TM], gh[TN], gl[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int col = xcol0 + a * 32;
      xh[a] = tr_frag(Xhi, row0 * PA + col, (row0 + 4) * PA + col);
      xl[a] = tr_frag(Xlo, row0 * PA + col, (row0 + 4) * PA + col);
    }
#pragma unroll
    for (int c = 0; c < TN; ++c) {
      const int col = gcol0 + c * 32;
      gh[c] = tr_frag(Ghi, row0 * PB + col, (row0 + 4) * PB + col);
      gl[c] = tr_frag(Glo, row0 * PB + col, (row0 + 4) * PB + col);
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int c = 0; c < TN; ++c) {
        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[a], gh[c], acc[a][c], 0, 0, 0);
        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[a], gl[c], acc[a][c], 0, 0, 0);
        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[a], gh[c], acc[a][c], 0, 0, 0);
      }
    if (s == 0) __builtin_amdgcn_sched_barrier(0);
  }
}
#endif
