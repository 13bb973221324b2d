// Building blocks of the "f16c8" convolution arithmetic (csrc/planes_fmt.h, PP_FMT == 1), checked on the hardware before the kernel relies on them:
//   x = x_hi + x_lo, x_hi = f16(x), x_lo8 = e5m2(x_lo * 2^12); the same for w
//   x*w ~= x_hi*w_hi (v_mfma_f32_32x32x16_f16) + [x_hi8*w_lo8 + x_lo8*w_hi8] * 2^-12 (v_mfma_scale_f32_32x32x64_f8f6f4, e5m2
//   operands, E8M0 scale 2^-12 on the "lo" side), x_hi8 = the top byte of x_hi (e5m2 IS the top byte of an IEEE half).
// 1. lane map of the scaled MFMA with e5m2 operands: lane (r = l & 31, h = l >> 5) holds A[r][32h + j], B[32h + j][r] in byte j
// 2. the E8M0 scale operand (byte selected by opsel; 127 = 1.0, 115 = 2^-12)
// 3. the error of the whole scheme against float64 on random data (beside bf16x3 and plain f16 / bf16)
// 4. issue rate of the 16 + 8 MFMA mix of one K = 64 step
// 5. the lane map of ds_read_b64_tr_b8
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef int intx2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e = (x);                                                               \
    if (e != hipSuccess) {                                                            \
      printf("%s: %s\n", #x, hipGetErrorString(e));                                   \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)

// ---- host helpers: e5m2 / f16 conversions ----
static unsigned short f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;
  unsigned short u;
  memcpy(&u, &h, 2);
  return u;
}
static float f16_bits_to_f32(unsigned short u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}
static unsigned char e5m2_rne(float f) {  // round to nearest even through the half encoding
  unsigned short u = f32_to_f16_bits(f);
  unsigned r = u + 0x7f + ((u >> 8) & 1);
  return (unsigned char)(r >> 8);
}
static float e5m2_to_f32(unsigned char b) { return f16_bits_to_f32((unsigned short)(b << 8)); }

// ---- 1 / 2: one scaled MFMA, operands given per lane as 32 bytes ----
__global__ void scaled_once(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, float* __restrict__ C, int scale_a,
                            int scale_b) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  intx8 a, b;
  for (int v = 0; v < 8; ++v) {
    unsigned ua = 0, ub = 0;
    for (int j = 0; j < 4; ++j) {
      const int k = 32 * h + 4 * v + j;
      ua |= (unsigned)A[r * 64 + k] << (8 * j);
      ub |= (unsigned)B[k * 32 + r] << (8 * j);
    }
    a[v] = (int)ua;
    b[v] = (int)ub;
  }
  floatx16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  // scale operands: byte 0 of the VGPR (opsel 0)
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, scale_a, 0, scale_b);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[row * 32 + r] = c[i];
  }
}

// ---- 3: a 32x32 output tile over K (multiple of 64); A [32][K] f32, B [K][32] f32 ----
__device__ __forceinline__ unsigned perm_hi8(unsigned lo_pair, unsigned hi_pair) {
  // bytes 1, 3 of lo_pair and bytes 1, 3 of hi_pair (the top bytes of four halves) -> one dword
  return __builtin_amdgcn_perm(hi_pair, lo_pair, 0x07050301u);
}
// the comparison arithmetics, one accumulator each: bf16x3 (conv3.hip) and plain f16
__global__ void other_tiles(const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ C_bf16x3, float* __restrict__ C_f16) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  floatx16 c1, c2;
  for (int i = 0; i < 16; ++i) c1[i] = c2[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    halfx8 ah, bh;
    bf16x8 ahb, bhb, alb, blb;
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + 8 * h + j;
      const float xa = A[r * K + k], xb = B[k * 32 + r];
      ah[j] = (_Float16)xa;
      bh[j] = (_Float16)xb;
      const __bf16 hab = (__bf16)xa, hbb = (__bf16)xb;
      ahb[j] = hab;
      bhb[j] = hbb;
      alb[j] = (__bf16)(xa - (float)hab);
      blb[j] = (__bf16)(xb - (float)hbb);
    }
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alb, bhb, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahb, blb, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahb, bhb, c1, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C_bf16x3[row * 32 + r] = c1[i];
    C_f16[row * 32 + r] = c2[i];
  }
}

// the f16c8 tile (hi8 = truncated top bytes, through v_perm) and its two cross terms alone, through the scaled MFMA (K = 64) and through
// the unscaled bf8 MFMA (K = 16), beside the host's evaluation of the same 8-bit operands
__global__ void cross_debug(const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ X1s, float* __restrict__ X2s,
                            float* __restrict__ X1u, float* __restrict__ X2u, float* __restrict__ Csimple) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  floatx16 c0, c1, c2, c3, c4;
  for (int i = 0; i < 16; ++i) c0[i] = c1[i] = c2[i] = c3[i] = c4[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 64) {
    intx8 a_hi8, b_hi8, a_lo8, b_lo8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      unsigned char ha8[8], hb8[8], la[8], lb[8];
      halfx8 ahv, bhv;
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + 16 * s + 8 * h + j;
        const float xa = A[r * K + k], xb = B[k * 32 + r];
        const _Float16 ha = (_Float16)xa, hb = (_Float16)xb;
        ahv[j] = ha;
        bhv[j] = hb;
        const _Float16 al = (_Float16)((xa - (float)ha) * 4096.f), bl = (_Float16)((xb - (float)hb) * 4096.f);
        unsigned short ua = __builtin_bit_cast(unsigned short, al), ub = __builtin_bit_cast(unsigned short, bl);
        la[j] = (unsigned char)((ua + 0x7f + ((ua >> 8) & 1)) >> 8);
        lb[j] = (unsigned char)((ub + 0x7f + ((ub >> 8) & 1)) >> 8);
        ha8[j] = (unsigned char)(__builtin_bit_cast(unsigned short, ha) >> 8);
        hb8[j] = (unsigned char)(__builtin_bit_cast(unsigned short, hb) >> 8);
      }
      unsigned w[8];
      memcpy(&w[0], ha8, 8);
      memcpy(&w[2], hb8, 8);
      memcpy(&w[4], la, 8);
      memcpy(&w[6], lb, 8);
      a_hi8[2 * s] = (int)w[0]; a_hi8[2 * s + 1] = (int)w[1];
      b_hi8[2 * s] = (int)w[2]; b_hi8[2 * s + 1] = (int)w[3];
      a_lo8[2 * s] = (int)w[4]; a_lo8[2 * s + 1] = (int)w[5];
      b_lo8[2 * s] = (int)w[6]; b_lo8[2 * s + 1] = (int)w[7];
      long ah8, bh8, al8, bl8;
      memcpy(&ah8, ha8, 8); memcpy(&bh8, hb8, 8); memcpy(&al8, la, 8); memcpy(&bl8, lb, 8);
      c4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahv, bhv, c4, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(ah8, bl8, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(al8, bh8, c3, 0, 0, 0);
    }
    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_hi8, b_lo8, c0, 1, 1, 0, 127, 0, 115);
    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_lo8, b_hi8, c1, 1, 1, 0, 115, 0, 127);
    c4 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_hi8, b_lo8, c4, 1, 1, 0, 127, 0, 115);
    c4 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_lo8, b_hi8, c4, 1, 1, 0, 115, 0, 127);
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    Csimple[row * 32 + r] = c4[i];
    X1s[row * 32 + r] = c0[i];
    X2s[row * 32 + r] = c1[i];
    X1u[row * 32 + r] = c2[i] * (1.f / 4096.f);
    X2u[row * 32 + r] = c3[i] * (1.f / 4096.f);
  }
}

__global__ void perm_check(unsigned* out) { out[0] = perm_hi8(0x44332211u, 0x88776655u); }

// ---- 4: issue rate: per iteration 16 f16 MFMAs + 8 scaled MFMAs on 4 accumulators (the K = 64 step of a 64x64 wave tile) ----
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, unsigned seed) {
  const int l = threadIdx.x;
  halfx8 a[2], b[2];
  intx8 a8[2], b8[2];
  bf16x8 ab[2], bb[2];
  for (int i = 0; i < 2; ++i) {
    for (int j = 0; j < 8; ++j) {
      unsigned s = (seed + l * 977u + i * 131u + j * 17u) * 2654435761u;
      a[i][j] = (_Float16)(((s >> 8) & 1023) / 512.f - 1.f);
      b[i][j] = (_Float16)(((s >> 18) & 1023) / 512.f - 1.f);
      ab[i][j] = (__bf16)(((s >> 8) & 1023) / 512.f - 1.f);
      bb[i][j] = (__bf16)(((s >> 18) & 1023) / 512.f - 1.f);
      a8[i][j] = (int)((s & 0x3b3b3b3bu) | 0x30303030u);
      b8[i][j] = (int)(((s >> 3) & 0x3b3b3b3bu) | 0x30303030u);
    }
  }
  floatx16 c[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 16 f16 + 8 scaled (the f16c8 step)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], c[i][j], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            c[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], c[i][j], 1, 1, 0, 115, 0, 127);
    } else if (MODE == 1) {  // 48 bf16 (the bf16x3 work of the same K = 64)
#pragma unroll
      for (int s = 0; s < 12; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[i], bb[j], c[i][j], 0, 0, 0);
    } else {  // 8 scaled alone
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            c[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], c[i][j], 1, 1, 0, 115, 0, 127);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += c[i][j][r];
  if (s == 123.456f) out[0] = s;
}

// ---- 5: raw lane map of ds_read_b64_tr_b8: lane i supplies the address of the 8 bytes [8i, 8i + 8) ----
__global__ void tr8_dump(unsigned long long* out_src_lane, unsigned long long* out_src_byte) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][512];
  for (int i = threadIdx.x; i < 512; i += 64) {
    lds[0][i] = (unsigned char)(i >> 3);
    lds[1][i] = (unsigned char)(i & 7);
  }
  __syncthreads();
  const int lane = threadIdx.x;
  __attribute__((address_space(3))) intx2* p0 = (__attribute__((address_space(3))) intx2*)(&lds[0][8 * lane]);
  __attribute__((address_space(3))) intx2* p1 = (__attribute__((address_space(3))) intx2*)(&lds[1][8 * lane]);
  intx2 v0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32(p0);
  intx2 v1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32(p1);
  out_src_lane[lane] = *reinterpret_cast<unsigned long long*>(&v0);
  out_src_byte[lane] = *reinterpret_cast<unsigned long long*>(&v1);
}

static double rel_l2(const std::vector<float>& got, const std::vector<double>& ref) {
  double n = 0, d = 0;
  for (size_t i = 0; i < ref.size(); ++i) {
    n += (got[i] - ref[i]) * (got[i] - ref[i]);
    d += ref[i] * ref[i];
  }
  return sqrt(n / d);
}
static double max_rel_to_absdot(const std::vector<float>& got, const std::vector<double>& ref, const std::vector<double>& absdot) {
  double m = 0;
  for (size_t i = 0; i < ref.size(); ++i) m = fmax(m, fabs(got[i] - ref[i]) / absdot[i]);
  return m;
}

int main() {
  // ---- 1 / 2 ----
  {
    std::vector<unsigned char> A(32 * 64), B(64 * 32);
    srand(1);
    // small exactly representable e5m2 values: +-{0, 0.5, 1, 1.5, 2, 3}
    const float vals[6] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f};
    std::vector<float> Af(32 * 64), Bf(64 * 32);
    for (int i = 0; i < 32 * 64; ++i) {
      float a = vals[rand() % 6] * ((rand() & 1) ? 1.f : -1.f), b = vals[rand() % 6] * ((rand() & 1) ? 1.f : -1.f);
      Af[i] = a;
      Bf[i] = b;
      A[i] = e5m2_rne(a);
      B[i] = e5m2_rne(b);
      if (e5m2_to_f32(A[i]) != a || e5m2_to_f32(B[i]) != b) {
        printf("host e5m2 conversion broken\n");
        return 2;
      }
    }
    unsigned char *dA, *dB;
    float* dC;
    CK(hipMalloc(&dA, A.size()));
    CK(hipMalloc(&dB, B.size()));
    CK(hipMalloc(&dC, 32 * 32 * 4));
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    const int scales[3][2] = {{127, 127}, {115, 127}, {127 | (99 << 8), 115 | (7 << 8)}};
    const double mult[3] = {1.0, 1.0 / 4096, 1.0 / 4096};
    for (int t = 0; t < 3; ++t) {
      hipLaunchKernelGGL(scaled_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, scales[t][0], scales[t][1]);
      std::vector<float> C(32 * 32);
      CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
      int bad = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double ref = 0;
          for (int k = 0; k < 64; ++k) ref += (double)Af[i * 64 + k] * Bf[k * 32 + j];
          ref *= mult[t];
          if (C[i * 32 + j] != (float)ref) {
            if (bad < 4) printf("  scaled MFMA [%d][%d]: got %g want %g\n", i, j, C[i * 32 + j], ref);
            ++bad;
          }
        }
      printf("scaled e5m2 MFMA, E8M0 scales (%d, %d) [low byte]: %s (%d wrong)\n", scales[t][0] & 255, scales[t][1] & 255,
             bad ? "MISMATCH" : "lane map + scale OK", bad);
    }
  }
  {
    unsigned* d;
    CK(hipMalloc(&d, 4));
    hipLaunchKernelGGL(perm_check, dim3(1), dim3(1), 0, 0, d);
    unsigned h;
    CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("perm_hi8(0x44332211, 0x88776655) = 0x%08x (want 0x88664422)\n", h);
  }
  // ---- 3 ----
  {
    for (int cfg = 0; cfg < 4; ++cfg) {  // 0 gaussian, 1 post-ReLU activations, 2 heavy-tailed magnitudes, 3 small activations
      const int K = 4608;
      std::vector<float> A(32 * K), B(K * 32);
      srand(7 + cfg);
      auto gauss = []() {
        double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
        return sqrt(-2 * log(u)) * cos(6.283185307179586 * v);
      };
      for (int i = 0; i < 32 * K; ++i) {
        double a = gauss(), b = gauss() * 0.02;
        if (cfg == 1) a = a > 0 ? a : 0;             // post-ReLU activations
        if (cfg == 2) a *= exp(1.5 * gauss());       // heavy-tailed magnitudes
        if (cfg == 3) { a *= 1e-3; b *= 30; }        // small activations
        A[i] = (float)a;
        B[i] = (float)b;
      }
      float *dA, *dB, *dC[4];
      CK(hipMalloc(&dA, A.size() * 4));
      CK(hipMalloc(&dB, B.size() * 4));
      for (int i = 0; i < 4; ++i) CK(hipMalloc(&dC[i], 32 * 32 * 4));
      CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
      std::vector<double> ref(32 * 32), absdot(32 * 32);
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double s = 0, t = 0;
          for (int k = 0; k < K; ++k) {
            s += (double)A[i * K + k] * B[k * 32 + j];
            t += fabs((double)A[i * K + k] * B[k * 32 + j]);
          }
          ref[i * 32 + j] = s;
          absdot[i * 32 + j] = t;
        }
      {
        // the cross terms alone against the host's evaluation of the same 8-bit operands
        float* dX[5];
        for (int i = 0; i < 5; ++i) CK(hipMalloc(&dX[i], 32 * 32 * 4));
        hipLaunchKernelGGL(cross_debug, dim3(1), dim3(64), 0, 0, dA, dB, K, dX[0], dX[1], dX[2], dX[3], dX[4]);
        std::vector<float> X[5];
        for (int i = 0; i < 5; ++i) {
          X[i].resize(32 * 32);
          CK(hipMemcpy(X[i].data(), dX[i], 32 * 32 * 4, hipMemcpyDeviceToHost));
        }
        std::vector<double> r1(32 * 32), r2(32 * 32), t1(32 * 32), t2(32 * 32);
        for (int i = 0; i < 32; ++i)
          for (int j = 0; j < 32; ++j) {
            double s1 = 0, s2 = 0, e1 = 0, e2 = 0;
            for (int k = 0; k < K; ++k) {
              const float xa = A[i * K + k], xb = B[k * 32 + j];
              const unsigned short ha = f32_to_f16_bits(xa), hb = f32_to_f16_bits(xb);
              const float la = (xa - f16_bits_to_f32(ha)) * 4096.f, lb = (xb - f16_bits_to_f32(hb)) * 4096.f;
              const double ah8 = e5m2_to_f32((unsigned char)(ha >> 8)), bh8 = e5m2_to_f32((unsigned char)(hb >> 8));
              const double al8 = e5m2_to_f32(e5m2_rne(la)), bl8 = e5m2_to_f32(e5m2_rne(lb));
              s1 += ah8 * bl8 / 4096;
              s2 += al8 * bh8 / 4096;
              e1 += (double)f16_bits_to_f32(ha) * lb / 4096;
              e2 += (double)la / 4096 * f16_bits_to_f32(hb);
            }
            r1[i * 32 + j] = s1; r2[i * 32 + j] = s2; t1[i * 32 + j] = e1; t2[i * 32 + j] = e2;
          }
        printf("cross terms alone, rel-L2 to the host's sum of the same 8-bit operands: scaled X1 %.3e X2 %.3e | unscaled bf8 X1 %.3e X2 %.3e\n",
               rel_l2(X[0], r1), rel_l2(X[1], r2), rel_l2(X[2], r1), rel_l2(X[3], r2));
        printf("                    rel-L2 to the exact cross terms: scaled X1 %.3e X2 %.3e\n", rel_l2(X[0], t1), rel_l2(X[1], t2));
        printf("cfg %d: f16c8 in a plain kernel (hi8 truncated): rel-L2 %.3e  max|err|/sum|ab| %.3e\n", cfg, rel_l2(X[4], ref), max_rel_to_absdot(X[4], ref, absdot));
      }
      {
        hipLaunchKernelGGL(other_tiles, dim3(1), dim3(64), 0, 0, dA, dB, K, dC[1], dC[2]);
        std::vector<float> C1(32 * 32), C2(32 * 32);
        CK(hipMemcpy(C1.data(), dC[1], 32 * 32 * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(C2.data(), dC[2], 32 * 32 * 4, hipMemcpyDeviceToHost));
        printf("cfg %d: for comparison, rel-L2 / max|err|/sum|ab|:  bf16x3 %.3e / %.3e   plain f16 %.3e / %.3e\n", cfg, rel_l2(C1, ref),
               max_rel_to_absdot(C1, ref, absdot), rel_l2(C2, ref), max_rel_to_absdot(C2, ref, absdot));
      }
    }
  }
  // ---- 4 ----
  {
    float* d;
    CK(hipMalloc(&d, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 20000, blocks = 256 * 2;  // 8 waves per CU = 2 per SIMD
    const char* names[3] = {"16 f16 + 8 scaled e5m2 (f16c8 step, K = 64)", "48 bf16 (bf16x3 step, K = 64)", "8 scaled e5m2 alone"};
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
        if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
        if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        // algorithmic flops of the step: 4 tiles of 32x32 over K = 64
        const double flops = 2.0 * 4 * 32 * 32 * 64 * (double)iters * blocks * 4;
        if (rep == 1) printf("rate %-48s %8.3f ms  %7.1f algorithmic TFLOP/s\n", names[mode], ms, flops / ms * 1e-9);
      }
    }
  }
  // ---- 5 ----
  {
    unsigned long long *d0, *d1;
    CK(hipMalloc(&d0, 64 * 8));
    CK(hipMalloc(&d1, 64 * 8));
    hipLaunchKernelGGL(tr8_dump, dim3(1), dim3(64), 0, 0, d0, d1);
    unsigned long long h0[64], h1[64];
    CK(hipMemcpy(h0, d0, sizeof(h0), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1, d1, sizeof(h1), hipMemcpyDeviceToHost));
    printf("ds_read_b64_tr_b8: dest lane: (source lane . source byte) for result bytes 0..7\n");
    for (int lane = 0; lane < 64; ++lane) {
      printf("  lane %2d:", lane);
      for (int j = 0; j < 8; ++j) printf(" %2d.%d", (int)((h0[lane] >> (8 * j)) & 255), (int)((h1[lane] >> (8 * j)) & 255));
      printf("\n");
    }
  }
  return 0;
}
