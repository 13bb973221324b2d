// Device helpers shared by the translation units of the plane-format convolution family (conv3.hip, conv4.hip): included INSIDE
// the includer's anonymous namespace, after planes_fmt.h (everything here depends on the plane format of the build, PP_FMT).
// The packed plane layout, the LDS-staged epilogue of every implicit-GEMM kernel, raw buffer loads / stores and the LDS-DMA piece.
#pragma once

__device__ __forceinline__ void split8(const float4& lo4, const float4& hi4, uint4* out_hi, uint4* out_lo) {
  fmt_encode2(lo4.x, lo4.y, &out_hi->x, &out_lo->x);
  fmt_encode2(lo4.z, lo4.w, &out_hi->y, &out_lo->y);
  fmt_encode2(hi4.x, hi4.y, &out_hi->z, &out_lo->z);
  fmt_encode2(hi4.z, hi4.w, &out_hi->w, &out_lo->w);
}

// AP: the gathered operand comes from pre-split (hi, lo) planes (pp_split_planes_bf16x3, same [rows][ld]
// geometry as the f32 tensor) instead of being split from f32 while staging: no conversion VALU in the loop.
__device__ __forceinline__ void split4(const float4& v4, uint2* out_hi, uint2* out_lo) {
  fmt_encode2(v4.x, v4.y, &out_hi->x, &out_lo->x);
  fmt_encode2(v4.z, v4.w, &out_hi->y, &out_lo->y);
}

// ---- the PACKED plane layout ----
// A tensor [rows][ld] (ld % 8 == 0) stored as (hi, lo) planes occupies rows * ld * 4 bytes, like float32, cut into 32-byte
// groups of 8 consecutive channels: bytes 0..15 = the 8 hi halves, bytes 16..31 = the 8 lo units (value = hi + lo8 * 2^-12, within
// 2^-15 of the float32 that was split).  "hi" points at the buffer, "lo" 16 bytes behind it, and BOTH are addressed with the
// float32 byte offsets of the element's group -- so a staging thread that used to fetch the two float4 of an octet fetches
// the same 32 contiguous bytes, and 128-byte row segments stay whole (two separate planes cut every access into 64-byte
// halves and cost 5 % on the forward launches: profiles/r02_planes_separate_vs_packed.txt).
// index of the 8-byte half-group (4 channels) i4 = element / 4 of a plane, in uint2 units from the plane's pointer
__device__ __forceinline__ long long pk4(long long i4) { return ((i4 >> 1) << 2) + (i4 & 1); }

// four consecutive elements (element index 4 * i4) of a tensor stored as planes
__device__ __forceinline__ float4 planes_ld4(const void* hi, const void* lo, long long i4) {
  const long long q = pk4(i4);
  const uint2 h = reinterpret_cast<const uint2*>(hi)[q], l = reinterpret_cast<const uint2*>(lo)[q];
  float4 v;
  fmt_value2(h.x, l.x, &v.x, &v.y);
  fmt_value2(h.y, l.y, &v.z, &v.w);
  return v;
}
// the hi plane alone: enough for the sign / zero test of a ReLU source (+1 / 0 per element)
__device__ __forceinline__ float4 hi_ld4(const void* hi, long long i4) {
  const uint2 h = reinterpret_cast<const uint2*>(hi)[pk4(i4)];
  return make_float4(fmt_pos(h.x & 0xffffu) ? 1.f : 0.f, fmt_pos(h.x >> 16) ? 1.f : 0.f, fmt_pos(h.y & 0xffffu) ? 1.f : 0.f,
                     fmt_pos(h.y >> 16) ? 1.f : 0.f);
}
__device__ __forceinline__ void planes_st4(void* hi, void* lo, long long i4, const float4& v) {
  uint2 oh, ol;
  split4(v, &oh, &ol);
  const long long q = pk4(i4);
  reinterpret_cast<uint2*>(hi)[q] = oh;
  reinterpret_cast<uint2*>(lo)[q] = ol;
}

// exchange with the neighbouring lane (lane ^ 1) inside a quad: DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ unsigned lane_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }

// exact a / b for 0 <= a < 2^24, b >= 1 by one reciprocal multiply and a +-1 correction
__device__ __forceinline__ int div_small(int a, int b, float rcp_b, int* rem) {
  int q = (int)((float)a * rcp_b);
  int r = a - q * b;
  const bool lo = r < 0;
  q -= lo ? 1 : 0;
  r += lo ? b : 0;
  const bool hi = r >= b;
  q += hi ? 1 : 0;
  r -= hi ? b : 0;
  *rem = r;
  return q;
}

// ---- epilogue through LDS (see conv.hip), shared by both main loops ----
// (Measured and not kept: non-temporal stores for the output tile -- the next launch reads it back, and the step lost 1 %.)
// RL: the tile's BM rows are BM / 32 listed 32-row blocks (rl_blk[j] = first row of block j of this tile, >= M when the
// list has ended) instead of the consecutive rows m0 ..
template <int TM, int TN, bool OP, int GOP = 1, bool SC = false, bool XR = false, bool RL = false, int NWM = 2, int NT = 256, int SUBF = 0>
__device__ __forceinline__ void epilogue3(const IgemmParams& p, floatx16 (&acc)[TM][TN], uint4* smem, int m0, int n0, int tid, int wm,
                                          int wn, int il, int h, const float* __restrict__ g_bias,
                                          const float* __restrict__ g_addend, const float* __restrict__ g_mask,
                                          float* __restrict__ g_out, uint2* __restrict__ g_ohi, uint2* __restrict__ g_olo,
                                          const int* rl_blk = nullptr) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32, NO = BK / 8;
  constexpr int SMEM_U4 = 2 * NO * (BM + BN);
  // blocked sub-tiles here, so staged row = (a - a0) * 32 + i
  constexpr int SUB = SUBF ? SUBF : ((TM == 4) ? 2 : ((TM * BN > 256 * 1) ? 1 : TM));  // 32*SUB rows x BN floats must fit the LDS buffer
  constexpr int ROWS = 32 * SUB;
  constexpr int C4 = BN / 4;
  constexpr int RPI = NT / C4;
  constexpr int SWEEPS = ROWS / RPI;
  static_assert(ROWS * BN * 4 <= SMEM_U4 * 16, "epilogue staging does not fit");
  static_assert(SWEEPS >= 1 && ROWS % RPI == 0, "epilogue sweep geometry");
  float* stage = reinterpret_cast<float*>(smem);
  const int e_c4 = tid % C4, e_r = tid / C4;
  const int co = n0 + 4 * e_c4;
  const bool odd = (e_c4 & 1) != 0;  // lanes 2j / 2j + 1 hold channels 8j .. 8j+3 / 8j+4 .. 8j+7 of the same row
  const bool col_ok = co < ((p.Nout + 3) & ~3);
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g_bias && col_ok) bias4 = *reinterpret_cast<const float4*>(g_bias + co);
  const int m_last = p.M - 1;
  const bool add_pl = p.add_hi != nullptr, mask_pl = p.mask_hi != nullptr;  // (uniform) operands stored as bf16 planes
#pragma unroll
  for (int hm = 0; hm < NWM; ++hm) {
#pragma unroll
    for (int a0 = 0; a0 < TM; a0 += SUB) {
      __syncthreads();
      if (wm == hm) {
#pragma unroll
        for (int as = 0; as < SUB; ++as)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = as * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int b = 0; b < TN; ++b) stage[row * BN + wn * 32 * TN + b * 32 + il] = acc[a0 + as][b][r];
          }
      }
      __syncthreads();
      if (col_ok) {
        constexpr int G = OP ? (SWEEPS < GOP ? SWEEPS : GOP) : (SWEEPS < 4 ? SWEEPS : 4);  // rows in flight per thread
        const int base_row = m0 + hm * 32 * TM + a0 * 32;
        auto sweep = [&](auto has_add, auto has_mask) {
#pragma unroll
          for (int s0 = 0; s0 < SWEEPS; s0 += G) {
            float4 ad[G], mk[G];
            int mo[G];  // row of the output / addend / mask tensors (SC: the class row scattered into the full grid)
#pragma unroll
            for (int g = 0; g < G; ++g) {
              int m = XR ? max(min(base_row + e_r + RPI * (s0 + g), m_last), 0) : min(base_row + e_r + RPI * (s0 + g), m_last);
              if (RL) {
                const int t = hm * 32 * TM + a0 * 32 + e_r + RPI * (s0 + g);
                m = min(rl_blk[t >> 5] + (t & 31), m_last);
              }
              mo[g] = m;
              if (SC) {
                const int hw = p.seg[0].OH * p.seg[0].OW;
                int rem, xq;
                const int n = div_small(m, hw, __frcp_rn((float)hw), &rem);
                const int yq = div_small(rem, p.seg[0].OW, __frcp_rn((float)p.seg[0].OW), &xq);
                mo[g] = (n * p.sc_H + 2 * yq + p.sc_cy) * p.sc_W + 2 * xq + p.sc_cx;
              }
              if (has_add) {
                if (add_pl) {
                  // packed planes, 16 bytes per lane: the even lane of a pair fetches the group's hi half, the odd lane its lo half
                  // (8-byte loads run at 0.5-0.7 of the 16-byte rate: the HBM-bound 1x1 launches were 6-25 % slower with them)
                  const long long grp = ((long long)mo[g] * p.ld_add + (co & ~7)) >> 3;
                  const uint4 q = reinterpret_cast<const uint4*>(p.add_hi)[2 * grp + (odd ? 1 : 0)];
                  ad[g] = *reinterpret_cast<const float4*>(&q);  // raw halves: combined after the exchange below
                } else {
                  ad[g] = *reinterpret_cast<const float4*>(g_addend + (long long)mo[g] * p.ld_add + co);
                }
              }
              if (has_mask) {
                if (mask_pl) {
                  // the hi half of the group alone (hi > 0 <=> value > 0): the even lane fetches it for the pair
                  const long long grp = ((long long)mo[g] * p.ld_mask + (co & ~7)) >> 3;
                  uint4 q = make_uint4(0u, 0u, 0u, 0u);
                  if (!odd) q = reinterpret_cast<const uint4*>(p.mask_hi)[2 * grp];
                  mk[g] = *reinterpret_cast<const float4*>(&q);
                } else {
                  mk[g] = *reinterpret_cast<const float4*>(g_mask + (long long)mo[g] * p.ld_mask + co);
                }
              }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const int row = e_r + RPI * (s0 + g);
              int m = base_row + row;
              if (RL) {
                const int t = hm * 32 * TM + a0 * 32 + row;
                m = rl_blk[t >> 5] + (t & 31);
              }
              float4 v = *reinterpret_cast<const float4*>(stage + row * BN + 4 * e_c4);
              v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
              if (has_add && add_pl) {
                // even lane holds hi[0..7], odd lane lo[0..7] of the pair's 8 channels: the even lane needs lo[0..3] (odd's first
                // half), the odd lane hi[4..7] (even's second half)
                const uint4 q = *reinterpret_cast<const uint4*>(&ad[g]);
                const unsigned sx = odd ? q.x : q.z, sy = odd ? q.y : q.w;
                const unsigned rx = lane_xor1(sx), ry = lane_xor1(sy);
                const unsigned hx = odd ? rx : q.x, hy = odd ? ry : q.y, lx = odd ? q.z : rx, ly = odd ? q.w : ry;
                fmt_value2(hx, lx, &ad[g].x, &ad[g].y);
                fmt_value2(hy, ly, &ad[g].z, &ad[g].w);
              }
              if (has_mask && mask_pl) {
                const uint4 q = *reinterpret_cast<const uint4*>(&mk[g]);
                const unsigned rx = lane_xor1(q.z), ry = lane_xor1(q.w);  // the even lane's second half
                const unsigned hx = odd ? rx : q.x, hy = odd ? ry : q.y;
                mk[g] = make_float4(fmt_pos(hx & 0xffffu) ? 1.f : 0.f, fmt_pos(hx >> 16) ? 1.f : 0.f, fmt_pos(hy & 0xffffu) ? 1.f : 0.f,
                                    fmt_pos(hy >> 16) ? 1.f : 0.f);
              }
              if (has_add) { v.x += ad[g].x; v.y += ad[g].y; v.z += ad[g].z; v.w += ad[g].w; }
              if (has_mask) {
                v.x = mk[g].x > 0.f ? v.x : 0.f; v.y = mk[g].y > 0.f ? v.y : 0.f;
                v.z = mk[g].z > 0.f ? v.z : 0.f; v.w = mk[g].w > 0.f ? v.w : 0.f;
              }
              if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
              // XR (igemm3x): tile rows 0 and BM - 1 are halo rows that the neighbouring tiles own
              const int trow = hm * 32 * TM + a0 * 32 + row;
              if (m <= m_last && (!XR || (trow >= 1 && trow <= 32 * TM * NWM - 2))) {
                if (!OP || g_out != nullptr) *reinterpret_cast<float4*>(g_out + (long long)mo[g] * p.ld_out + co) = v;
              }
              if (OP) {
                // the output as packed planes: lanes 2j / 2j + 1 hold channels 8j .. 8j+3 / 8j+4 .. 8j+7 of the same row; they swap
                // halves so that the even lane stores the group's 16 hi bytes and the odd lane its 16 lo bytes (one 16-byte
                // store per lane, like the float32 tile; Nout % 8 == 0 is host-checked, so both lanes of a pair are in range)
                uint2 oh, ol;
                split4(v, &oh, &ol);
                const unsigned sx = odd ? oh.x : ol.x, sy = odd ? oh.y : ol.y;
                const unsigned rx = lane_xor1(sx), ry = lane_xor1(sy);
                const uint4 o16 = odd ? make_uint4(rx, ry, ol.x, ol.y) : make_uint4(oh.x, oh.y, rx, ry);
                if (m <= m_last && (!XR || (trow >= 1 && trow <= 32 * TM * NWM - 2))) {
                  const long long grp = ((long long)mo[g] * p.ld_out + (co & ~7)) >> 3;  // 32-byte group
                  reinterpret_cast<uint4*>(g_ohi)[2 * grp + (odd ? 1 : 0)] = o16;
                }
              }
            }
          }
        };
        const bool any_add = g_addend != nullptr || add_pl, any_mask = g_mask != nullptr || mask_pl;
        if (any_add && any_mask) sweep(std::true_type{}, std::true_type{});
        else if (any_add) sweep(std::true_type{}, std::false_type{});
        else if (any_mask) sweep(std::false_type{}, std::true_type{});
        else sweep(std::false_type{}, std::false_type{});
      }
    }
  }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define PP_BUF_OOB ((int)0x80000000)

__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, const uint4& v, int voff) {
  u32x4 q;
  q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
  __builtin_amdgcn_raw_buffer_store_b128(q, r, voff, 0, 0);  // out-of-range offsets are dropped
}


__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, uint4* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
