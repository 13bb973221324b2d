// Implicit-GEMM convolution in the "f16c8" arithmetic: float32-class accuracy at TWO matrix-core units per product
// (conv3.hip's bf16x3 needs three):
//   x = x_hi + x_lo,   x_hi = f16(x) (11-bit significand),   x_lo8 = e5m2(x_lo * 2^12)      (the same for w)
//   x*w ~= x_hi*w_hi                                   v_mfma_f32_32x32x16_f16, K = 16, 8 passes
//        + (x_hi8*w_lo8 + x_lo8*w_hi8) * 2^-12         v_mfma_scale_f32_32x32x64_f8f6f4 on e5m2 operands, K = 64, 16 passes:
//                                                      twice the cycles of the f16 form for four times the K
// x_hi8 = e5m2(x_hi) costs no memory and no LDS traffic: an e5m2 number IS the top byte of an IEEE half, so one v_perm_b32
// turns four halves of a lane's f16 fragment into four bytes of its e5m2 fragment (truncation: the cross terms are 2^-12 of
// the product, their 2^-3 relative error ~2^-15 of it).  The E8M0 block scale of the scaled MFMA applies the 2^-12.
// Per product: 1 + 1/2 + 1/2 MFMA units, error ~3e-5 relative L2 on a K = 4608 reduction of random data (bf16x3: 4.5e-6,
// plain f16: 2.9e-4; tools/ubench/f16c8_check.hip measures all of it on the hardware, profiles/r03_f16c8_check.txt).
//
// STATUS (round 3): a measured experiment, NOT wired into the engine.  The kernel below is exact to the scheme (3.1e-5 against
// float64 on the head shapes) and runs the 512-wide head conv 1.27-1.41x faster than igemm3x on the same box (436-489 vs
// 335-346 algorithmic TFLOP/s in tools/conv_bench.py); the go / no-go gate was 1.34x (520 vs 389).  Ablations
// (profiles/r03_f16c8_ablations.txt): with 1/3 fewer MFMA cycles the loop is co-limited by staging + fragment reads (298 us
// without any MFMA, 305 us without fragment reads, 462 us whole, tail-free shape) -- DESIGN.md section 7d.
//
// ---- the H16L8 storage format ("hl") ----
// A tensor [rows][ld], ld % 64 == 0, occupies rows * ld * 3 bytes, cut into 192-byte groups of 64 consecutive channels:
//   bytes   0..127  the 64 f16 hi values in channel order
//   bytes 128..191  the 64 e5m2 lo bytes, ordered for the MFMA lane halves: channel c of the group (octet o = c >> 3, j = c & 7)
//                   sits at 128 + 32 * (o & 1) + 8 * (o >> 1) + j  -- lane half h of a fragment owns the octets h, 2 + h, 4 + h,
//                   6 + h of a 64-deep k-step (its four f16 fragments), so its 32 lo bytes are one contiguous run.
// Rows are 3 * ld bytes apart; every 16-byte piece of a group is 16-byte aligned.
#include "conv_common.h"

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define PP_BUF_OOB ((int)0x80000000)

namespace {

__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- f32 -> (f16 hi, e5m2 lo) ----
// hi saturates at the largest half (a value beyond +-65504 is outside the format: documented limit of the mode)
__device__ __forceinline__ void hl_split1(float v, unsigned short* hi, unsigned char* lo) {
  v = fminf(fmaxf(v, -65504.f), 65504.f);
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)((v - (float)h) * 4096.f);
  const unsigned short ul = __builtin_bit_cast(unsigned short, l);
  *hi = __builtin_bit_cast(unsigned short, h);
  *lo = (unsigned char)((ul + 0x7fu + ((ul >> 8) & 1u)) >> 8);  // round to nearest even onto the top byte
}

__device__ __forceinline__ void hl_split8(const float4& a, const float4& b, uint4* hi, uint2* lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned short h[8];
  unsigned char l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) hl_split1(v[j], &h[j], &l[j]);
  hi->x = h[0] | ((unsigned)h[1] << 16);
  hi->y = h[2] | ((unsigned)h[3] << 16);
  hi->z = h[4] | ((unsigned)h[5] << 16);
  hi->w = h[6] | ((unsigned)h[7] << 16);
  lo->x = l[0] | ((unsigned)l[1] << 8) | ((unsigned)l[2] << 16) | ((unsigned)l[3] << 24);
  lo->y = l[4] | ((unsigned)l[5] << 8) | ((unsigned)l[6] << 16) | ((unsigned)l[7] << 24);
}

// byte offset of octet o (8 channels) of a 64-channel group: hi part, lo part
__device__ __forceinline__ int hl_hi_off(int o) { return 16 * o; }
__device__ __forceinline__ int hl_lo_off(int o) { return 128 + 32 * (o & 1) + 8 * (o >> 1); }

// the top bytes of four halves (two dwords) -> one dword of e5m2
__device__ __forceinline__ int hi8_of(unsigned lo_pair, unsigned hi_pair) { return (int)__builtin_amdgcn_perm(hi_pair, lo_pair, 0x07050301u); }

}  // namespace

// ---- f32 [rows][ld] -> hl ----
__global__ void split_hl_kernel(long long n8, int ld8, const float4* __restrict__ src, unsigned char* __restrict__ dst) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const float4 a = src[2 * i], b = src[2 * i + 1];
    uint4 hi;
    uint2 lo;
    hl_split8(a, b, &hi, &lo);
    const long long row = i / ld8;
    const int o_row = (int)(i - row * ld8);  // octet within the row
    unsigned char* g = dst + (row * (ld8 >> 3) + (o_row >> 3)) * 192;
    const int o = o_row & 7;
    *reinterpret_cast<uint4*>(g + hl_hi_off(o)) = hi;
    *reinterpret_cast<uint2*>(g + hl_lo_off(o)) = lo;
  }
}

// hl -> f32 (tests, debugging)
__global__ void merge_hl_kernel(long long n8, int ld8, const unsigned char* __restrict__ src, float4* __restrict__ dst) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / ld8;
    const int o_row = (int)(i - row * ld8);
    const unsigned char* g = src + (row * (ld8 >> 3) + (o_row >> 3)) * 192;
    const int o = o_row & 7;
    const uint4 hi = *reinterpret_cast<const uint4*>(g + hl_hi_off(o));
    const uint2 lo = *reinterpret_cast<const uint2*>(g + hl_lo_off(o));
    const unsigned hw[4] = {hi.x, hi.y, hi.z, hi.w};
    const unsigned lw[2] = {lo.x, lo.y};
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned short hb = (unsigned short)(hw[j >> 1] >> (16 * (j & 1)));
      const unsigned short lb = (unsigned short)(((lw[j >> 2] >> (8 * (j & 3))) & 0xffu) << 8);
      v[j] = (float)__builtin_bit_cast(_Float16, hb) + (float)__builtin_bit_cast(_Float16, lb) * (1.f / 4096.f);
    }
    dst[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
    dst[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// ---- weights: f32 HWIO [tap * cin + ci][ld_w] -> hl, both k-contiguous layouts: forward [tap][cout rows][cin], bwd-data
// [tap][cin][cout rounded up to 64] ----
__global__ void split_weights_hl_kernel(int cin, int cout, int ld_w, const float* __restrict__ w, unsigned char* __restrict__ fwd,
                                        int cout_rows, unsigned char* __restrict__ dg, int dg_ld) {
  // one block: 64 (ci) x 64 (co) of one tap through LDS, so that both layouts are written as whole 192-byte groups
  __shared__ unsigned short t_hi[64][65];
  __shared__ unsigned char t_lo[64][68];
  const int tap = blockIdx.z, ci0 = blockIdx.y * 64, co0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (int j = ty; j < 64; j += 4) {
    const int ci = ci0 + j, co = co0 + tx;
    float v = 0.f;
    if (ci < cin && co < cout) v = w[((long long)tap * cin + ci) * ld_w + co];
    unsigned short h;
    unsigned char l;
    hl_split1(v, &h, &l);
    t_hi[j][tx] = h;
    t_lo[j][tx] = l;
  }
  __syncthreads();
  // the group of (row, 64 k-values): channel c -> hi at 2c, lo at hl_lo_off(c >> 3) + (c & 7)
  if (fwd) {  // row = output channel, k = input channel
    for (int j = ty; j < 64; j += 4) {
      const int co = co0 + j;
      if (co < cout_rows) {
        unsigned char* g = fwd + (((long long)tap * cout_rows + co) * (cin >> 6) + (ci0 >> 6)) * 192;
        reinterpret_cast<unsigned short*>(g)[tx] = t_hi[tx][j];
        g[hl_lo_off(tx >> 3) + (tx & 7)] = t_lo[tx][j];
      }
    }
  }
  if (dg) {  // row = input channel, k = output channel
    for (int j = ty; j < 64; j += 4) {
      const int ci = ci0 + j;
      if (ci < cin) {
        unsigned char* g = dg + (((long long)tap * cin + ci) * (dg_ld >> 6) + (co0 >> 6)) * 192;
        reinterpret_cast<unsigned short*>(g)[tx] = t_hi[j][tx];
        g[hl_lo_off(tx >> 3) + (tx & 7)] = t_lo[j][tx];
      }
    }
  }
}

// ---- the tap-row-reuse kernel (3-wide stride-1 "same" convs: heads, FPN 3x3, bottleneck 3x3; forward and bwd-data) ----
// Same geometry as conv3.hip's igemm3x_kernel: tile row j <-> output row m0 + j, rows 0 and BM - 1 are halo; the workgroup
// stages its BM gathered rows once per (kernel row, 64-channel chunk) and the three taps of the row read fragments at row
// offsets -1 / 0 / +1, or the all-zero row of the image where the tap pads.  64-deep k-step: per tap and 64x64 wave tile
// 16 f16 MFMAs + 8 scaled MFMAs (1024 MFMA cycles) from 24 ds_read_b128; 12 16-byte pieces per staged row (8 hi + 4 lo).
// LDS image per operand: 12 slots of (rows + 2) x 16 B, un-rotated, so that every fragment read of a lane is ONE base
// address + a compile-time offset (the rotated image of igemm3x cost ~40 address VALU per tap here, with twice the reads per
// MFMA).  Slots are grouped by lane half: half h reads slots 6h .. 6h + 3 (its f16 octets h, 2 + h, 4 + h, 6 + h) and
// 6h + 4, 6h + 5 (its 32 lo bytes); the h = 1 group starts 32 banks off the h = 0 group, the slot pitch is 8 banks off a
// multiple of 64: the 16-byte stores of a staging quad-pair and the 512-byte fragment reads are conflict-free.
// Row `rows` of every slot stays zero: a padded tap reads it (one select of the base address per tap and row block).
// WR: rows of the wave grid (2 columns): 2 * WR waves, tile 32 TM WR x 64 TN.  WR = 2: 128x128, 4 waves, two workgroups per CU.
// WR = 4: 256x128, 8 waves, ONE workgroup per CU (100 KB of LDS): the weight tile is staged once for twice the MFMAs.
template <int TM, int TN, int DB, int WR = 2>
__global__ __launch_bounds__(128 * WR, WR == 2 ? 2 : 1) void igemm2x_kernel(const IgemmParams p, const void* __restrict__ g_a, unsigned a_bytes,
                                                         const void* __restrict__ g_w, unsigned w_bytes, const float* __restrict__ g_bias,
                                                         float* __restrict__ g_out, int w_rows, int w_groups) {
  constexpr int BM = 32 * TM * WR, BN = 64 * TN, NT = 128 * WR, RS = NT / 4;  // RS: rows staged per pass
  constexpr int PA = BM + 2, PB = BN + 2;          // slot pitch in 16-byte units
  constexpr int HA = 6 * PA + 12, HB = 6 * PB + 12;  // start of the h = 1 slot group
  constexpr int A_U4 = 12 * PA + 12, B_U4 = 12 * PB + 12;
  constexpr int SMEM_U4 = A_U4 + (DB != 0 ? 2 : 1) * B_U4;  // DB: the weight tile is double-buffered, one barrier per tap
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];  // SMEM_U4 of them (dynamic: the 8-wave form needs 100 KB)
  uint4* As = smem;
  uint4* Bs0 = smem + A_U4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lb = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_n = lb % p.n_tiles_n, tile_m = lb / p.n_tiles_n;
  const int m0 = tile_m * (BM - 2) - 1, n0 = tile_n * BN;
  const int oct = tid & 3, r0 = tid >> 2;
  const int il = lane & 31, h = lane >> 5;
  const int n_chunks = p.Cred >> 6;
  const int n_groups = p.kh * n_chunks;  // one group = the three taps of (kernel row, chunk); kernel row innermost

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_a), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_w), 0, w_bytes, 0x00020000);

  // staged rows r0 + 64 i: byte offset of the dx = 0 source pixel at ty = 0 (+ this thread's piece), row pitch | validity per ty
  const int row_bytes = p.ld_src * 3;
  constexpr int PA_N = BM / RS, PB_N = BN / RS;  // staging passes over the gathered / the weight tile
  int s_base[PA_N], s_pitch[PA_N];
#pragma unroll
  for (int i = 0; i < PA_N; ++i) {
    const int q = m0 + r0 + RS * i;
    const RowPos r = decode_row(p, q < 0 ? 0 : q);
    const bool ok = q >= 0 && r.ok;
    const int x = r.xbase - p.off_x;
    s_base[i] = (r.rowbase + r.ybase * r.SW + x) * row_bytes + 16 * oct;
    int v = 0;
    for (int ty = 0; ty < p.kh; ++ty)
      if (ok && (unsigned)(r.ybase + ty * p.tsign) < (unsigned)r.SH) v |= 1 << ty;
    s_pitch[i] = (p.tsign * r.SW * row_bytes) | v;  // row_bytes % 192 == 0: the low 4 bits are free
  }
  unsigned f_valid[(TM + 1) / 2];
#pragma unroll
  for (int a = 0; a < (TM + 1) / 2; ++a) f_valid[a] = 0;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int q = m0 + wm * 32 * TM + a * 32 + il;
    const RowPos r = decode_row(p, q < 0 ? 0 : q);
    unsigned v = 0;
    int t = 0;
    for (int ty = 0; ty < p.kh; ++ty)
      for (int tx = 0; tx < 3; ++tx, ++t) {
        const int sy = r.ybase + ty * p.tsign, sx = r.xbase + tx * p.tsign;
        if (q >= 0 && r.ok && (unsigned)sy < (unsigned)r.SH && (unsigned)sx < (unsigned)r.SW) v |= 1u << t;
      }
    f_valid[a >> 1] |= v << (16 * (a & 1));
  }
  int b_base[PB_N];
#pragma unroll
  for (int i = 0; i < PB_N; ++i) {
    const int n = n0 + r0 + RS * i;
    b_base[i] = n < w_rows ? n * w_groups * 192 + 16 * oct : PP_BUF_OOB;
  }
  const int b_tap = w_rows * w_groups * 192;

  // LDS slot of the memory piece 4q + oct this thread stages (memory order: hi octets 0..7, lo of half 0 (2 pieces), lo of half 1)
  int st_a[3], st_b[3];
  {
    const int sl[3] = {6 * (oct & 1) + (oct >> 1), 6 * (oct & 1) + 2 + (oct >> 1), 4 + (oct & 1) + 6 * (oct >> 1)};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      st_a[q] = sl[q] * PA + (sl[q] >= 6 ? 12 : 0) + r0;
      st_b[q] = sl[q] * PB + (sl[q] >= 6 ? 12 : 0) + r0;
    }
  }

  uint4 ra[PA_N][3], rb[PB_N][3];
  int ty = 0, chunk = 0;  // group being LOADED
  auto load_a = [&]() {
#pragma unroll
    for (int i = 0; i < PA_N; ++i) {
      int vo = s_base[i] + __mul24(ty, s_pitch[i] & ~15) + chunk * 192;
      vo = ((s_pitch[i] >> ty) & 1) ? vo : PP_BUF_OOB;
#pragma unroll
      for (int q = 0; q < 3; ++q) ra[i][q] = buf_load16(rs_a, vo, 64 * q);
    }
  };
  auto load_b = [&](int tx) {
    const int b_uni = ((p.w_ty0 + ty) * p.w_kw + tx) * b_tap + chunk * 192;
#pragma unroll
    for (int i = 0; i < PB_N; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) rb[i][q] = buf_load16(rs_w, b_base[i], b_uni + 64 * q);
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < PA_N; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) As[st_a[q] + RS * i] = ra[i][q];
  };
  auto store_b = [&](int buf) {
    uint4* Bs = Bs0 + buf * B_U4;
#pragma unroll
    for (int i = 0; i < PB_N; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) Bs[st_b[q] + RS * i] = rb[i][q];
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int a_lane = h * HA + wm * 32 * TM + il;   // + a * 32 + dx, or the zero row
  const int a_zero = h * HA + BM;
  const uint4* b_lane0 = Bs0 + h * HB + wn * 32 * TN + il;

  auto mma_tile = [&](int c_ty, auto tx_c, int buf) {
    constexpr int TX = decltype(tx_c)::value;
    const int dx = p.off_x + TX * p.tsign;
    const unsigned tap_bits = (1u << (c_ty * 3 + TX)) * 0x10001u;
    const uint4* ap[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const bool ok = ((f_valid[a >> 1] & tap_bits) >> (16 * (a & 1)) & 0xffffu) != 0;
      ap[a] = As + (ok ? a_lane + a * 32 + dx : a_zero);
    }
    const uint4* bp = b_lane0 + buf * B_U4;
    intx8 a8[TM], b8[TN];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 af[TM], bf[TN];
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[b] = bp[s * PB + b * 32];
#pragma unroll
      for (int a = 0; a < TM; ++a) af[a] = ap[a][s * PA];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<halfx8*>(&af[a]), *reinterpret_cast<halfx8*>(&bf[b]), acc[a][b],
                                                             0, 0, 0);
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        a8[a][2 * s] = hi8_of(af[a].x, af[a].y);
        a8[a][2 * s + 1] = hi8_of(af[a].z, af[a].w);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        b8[b][2 * s] = hi8_of(bf[b].x, bf[b].y);
        b8[b][2 * s + 1] = hi8_of(bf[b].z, bf[b].w);
      }
    }
    // cross term 1: x_hi8 * w_lo8 (the lane half's 32 lo bytes: slots 4 and 5 of its group)
    {
      intx8 bl[TN];
#pragma unroll
      for (int b = 0; b < TN; ++b) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const uint4 t = bp[(4 + e) * PB + b * 32];
          bl[b][4 * e] = (int)t.x; bl[b][4 * e + 1] = (int)t.y; bl[b][4 * e + 2] = (int)t.z; bl[b][4 * e + 3] = (int)t.w;
        }
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[a], bl[b], acc[a][b], 1, 1, 0, 127, 0, 115);
    }
    // cross term 2: x_lo8 * w_hi8
    {
      intx8 al[TM];
#pragma unroll
      for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const uint4 t = ap[a][(4 + e) * PA];
          al[a][4 * e] = (int)t.x; al[a][4 * e + 1] = (int)t.y; al[a][4 * e + 2] = (int)t.z; al[a][4 * e + 3] = (int)t.w;
        }
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(al[a], b8[b], acc[a][b], 1, 1, 0, 115, 0, 127);
    }
  };

  std::integral_constant<int, 0> t0;
  std::integral_constant<int, 1> t1;
  std::integral_constant<int, 2> t2;
  if (tid < 12) As[tid * PA + (tid >= 6 ? 12 : 0) + BM] = make_uint4(0u, 0u, 0u, 0u);  // the zero row of every slot
  if constexpr (DB == 1) {
    // Pipeline: per tap [store the weight tile loaded one tap ago into the free buffer] [issue the loads of the tile after it]
    // [MMA on the current buffer] [one barrier].  The gathered tile is single-buffered: a group ends barrier / store / barrier.
    load_a();
    load_b(0);
    store_a();
    store_b(0);
    load_b(1);
    __syncthreads();
    int cur = 0;
    for (int g = 0; g < n_groups; ++g) {
      const int c_ty = ty;
      // tx = 0: tile(tx = 1) is in registers
      store_b(cur ^ 1);
      load_b(2);
      __builtin_amdgcn_sched_barrier(0);
      mma_tile(c_ty, t0, cur);
      __syncthreads();
      cur ^= 1;
      // tx = 1: tile(tx = 2) is in registers
      store_b(cur ^ 1);
      {
        const bool more = g + 1 < n_groups;  // past the end: rewind to group 0 (a harmless re-load)
        ty += 1;
        const bool wt = ty == p.kh;
        ty = wt ? 0 : ty;
        chunk += wt ? 1 : 0;
        chunk = more ? chunk : 0;
        ty = more ? ty : 0;
      }
      load_b(0);
      load_a();
      __builtin_amdgcn_sched_barrier(0);
      mma_tile(c_ty, t1, cur);
      __syncthreads();
      cur ^= 1;
      // tx = 2: tile(next group, tx = 0) and the next gathered rows are in registers
      store_b(cur ^ 1);
      load_b(1);
      __builtin_amdgcn_sched_barrier(0);
      mma_tile(c_ty, t2, cur);
      __syncthreads();
      store_a();
      __syncthreads();
      cur ^= 1;
    }
  } else {
    load_a();
    load_b(0);
    store_a();
    store_b(0);
    __syncthreads();
    for (int g = 0; g < n_groups; ++g) {
      const int c_ty = ty;
      load_b(1);
      __builtin_amdgcn_sched_barrier(0);
      mma_tile(c_ty, t0, 0);
      __syncthreads();
      store_b(0);
      __syncthreads();
      load_b(2);
      __builtin_amdgcn_sched_barrier(0);
      mma_tile(c_ty, t1, 0);
      __syncthreads();
      store_b(0);
      __syncthreads();
      {
        const bool more = g + 1 < n_groups;  // past the end: rewind to group 0 (a harmless re-load)
        ty += 1;
        const bool wt = ty == p.kh;
        ty = wt ? 0 : ty;
        chunk += wt ? 1 : 0;
        chunk = more ? chunk : 0;
        ty = more ? ty : 0;
      }
      load_b(0);
      load_a();
      __builtin_amdgcn_sched_barrier(0);
      mma_tile(c_ty, t2, 0);
      __syncthreads();
      store_b(0);
      store_a();
      __syncthreads();
    }
  }

  // ---- epilogue through LDS: + bias, ReLU, f32 rows of 16 bytes per lane (halo rows 0 and BM - 1 are not written) ----
  float* stage = reinterpret_cast<float*>(smem);
  constexpr int C4 = BN / 4, RPI = NT / C4, ROWS = 32 * TM, SWEEPS = ROWS / RPI;
  static_assert(ROWS * BN * 4 <= SMEM_U4 * 16, "epilogue staging does not fit");
  const int e_c4 = tid % C4, e_r = tid / C4;
  const int co = n0 + 4 * e_c4;
  const bool col_ok = co < ((p.Nout + 3) & ~3);
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g_bias && col_ok) bias4 = *reinterpret_cast<const float4*>(g_bias + co);
#pragma unroll
  for (int hm = 0; hm < WR; ++hm) {
    __syncthreads();
    if (wm == hm) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
          for (int b = 0; b < TN; ++b) stage[row * BN + wn * 32 * TN + b * 32 + il] = acc[a][b][r];
        }
    }
    __syncthreads();
    if (col_ok) {
#pragma unroll
      for (int s = 0; s < SWEEPS; ++s) {
        const int row = e_r + RPI * s;
        const int trow = hm * 32 * TM + row;
        const int m = m0 + trow;
        float4 v = *reinterpret_cast<const float4*>(stage + row * BN + 4 * e_c4);
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (m >= 0 && m < p.M && trow >= 1 && trow <= BM - 2) *reinterpret_cast<float4*>(g_out + (long long)m * p.ld_out + co) = v;
      }
    }
  }
}

// ---- C ABI ----
extern "C" int pp_split_h16l8(pp_ctx* ctx, long long rows, int ld, const float* src, void* dst) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, src && dst && rows >= 0 && ld > 0 && ld % 64 == 0, PP_ERR_ARG, "pp_split_h16l8: ld must be a multiple of 64");
  PP_CHECK_ARG(ctx, pp_is_aligned16(src) && pp_is_aligned16(dst), PP_ERR_ALIGN, "pp_split_h16l8: alignment");
  const long long n8 = rows * (ld / 8);
  if (n8 == 0) return PP_OK;
  long long blocks = (n8 + 255) / 256;
  const long long cap = (long long)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(split_hl_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, n8, ld / 8, (const float4*)src, (unsigned char*)dst);
  PP_CHECK_LAUNCH(ctx, "pp_split_h16l8");
  return PP_OK;
}

extern "C" int pp_merge_h16l8(pp_ctx* ctx, long long rows, int ld, const void* src, float* dst) {
  PP_REQUIRE_CTX(ctx);
  PP_CHECK_ARG(ctx, src && dst && rows >= 0 && ld > 0 && ld % 64 == 0, PP_ERR_ARG, "pp_merge_h16l8: ld must be a multiple of 64");
  PP_CHECK_ARG(ctx, pp_is_aligned16(src) && pp_is_aligned16(dst), PP_ERR_ALIGN, "pp_merge_h16l8: alignment");
  const long long n8 = rows * (ld / 8);
  if (n8 == 0) return PP_OK;
  long long blocks = (n8 + 255) / 256;
  const long long cap = (long long)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(merge_hl_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, n8, ld / 8, (const unsigned char*)src, (float4*)dst);
  PP_CHECK_LAUNCH(ctx, "pp_merge_h16l8");
  return PP_OK;
}

extern "C" int pp_conv_split_weights_f16c8(pp_ctx* ctx, const pp_conv_desc* d, const float* w, void* fwd, void* dgrad) {
  PP_REQUIRE_CTX(ctx);
  int rc = check_desc(ctx, d, "pp_conv_split_weights_f16c8");
  if (rc) return rc;
  PP_CHECK_ARG(ctx, w && (fwd || dgrad), PP_ERR_ARG, "pp_conv_split_weights_f16c8: null tensor");
  PP_CHECK_ARG(ctx, d->cin % 64 == 0, PP_ERR_SHAPE, "pp_conv_split_weights_f16c8: cin %d must be a multiple of 64", d->cin);
  PP_CHECK_ARG(ctx, !fwd || pp_is_aligned16(fwd), PP_ERR_ALIGN, "pp_conv_split_weights_f16c8: alignment");
  PP_CHECK_ARG(ctx, !dgrad || pp_is_aligned16(dgrad), PP_ERR_ALIGN, "pp_conv_split_weights_f16c8: alignment");
  const int taps = d->kh * d->kw;
  const int dg_ld = (d->cout + 63) / 64 * 64;
  dim3 grid((unsigned)(dg_ld / 64), (unsigned)(d->cin / 64), (unsigned)taps);
  hipLaunchKernelGGL(split_weights_hl_kernel, grid, dim3(256), 0, ctx->stream, d->cin, d->cout, d->ld_w, w, (unsigned char*)fwd, d->cout,
                     (unsigned char*)dgrad, dg_ld);
  PP_CHECK_LAUNCH(ctx, "pp_conv_split_weights_f16c8");
  return PP_OK;
}

extern "C" int pp_conv2d_nhwc_fwd_f16c8(pp_ctx* ctx, const pp_conv_desc* d, const void* x_hl, const void* w_hl, const float* bias, int relu,
                                        float* y) {
  PP_REQUIRE_CTX(ctx);
  int rc = check_desc(ctx, d, "pp_conv2d_nhwc_fwd_f16c8");
  if (rc) return rc;
  PP_CHECK_ARG(ctx, x_hl && w_hl && y, PP_ERR_ARG, "pp_conv2d_nhwc_fwd_f16c8: null tensor");
  PP_CHECK_ARG(ctx, d->cin % 64 == 0 && d->ld_x == d->cin, PP_ERR_SHAPE, "pp_conv2d_nhwc_fwd_f16c8: cin %d must be a multiple of 64 and ld_x == cin",
               d->cin);
  PP_CHECK_ARG(ctx, pp_is_aligned16(x_hl) && pp_is_aligned16(w_hl) && pp_is_aligned16(y) && (!bias || pp_is_aligned16(bias)), PP_ERR_ALIGN,
               "pp_conv2d_nhwc_fwd_f16c8: tensors must be 16-byte aligned");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.out = y; p.bias = bias;
  p.ld_src = d->ld_x; p.ld_w = d->ld_w; p.ld_out = d->ld_y;
  p.relu = relu;
  p.n_seg = d->in.n_seg;
  fill_segs(ctx, d, true, p.seg, &p.M, &p.src_rows);
  p.Cred = d->cin; p.Nout = d->cout; p.w_tap_rows = d->cin;
  p.kh = d->kh; p.kw = d->kw;
  p.mul = d->stride; p.tsign = 1; p.off_y = -d->pad_t; p.off_x = -d->pad_l; p.div = 1;
  p.w_ty0 = 0; p.w_tx0 = 0; p.w_tstep = 1; p.w_kw = d->kw; p.w_taps = d->kh * d->kw;
  bool same = d->stride == 1 && d->kw == 3 && d->kh <= 3 && d->pad_l == 1;
  for (int i = 0; i < p.n_seg && same; ++i)
    same = p.seg[i].OH == p.seg[i].SH && p.seg[i].OW == p.seg[i].SW && p.seg[i].row_begin == p.seg[i].src_row_begin;
  const long long a_bytes = p.src_rows * (long long)p.ld_src * 3;
  const long long w_bytes = (long long)p.w_taps * d->cout * d->cin * 3;
  int max_sw = 0;
  for (int i = 0; i < p.n_seg; ++i) max_sw = p.seg[i].SW > max_sw ? p.seg[i].SW : max_sw;
  PP_CHECK_ARG(ctx, same && a_bytes < (1ll << 31) && w_bytes < (1ll << 31) && (long long)max_sw * p.ld_src * 3 < (1ll << 23), PP_ERR_SHAPE,
               "pp_conv2d_nhwc_fwd_f16c8: (prototype) 3-wide stride-1 'same' convolutions below 2 GiB only");
  static const int db = []() { const char* e = getenv("PP_CONV2_DB"); return e ? atoi(e) : 1; }();
  static const int wr = []() { const char* e = getenv("PP_CONV2_WR"); return e ? atoi(e) : 2; }();
  auto launch = [&](auto kern, int bm, int bn, int threads, size_t smem_bytes) {
    p.n_tiles_n = (p.Nout + bn - 1) / bn;
    const int n_tiles_mx = (p.M + bm - 3) / (bm - 2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes);
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_tiles_mx * p.n_tiles_n)), dim3(threads), smem_bytes, ctx->stream, p, x_hl, (unsigned)a_bytes, w_hl,
                       (unsigned)w_bytes, bias, y, d->cout, d->cin / 64);
  };
  auto smem_of = [](int bm, int bn, int nb) { return (size_t)((12 * (bm + 2) + 12) + nb * (12 * (bn + 2) + 12)) * 16; };
  if (wr == 4) launch(igemm2x_kernel<2, 2, 1, 4>, 256, 128, 512, smem_of(256, 128, 2));
  else if (db) launch(igemm2x_kernel<2, 2, 1, 2>, 128, 128, 256, smem_of(128, 128, 2));
  else launch(igemm2x_kernel<2, 2, 0, 2>, 128, 128, 256, smem_of(128, 128, 1));
  PP_CHECK_LAUNCH(ctx, "pp_conv2d_nhwc_fwd_f16c8");
  return PP_OK;
}
