// Weight-gradient GEMM on the CDNA4 bf16 matrix cores, fp32-accurate ("bf16x6", see gemm_x6.hip for the arithmetic).
//
//   paths_gemm_tn_x6    C[N1,N2] (+)= A[M,N1]^T * [B0 | B1][M,N2]          same contract as paths_gemm_tn_f32 (gemm_bwd.hip)
//
// It replaces aten::mm(dY^T, X) of the reference's loss.backward() (train.py:65) for every nn.Linear weight on the path; the
// gate matrix of the LSTM cell alone (dW[1792, 2048] over M = 16,384 rows) is 108 GFLOP per level.
//
// TN mapping.  Both operands are row-major with the reduction index m as the slow dimension, so a k16 stage is 16 consecutive
// rows of each, loaded with 16-byte coalesced reads ALONG n.  Every value is split in registers into its three bf16 planes
// (hi | mid | lo, exact) and written to LDS as [plane][m][n] 16-bit rows; the MFMA operand of v_mfma_f32_32x32x16_bf16 is "8
// consecutive m of one n per lane" for A^T and for B alike, which is exactly what ds_read_b64_tr_b16 delivers from such rows
// (two transposed reads per fragment, no shuffles, no second copy).  The image is the 256-byte-row XOR layout whose transposed
// reads are bank-conflict free: off(row, chunk) = 256 row + 16 (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))).
//
// One workgroup = 4 waves (2 x 2), wave tile 32 WT1 x 32 WT2, workgroup tile 64 WT1 x 64 WT2 (256 x 256 for the big products:
// 64 flop per operand byte, the 128 x 128 tile of the fp32-MFMA kernel would need > 9 TB/s of L2 at this MFMA rate).  k16
// stages are double-buffered in LDS with one barrier per stage; a thread's fp32 chunk of stage kt+2 is loaded right after its
// registers were split into stage kt+1's LDS buffer (a full stage of latency cover), and the split runs as 1-2 instruction
// micro-steps pinned into the gaps between MFMAs.  The reduction over M is split across workgroups; every split writes its own
// fp32 slab and reduce_slabs adds them in a fixed order (deterministic, rank-count independent: no float atomics).
#include <type_traits>

#include "common.h"
#include "reduce.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32: round to nearest even
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

constexpr int TK = 16;                 // rows of the reduction dimension per stage
constexpr int IMG = TK * 256;          // bytes of one [16 m][128 n] 16-bit image

struct TnX6Operands {
  const float* A; int64_t lda; uint32_t a_bytes;          // [M, N1]; *_bytes = extent the buffer descriptor may read
  const float* B0; int64_t ldb0; int NB0; uint32_t b0_bytes;
  const float* B1; int64_t ldb1; uint32_t b1_bytes;       // columns [NB0, N2) (may be null)
  int M, N1, N2;
  float* slabs; int64_t ld_out;        // [splits][N1][ld_out]  (or the output itself when there is one split)
  int rows_per_split;                  // multiple of 2 TK
  int splits;
};

// PL = bf16 planes per operand: 3 (hi | mid | lo, six partial products: exact fp32 products) or 2 (hi | mid, three partial products
// hi*hi + hi*mid + mid*hi: 16 significant bits per operand at fp32's exponent range, relative error of a product ~2e-5; half the
// MFMAs, 7 instead of 12 split micro-steps, two thirds of the LDS traffic - the default of the training step since round 3)
template <int WT1, int WT2, int OCC, int PL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
gemm_tn_x6_kernel(TnX6Operands g) {
  static_assert(PL == 2 || PL == 3, "two or three bf16 planes");
  constexpr int NPRD = PL == 3 ? 6 : 3;                   // partial products per operand pair
  constexpr int B1 = 64 * WT1, B2 = 64 * WT2;             // workgroup tile
  constexpr int NIA = (B1 + 127) / 128, NIB = (B2 + 127) / 128;   // 128-column images per plane
  constexpr int OPA = PL * NIA * IMG, STAGE = PL * (NIA + NIB) * IMG;
  constexpr int NCA = WT1, NCB = WT2, NC = NCA + NCB;     // fp32 chunks (4 floats) per thread per stage
  constexpr int RG = WT2 * NPRD;                          // MFMA gaps per accumulator row
  constexpr int NG = WT1 * RG, AG = (WT1 - 1) * RG;       // gaps per stage / before the barrier
  constexpr int NS = PL == 3 ? 12 : 7;                    // split micro-steps per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;

  // XCD-aware order (speed only): XCD x (= linear id % 8) owns a contiguous run of the (split, tile) sequence, tile fastest: the
  // workgroups that share one split's rows of A and B run on one XCD and re-use them from its private L2.
  const int nt2 = g.N2 / B2, ntiles = (g.N1 / B1) * nt2, nblk = ntiles * g.splits;
  int lin = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = lin & 7, j = lin >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int split = lin / ntiles, tile = lin - split * ntiles;
  const int n1_0 = (tile / nt2) * B1, n2_0 = (tile % nt2) * B2;
  const int m_begin = split * g.rows_per_split;
  const int rows = min(g.M, m_begin + g.rows_per_split) - m_begin;
  const int nk = ((rows + TK - 1) / TK + 1) & ~1;        // even; a stage past the end reads rows >= M = zeros (buffer range check)

  const bool second = g.B1 != nullptr && n2_0 >= g.NB0;   // the B column block comes from one panel (NB0 % B2 == 0)
  const float* Bp = second ? g.B1 : g.B0;
  const int64_t ldb = second ? g.ldb1 : g.ldb0;
  const int bcol0 = second ? n2_0 - g.NB0 : n2_0;
  // descriptors with the true extents: a read of a row >= M returns zeros (no clamping, no select in the loop)
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, g.a_bytes, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bp), 0, second ? g.b1_bytes : g.b0_bytes, 0x00020000);

  // ---- staging: chunk q of an operand = float4 number q*256 + tid of the stage's [16][B] fp32 tile
  uint32_t goff[NC]; int lwr[NC];      // per-lane byte offset at stage 0 of this split; LDS byte offset inside a stage buffer
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    const bool isA = q < NCA;
    const int Bw = isA ? B1 : B2, qq = isA ? q : q - NCA;
    const int idx = qq * 256 + tid, row = idx / (Bw / 4), c4 = idx % (Bw / 4);
    const int64_t ld = isA ? g.lda : ldb;
    goff[q] = (uint32_t)((((int64_t)(m_begin + row)) * ld + (isA ? n1_0 : bcol0) + 4 * c4) * 4);
    const int col = 4 * c4, im = col >> 7, ch = (col & 127) >> 3, half = (col >> 2) & 1;
    const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
    lwr[q] = (isA ? 0 : OPA) + im * IMG + 256 * row + 16 * (ch ^ sw) + 8 * half;
  }
  const uint32_t gstepA = (uint32_t)(TK * g.lda * 4), gstepB = (uint32_t)(TK * ldb * 4);
  f32x4 sa[NC];
  uint32_t hi[2], mid[2], lo[2];
  float tf[2];
  auto gload = [&](int q, int kt) {
    const bool isA = q < NCA;
    const uint32_t off = goff[q] + (uint32_t)kt * (isA ? gstepA : gstepB);
    sa[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(isA ? rsA : rsB, (int)off, 0, 0));
  };
  // split of one chunk in NS micro-steps of at most 2 VALU instructions; the last one is the three LDS writes
  auto a_step = [&](int q, int st, int buf) __attribute__((always_inline)) {
    f32x4& v = sa[q];
    const int ps = (q < NCA ? NIA : NIB) * IMG;           // plane stride of this operand
    if (st == 0) { hi[0] = pk_bf16(v[0], v[1]); hi[1] = pk_bf16(v[2], v[3]); }
    if (st == 1) { tf[0] = bf_lo(hi[0]); tf[1] = bf_hi(hi[0]); }
    if (st == 2) { v[0] -= tf[0]; v[1] -= tf[1]; }
    if (st == 3) { tf[0] = bf_lo(hi[1]); tf[1] = bf_hi(hi[1]); }
    if (st == 4) { v[2] -= tf[0]; v[3] -= tf[1]; }
    if (st == 5) { mid[0] = pk_bf16(v[0], v[1]); mid[1] = pk_bf16(v[2], v[3]); }
    if constexpr (PL == 2) {
      if (st == 6) {
        char* d = smem + buf * STAGE + lwr[q];
        *reinterpret_cast<u32x2*>(d) = u32x2{hi[0], hi[1]};
        *reinterpret_cast<u32x2*>(d + ps) = u32x2{mid[0], mid[1]};
      }
    } else {
      if (st == 6) { tf[0] = bf_lo(mid[0]); tf[1] = bf_hi(mid[0]); }
      if (st == 7) { v[0] -= tf[0]; v[1] -= tf[1]; }
      if (st == 8) { tf[0] = bf_lo(mid[1]); tf[1] = bf_hi(mid[1]); }
      if (st == 9) { v[2] -= tf[0]; v[3] -= tf[1]; }
      if (st == 10) { lo[0] = pk_bf16(v[0], v[1]); lo[1] = pk_bf16(v[2], v[3]); }
      if (st == 11) {
        char* d = smem + buf * STAGE + lwr[q];
        *reinterpret_cast<u32x2*>(d) = u32x2{hi[0], hi[1]};
        *reinterpret_cast<u32x2*>(d + ps) = u32x2{mid[0], mid[1]};
        *reinterpret_cast<u32x2*>(d + 2 * ps) = u32x2{lo[0], lo[1]};
      }
    }
  };

  // ---- fragment reads.  Transposed read t (0, 1) of the 32-column fragment starting at column c of an operand: the 16-lane group
  // gi takes the block rows 8 (gi >> 1) + 4 t .. + 3, columns c + 16 (gi & 1) .. + 15; lane 4 q + p of the group supplies the
  // address of row q, columns 4 p .. 4 p + 3 of the block and receives column (lane & 15), the four rows in its four elements.
  uint32_t fra[WT1][2], frb[WT2][2];
  {
    const int gi = lane >> 4, li = lane & 15, qr = li >> 2, pp = li & 3;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 8 * (gi >> 1) + 4 * t + qr;
      const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
      for (int i = 0; i < WT1; ++i) {
        const int col = wm * 32 * WT1 + 32 * i + 16 * (gi & 1), im = col >> 7, ch = ((col & 127) >> 3) + (pp >> 1);
        fra[i][t] = im * IMG + 256 * row + 16 * (ch ^ sw) + 8 * (pp & 1);
      }
#pragma unroll
      for (int j = 0; j < WT2; ++j) {
        const int col = wn * 32 * WT2 + 32 * j + 16 * (gi & 1), im = col >> 7, ch = ((col & 127) >> 3) + (pp >> 1);
        frb[j][t] = OPA + im * IMG + 256 * row + 16 * (ch ^ sw) + 8 * (pp & 1);
      }
    }
  }
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  auto tr_read = [&](uint32_t byte_off) -> u32x2 {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(smem + byte_off)));
  };
  u32x4 fa[2][PL], fb[2][WT2][PL];
  auto read_a = [&](int buf, int i, int slot, int p) {
    const u32x2 x = tr_read(buf * STAGE + p * NIA * IMG + fra[i][0]), y = tr_read(buf * STAGE + p * NIA * IMG + fra[i][1]);
    fa[slot][p] = u32x4{x[0], x[1], y[0], y[1]};
  };
  auto read_b = [&](int buf, int j, int slot, int p) {
    const u32x2 x = tr_read(buf * STAGE + p * NIB * IMG + frb[j][0]), y = tr_read(buf * STAGE + p * NIB * IMG + frb[j][1]);
    fb[slot][j][p] = u32x4{x[0], x[1], y[0], y[1]};
  };

  f32x16 acc[WT1][WT2];
#pragma unroll
  for (int i = 0; i < WT1; ++i)
#pragma unroll
    for (int j = 0; j < WT2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto one_mfma = [&](int gq, int sb) __attribute__((always_inline)) {
    const int i = gq / RG, j = (gq % RG) / NPRD, t = gq % NPRD, sl = i & 1;
    // smallest partial products first.  PL == 3: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi;  PL == 2: mid*hi, hi*mid, hi*hi
    constexpr int PA3[6] = {2, 0, 1, 1, 0, 0}, PB3[6] = {0, 2, 1, 0, 1, 0}, PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};
    const int pa = PL == 3 ? PA3[t] : PA2[t % 3], pb = PL == 3 ? PB3[t] : PB2[t % 3];
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[sl][pa]), __builtin_bit_cast(bf16x8, fb[sb][j][pb]), acc[i][j], 0, 0, 0);
  };

  // staging slots: chunk q owns GPC consecutive gaps (SPG micro-steps each); its reload with stage kt+2 shares the last one
  constexpr auto fits = [](int spg) constexpr { return ((NS + spg - 1) / spg) * NC <= AG; };
  constexpr int SPG = fits(1) ? 1 : fits(2) ? 2 : fits(3) ? 3 : fits(4) ? 4 : fits(6) ? 6 : NS;   // (128 x 128 tiles at two planes: few gaps)
  static_assert(fits(SPG), "staging does not fit before the barrier");
  constexpr int GPC = (NS + SPG - 1) / SPG;
  constexpr int AR0 = RG / 2;                            // gaps AR0 .. AR0+PL-1 of row i: fragment reads of A row i+1
  auto staging_slot = [&](auto sc, int kt, auto bufc, auto m1c, auto m2c) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value, buf = decltype(bufc)::value;
    constexpr bool more1 = decltype(m1c)::value, more2 = decltype(m2c)::value;
    if constexpr (s < GPC * NC) {
      constexpr int q = s / GPC, g0 = s % GPC;
      if constexpr (more1) {
        static_for<g0 * SPG, (g0 + 1) * SPG < NS ? (g0 + 1) * SPG : NS>([&](auto mc) __attribute__((always_inline)) {
          a_step(q, decltype(mc)::value, buf ^ 1);
        });
      }
      if constexpr (more2 && g0 == GPC - 1) gload(q, kt + 2);
    }
  };
  auto stage_body = [&](int kt, auto bufc, auto m1c, auto m2c) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value, sb = buf;
    constexpr bool more1 = decltype(m1c)::value;
    static_for<0, AG>([&](auto gc) __attribute__((always_inline)) {
      constexpr int gq = decltype(gc)::value, i = gq / RG, gr = gq % RG;
      one_mfma(gq, sb);
      if constexpr (gr >= AR0 && gr < AR0 + PL) read_a(buf, i + 1, (i + 1) & 1, gr - AR0);
      staging_slot(std::integral_constant<int, gq>{}, kt, bufc, m1c, m2c);
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    static_for<AG, NG>([&](auto gc) __attribute__((always_inline)) {
      constexpr int gq = decltype(gc)::value, gr = gq % RG;
      one_mfma(gq, sb);
      if constexpr (more1) {
        static_for<2 * gr, 2 * gr + 2>([&](auto fc) __attribute__((always_inline)) {
          constexpr int f = decltype(fc)::value;
          if constexpr (f < PL) read_a(buf ^ 1, 0, 0, f);
          else if constexpr (f < PL + PL * WT2) read_b(buf ^ 1, (f - PL) / PL, sb ^ 1, (f - PL) % PL);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  static_assert(2 * RG >= PL + PL * WT2, "next stage's first fragments do not fit behind the barrier");
  constexpr std::integral_constant<int, 0> I0{};
  constexpr std::integral_constant<int, 1> I1{};
  constexpr std::true_type T{};
  constexpr std::false_type F{};

  // prologue: stage 0 -> LDS buffer 0, stage 1 -> registers
#pragma unroll
  for (int q = 0; q < NC; ++q) gload(q, 0);
#pragma unroll
  for (int q = 0; q < NC; ++q) {
#pragma unroll
    for (int st = 0; st < NS; ++st) a_step(q, st, 0);
    gload(q, 1);
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < PL; ++p) read_a(0, 0, 0, p);
#pragma unroll
  for (int j = 0; j < WT2; ++j)
#pragma unroll
    for (int p = 0; p < PL; ++p) read_b(0, j, 0, p);
  for (int kt = 0; kt < nk - 2; kt += 2) {
    __builtin_amdgcn_sched_barrier(0);
    stage_body(kt, I0, T, T);
    stage_body(kt + 1, I1, T, T);
  }
  stage_body(nk - 2, I0, T, F);
  stage_body(nk - 1, I1, F, F);

  float* out = g.slabs + (int64_t)split * g.N1 * g.ld_out;
#pragma unroll
  for (int i = 0; i < WT1; ++i)
#pragma unroll
    for (int j = 0; j < WT2; ++j) {
      const int col = n2_0 + wn * 32 * WT2 + 32 * j + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n1_0 + wm * 32 * WT1 + 32 * i + c32_row(r, lane);
        out[(int64_t)row * g.ld_out + col] = acc[i][j][r];
      }
    }
}

// out[i] (+)= sum_s slabs[s][i]   (fixed order: deterministic; 8 loads in flight per thread)
__global__ void __launch_bounds__(256)
reduce_slabs_x6_kernel(const float* __restrict__ slabs, int splits, int64_t n, float* __restrict__ out, int64_t ldo, int ncols,
                       int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= splits; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] += slabs[(int64_t)(k + u) * n + i];
  }
  for (; k < splits; ++k) p[k & 7] += slabs[(int64_t)k * n + i];
  const float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  const int64_t r = i / ncols, c = i % ncols;
  float* o = out + r * ldo + c;
  *o = accumulate ? *o + s : s;
}

template <int WT1, int WT2, int PL>
int launch_tn(const TnX6Operands& g, hipStream_t stream) {
  constexpr int lds = 2 * PL * ((64 * WT1 + 127) / 128 + (64 * WT2 + 127) / 128) * IMG;
  auto kern = gemm_tn_x6_kernel<WT1, WT2, (WT1 * WT2 < 16 ? 2 : 1), PL>;
  PATHS_LDS_OPT_IN(kern, lds, "gemm_tn_x6");
  const int nblk = (g.N1 / (64 * WT1)) * (g.N2 / (64 * WT2)) * g.splits;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, stream, g);
  return PATHS_OK;
}

}  // namespace

extern "C" {

// Same arguments as paths_gemm_tn_f32 + planes: 3 = three bf16 planes per operand (exact fp32 products), 2 = two (16 significant bits
// per operand, half the MFMAs).  `splits` is an upper bound (the kernel needs >= 32 rows per split); `workspace` holds
// paths_gemm_tn_workspace(N1, N2, splits) floats.
int paths_gemm_tn_x6(const float* a, int64_t lda, const float* b0, int64_t ldb0, int nb0, const float* b1, int64_t ldb1,
                     float* out, int64_t ldo, int M, int N1, int N2, int splits, int accumulate, float* workspace, int planes,
                     hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && N1 > 0 && N2 > 0 && splits > 0 && a && b0 && out && workspace, "gemm_tn_x6: bad arguments");
  PATHS_REQUIRE(planes == 2 || planes == 3, "gemm_tn_x6: planes must be 2 or 3 (bf16 planes per operand)");
  PATHS_REQUIRE(N1 % 128 == 0 && N2 % 128 == 0, "gemm_tn_x6: N1 (%d) and N2 (%d) must be multiples of 128", N1, N2);
  PATHS_REQUIRE(lda % 4 == 0 && ldb0 % 4 == 0 && (b1 == nullptr || ldb1 % 4 == 0), "gemm_tn_x6: leading dimensions must be multiples of 4");
  PATHS_REQUIRE(b1 == nullptr || (nb0 % 128 == 0 && nb0 > 0 && nb0 < N2), "gemm_tn_x6: panel split must be a multiple of 128");
  PATHS_REQUIRE(((uintptr_t)a | (uintptr_t)b0 | (uintptr_t)b1) % 16 == 0, "gemm_tn_x6: operands must be 16-byte aligned");
  const int nb0e = b1 ? nb0 : N2;
  const int64_t a_bytes = ((int64_t)(M - 1) * lda + N1) * 4, b0_bytes = ((int64_t)(M - 1) * ldb0 + nb0e) * 4;
  const int64_t b1_bytes = b1 ? ((int64_t)(M - 1) * ldb1 + (N2 - nb0)) * 4 : 0;
  // 32-bit buffer offsets; two stages of read-ahead past the last row must not wrap
  const int64_t lim = (int64_t)0xFFFFFFFF - 64ll * 4 * (lda > ldb0 ? (lda > ldb1 ? lda : ldb1) : (ldb0 > ldb1 ? ldb0 : ldb1));
  PATHS_REQUIRE(a_bytes < lim && b0_bytes < lim && b1_bytes < lim, "gemm_tn_x6: operand larger than the 32-bit buffer offsets cover");
  int rps = (M + splits - 1) / splits;
  rps = (rps + 2 * TK - 1) / (2 * TK) * (2 * TK);
  const int nsplit = (M + rps - 1) / rps;
  const bool big = N1 % 256 == 0 && N2 % 256 == 0 && (b1 == nullptr || nb0 % 256 == 0);
  const bool direct = nsplit == 1 && !accumulate;
  TnX6Operands g{a, lda, (uint32_t)a_bytes, b0, ldb0, nb0e, (uint32_t)b0_bytes, b1, ldb1, (uint32_t)b1_bytes, M, N1, N2,
                 direct ? out : workspace, direct ? ldo : (int64_t)N2, rps, nsplit};
  const int rc_ = planes == 3 ? (big ? launch_tn<4, 4, 3>(g, stream) : launch_tn<2, 2, 3>(g, stream))
                              : (big ? launch_tn<4, 4, 2>(g, stream) : launch_tn<2, 2, 2>(g, stream));
  if (rc_ != PATHS_OK) return rc_;
  PATHS_LAUNCH_CHECK("gemm_tn_x6");
  if (!direct) {
    const int64_t n = (int64_t)N1 * N2;
    if (const int rd = paths_reduce_try_defer(workspace, nsplit, n, out, ldo, N2, accumulate, 0, stream)) return rd == PATHS_DEFERRED ? PATHS_OK : rd;
    hipLaunchKernelGGL(reduce_slabs_x6_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, workspace, nsplit, n, out, ldo, N2, accumulate);
    PATHS_LAUNCH_CHECK("gemm_tn_x6(reduce)");
  }
  return PATHS_OK;
}

}  // extern "C"
