// Fused masked self-attention on the bf16 matrix cores with fp32 accuracy (the "x6" operand split of gemm_x6.hip).
//
// Same contract as paths_attention_f32 (attn_f32.hip; reference model/aggregator.py:70-72 + utils.py:97-103): q, k, v
// head-major [B][h][T][32] fp32, q pre-scaled by log2(e)/sqrt(hd), o token-major [B][T][h*32], keys >= num_ims[b]+1 masked.
//
// Two launches:
//  1. attn_x6_prep_kernel: every fp32 value of q, k, v becomes three bf16 (hi + mid + lo = the value, exactly), stored as
//     MFMA FRAGMENTS: one 16-row x 32-k operand of v_mfma_f32_16x16x32_bf16 for one plane is 1 KiB, lane l owns bytes
//     [16 l, 16 l + 16) = its 8 k-values.  Q and K fragments: rows = tokens, k = the 32 head dims.  V fragments are
//     TRANSPOSED (rows = 16 head dims, k = 32 keys) and key-PERMUTED, k-slot (g, j) <-> key 4g + (j&3) + 16 (j>>2), which is
//     the order in which the S^T accumulators of two 16-key tiles sit in a lane - so P never leaves the registers.
//     Keys that are masked (or beyond T) are written as zeros.
//  2. attn_x6_kernel: one wave = 32 queries (two 16-query tiles) of one (slide, head); a 4-wave workgroup shares 64-key
//     K / V^T fragment sets through LDS (plain 16-byte copies, double-buffered).  Per 32-key group and wave:
//        S^T[key][q] = K Q^T          2 key tiles x 2 query tiles x 6 MFMAs (A = K fragment, B = Q fragment in registers)
//        online softmax               per lane: one query of each query tile, keys in registers, 2 cross-lane swaps
//        P^T split in registers       3 x bf16x8 per query tile
//        O^T[dv][q] += V^T P^T        2 dv tiles x 2 query tiles x 6 MFMAs
//     48 MFMAs of 16 cycles against 128 of 32 cycles (v_mfma_f32_16x16x4_f32) in the f32 kernel for the same 32 x 32 block.
// The six kept partial products per operand pair are accumulated smallest first, exactly as in gemm_x6.hip.
#include "common.h"
#include "dropout.h"

DropSite paths_make_drop_site(uint64_t key, float p);      // dropout.hip
#if defined(PATHS_ATTN_STAMPS) && !defined(PATHS_ATTN_DEBUG)
#define PATHS_ATTN_DEBUG 1
#endif
#ifdef PATHS_M32P_STAMPS
unsigned long long* g_m32p_dbg = nullptr;     // diagnostic builds only (tools/attn_pair_stamps.py): segment / barrier cycles per wave
#define M32P_T(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define M32P_T(i) do { } while (0)
#endif
#ifdef PATHS_ATTN_DEBUG
unsigned long long* g_attn_dbg = nullptr;
#endif
// PATHS_ATTN_STAMPS = n (diagnostic builds only, tools/attn_stamps.py): s_memtime stamps between the phases of a key step, summed
// per wave 0 of every workgroup into the debug buffer (16 words per workgroup).  1: around the barrier only (the schedule of the
// step stays hipcc's), 2: every phase (pins the phases apart: shares, never quoted as run time).
#ifdef PATHS_ATTN_STAMPS
#define ATTN_STAMP(i, lvl) do { if (PATHS_ATTN_STAMPS >= (lvl)) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    st_acc[i] += (unsigned)(t_ - st_prev); st_prev = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define ATTN_STAMP(i, lvl) do { } while (0)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 32;
constexpr int FRAG = 1024;                 // bytes of one fragment (64 lanes x 8 bf16)
constexpr int KSTEP = 64;                  // keys staged per LDS buffer
// per 64 keys: K = 4 key tiles x NP planes, V^T = 2 key groups x 2 dv tiles x NP planes -> 8 NP fragments of 1 KiB
// NP = 3: bf16 hi|mid|lo, 6 MFMAs per product (x6);  NP = 2: fp16 hi|lo, 3 MFMAs (h3; q, k, v and P are O(1): no scaling)
template <int NP> constexpr int step_bytes() { return 8 * NP * FRAG; }

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

__device__ __forceinline__ uint32_t pk_f16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// 8 fp32 -> two planes of 8 fp16 (hi, lo): 22 significant bits
__device__ __forceinline__ void split8h(const float (&x)[8], u32x4& hi, u32x4& lo) {
  // residual + rounding of the lo plane: one v_fma_mixlo_f16 / v_fma_mixhi_f16 per value (common.h: f16_pair_residuals_pk), the four
  // low halves first: a half-register write directly in front of the other half's costs a wait state each (16 s_nop per key step)
#pragma unroll
  for (int i = 0; i < 4; ++i) hi[i] = pk_f16(x[2 * i], x[2 * i + 1]);
  uint32_t r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[i]) : "v"(hi[i]), "v"(x[2 * i]));
#pragma unroll
  for (int i = 0; i < 4; ++i) asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r[i]) : "v"(hi[i]), "v"(x[2 * i + 1]));
#pragma unroll
  for (int i = 0; i < 4; ++i) lo[i] = r[i];
}
// planes[0..NP) of 8 values
template <int NP>
__device__ __forceinline__ void split_planes(const float (&x)[8], u32x4 (&pl)[NP]);
// 8 fp32 -> three planes of 8 bf16 (hi, mid, lo), exact
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const uint32_t h = pk_bf16(a, b);
    const float ra = a - bf_lo(h), rb = b - bf_hi(h);
    const uint32_t m = pk_bf16(ra, rb);
    const float sa = ra - bf_lo(m), sb = rb - bf_hi(m);
    hi[i] = h; mid[i] = m; lo[i] = pk_bf16(sa, sb);
  }
}

template <> __device__ __forceinline__ void split_planes<3>(const float (&x)[8], u32x4 (&pl)[3]) { split8(x, pl[0], pl[1], pl[2]); }
template <> __device__ __forceinline__ void split_planes<2>(const float (&x)[8], u32x4 (&pl)[2]) { split8h(x, pl[0], pl[1]); }

// Fragment images (per (slide, head), Tp = T rounded up to 64):
//   Q6 / K6 : [Tp/16 tiles][3 planes][64 lanes][8 bf16]        lane (r = l&15, g = l>>4): token 16 tile + r, dims 8g .. 8g+7
//   V6      : [Tp/32 groups][2 dv tiles][3 planes][64 lanes][8] lane (dv = l&15, g):      dim 16 dvt + dv, keys 32 grp + kappa(g, j)
template <int NP>
__global__ void __launch_bounds__(256)
attn_x6_prep_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                    char* __restrict__ q6, char* __restrict__ k6, char* __restrict__ v6,
                    const int64_t* __restrict__ num_ims, int T, int Tp, int H, int nq) {
  __shared__ float sv[KSTEP][HD + 1];
  const int b = blockIdx.z, head = blockIdx.y, t0 = blockIdx.x * KSTEP;
  const int len = min((int)num_ims[b] + 1, T);
  const int tid = threadIdx.x;
  const int64_t base = ((int64_t)b * H + head) * T * HD;
  const int64_t ibase = ((int64_t)b * H + head) * (int64_t)Tp * HD * 2 * NP;     // bytes of one (slide, head) image
  // ---- Q and K: thread = (token t0 + tid/4, dims 8 (tid%4) ..)
  {
    const int tl = tid >> 2, g = tid & 3, tok = t0 + tl;
    float xq[8], xk[8];
    const bool kvalid = tok < len, qvalid = tok < T && tok < nq;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xk[i] = kvalid ? k[base + (int64_t)tok * HD + 8 * g + i] : 0.f;
      xq[i] = qvalid ? q[base + (int64_t)tok * HD + 8 * g + i] : 0.f;
    }
    u32x4 pl[NP];
    const int64_t off = ibase + ((int64_t)(tok >> 4) * NP) * FRAG + ((tok & 15) + 16 * g) * 16;
    split_planes<NP>(xk, pl);
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(k6 + off + p * FRAG) = pl[p];
    split_planes<NP>(xq, pl);
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(q6 + off + p * FRAG) = pl[p];
  }
  // ---- V^T: through LDS (coalesced rows in, transposed + key-permuted fragments out)
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int idx = tid + 256 * p, tl = idx >> 5, dcol = idx & 31, tok = t0 + tl;
    sv[tl][dcol] = tok < len ? v[base + (int64_t)tok * HD + dcol] : 0.f;
  }
  __syncthreads();
  {
    const int kg = tid >> 7, dvt = (tid >> 6) & 1, l = tid & 63, dv = l & 15, g = l >> 4;
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = sv[32 * kg + 4 * g + (j & 3) + 16 * (j >> 2)][16 * dvt + dv];
    u32x4 pl[NP];
    split_planes<NP>(xv, pl);
    const int64_t off = ibase + ((int64_t)(((t0 >> 5) + kg) * 2 + dvt) * NP) * FRAG + l * 16;
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(v6 + off + p * FRAG) = pl[p];
  }
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// acc += A * B from split operands, smallest partial products first
__device__ __forceinline__ f32x4 mfma_split(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4 c) {   // hi, mid, lo: six largest of nine
  c = mfma_bf16(a[2], b[0], c);
  c = mfma_bf16(a[0], b[2], c);
  c = mfma_bf16(a[1], b[1], c);
  c = mfma_bf16(a[1], b[0], c);
  c = mfma_bf16(a[0], b[1], c);
  c = mfma_bf16(a[0], b[0], c);
  return c;
}
__device__ __forceinline__ f32x4 mfma_split(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x4 c) {   // hi, lo: all but lo*lo
#if PATHS_ATTN_WHATIF & 4
  return mfma_f16(a[0], b[0], c);
#endif
  c = mfma_f16(a[1], b[0], c);
  c = mfma_f16(a[0], b[1], c);
  c = mfma_f16(a[0], b[0], c);
  return c;
}

__device__ __forceinline__ float rows_max(float x) { x = fmaxf(x, __shfl_xor(x, 16)); return fmaxf(x, __shfl_xor(x, 32)); }
__device__ __forceinline__ float rows_sum(float x) { x += __shfl_xor(x, 16); return x + __shfl_xor(x, 32); }

// DROP (training with dropout > 0, reference nn.MultiheadAttention(dropout=p)): the softmax probabilities that enter the PV product
// are multiplied by the regenerated mask / (1 - p) (element ((slide*H + head)*T + query)*T + key of the site, csrc/dropout.h); the
// normaliser l and the saved log-sum-exp stay those of the un-dropped softmax, as in the reference.
#ifndef PATHS_ATTN_OCC
#define PATHS_ATTN_OCC 2
#endif
// PATHS_ATTN_OCC = waves per SIMD the kernel is built for.  2: the software-pipelined loop (S of the next step computed before the
// softmax of this one; two score buffers).  3: one score buffer (32 registers less: fits 168), three workgroups per CU - the same
// work on two thirds of the CUs, which leaves more of the chip to the selection chain's GEMMs running beside it.
constexpr int ATTN_OCC = PATHS_ATTN_OCC;
#ifndef PATHS_ATTN_QT
#define PATHS_ATTN_QT 2
#endif
// QT = 16-query tiles per wave (a workgroup = 4 waves = 64 QT queries).  2: every K / V^T fragment read from LDS feeds two query
// tiles.  1 (with ATTN_OCC >= 3): twice the workgroups at half the registers - four or more waves per SIMD hide each other's
// LDS / softmax latencies (the loop is latency-bound at two), at twice the LDS fragment traffic per MFMA.
constexpr int QT = PATHS_ATTN_QT;
#ifndef PATHS_ATTN_P1
#define PATHS_ATTN_P1 0
#endif
constexpr bool ATTN_P1 = PATHS_ATTN_P1 != 0;      // P as one fp16 plane (see the kernel)
#ifndef PATHS_ATTN_DEFER
#define PATHS_ATTN_DEFER 8
#endif
constexpr float ATTN_DEFER = (float)(PATHS_ATTN_DEFER);   // deferred-rescale threshold in log2 units (0 = rescale every step)
#ifndef PATHS_ATTN_WHATIF
#define PATHS_ATTN_WHATIF 0
#endif
// PATHS_ATTN_WHATIF (diagnostic builds, WRONG results, tools/attn_time.py): 1 no exp2, 2 no lo plane of P, 4 one MFMA per product
// block, 8 no PV products, 16 no score products, 32 no LDS fragment reads (one fragment set re-used), 64 no staging (loads, LDS writes)
constexpr int WHATIF = PATHS_ATTN_WHATIF;
#ifndef PATHS_ATTN_DELAY_PV
#define PATHS_ATTN_DELAY_PV 0
#endif
// DELAY_PV (FAST path): the PV product of key step k is issued in step k+1, in the same basic block as the score product of step k+2
// and the exp2 / split work of step k+1 - every MFMA of the loop then has vector work to hide behind (before, the 24 PV MFMAs of a
// step ran bare, after the step's vector work: the probabilities they multiply did not exist earlier).  V^T tiles live one step
// longer (ring of three LDS buffers instead of two); a revision of the running maximum in step k+1 happens AFTER the pending product
// was added, so one rescale covers it.  MEASURED (round 4) and OFF: hipcc does interleave the 48 MFMAs with ~100 vector instructions
// then, results are bit-identical, and the kernel takes 57.6 us against 56.6 (A/B/A/B on one box) - the SIMD's aggregate issue
// capacity, not the order inside one wave, is what bounds the step (tools/simd_probe.hip).
constexpr bool ATTN_DELAY_PV = PATHS_ATTN_DELAY_PV != 0;
#ifndef PATHS_ATTN_FAST
#define PATHS_ATTN_FAST 1
#endif
// FAST (two fp16 planes, no dropout): the running maximum is SUBTRACTED INSIDE the score product (the accumulators of S^T = K Q^T
// start at -m_run instead of 0) and is only revised when a probability sum says it has to be: the common key step has no maximum
// chain, no subtraction, no vote on the scores - exp2, the sum, the fp16 hi | lo split and nothing else (VALU instructions per
// score element 6.7 -> ~4.5; the kernel is VALU-issue bound: round-4 measurement).
constexpr bool ATTN_FAST = PATHS_ATTN_FAST != 0 && !ATTN_P1;
template <int NP, bool DROP>
__global__ void __launch_bounds__(256, ATTN_OCC)
attn_x6_kernel(const char* __restrict__ q6, const char* __restrict__ k6, const char* __restrict__ v6,
               float* __restrict__ o, float* __restrict__ lse, const int64_t* __restrict__ num_ims, int T, int Tp, int H,
               int npairs_arg, int nqb_arg, DropSite drop, char* __restrict__ o_img
#ifdef PATHS_ATTN_DEBUG
               , unsigned long long* dbg
#endif
               ) {
  constexpr int STEP_BYTES = step_bytes<NP>();
#ifdef PATHS_ATTN_DEBUG
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];          // 2 x STEP_BYTES (+ occupancy padding, see the launcher)
  // XCD-aware placement (speed only): every workgroup of one (slide, head) pair streams that pair's whole K / V^T images (0.5 MB
  // at T = 2049), and blocks are dealt round-robin over the 8 XCDs.  With the query block as the fastest grid index each XCD's
  // 4 MiB L2 saw ALL pairs (8.4 MB at 8 slides): the images were re-fetched ~16 x from beyond L2 (133 MB per launch, measured)
  // and a key step took 1.5 x as long at 8 slides as at 2.  Here pair p only ever runs on the XCD group p % 8: linear id ->
  // (group x = id % 8, j = id / 8), the group's pairs are x, x + 8, ..., pair index fastest within the group.
  const int npairs = npairs_arg, nqb = nqb_arg;         // grid.x = 8 * ceil(npairs / 8) * nqb
  const int lin = blockIdx.x, xg = lin & 7, jx = lin >> 3;
  const int cnt = (npairs - xg + 7) >> 3;               // pairs of this group
  if (cnt <= 0) return;
  const int pair = xg + 8 * (jx % cnt), qb = jx / cnt;
  if (qb >= nqb) return;
  const int b = pair / H, head = pair - b * H, q0 = qb * 64 * QT;
  const int len = min((int)num_ims[b] + 1, T);          // valid keys = special token + patches
  const DropWin dwin = drop_window(drop, drop_attn_row((uint64_t)pair, T, 0));      // (this pair's T x T' mask elements: csrc/dropout.h)
  if (q0 >= len) return;                                // every query of this block is padding
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 15, g4 = lane >> 4;
  const int64_t ibase = ((int64_t)b * H + head) * (int64_t)Tp * HD * 2 * NP;
  const int qw = q0 + wave * 16 * QT;                   // this wave's first query

  // Q fragments (B operand of S^T): two 16-query tiles x 3 planes, kept in registers
  u32x4 qf[QT][NP];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int p = 0; p < NP; ++p)
      qf[qt][p] = *reinterpret_cast<const u32x4*>(q6 + ibase + ((int64_t)(min(qw + 16 * qt, Tp - 16) >> 4) * NP + p) * FRAG + lane * 16);

  f32x4 oacc[2][QT];                                    // [dv tile][query tile]: rows = dims 4 g4 .. +3, col = query ql
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < QT; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr bool FAST = ATTN_FAST && NP == 2 && !DROP;
  float m_run[QT], l_run[QT];
  f32x4 negm[QT];                                       // FAST: -m_run in all four elements = the C operand of the score products
#pragma unroll
  for (int j = 0; j < QT; ++j) { m_run[j] = FAST ? 0.f : -INFINITY; l_run[j] = 0.f; negm[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // staging: one 64-key step = 4 NP KiB of K fragments + 4 NP KiB of V^T fragments, both contiguous in their images.
  // Software pipeline: S(k+1) = K(k+1) Q^T is issued BEFORE the softmax of S(k), so one wave's MFMAs run under its own VALU work
  // (as one block, QK^T -> softmax -> PV, a wave ran them strictly one after the other and only the SIMD's second wave overlapped
  // them).  K therefore runs one step ahead of V in LDS: K(k+1), K(k+2 being written) | V(k), V(k+1 being written).
  const int nkt = (len + KSTEP - 1) / KSTEP;
  constexpr int HALF = 4 * NP * FRAG;                   // bytes of the K (or V) fragments of one step
  constexpr bool FASTC = ATTN_FAST && NP == 2 && !DROP;
  constexpr bool DELAY = FASTC && ATTN_DELAY_PV && ATTN_OCC < 3;
  constexpr int NVB = DELAY ? 3 : 2;                    // V^T buffers (DELAY: a tile is read one step after its scores)
  auto vbuf = [&](int kt) { return DELAY ? kt % 3 : (kt & 1); };
  char* const sKb = smem_raw;                           // [2][HALF]
  char* const sVb = smem_raw + 2 * HALF;                // [NVB][HALF]
  u32x4 st[2 * NP];
  auto gload_k = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) st[i] = *reinterpret_cast<const u32x4*>(k6 + ibase + (int64_t)kt * HALF + (tid + 256 * i) * 16);
  };
  auto gload_v = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) st[NP + i] = *reinterpret_cast<const u32x4*>(v6 + ibase + (int64_t)kt * HALF + (tid + 256 * i) * 16);
  };
  auto swrite_k = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) *reinterpret_cast<u32x4*>(sKb + (kt & 1) * HALF + (tid + 256 * i) * 16) = st[i];
  };
  auto swrite_v = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) *reinterpret_cast<u32x4*>(sVb + vbuf(kt) * HALF + (tid + 256 * i) * 16) = st[NP + i];
  };
  auto qk = [&](int kt, f32x4 (&s)[QT][4]) __attribute__((always_inline)) {      // S^T = K Q^T for the 4 key tiles of step kt
    const char* sK = sKb + (kt & 1) * HALF + lane * 16;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      u32x4 kf[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) kf[p] = *reinterpret_cast<const u32x4*>(sK + (((WHATIF & 32) ? 0 : t) * NP + p) * FRAG);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        if constexpr ((WHATIF & 16) != 0) { s[qt][t] = negm[qt]; asm("" : "+v"(s[qt][t])); }
        else s[qt][t] = mfma_split(kf, qf[qt], negm[qt]);      // (zero outside FAST)
      }
    }
  };
  gload_k(0); gload_v(0);
  swrite_k(0); swrite_v(0);
  if (nkt > 1) { gload_k(1); swrite_k(1); }
  __syncthreads();
  f32x4 sA[QT][4], sB[QT][4];                             // [query tile][key tile]: rows = keys 4 g4 .. +3, col = query ql
  if constexpr (ATTN_OCC < 3) qk(0, sA);
  // one step: s = S(kt) (ready), sn receives S(kt+1)
  // Fair share for the SECOND workgroup of a CU.  Issue arbitration on a SIMD is oldest-first: measured per workgroup
  // (tools/attn_wg_times.py), the 256 first-dispatched workgroups ran 46 us and their 224 younger CU mates 58 us - the last
  // 12 us with a single workgroup per CU.  Alternating the priority between the two every few key steps lets both finish together.
#ifndef PATHS_ATTN_PRIO_SHIFT
#define PATHS_ATTN_PRIO_SHIFT 2
#endif
  const int prio_parity = (qb * npairs + pair) >= 256 ? 1 : 0;
  auto step = [&](int kt, f32x4 (&s)[QT][4], f32x4 (&sn)[QT][4], auto lastc) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(lastc)::value;       // the masked step is peeled: 26 selects per step otherwise
    if constexpr (PATHS_ATTN_PRIO_SHIFT >= 0) {
      if ((((kt >> PATHS_ATTN_PRIO_SHIFT) ^ prio_parity) & 1) != 0) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    if (kt + 2 < nkt) gload_k(kt + 2);
    if (kt + 1 < nkt) gload_v(kt + 1);
    if constexpr (ATTN_OCC >= 3) qk(kt, s);
    const char* sV = sVb + (kt & 1) * HALF + lane * 16;
    // ---- mask (last step only) + online softmax (lane: query ql of each tile; keys 16 t + 4 g4 + r)
    if constexpr (LAST) {
      const int kbase = kt * KSTEP + 4 * g4;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kbase + 16 * t + r >= len) s[qt][t][r] = -INFINITY;
    }
    if constexpr (ATTN_OCC < 3) qk(kt + 1, sn);         // (past the end: stale K fragments, finite garbage nobody reads)
    // P1 (two-plane mode, no dropout): P enters the PV product as ONE fp16 plane (P^ = fp16(P), 11 bits) against V hi | lo: two
    // MFMAs per block instead of three and no residual plane to build (1.5 of ~4.5 VALU per score element).  The normaliser sums
    // the SAME rounded values (v_fma_mix reads them out of the packed register), so the result is an exact softmax-weighted mean
    // with weights p^_k / sum p^: rounding perturbs each weight by <= 2^-12 relative and the perturbations largely cancel between
    // numerator and denominator (error ~ 2^-12 |v - o| / sqrt(effective keys)).
    constexpr bool P1 = ATTN_P1 && NP == 2 && !DROP;
    constexpr int NPP = P1 ? 1 : NP;
    u32x4 pf[QT][2][NPP];                               // [query tile][32-key group][plane]
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 4; ++t) {                     // two v_max3_f32 per key tile
        mx = fmaxf(fmaxf(mx, s[qt][t][0]), s[qt][t][1]);
        mx = fmaxf(fmaxf(mx, s[qt][t][2]), s[qt][t][3]);
      }
      // deferred rescale: while no lane's new maximum exceeds the running one by more than ATTN_DEFER (in log2 units) the running
      // maximum is kept - no cross-lane max, no exp2 of the correction, no pass over the output accumulators.  P then reaches
      // 2^ATTN_DEFER at most, far inside fp16 / fp32 range; the first step (m_run = -inf) always takes the full path.
      float m_new = m_run[qt];
      const bool keep = ATTN_DEFER > 0.f && __all(mx - m_run[qt] <= ATTN_DEFER);
      if (!keep) {
        mx = rows_max(mx);
        m_new = fmaxf(m_run[qt], mx);                   // finite: key 0 (special token) is always valid
        const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
        l_run[qt] *= alpha;
        oacc[0][qt] *= alpha;
        oacc[1][qt] *= alpha;
        m_run[qt] = m_new;
      }
      float psum = 0.f;
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {                // k-slot (g4, j) of the PV product = key 4 g4 + (j&3) + 16 (j>>2) of the group
          const f32x2 d = f32x2{s[qt][2 * kg + (j >> 2)][j & 3], s[qt][2 * kg + (j >> 2)][(j & 3) + 1]} - f32x2{m_new, m_new};   // v_pk_add_f32
          pv[j] = __builtin_amdgcn_exp2f(d[0]);
          pv[j + 1] = __builtin_amdgcn_exp2f(d[1]);
          if constexpr (!P1) psum += pv[j] + pv[j + 1];
        }
        if constexpr (DROP) {
          const uint64_t row = drop_attn_row((uint64_t)pair, T, min(qw + 16 * qt + ql, T - 1));
#pragma unroll
          for (int j = 0; j < 8; j += 2) {              // keys 4 g4 + (j & 3), + 1: one hash per pair
            float m0, m1;
            drop_mult2_w(drop, dwin, row + (uint64_t)(kt * KSTEP + 16 * (2 * kg + (j >> 2)) + 4 * g4 + (j & 3)), m0, m1);
            pv[j] *= m0; pv[j + 1] *= m1;
          }
        }
        if constexpr (P1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t h = pk_f16(pv[2 * i], pv[2 * i + 1]);
            pf[qt][kg][0][i] = h;
            asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(psum) : "v"(h));                    // psum += (float)h.lo
            asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(psum) : "v"(h));      // psum += (float)h.hi
          }
        } else {
          split_planes<NP>(pv, pf[qt][kg]);
        }
      }
      l_run[qt] += psum;
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int dvt = 0; dvt < 2; ++dvt) {
        u32x4 vf[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) vf[p] = *reinterpret_cast<const u32x4*>(sV + ((kg * 2 + dvt) * NP + p) * FRAG);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          if constexpr (P1) {
            oacc[dvt][qt] = mfma_f16(vf[1], pf[qt][kg][0], oacc[dvt][qt]);      // V lo * P^
            oacc[dvt][qt] = mfma_f16(vf[0], pf[qt][kg][0], oacc[dvt][qt]);      // V hi * P^
          } else {
            oacc[dvt][qt] = mfma_split(vf, pf[qt][kg], oacc[dvt][qt]);
          }
        }
      }
    if (kt + 2 < nkt) swrite_k(kt + 2);                 // over K(kt): read one step ago
    if (kt + 1 < nkt) swrite_v(kt + 1);                 // over V(kt-1)
    __syncthreads();
  };
  // FAST step.  Invariant on entry: s = S(kt) - m_run (per query tile), as the MFMA left it.  P = exp2(s) is taken optimistically;
  // a lane whose 16-key probability sum exceeds 2^ATTN_DEFER (or is inf: a score far above the running maximum) sends the wave
  // through the revision path: true maximum of the step, m_run += d, everything already computed relative to the old value
  // (l, O^T, these scores and the next step's, which are in flight) re-based, probabilities recomputed.  The first key step always
  // revises (m_run starts at 0; there d may be negative).  P <= 2^ATTN_DEFER keeps the fp16 planes far from their range.
#ifdef PATHS_ATTN_STAMPS
  unsigned st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#endif
  auto pv_product = [&](const char* sV, u32x4 (&pfr)[QT][2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int dvt = 0; dvt < 2; ++dvt) {
        u32x4 vf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) vf[p] = *reinterpret_cast<const u32x4*>(sV + (((WHATIF & 32) ? 0 : (kg * 2 + dvt)) * NP + p) * FRAG);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          if constexpr ((WHATIF & 8) != 0) { oacc[dvt][qt] += __builtin_bit_cast(f32x4, pfr[qt][kg][0]) + __builtin_bit_cast(f32x4, pfr[qt][kg][1]) + __builtin_bit_cast(f32x4, vf[0]); }
          else oacc[dvt][qt] = mfma_split(vf, pfr[qt][kg], oacc[dvt][qt]);
        }
      }
  };
  // pf: receives this step's probability fragments; pfp (DELAY): the previous step's, whose PV product is issued here (pendc: there is one)
  auto step_fast = [&](int kt, f32x4 (&s)[QT][4], f32x4 (&sn)[QT][4], u32x4 (&pf)[QT][2][2], u32x4 (&pfp)[QT][2][2], auto lastc, auto pendc) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(lastc)::value, PEND = decltype(pendc)::value;
    if constexpr (PATHS_ATTN_PRIO_SHIFT >= 0) {
      if ((((kt >> PATHS_ATTN_PRIO_SHIFT) ^ prio_parity) & 1) != 0) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    ATTN_STAMP(0, 1);
    if constexpr ((WHATIF & 64) == 0) {
    if (kt + 2 < nkt) gload_k(kt + 2);
    if (kt + 1 < nkt) gload_v(kt + 1);
    }
    if constexpr (ATTN_OCC >= 3) qk(kt, s);
    const char* sV = sVb + vbuf(DELAY ? kt - 1 : kt) * HALF + lane * 16;      // (DELAY: the tile of the pending product)
    if constexpr (LAST) {
      const int kbase = kt * KSTEP + 4 * g4;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kbase + 16 * t + r >= len) s[qt][t][r] = -INFINITY;
    }
    if constexpr (ATTN_OCC < 3) qk(kt + 1, sn);
    if constexpr (DELAY && PEND) pv_product(sV, pfp);   // O^T += V^T(kt-1) P^T(kt-1): same basic block as the score product above and exp2 / split below
    ATTN_STAMP(1, 2);
    float psum[QT];
    auto probs = [&](int qt) __attribute__((always_inline)) {
      // four independent partial sums (a 16-deep dependent chain sat on the step's critical path: hipcc packed the two query tiles'
      // chains into v_pk_add_f32, each followed by a wait state); the empty asm keeps them scalar
      float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                   // k-slot (g4, j) of the PV product = key 4 g4 + (j&3) + 16 (j>>2) of the group
          pv[j] = (WHATIF & 1) ? s[qt][2 * kg + (j >> 2)][j & 3] : __builtin_amdgcn_exp2f(s[qt][2 * kg + (j >> 2)][j & 3]);
          ps[j & 3] += pv[j];
          asm("" : "+v"(ps[j & 3]));
        }
        if constexpr ((WHATIF & 2) != 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) pf[qt][kg][0][i] = pf[qt][kg][1][i] = pk_f16(pv[2 * i], pv[2 * i + 1]);
        } else split_planes<2>(pv, pf[qt][kg]);
      }
      psum[qt] = (ps[0] + ps[1]) + (ps[2] + ps[3]);
    };
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) probs(qt);
    ATTN_STAMP(2, 2);
    float pmax = psum[0];
#pragma unroll
    for (int qt = 1; qt < QT; ++qt) pmax = fmaxf(pmax, psum[qt]);
    if (kt == 0 || __any(!(pmax <= 256.0f))) {          // 2^8 (ATTN_DEFER); !(<=) also catches a NaN
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          mx = fmaxf(fmaxf(mx, s[qt][t][0]), s[qt][t][1]);
          mx = fmaxf(fmaxf(mx, s[qt][t][2]), s[qt][t][3]);
        }
        mx = rows_max(mx);                              // finite: key 0 (special token) is valid for every query
        const float d = kt == 0 ? mx : fmaxf(mx, 0.f);
        const float alpha = kt == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);
        l_run[qt] *= alpha;
        oacc[0][qt] *= alpha;
        oacc[1][qt] *= alpha;
        m_run[qt] += d;
        negm[qt] = f32x4{-m_run[qt], -m_run[qt], -m_run[qt], -m_run[qt]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s[qt][t] -= f32x4{d, d, d, d};
          if constexpr (ATTN_OCC < 3) sn[qt][t] -= f32x4{d, d, d, d};
        }
        probs(qt);
      }
    }
    ATTN_STAMP(3, 2);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) l_run[qt] += psum[qt];
    if constexpr (!DELAY) pv_product(sV, pf);
    ATTN_STAMP(4, 2);
    if constexpr ((WHATIF & 64) == 0) {
    if (kt + 2 < nkt) swrite_k(kt + 2);
    if (kt + 1 < nkt) swrite_v(kt + 1);
    }
    ATTN_STAMP(5, 1);
    __syncthreads();
    ATTN_STAMP(6, 1);
  };
  u32x4 pfA[QT][2][2], pfB[QT][2][2];                  // probability fragments of two consecutive steps (DELAY keeps one pending)
  auto stepx = [&](int kt, f32x4 (&s)[QT][4], f32x4 (&sn)[QT][4], u32x4 (&pf)[QT][2][2], u32x4 (&pfp)[QT][2][2], auto lastc, auto pendc) __attribute__((always_inline)) {
    if constexpr (FAST) step_fast(kt, s, sn, pf, pfp, lastc, pendc);
    else step(kt, s, sn, lastc);
  };
  {
    constexpr std::false_type MID{};
    constexpr std::true_type END{};
    constexpr std::false_type NOPEND{};
    constexpr std::true_type PEND{};
    int kt = 0;
    if constexpr (ATTN_OCC >= 3) {
      for (; kt + 1 < nkt; ++kt) stepx(kt, sA, sA, pfA, pfA, MID, NOPEND);
      stepx(kt, sA, sA, pfA, pfA, END, NOPEND);
    } else {
      // steps alternate the score buffers (sA, sB) and the fragment buffers (pfA, pfB); the first step has no pending product
      if (nkt == 1) {
        stepx(0, sA, sB, pfA, pfB, END, NOPEND);
        if constexpr (DELAY) pv_product(sVb + vbuf(0) * HALF + lane * 16, pfA);
      } else {
        stepx(0, sA, sB, pfA, pfB, MID, NOPEND);
        kt = 1;
        for (; kt + 2 < nkt; kt += 2) {
          stepx(kt, sB, sA, pfB, pfA, MID, PEND);
          stepx(kt + 1, sA, sB, pfA, pfB, MID, PEND);
        }
        // kt = first step not yet done (odd); one or two steps left
        if (kt + 1 < nkt) {
          stepx(kt, sB, sA, pfB, pfA, MID, PEND);
          stepx(kt + 1, sA, sB, pfA, pfB, END, PEND);
          if constexpr (DELAY) pv_product(sVb + vbuf(kt + 1) * HALF + lane * 16, pfA);
        } else {
          stepx(kt, sB, sA, pfB, pfA, END, PEND);
          if constexpr (DELAY) pv_product(sVb + vbuf(kt) * HALF + lane * 16, pfB);
        }
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l = rows_sum(l_run[qt]);
    const float inv = 1.0f / l;
    const int qi = qw + 16 * qt + ql;
    if constexpr (NP == 2) {
      // O as the B-operand fragments of the out_proj product of tlayer_ws.hip: this lane's 8 values (dims 4 g4 + r of both dv tiles
      // of query ql) are k-slot (g4, j) of k32 block `head`; image [slide][64-token group][head][16-token tile][plane][lane][16 B]
      if (o_img != nullptr) {
        const int tq = qw + 16 * qt;
        if (tq < Tp) {
          const float v[8] = {oacc[0][qt][0] * inv, oacc[0][qt][1] * inv, oacc[0][qt][2] * inv, oacc[0][qt][3] * inv,
                              oacc[1][qt][0] * inv, oacc[1][qt][1] * inv, oacc[1][qt][2] * inv, oacc[1][qt][3] * inv};
          u32x4 hi, lo;
          split8h(v, hi, lo);
          char* dst = o_img + ((((int64_t)b * (Tp >> 6) + (tq >> 6)) * H + head) * 4 + ((tq >> 4) & 3)) * (2 * FRAG) + lane * 16;
          *reinterpret_cast<u32x4*>(dst) = hi;
          *reinterpret_cast<u32x4*>(dst + FRAG) = lo;
        }
        continue;
      }
    }
    if (qi < T) {
      float* op = o + ((int64_t)b * T + qi) * (H * HD) + head * HD + 4 * g4;
      *reinterpret_cast<f32x4*>(op) = oacc[0][qt] * inv;
      *reinterpret_cast<f32x4*>(op + 16) = oacc[1][qt] * inv;
      if (lse && g4 == 0) lse[((int64_t)b * H + head) * T + qi] = m_run[qt] + log2f(l);
    }
  }
#ifdef PATHS_ATTN_DEBUG
  if (dbg && threadIdx.x == 0) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
#ifdef PATHS_ATTN_STAMPS
    unsigned long long* d = dbg + 16 * blockIdx.x;
    d[0] = dbg_t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = ((unsigned long long)xcc << 32) | hwid;
    for (int i = 0; i < 8; ++i) d[3 + i] = st_acc[i];
    d[11] = (unsigned long long)nkt;
#else
    dbg[3 * blockIdx.x] = dbg_t0; dbg[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); dbg[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hwid;
#endif
  }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same attention on v_mfma_f32_32x32x16_f16 (two fp16 planes, no dropout; round 4).  Why: a 16x16x32 MFMA holds the SIMD's
// vector issue for 8 of its 16 cycles, a 32x32x16 for 8 of its 32 - at equal matrix-pipe time the softmax's vector work (exp2, sums,
// operand split) finds three times the issue slots under the MFMAs (tools/simd_probe.hip: "12 x (32x32x16 + 6 fma)" 404 cycles per
// SIMD against "24 x (16x16x32 + 3 fma)" 499).  NOTHING changes for the producers and the consumer: the operand images are the
// 16x16x32 fragment images of the kernel above, read with a permuted lane address -
//   K / Q (32 tokens x 16 dims of k-step ks): lane (r = l & 31, h = l >> 5) takes the 16 bytes of old lane (r & 15) + 16 (2 ks + h) of
//         16-token tile r >> 4  (its dims 16 ks + 8 h .. + 7);
//   V^T   (32 dims x 16 keys of step u of a 32-key group): lane (dv = l & 31, h) takes the 16 bytes of old lane (dv & 15) + 16 (h + 2 u)
//         of dv tile dv >> 4: keys 4 h + 8 u + (j & 3) + 16 (j >> 2), j = 0..7 - the S^T accumulators 4 u + (j & 3) + 8 (j >> 2) of the
//         lane (accumulator i of a 32 x 32 tile = row (i & 3) + 8 (i >> 2) + 4 h): the contraction order is free, only P and V^T must
//         agree on it (two 8-byte reads per fragment in "natural" order cost a 4-way bank conflict: 58 us against 55);
//   O     lane (q, h) holds dims 4 h + r + 8 m (m = 0..3) of query q = the contents of old lanes (q & 15) + 16 h and + 16 (h + 2) of
//         the chain kernel's out_proj image: two 16-byte stores per plane, no shuffle.
// One wave = 32 queries (one accumulator column per lane: the softmax state is one scalar per lane, the two lane halves hold
// different keys of the same query); per 64-key step 12 + 12 MFMAs of 32 cycles.  Same FAST softmax as above (running maximum
// subtracted inside the score product, optimistic probabilities, revision path).
__device__ __forceinline__ f32x16 mfma32_f16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32_split(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x16 c) {   // hi, lo: all but lo*lo, smallest first
  c = mfma32_f16(a[1], b[0], c);
  c = mfma32_f16(a[0], b[1], c);
  return mfma32_f16(a[0], b[0], c);
}

__global__ void __launch_bounds__(256, 2)
attn_m32_kernel(const char* __restrict__ q6, const char* __restrict__ k6, const char* __restrict__ v6, float* __restrict__ o,
                float* __restrict__ lse, const int64_t* __restrict__ num_ims, int T, int Tp, int H, int npairs_arg, int nqb_arg,
                char* __restrict__ o_img) {
  constexpr int NP = 2;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int npairs = npairs_arg, nqb = nqb_arg;
  const int lin = blockIdx.x, xg = lin & 7, jx = lin >> 3;
  const int cnt = (npairs - xg + 7) >> 3;
  if (cnt <= 0) return;
  const int pair = xg + 8 * (jx % cnt), qb = jx / cnt;
  if (qb >= nqb) return;
  const int b = pair / H, head = pair - b * H, q0 = qb * 128;
  const int len = min((int)num_ims[b] + 1, T);
  if (q0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qc = lane & 31, h = lane >> 5, l15 = lane & 15, sub = (lane >> 4) & 1;      // sub: which 16-token / 16-dim tile of the 32
  const int64_t ibase = ((int64_t)b * H + head) * (int64_t)Tp * HD * 2 * NP;
  const int qw = q0 + wave * 32;

  // Q fragments [k-step][plane] (B operand: 16 dims x 32 queries)
  u32x4 qf[2][NP];
  {
    const int64_t tq = min(qw + 16 * sub, Tp - 16) >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < NP; ++p)
        qf[ks][p] = *reinterpret_cast<const u32x4*>(q6 + ibase + (tq * NP + p) * FRAG + (l15 + 16 * (2 * ks + h)) * 16);
  }
  f32x16 oacc, negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) { oacc[i] = 0.f; negm[i] = 0.f; }
  float m_run = 0.f, l_run = 0.f;

  const int nkt = (len + KSTEP - 1) / KSTEP;
  constexpr int HALF = 4 * NP * FRAG;
  char* const sKb = smem_raw;                           // [2][HALF]
  char* const sVb = smem_raw + 2 * HALF;                // [2][HALF]
  u32x4 st[2 * NP];
  auto gload_k = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) st[i] = *reinterpret_cast<const u32x4*>(k6 + ibase + (int64_t)kt * HALF + (tid + 256 * i) * 16);
  };
  auto gload_v = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) st[NP + i] = *reinterpret_cast<const u32x4*>(v6 + ibase + (int64_t)kt * HALF + (tid + 256 * i) * 16);
  };
  auto swrite_k = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) *reinterpret_cast<u32x4*>(sKb + (kt & 1) * HALF + (tid + 256 * i) * 16) = st[i];
  };
  auto swrite_v = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) *reinterpret_cast<u32x4*>(sVb + (kt & 1) * HALF + (tid + 256 * i) * 16) = st[NP + i];
  };
  // S^T of the two 32-key tiles of step kt: s[t] = K_t Q^T - m_run
  auto qk = [&](int kt, f32x16 (&s)[2]) __attribute__((always_inline)) {
    const char* sK = sKb + (kt & 1) * HALF;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 kf[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
          kf[p] = *reinterpret_cast<const u32x4*>(sK + ((2 * t + sub) * NP + p) * FRAG + (l15 + 16 * (2 * ks + h)) * 16);
        s[t] = mfma32_split(kf, qf[ks], ks == 0 ? negm : s[t]);
      }
  };
  gload_k(0); gload_v(0);
  swrite_k(0); swrite_v(0);
  if (nkt > 1) { gload_k(1); swrite_k(1); }
  __syncthreads();
  f32x16 sA[2], sB[2];
  qk(0, sA);
  auto step = [&](int kt, f32x16 (&s)[2], f32x16 (&sn)[2], auto lastc) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(lastc)::value;
    if (kt + 2 < nkt) gload_k(kt + 2);
    if (kt + 1 < nkt) gload_v(kt + 1);
    const char* sV = sVb + (kt & 1) * HALF;
    if constexpr (LAST) {
      const int kbase = kt * KSTEP + 4 * h;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kbase + 32 * t + (i & 3) + 8 * (i >> 2) >= len) s[t][i] = -INFINITY;
    }
    qk(kt + 1, sn);                                     // (past the end: stale K fragments, finite garbage nobody reads)
    u32x4 pf[2][2][NP];                                 // [key tile][16-key step][plane]
    float psum;
    auto probs = [&]() __attribute__((always_inline)) {
      float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float pv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {                 // k-slot (h, j) of PV step u = key 4 h + 8 u + (j & 3) + 16 (j >> 2) of the tile
            pv[j] = __builtin_amdgcn_exp2f(s[t][4 * u + (j & 3) + 8 * (j >> 2)]);
            ps[j & 3] += pv[j];
            asm("" : "+v"(ps[j & 3]));
          }
          split8h(pv, pf[t][u][0], pf[t][u][1]);
        }
      psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
    };
    probs();
    if (kt == 0 || __any(!(psum <= 256.0f))) {
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s[t][i]), s[t][i + 1]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));               // the other half of the query's keys; finite: key 0 is valid for every query
      const float d = kt == 0 ? mx : fmaxf(mx, 0.f);
      const float alpha = kt == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);
      l_run *= alpha;
      m_run += d;
#pragma unroll
      for (int i = 0; i < 16; ++i) { oacc[i] *= alpha; negm[i] = -m_run; }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[t][i] -= d; sn[t][i] -= d; }
      probs();
    }
    l_run += psum;
    // O^T += V^T P^T: four 16-key steps
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4 vf[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
          vf[p] = *reinterpret_cast<const u32x4*>(sV + ((t * 2 + sub) * NP + p) * FRAG + (l15 + 16 * (h + 2 * u)) * 16);
        oacc = mfma32_split(vf, pf[t][u], oacc);
      }
    if (kt + 2 < nkt) swrite_k(kt + 2);
    if (kt + 1 < nkt) swrite_v(kt + 1);
    __syncthreads();
  };
  {
    constexpr std::false_type MID{};
    constexpr std::true_type END{};
    int kt = 0;
    for (; kt + 2 < nkt; kt += 2) {
      step(kt, sA, sB, MID);
      step(kt + 1, sB, sA, MID);
    }
    if (kt + 1 < nkt) { step(kt, sA, sB, MID); step(kt + 1, sB, sA, END); }
    else step(kt, sA, sB, END);
  }
  const float l = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l;
  if (o_img != nullptr) {
    const int tq = qw + 16 * sub;
    if (tq < Tp) {
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {                  // old lanes (q & 15) + 16 g4, g4 = h + 2 gg: dims 4 g4 + r and 16 + 4 g4 + r
        const int i0 = 4 * gg, i1 = 8 + 4 * gg;
        const float v[8] = {oacc[i0] * inv, oacc[i0 + 1] * inv, oacc[i0 + 2] * inv, oacc[i0 + 3] * inv,
                            oacc[i1] * inv, oacc[i1 + 1] * inv, oacc[i1 + 2] * inv, oacc[i1 + 3] * inv};
        u32x4 hi, lo;
        split8h(v, hi, lo);
        char* dst = o_img + ((((int64_t)b * (Tp >> 6) + (tq >> 6)) * H + head) * 4 + ((tq >> 4) & 3)) * (2 * FRAG) + (l15 + 16 * (h + 2 * gg)) * 16;
        *reinterpret_cast<u32x4*>(dst) = hi;
        *reinterpret_cast<u32x4*>(dst + FRAG) = lo;
      }
    }
    return;
  }
  const int qi = qw + qc;
  if (qi < T) {
    float* op = o + ((int64_t)b * T + qi) * (H * HD) + head * HD + 4 * h;
#pragma unroll
    for (int m = 0; m < 4; ++m)
      *reinterpret_cast<f32x4*>(op + 8 * m) = f32x4{oacc[4 * m] * inv, oacc[4 * m + 1] * inv, oacc[4 * m + 2] * inv, oacc[4 * m + 3] * inv};
    if (lse && h == 0) lse[((int64_t)b * H + head) * T + qi] = m_run + log2f(l);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The 32x32x16 kernel as a PHASE-LOCKED WAVE PAIR (round 4): 512 threads = 8 waves of 32 queries, waves w and w + 4 share a SIMD.
// A wave's key step is two segments - V: the softmax of S(kt) (vector unit only: 32 exp2, sums, the fp16 split) and M: O += V^T P(kt),
// S(kt+1) = K Q^T (matrix unit only, 24 MFMAs of 32 cycles) - with a workgroup barrier after each, and waves 4-7 run one segment
// behind waves 0-3: on every SIMD one wave is in its M segment while the other is in its V segment, all the time.  Why: two
// free-running waves that each interleave MFMAs and vector work do not overlap them - an in-order wave whose next MFMA waits for the
// partner's MFMA cannot issue the vector instructions behind it, and the measured step was the SUM of both pipes' time
// (profiles/r04_experiments.md: 3056 cycles per SIMD and pair of wave-steps = 1536 MFMA + ~1400 vector); tools/simd_probe.hip's
// LOCKED rows give 1806 for the pair in complementary segments.  S needs one buffer only (S(kt+1) is produced after P(kt) was taken).
//   slot (barrier interval):   2kt                          2kt+1                        2kt+2
//   waves 0-3                  V(kt)                        M(kt)                        V(kt+1)
//   waves 4-7                  M(kt-1)                      V(kt)                        M(kt)
//   waves 8-11 (loaders)       -                            DMA of bundle kt+3, wait for bundle kt+1
// Staging is LDS-DMA (global_load_lds_dwordx4: the fragment images are copied verbatim, 1 KiB per wave instruction) by four loader
// waves, one per SIMD, that do nothing else (768 threads; 159 VGPRs x 3 waves fit a SIMD).  Bundle j = {V^T(j), K(j+1)} (what M(j)
// reads) is issued in slot 2j-5, waited for (counted vmcnt: the next two bundles stay in flight) at the end of slot 2j-1 and published
// by that slot's barrier, so BOTH groups may issue the first fragment reads of M(j) before the barrier that opens it (end of slot 2j
// / 2j+1); read until slot 2j+2, bundle j+4 takes its place from slot 2j+3: rings of four steps (64 KiB).
// (Measured on the way, profiles/r04_experiments.md: issued by waves 4-7 themselves, the four DMAs of a step took 650-1000 cycles to
// ISSUE - in front of their vector segment or of their MFMAs alike, with or without s_setprio - whenever the SIMD partner was in its
// dense vector segment; through registers (global_load + ds_write) ~300.)
// Barriers are raw s_barrier (a __syncthreads would drain the DMA in flight); nothing in the loop writes LDS by ds_write.
__device__ __forceinline__ void glds16(const char* sbase, uint32_t voff, uint32_t lds_dst) {
  // sbase + voff: wave-uniform base (SGPR pair) + per-lane byte offset - no vector instruction at the issue: a 64-bit VALU add per DMA
  // in front of the M segment waited for issue slots behind the partner's dense vector stream (~250 cycles per DMA);
  // lds_dst: wave-uniform byte address, lane l lands at + 16 l
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void pair_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

#ifndef PATHS_M32P_PKADD
#define PATHS_M32P_PKADD 0
#endif
#ifndef PATHS_M32P_EARLYPRE
#define PATHS_M32P_EARLYPRE 1
#endif
#ifndef PATHS_M32P_PIPE
#define PATHS_M32P_PIPE 0                 // explicit two-chunks-ahead fragment reads in M: over the 168-VGPR cap of three waves per SIMD (spills, 65 us)
#endif
#ifndef PATHS_M32P_LOADERS
#define PATHS_M32P_LOADERS 1              // 1: four loader waves (768 threads); 0: waves 0-3 issue the DMAs at the head of their M segment (512 threads)
#endif
__global__ void __launch_bounds__(PATHS_M32P_LOADERS ? 768 : 512)
attn_m32p_kernel(const char* __restrict__ q6, const char* __restrict__ k6, const char* __restrict__ v6, float* __restrict__ o,
                 float* __restrict__ lse, const int64_t* __restrict__ num_ims, int T, int Tp, int H, int npairs_arg, int nqb_arg,
                 char* __restrict__ o_img
#ifdef PATHS_M32P_STAMPS
                 , unsigned long long* __restrict__ dbg
#endif
                 ) {
  constexpr int NP = 2;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
#ifdef PATHS_M32P_STAMPS
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
  const int npairs = npairs_arg, nqb = nqb_arg;
  const int lin = blockIdx.x, xg = lin & 7, jx = lin >> 3;
  const int cnt = (npairs - xg + 7) >> 3;
  if (cnt <= 0) return;
  const int pair = xg + 8 * (jx % cnt), qb = jx / cnt;
  if (qb >= nqb) return;
  const int b = pair / H, head = pair - b * H, q0 = qb * 256;
  const int len = min((int)num_ims[b] + 1, T);
  if (q0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                            // 1 = the half that runs one segment behind; 2 = the four loader waves
  const int qc = lane & 31, h = lane >> 5, l15 = lane & 15, sub = (lane >> 4) & 1;
  const int64_t ibase = ((int64_t)b * H + head) * (int64_t)Tp * HD * 2 * NP;
  // A block with at most 32 queries (the ninth block of a completely full slide: 2,049 = 8 x 256 + 1 tokens) takes the SMALL path below:
  // all its waves share the one query tile and split the KEYS (block-uniform branch)
  const bool small = len - q0 <= 32;
  const int qw = small ? q0 : q0 + wave * 32;
  const bool active = qw < len && grp < 2;              // (a wave without queries still meets every barrier)

  u32x4 qf[2][NP];
  {
    const int64_t tq = min(qw + 16 * sub, Tp - 16) >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < NP; ++p)
        qf[ks][p] = *reinterpret_cast<const u32x4*>(q6 + ibase + (tq * NP + p) * FRAG + (l15 + 16 * (2 * ks + h)) * 16);
  }
  f32x16 oacc, negm, s[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { oacc[i] = 0.f; negm[i] = 0.f; }
  float m_run = 0.f, l_run = 0.f;
  u32x4 pf[2][2][NP];                                   // P^T fragments [key tile][16-key step][plane]
#if PATHS_M32P_PIPE
  u32x4 fr[3][NP];                                      // operand fragments of M in flight (chunk c in fr[c % 3]); fr[0], fr[1] = V^T of key tile 0, read ahead of the barrier
  auto& vpre = fr;
#else
  u32x4 vpre[2][NP];                                    // V^T fragments of key tile 0, read ahead of the barrier that opens M
#endif

  const int nkt = (len + KSTEP - 1) / KSTEP;
  constexpr int HALF = 4 * NP * FRAG;                   // one 64-key step of K or of V^T
  // normalise and store this wave's query tile (both paths end here)
  auto write_out = [&]() __attribute__((always_inline)) {
    const float l = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l;
    if (o_img != nullptr) {
      const int tq = qw + 16 * sub;
      if (tq < Tp) {
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int i0 = 4 * gg, i1 = 8 + 4 * gg;
          const float v[8] = {oacc[i0] * inv, oacc[i0 + 1] * inv, oacc[i0 + 2] * inv, oacc[i0 + 3] * inv,
                              oacc[i1] * inv, oacc[i1 + 1] * inv, oacc[i1 + 2] * inv, oacc[i1 + 3] * inv};
          u32x4 hi, lo;
          split8h(v, hi, lo);
          char* dst = o_img + ((((int64_t)b * (Tp >> 6) + (tq >> 6)) * H + head) * 4 + ((tq >> 4) & 3)) * (2 * FRAG) + (l15 + 16 * (h + 2 * gg)) * 16;
          *reinterpret_cast<u32x4*>(dst) = hi;
          *reinterpret_cast<u32x4*>(dst + FRAG) = lo;
        }
      }
      return;
    }
    const int qi = qw + qc;
    if (qi < T) {
      float* op = o + ((int64_t)b * T + qi) * (H * HD) + head * HD + 4 * h;
#pragma unroll
      for (int m = 0; m < 4; ++m)
        *reinterpret_cast<f32x4*>(op + 8 * m) = f32x4{oacc[4 * m] * inv, oacc[4 * m + 1] * inv, oacc[4 * m + 2] * inv, oacc[4 * m + 3] * inv};
      if (lse && h == 0) lse[((int64_t)b * H + head) * T + qi] = m_run + log2f(l);
    }
  };
  if (small) {
    // ---- SMALL path: the block's one query tile against key steps wave, wave + NWS, ... per wave, fragments straight from the
    // images in global memory (they are stored in operand order), the (m, l, O) states merged through LDS by wave 0.  Run as a
    // ninth block at the pace of a full one (one active wave meeting every barrier of 33 key steps) this took ~35 us as a SECOND ROUND
    // of workgroups behind the 256 full blocks of eight full slides: 46 -> 82 us for the launch (tools/attn_full_slides.py).
    constexpr int NWS = PATHS_M32P_LOADERS ? 12 : 8;    // every wave of the workgroup takes key steps (nothing is staged through LDS here)
    const char* const kg = k6 + ibase;
    const char* const vg = v6 + ibase;
    bool first = true;
    for (int kt = wave; kt < nkt; kt += NWS) {
      const char* gK = kg + (int64_t)kt * HALF;
      const char* gV = vg + (int64_t)kt * HALF;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 kf[NP];
#pragma unroll
          for (int p = 0; p < NP; ++p)
            kf[p] = *reinterpret_cast<const u32x4*>(gK + ((2 * t + sub) * NP + p) * FRAG + (l15 + 16 * (2 * ks + h)) * 16);
          s[t] = mfma32_split(kf, qf[ks], ks == 0 ? negm : s[t]);
        }
      if (kt == nkt - 1) {
        const int kbase = kt * KSTEP + 4 * h;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (kbase + 32 * t + (i & 3) + 8 * (i >> 2) >= len) s[t][i] = -INFINITY;
      }
      float psum;
      auto probs = [&]() __attribute__((always_inline)) {
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              pv[j] = __builtin_amdgcn_exp2f(s[t][4 * u + (j & 3) + 8 * (j >> 2)]);
              ps[j & 3] += pv[j];
            }
            split8h(pv, pf[t][u][0], pf[t][u][1]);
          }
        psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
      };
      probs();
      if (first || __any(!(psum <= 256.0f))) {
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s[t][i]), s[t][i + 1]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float d = first ? mx : fmaxf(mx, 0.f);
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-d);
        l_run *= alpha;
        m_run += d;
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[i] *= alpha; negm[i] -= d; }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) s[t][i] -= d;
        probs();
      }
      l_run += psum;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          u32x4 vf[NP];
#pragma unroll
          for (int p = 0; p < NP; ++p) vf[p] = *reinterpret_cast<const u32x4*>(gV + ((t * 2 + sub) * NP + p) * FRAG + (l15 + 16 * (h + 2 * u)) * 16);
          oacc = mfma32_split(vf, pf[t][u], oacc);
        }
      first = false;
    }
    // per-lane states -> LDS [NWS waves][18][64]; a wave without a key step publishes the neutral element
    float* const st = reinterpret_cast<float*>(smem_raw) + wave * (18 * 64);
    st[lane] = first ? -1e30f : m_run;
    st[64 + lane] = first ? 0.f : l_run;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[(2 + i) * 64 + lane] = first ? 0.f : oacc[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    pair_barrier();
    if (wave != 0) return;
    const float* const all = reinterpret_cast<const float*>(smem_raw);
    float M = -1e30f;
#pragma nounroll
    for (int w = 0; w < NWS; ++w) M = fmaxf(M, all[w * (18 * 64) + lane]);
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[i] = 0.f;
#pragma nounroll
    for (int w = 0; w < NWS; ++w) {
      const float sc = __builtin_amdgcn_exp2f(all[w * (18 * 64) + lane] - M);
      lsum = fmaf(all[w * (18 * 64) + 64 + lane], sc, lsum);
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[i] = fmaf(all[w * (18 * 64) + (2 + i) * 64 + lane], sc, oacc[i]);
    }
    m_run = M;
    l_run = lsum;
    write_out();
    return;
  }
  char* const sKb = smem_raw;                           // [4][HALF]
  char* const sVb = smem_raw + 4 * HALF;                // [4][HALF]
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem_raw);
  // prologue: K(0) and bundle 0 = {V^T(0), K(1)} by all 512 threads through registers
  {
    const int t5 = tid & 511;
    const u32x4 a = *reinterpret_cast<const u32x4*>(k6 + ibase + t5 * 16);
    const u32x4 c = *reinterpret_cast<const u32x4*>(v6 + ibase + t5 * 16);
    u32x4 d = a;
    if (nkt > 1) d = *reinterpret_cast<const u32x4*>(k6 + ibase + HALF + t5 * 16);
    if (grp < 2) {
      *reinterpret_cast<u32x4*>(sKb + t5 * 16) = a;
      *reinterpret_cast<u32x4*>(sVb + t5 * 16) = c;
      *reinterpret_cast<u32x4*>(sKb + HALF + t5 * 16) = d;
    }
  }
  // bundle j by waves 4-7: V^T(j) -> V ring position j & 3, K(j + 1) -> K ring position (j + 1) & 3; each wave 2 + 2 KiB
  const int wq = wave & 3;
  const uint32_t voff = wq * 1024 + lane * 16;
  auto dma_bundle = [&](int j) {
#ifndef PATHS_M32P_NOSTAGE
    if (j >= nkt) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16(v6 + ibase + (int64_t)j * HALF + 4096 * i, voff, lds0 + (4 + (j & 3)) * HALF + (wq + 4 * i) * 1024);
    if (j + 1 < nkt) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        glds16(k6 + ibase + (int64_t)(j + 1) * HALF + 4096 * i, voff, lds0 + ((j + 1) & 3) * HALF + (wq + 4 * i) * 1024);
    }
#endif
  };
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  auto fly_wait = [&](int kt) {                         // bundle kt + 1 has landed; kt + 2 and kt + 3 may fly
    auto count = [&](int j) { return j >= nkt ? 0 : j + 1 < nkt ? 4 : 2; };
    const int fly = count(kt + 2) + count(kt + 3);
    if (fly == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (fly == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (fly == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (fly == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
#if !PATHS_M32P_LOADERS
  if (grp == 0) { dma_bundle(1); dma_bundle(2); }
#endif
  if (grp == 2) {
    // The loader waves (one per SIMD, at priority 3: they issue ~30 instructions per key step).  Issued by waves 4-7 themselves the
    // four DMAs of a step took 650-1000 cycles to ISSUE behind the partner's dense vector stream, in front of the wave's MFMAs.
    // Slot s ends with barrier s + 1; in slot 2kt+1: issue bundle kt+3 (its ring position was last read in slot 2kt), then wait until
    // bundle kt+1 has landed (bundles kt+2 and kt+3 may fly).
    __builtin_amdgcn_s_setprio(3);
    dma_bundle(1); dma_bundle(2);
    auto count = [&](int j) { return j >= nkt ? 0 : j + 1 < nkt ? 4 : 2; };
    pair_barrier();                                     // (the prologue's)
    pair_barrier();                                     // slot -1
    pair_barrier();                                     // slot 0
    for (int kt = 0; kt < nkt; ++kt) {
      dma_bundle(kt + 3);
      const int fly = count(kt + 2) + count(kt + 3);
      if (fly == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (fly == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (fly == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (fly == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pair_barrier();                                   // slot 2kt+1
      pair_barrier();                                   // slot 2kt+2
    }
    return;
  }
  pair_barrier();

  auto qk = [&](const char* sK) __attribute__((always_inline)) {   // s[t] = K_t Q^T - m_run
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 kf[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
          kf[p] = *reinterpret_cast<const u32x4*>(sK + ((2 * t + sub) * NP + p) * FRAG + (l15 + 16 * (2 * ks + h)) * 16);
        if (ks == 0)                                    // (C = -m_run read in place: hipcc would copy the 16 registers into s[t] first)
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s[t]) : "v"(kf[1]), "v"(qf[0][0]), "v"(negm));
        else
          s[t] = mfma32_f16(kf[1], qf[ks][0], s[t]);
        s[t] = mfma32_f16(kf[0], qf[ks][1], s[t]);
        s[t] = mfma32_f16(kf[0], qf[ks][0], s[t]);
      }
  };
  auto vfrag = [&](const char* sV, int t, int u, int p) __attribute__((always_inline)) {
    return *reinterpret_cast<const u32x4*>(sV + ((t * 2 + sub) * NP + p) * FRAG + (l15 + 16 * (h + 2 * u)) * 16);
  };
  // V segment: (waves 4-7: DMA issue), softmax of s -> pf, l_run (and, rarely, a new running maximum), first V^T fragments of M(kt)
  auto vseg = [&](int kt) __attribute__((always_inline)) {
    M32P_T(5);
#ifdef PATHS_M32P_WHATIF
    if (active && !((PATHS_M32P_WHATIF & 1) && grp == 0) && !((PATHS_M32P_WHATIF & 8) && grp == 1)) {
#else
    if (active) {
#endif
      if (kt == nkt - 1) {
        const int kbase = kt * KSTEP + 4 * h;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (kbase + 32 * t + (i & 3) + 8 * (i >> 2) >= len) s[t][i] = -INFINITY;
      }
#if PATHS_M32P_EARLYPRE
      {                                                 // first V^T fragments of M(kt): published one slot ago, read under the softmax
        const char* sV = sVb + (kt & 3) * HALF;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int p = 0; p < NP; ++p) vpre[u][p] = vfrag(sV, 0, u, p);
      }
#endif
      float psum;
      auto probs = [&]() __attribute__((always_inline)) {
#if PATHS_M32P_PKADD
        f32x2 ps[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};     // four independent v_pk_add_f32 chains
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = __builtin_amdgcn_exp2f(s[t][4 * u + (j & 3) + 8 * (j >> 2)]);
#pragma unroll
            for (int j = 0; j < 4; ++j) ps[j] += f32x2{pv[2 * j], pv[2 * j + 1]};
            split8h(pv, pf[t][u][0], pf[t][u][1]);
          }
        const f32x2 q = (ps[0] + ps[1]) + (ps[2] + ps[3]);
        psum = q[0] + q[1];
#else
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              pv[j] = __builtin_amdgcn_exp2f(s[t][4 * u + (j & 3) + 8 * (j >> 2)]);
              ps[j & 3] += pv[j];
              asm("" : "+v"(ps[j & 3]));
            }
            split8h(pv, pf[t][u][0], pf[t][u][1]);
          }
        psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
#endif
      };
      probs();
      M32P_T(6);
      if (kt == 0 || __any(!(psum <= 256.0f))) {
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s[t][i]), s[t][i + 1]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float d = kt == 0 ? mx : fmaxf(mx, 0.f);
        const float alpha = kt == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);
        l_run *= alpha;
        m_run += d;
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[i] *= alpha; negm[i] -= d; }     // (in place: = -m_run, bit for bit - m_run is the sum of the d's)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) s[t][i] -= d;
        probs();
      }
      l_run += psum;
#if !PATHS_M32P_EARLYPRE
      const char* sV = sVb + (kt & 3) * HALF;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int p = 0; p < NP; ++p) vpre[u][p] = vfrag(sV, 0, u, p);
#endif
    }
  };
  // M segment: O^T += V^T(kt) P^T(kt), then S(kt+1)
  auto mseg = [&](int kt, auto morec) __attribute__((always_inline)) {
    constexpr bool more = decltype(morec)::value;       // S(kt+1) wanted: all but the last step
    M32P_T(7);
    if (!active) return;
#ifdef PATHS_M32P_WHATIF
    if ((PATHS_M32P_WHATIF & 4) && grp == 0) return;
    if ((PATHS_M32P_WHATIF & 16) && grp == 1) return;
#endif
    const char* sV = sVb + (kt & 3) * HALF;
#if PATHS_M32P_PIPE
    // 12 chunks of 3 MFMAs: 0-3 = V^T (tile, step), 4-11 = K (tile, k-step); chunk c + 2 is read while chunk c multiplies (chunks 0, 1
    // came in before the barrier): the fragment reads run two chunks = 192+ matrix cycles ahead of their use
    const char* sK = sKb + ((kt + 1) & 3) * HALF;
    auto rd = [&](int c) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (c < 4) fr[c % 3][p] = vfrag(sV, c >> 1, c & 1, p);
        else { const int t = (c - 4) >> 1, ks = (c - 4) & 1;
               fr[c % 3][p] = *reinterpret_cast<const u32x4*>(sK + ((2 * t + sub) * NP + p) * FRAG + (l15 + 16 * (2 * ks + h)) * 16); }
      }
    };
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      if (c + 2 < 4 || (c + 2 < 12 && more)) rd(c + 2);
      if (c < 4) oacc = mfma32_split(fr[c % 3], pf[c >> 1][c & 1], oacc);
      else if (more) {
        const int t = (c - 4) >> 1, ks = (c - 4) & 1;
        if (ks == 0)
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s[t]) : "v"(fr[c % 3][1]), "v"(qf[0][0]), "v"(negm));
        else
          s[t] = mfma32_f16(fr[c % 3][1], qf[ks][0], s[t]);
        s[t] = mfma32_f16(fr[c % 3][0], qf[ks][1], s[t]);
        s[t] = mfma32_f16(fr[c % 3][0], qf[ks][0], s[t]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int u = 0; u < 2; ++u) oacc = mfma32_split(vpre[u], pf[0][u], oacc);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      u32x4 vf[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) vf[p] = vfrag(sV, 1, u, p);
      oacc = mfma32_split(vf, pf[1][u], oacc);
    }
    if (more) qk(sKb + ((kt + 1) & 3) * HALF);
#endif
  };
  if (grp) pair_barrier();                              // (slot -1: waves 0-3 compute S(0))
  if (active) qk(sKb);
  pair_barrier();
  M32P_T(0);
  for (int kt = 0; kt < nkt; ++kt) {
    vseg(kt);
    M32P_T(1);
    pair_barrier();
    M32P_T(2);
#if !PATHS_M32P_LOADERS
    if (grp == 0) dma_bundle(kt + 3);                   // (slot 2kt+1, like the loader waves)
#endif
    if (kt + 1 < nkt) mseg(kt, std::true_type{}); else mseg(kt, std::false_type{});
#if !PATHS_M32P_LOADERS
    if (grp == 0) fly_wait(kt);
#endif
    M32P_T(3);
    pair_barrier();
    M32P_T(4);
  }
  if (!grp) pair_barrier();                             // (the last slot belongs to waves 4-7)
#ifdef PATHS_M32P_STAMPS
  if (dbg != nullptr && lane == 0) {
    unsigned long long* d = dbg + ((int64_t)blockIdx.x * 8 + wave) * 8;
    for (int i = 0; i < 5; ++i) d[i] = tacc[i];
    d[5] = tacc[5]; d[6] = nkt; d[7] = tacc[6]; d[4] = tacc[7];
  }
#endif
  if (!active) return;
  write_out();
}

#ifndef PATHS_ATTN_M32
#define PATHS_ATTN_M32 2
#endif
static int attn_num_cus() {                 // CUs of the current device (asked once per device)
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cached[dev] = cus;
  }
  return cached[dev];
}
template <int NP>
int attention_split(const float* q, const float* k, const float* v, float* o, float* lse, const int64_t* num_ims, int B, int T,
                           int H, int max_queries, void* workspace, int images_ready, hipStream_t stream, uint64_t drop_key = 0, float drop_p = 0.f,
                           char* o_img = nullptr) {
  const int Tp = (T + KSTEP - 1) / KSTEP * KSTEP;
  const int64_t img = (int64_t)B * H * Tp * HD * 2 * NP;
  char* q6 = reinterpret_cast<char*>(workspace);
  char* k6 = q6 + img;
  char* v6 = k6 + img;
  const int nq = max_queries > 0 && max_queries < T ? max_queries : T;
  if (!images_ready) {
  // (the dispatcher packs a CU to its limit before it moves on: pad the LDS request so that the grid covers all 256 CUs)
  const int nprep = (Tp / KSTEP) * H * B;
  const int pdepth = (nprep + 255) / 256;
  const int plds = pdepth >= 8 ? 0 : (160 * 1024 / pdepth - 9 * 1024) / 1024 * 1024 < 55 * 1024 ? (160 * 1024 / pdepth - 9 * 1024) / 1024 * 1024 : 55 * 1024;
  hipLaunchKernelGGL(attn_x6_prep_kernel<NP>, dim3(Tp / KSTEP, H, B), dim3(256), plds, stream, q, k, v, q6, k6, v6, num_ims, T, Tp, H, nq);
  PATHS_LAUNCH_CHECK("attention_x6(prep)");
  }
  // Workgroups per CU: registers allow 2, LDS would allow more.  The dispatcher fills a CU to its limit before it moves on, so
  // small grids ask for more LDS than needed to spread out: depth ~ grid / 256.
  const int nblk = ((nq + 64 * QT - 1) / (64 * QT)) * H * B;
  static const int depth_env = getenv("PATHS_ATTN_DEPTH") ? atoi(getenv("PATHS_ATTN_DEPTH")) : 0;   // experiment
  const int depth = depth_env ? depth_env : ATTN_OCC >= 3 ? (nblk <= 256 ? 1 : nblk <= 512 ? 2 : 3) : nblk <= 256 ? 1 : nblk <= 640 ? 2 : 3;    // (544 workgroups at K = 2048 x 8 slides: 2 per CU on every CU beat 3 per CU on 2/3 of them by 2 %)
  const int lds = depth == 1 ? 96 * 1024 : depth == 2 ? 64 * 1024 : (5 * step_bytes<NP>()) / 2;     // (K ring of two + V^T ring of up to three half-step buffers)
  PATHS_LDS_OPT_IN((attn_x6_kernel<NP, false>), 96 * 1024, "attention_x6");
  PATHS_LDS_OPT_IN((attn_x6_kernel<NP, true>), 96 * 1024, "attention_x6(dropout)");
  const int nqb = (nq + 64 * QT - 1) / (64 * QT), npairs = H * B;
  const DropSite site = paths_make_drop_site(drop_key, drop_p);
  // 1-D grid walked in XCD-aware order (see the kernel)
#ifdef PATHS_ATTN_DEBUG
  hipLaunchKernelGGL((attn_x6_kernel<NP, false>), dim3(8 * ((npairs + 7) / 8) * nqb, 1, 1), dim3(256), lds, stream, q6, k6, v6, o, lse, num_ims, T, Tp, H, npairs, nqb, site, o_img, g_attn_dbg);
#else
  const char* m32_env = getenv("PATHS_ATTN_M32");       // (read per call: the tests switch kernels inside one process)
  const int m32 = m32_env ? atoi(m32_env) : PATHS_ATTN_M32;     // 0: 16x16x32 kernel, 1: 32x32x16, 2: 32x32x16 wave pairs, 3: wave pairs on any grid
  // The wave-pair kernel puts 256 queries on one CU: it wins where its grid covers most of the chip (K = 2048 x 8 slides: 256
  // workgroups, 50.9 us against 55.3; 1024 x 8: 160, 28.4 against 29.9) and loses on small grids, which the 128-query kernel spreads
  // over twice as many CUs (2048 x 4 slides: 43.6 against 36.3; 512 x 8: 17.3 against 13.2).
  const int nqb2 = (nq + 255) / 256;
  const int pair_min = attn_num_cus() * 5 / 8;
  if (drop_p > 0.f)
    hipLaunchKernelGGL((attn_x6_kernel<NP, true>), dim3(8 * ((npairs + 7) / 8) * nqb, 1, 1), dim3(256), lds, stream, q6, k6, v6, o, lse, num_ims, T, Tp, H, npairs, nqb, site, o_img);
  else if (NP == 2 && (m32 == 3 || (m32 == 2 && npairs * nqb2 >= pair_min))) {
    // one workgroup of 12 waves per CU (168 VGPRs a wave): 84 KiB of LDS asked for, 64 KiB used
    PATHS_LDS_OPT_IN(attn_m32p_kernel, 96 * 1024, "attention_x6(32x32x16, wave pairs)");
#ifdef PATHS_M32P_STAMPS
    hipLaunchKernelGGL(attn_m32p_kernel, dim3(8 * ((npairs + 7) / 8) * nqb2, 1, 1), dim3(PATHS_M32P_LOADERS ? 768 : 512), 84 * 1024, stream, q6, k6, v6, o, lse, num_ims, T, Tp, H, npairs, nqb2, o_img, g_m32p_dbg);
#else
    hipLaunchKernelGGL(attn_m32p_kernel, dim3(8 * ((npairs + 7) / 8) * nqb2, 1, 1), dim3(PATHS_M32P_LOADERS ? 768 : 512), 84 * 1024, stream, q6, k6, v6, o, lse, num_ims, T, Tp, H, npairs, nqb2, o_img);
#endif
  } else if (NP == 2 && QT == 2 && m32 == 1) {
    PATHS_LDS_OPT_IN(attn_m32_kernel, 96 * 1024, "attention_x6(32x32x16)");
    hipLaunchKernelGGL(attn_m32_kernel, dim3(8 * ((npairs + 7) / 8) * nqb, 1, 1), dim3(256), lds, stream, q6, k6, v6, o, lse, num_ims, T, Tp, H, npairs, nqb, o_img);
  } else
    hipLaunchKernelGGL((attn_x6_kernel<NP, false>), dim3(8 * ((npairs + 7) / 8) * nqb, 1, 1), dim3(256), lds, stream, q6, k6, v6, o, lse, num_ims, T, Tp, H, npairs, nqb, site, o_img);
#endif
  PATHS_LAUNCH_CHECK("attention_x6");
  return PATHS_OK;
}


}  // namespace

#ifdef PATHS_M32P_STAMPS
extern "C" void paths_attn_pair_debug_buffer(unsigned long long* p) { g_m32p_dbg = p; }
#endif
#ifdef PATHS_ATTN_DEBUG
extern "C" void paths_attn_debug_buffer(unsigned long long* p) { g_attn_dbg = p; }     // development hook (tools/attn_wg_times.py)
#endif

extern "C" {

// bytes of the workspace paths_attention_x6 needs (three fragment images of `planes` 16-bit planes)
int64_t paths_attention_x6_workspace(int B, int T, int H, int head_dim, int planes) {
  const int64_t Tp = ((int64_t)T + KSTEP - 1) / KSTEP * KSTEP;
  return 3 * (int64_t)B * H * Tp * head_dim * 2 * planes;
}

// planes 3: bf16 hi|mid|lo, 6 MFMAs per product block; planes 2: fp16 hi|lo, 3 MFMAs (q, k, v must stay below 65504 in magnitude)
int paths_attention_x6(const float* q, const float* k, const float* v, float* o, float* lse /*[B,H,T] or null*/,
                       const int64_t* num_ims, int B, int T, int H, int head_dim, int max_queries, void* workspace, int planes,
                       int images_ready, hipStream_t stream) {
  PATHS_REQUIRE(head_dim == HD, "attention_x6: head_dim must be %d (got %d)", HD, head_dim);
  PATHS_REQUIRE(!images_ready || planes == 2, "attention_x6: images_ready (workspace filled by paths_token_layer_h3) is the two-plane form");
  PATHS_REQUIRE(images_ready || (q && k && v), "attention_x6: q, k, v are required unless images_ready");
  PATHS_REQUIRE(B > 0 && T > 0 && H > 0 && num_ims != nullptr && workspace != nullptr, "attention_x6: bad arguments B=%d T=%d H=%d", B, T, H);
  PATHS_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)workspace) % 16 == 0, "attention_x6: buffers must be 16-byte aligned");
  PATHS_REQUIRE(planes == 2 || planes == 3, "attention_x6: planes must be 3 (bf16 x6) or 2 (fp16 x3)");
  return planes == 3 ? attention_split<3>(q, k, v, o, lse, num_ims, B, T, H, max_queries, workspace, 0, stream)
                     : attention_split<2>(q, k, v, o, lse, num_ims, B, T, H, max_queries, workspace, images_ready, stream);
}

// paths_attention_x6 with planes = 2 and images_ready (the workspace holds the operand images written by paths_token_layer_ws /
// _h3), with the output written as the fp16 hi | lo fragment image paths_token_layer_ws reads as its out_proj operand
// (B * ceil(T/64) * 64 * H * 32 * 4 bytes: [slide][64-token group][head][16-token tile][plane][64 lanes][16 B]) instead of fp32 o.
int paths_attention_h3_img(void* o_img, const int64_t* num_ims, int B, int T, int H, int head_dim, void* workspace, hipStream_t stream) {
  PATHS_REQUIRE(head_dim == HD && H == 4, "attention_h3_img: head_dim must be %d and H 4 (got %d, %d)", HD, head_dim, H);
  PATHS_REQUIRE(B > 0 && T > 0 && num_ims != nullptr && workspace != nullptr && o_img != nullptr, "attention_h3_img: bad arguments B=%d T=%d", B, T);
  PATHS_REQUIRE(((uintptr_t)o_img | (uintptr_t)workspace) % 16 == 0, "attention_h3_img: buffers must be 16-byte aligned");
  return attention_split<2>(nullptr, nullptr, nullptr, nullptr, nullptr, num_ims, B, T, H, 0, workspace, 1, stream, 0, 0.f, reinterpret_cast<char*>(o_img));
}

// paths_attention_x6 in train mode with dropout p on the attention probabilities (reference nn.MultiheadAttention dropout):
// O = (softmax(S) * mask / (1 - p)) V, lse = the un-dropped log-sum-exp; mask element ((b*H + h)*T + q)*T + k of site `drop_key`.
int paths_attention_x6_dropout(const float* q, const float* k, const float* v, float* o, float* lse, const int64_t* num_ims, int B, int T,
                               int H, int head_dim, int max_queries, void* workspace, int planes, uint64_t drop_key, float drop_p,
                               hipStream_t stream) {
  PATHS_REQUIRE(head_dim == HD && q && k && v, "attention_x6_dropout: head_dim must be %d and q, k, v given", HD);
  PATHS_REQUIRE(B > 0 && T > 0 && H > 0 && num_ims != nullptr && workspace != nullptr, "attention_x6_dropout: bad arguments");
  PATHS_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (planes == 2 || planes == 3), "attention_x6_dropout: p in [0, 1), planes 2 or 3");
  return planes == 3 ? attention_split<3>(q, k, v, o, lse, num_ims, B, T, H, max_queries, workspace, 0, stream, drop_key, drop_p)
                     : attention_split<2>(q, k, v, o, lse, num_ims, B, T, H, max_queries, workspace, 0, stream, drop_key, drop_p);
}

}  // extern "C"
