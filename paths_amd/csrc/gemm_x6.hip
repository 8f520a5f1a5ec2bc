// fp32-accurate GEMM on the CDNA4 bf16 matrix cores ("bf16x6"), with the fused epilogues of gemm_epi.h.
//
//   C[M, N] = [A0 | A1][M, K0+K1] * W[N, K0+K1]^T        same contract as gemm_f32.hip, same epilogues
//
// Why: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  Every fp32 value is the exact sum of three bf16
// (8 significant bits each = the 24 of fp32): x = hi + mid + lo.  Then
//     x*w = hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid) + [mid*lo + lo*mid + lo*lo <= 2^-25 |x w|]
// and the six kept products are exact in the MFMA's fp32 accumulator: 6 v_mfma_f32_32x32x16_bf16 (32 cycles each) replace
// 8 v_mfma_f32_32x32x2_f32 (64 cycles each) per 32x32x16 block = 2.67x the fp32-MFMA peak, with an error that is
// measurably NOT larger than an fp32 FMA chain's (tools/x6_bench.hip: rms 2.8e-7 vs 3.4e-7 against fp64 at K = 1024;
// tests/test_gpu_parity.py::test_x6_gemm_matches_fp64).  Products are accumulated smallest first.
//
// Data movement (what the microbenchmark showed matters on gfx950):
//   * A stays fp32 in HBM (no extra pass, no extra bytes): each thread loads 4 consecutive k of one row, splits them in
//     registers (v_cvt_pk_bf16_f32 + exact residuals, ~22 VALU ops riding in MFMA gaps) and writes the three planes to LDS.
//   * W is split once at pack time (paths_x6_pack_weights) into the MFMA-native tiled image
//         [n/32][k/16][plane 3][k-half 2][n%32][8 bf16]
//     where one 32-row x 16-k x 1-plane fragment is 1 KiB and lane l owns bytes [16 l, 16 l + 16): every W load
//     instruction is 1 KiB contiguous (8 full lines; the row-major split image cost 4x the TA line look-ups and capped
//     the loop at 58 % MFMA busy), every LDS fragment read is lane-linear = conflict-free, no padding.
//   * one workgroup = 4 waves (2 x 2), ONE wave per SIMD with the whole 512-register file (WTM x WTN accumulator tiles
//     in AGPRs), BM x BN = 64 WTM x 64 WTN (256 x 256 for the gate GEMMs), k16 stages double-buffered in LDS, one
//     barrier per stage, global loads issued a full stage (>= 3,000 cycles) before their LDS write.
//   * every non-MFMA instruction is pinned into a gap between two MFMAs (sched_barrier after each), at most two LDS
//     operations per gap; the last accumulator row of a stage runs after the barrier to cover the next stage's first
//     fragment reads.
// Measured (MI355X, M = 16384, N = K = 1024): 152 us = 226 TFLOP/s fp32-equivalent (1.36 PFLOP/s of bf16 MFMA issue,
// 86 % MFMA-busy in the loop at the 1.7-1.9 GHz the chip holds under this load) vs 281 us on the f32 MFMA.
#include "common.h"
#include "gemm_epi.h"
#include "finish_qkv.h"
#ifndef PATHS_X6_MIX16
#define PATHS_X6_MIX16 1
#endif

// PATHS_X6_PART = 1 / 2 / 3: this translation unit defines one third of the C entry points (LSTM cell + weight packing, importance /
// projection, the gemm_nt family) - the file takes 4.5 minutes to compile whole, __graft_entry__.build() compiles the parts in
// parallel; 0 (default): everything in one object
#ifndef PATHS_X6_PART
#define PATHS_X6_PART 0
#endif

namespace {
using namespace paths_epi;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32: round to nearest even
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// fp16 twin of the bf16 helpers (NP == 2 split, see below)
__device__ __forceinline__ uint32_t pk_f16(float a, float b) {       // v_cvt_pk_f16_f32 / 2 x v_cvt_f16_f32: round to nearest even
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ float h_lo(uint32_t p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
__device__ __forceinline__ float h_hi(uint32_t p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }

constexpr int FRAG = 1024;         // bytes of one 32-row x 16-k fragment of one plane (16-bit elements either way)
// NP = operand split: 3 = three bf16 planes hi|mid|lo, 6 MFMAs per product block ("x6", exact fp32 products); 2 = two fp16 planes
// hi|lo, 3 MFMAs ("h3": 22 bits, operands pre-scaled by powers of two); 4 = TWO bf16 planes hi|mid, 3 MFMAs (hi*hi + hi*mid +
// mid*hi: 16 significant bits per operand at fp32's exponent range - no scaling, no overflow: the gradient GEMMs of the training
// step, whose operands span too many binades for a fixed-scale fp16 split; relative error of a product ~2e-5)
template <int NP> constexpr int planes_of() { return NP == 4 ? 2 : NP; }
template <int NP> constexpr int subt() { return planes_of<NP>() * FRAG; }    // one 32-row x 16-k sub-tile

struct X6Operands {
  const float* A0; int64_t lda0; int K0;
  const int64_t* A0rows;           // ROWS kernels: address of every row of panel 0 (rows live wherever they are, e.g. in the slides'
                                   // resident feature grids: no gathered copy); A0 / lda0 unused then
  const float* A1; int64_t lda1; int K1;
  const float* Aadd; int64_t ldadd;   // ADD kernels only: panel 0 is A0 + Aadd, summed in fp32 on the way to the split
  const char* Wt;                  // packed weights, already advanced to the first 32-row group and first k16 step used
  int64_t w_group_stride;          // bytes between consecutive 32-row groups (= K_packed / 16 * SUBT)
  int M;
  const int64_t* num_ims;          // optional padding skip
  int rows_per_slide;
  float a_scale;                   // NP == 2: activations are multiplied by this power of two before the fp16 split
  int ksplit = 1;                  // > 1 (single-panel launches only): blockIdx.z owns the k window [z K0/ksplit, (z+1) K0/ksplit)
#ifdef PATHS_X6_DEBUG
  uint64_t* dbg;                   // tools/x6_stages.py only: per-wave {init, loop, epilogue} shader-clock ticks, 100 MHz ticks, start/end 100 MHz stamps
#endif
};
#ifdef PATHS_X6_DEBUG
uint64_t* g_x6_dbg = nullptr;
#endif

// PF = how many stages ahead of its LDS write a stage is loaded into registers (1 or 2 register sets).  Stages of the
// 128-row tiles are only ~1,500 cycles long, shorter than a loaded-L2 round trip, so those run two stages ahead.
// OCC = waves per SIMD the register allocation is sized for: 1 (the whole 512-register file: the 256-row tiles) or more for the
// small-tile instantiations of epilogue-bound products (K = 256 mem_to_out: several workgroups per CU hide each other's
// epilogue loads / stores, which one wave per SIMD cannot).
template <int NP, int WTM, int WTN, int PF, bool ADD, bool ROWS, class Epi, int OCC = 1>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
gemm_x6_kernel(X6Operands g, Epi epi) {
  static_assert(NP == 2 || NP == 3 || NP == 4, "two fp16 planes, three bf16 planes or two bf16 planes");
  constexpr int PL = planes_of<NP>();                  // planes per operand
  constexpr int SUBT = subt<NP>();
  constexpr int NPROD = NP == 3 ? 6 : 3;               // partial products kept per operand pair
  constexpr int BM = WTM * 64, BN = WTN * 64;
  constexpr int SA = 2 * WTM, SB = 2 * WTN;            // 32-row sub-tiles per block
  constexpr int STAGE = (SA + SB) * SUBT;
  constexpr int NA = BM / 64;                          // fp32 A chunks (4 floats) per thread per stage
  constexpr int NPB = SB * PL, NB = (NPB + 3) / 4;     // 1-KiB W pieces per stage, per wave
  static_assert(WTM % 2 == 0, "A fragment register slots alternate per accumulator row");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order (speed only): XCD x (= linear id % 8) owns a contiguous run of the tile sequence, column-block
  // fastest within groups of GM row-blocks, so the workgroups of one XCD re-use A and W stages from its private L2.
  const int nbx = gridDim.x, nby = gridDim.y, nblk = nbx * nby;
  int lin = blockIdx.y * nbx + blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = lin & 7, j = lin >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;   // bijective for any nblk
  }
  constexpr int GM = 8;
  const int per_group = GM * nbx;
  const int grp = lin / per_group, in_grp = lin - grp * per_group;
  const int rows_in_grp = min(GM, nby - grp * GM);
  const int by = grp * GM + in_grp % rows_in_grp, bx = in_grp / rows_in_grp;
  const int m0 = by * BM, n0 = bx * BN;
  if (block_all_padding(g.num_ims, g.rows_per_slide, m0, BM, g.M)) return;
#ifdef PATHS_X6_DEBUG
  const uint64_t dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
  uint64_t dbg_t1 = 0, dbg_t2 = 0, dbg_s[4] = {0, 0, 0, 0};
#endif

  // ---- staging addresses.  A: thread -> (row = 64 p + tid/4, floats 4 (tid%4) .. +3 of the stage); 32-bit byte offsets
  // against a wave-uniform panel base.  W: wave -> 1-KiB pieces wave + 4 i of the stage's SB x 3 fragments.
  const int arow = tid >> 2, ac = tid & 3;
  uint32_t aoff0[NA], aoff1[NA], aoffs[ADD ? NA : 1]; int awr[NA];
  typedef const f32x4 __attribute__((address_space(1))) * gptr_f4;
  gptr_f4 rowp[ROWS ? NA : 1];                         // ROWS: this thread's piece of each of its rows, advanced one stage at a time
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const int row = p * 64 + arow;
    const int64_t grow = min(m0 + row, g.M - 1);
    aoff0[p] = ROWS ? 0u : (uint32_t)((grow * g.lda0 + 4 * ac) * 4);
    if constexpr (ROWS) rowp[p] = reinterpret_cast<gptr_f4>(static_cast<uintptr_t>(g.A0rows[grow]) + 16 * ac);
    aoff1[p] = (uint32_t)((grow * g.lda1 + 4 * ac) * 4);
    if constexpr (ADD) aoffs[p] = (uint32_t)((grow * g.ldadd + 4 * ac) * 4);
    awr[p] = (row >> 5) * SUBT + (ac >> 1) * 512 + (row & 31) * 16 + (ac & 1) * 8;
  }
  // split-K launches: this block's k window of the single panel (A columns, Aadd columns, W stages all start kofs floats in)
  const int K0w = g.ksplit > 1 ? g.K0 / g.ksplit : g.K0;
  const int kofs = g.ksplit > 1 ? (int)blockIdx.z * K0w : 0;
  if (kofs != 0) {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      aoff0[p] += (uint32_t)kofs * 4u;
      if constexpr (ROWS) rowp[p] += kofs >> 2;
      if constexpr (ADD) aoffs[p] += (uint32_t)kofs * 4u;
    }
  }
  const int nk0 = K0w >> 4, nk = (K0w + g.K1) >> 4;
  const char* bbase[NB]; int bwr[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int pc = wave + 4 * i;
    if (pc >= NPB) pc -= 4;                            // re-stage this wave's previous piece (same bytes, same slot)
    const int sub = pc / PL, pl = pc % PL;
    bbase[i] = g.Wt + ((int64_t)(n0 >> 5) + sub) * g.w_group_stride + pl * FRAG;
    bwr[i] = (SA + sub) * SUBT + pl * FRAG + lane * 16;
  }
  static_assert(PF == 1 || PF == 2, "one or two register sets");
  f32x4 sa[PF][NA], sadd[ADD ? PF : 1][ADD ? NA : 1]; u32x4 sbr[PF][NB];
  uint32_t hi[NA][2], mid[NP != 2 ? NA : 1][2], lo[NA][2];
  // Loads go through buffer descriptors: a wave-uniform base (SGPRs), a scalar byte offset (stage, panel, W piece) and ONE
  // per-lane 32-bit offset computed once.  As 64-bit pointers hipcc strength-reduced each load's address into a VGPR pair it
  // then bumped with 3-4 VALU instructions per load; packed into the MFMA gaps of the 128-row tiles that stretched the first
  // gaps of every stage to ~60 cycles.
  float* const anyA = const_cast<float*>(ROWS ? reinterpret_cast<const float*>(g.Wt) : g.A0);   // (descriptor placeholder when unused)
  const auto rsA0 = __builtin_amdgcn_make_buffer_rsrc(anyA, 0, 0xFFFFFFFFu, 0x00020000);
  const auto rsA1 = __builtin_amdgcn_make_buffer_rsrc(g.A1 ? const_cast<float*>(g.A1) : anyA, 0, 0xFFFFFFFFu, 0x00020000);
  const auto rsAdd = __builtin_amdgcn_make_buffer_rsrc(ADD ? const_cast<float*>(g.Aadd) : anyA, 0, 0xFFFFFFFFu, 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g.Wt), 0, 0xFFFFFFFFu, 0x00020000);
  const int lane16 = lane * 16;
  int bsoff[NB];                                       // scalar byte offset of W piece i at stage 0
#pragma unroll
  for (int i = 0; i < NB; ++i) bsoff[i] = (int)(bbase[i] - g.Wt) + (kofs >> 4) * SUBT;
  auto gload_a = [&](int set, int q, int kt) {
    const bool first = kt < nk0;                       // wave-uniform panel select
#ifdef PATHS_X6_EXP_HOTA
    const int soff = 0;                                // experiment: every stage re-reads stage 0 of A (cache-hot): is the pre-barrier loss load latency?
#else
    const int soff = (first ? kt : kt - nk0) * 64;
#endif
    if constexpr (ROWS) {                              // single panel: 64-bit per-lane row address + the stage's 64 bytes
      sa[set][q] = rowp[q][kt * 4];
    } else {
      const u32x4 raw = first ? __builtin_amdgcn_raw_buffer_load_b128(rsA0, (int)aoff0[q], soff, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(rsA1, (int)aoff1[q], soff, 0);
      sa[set][q] = __builtin_bit_cast(f32x4, raw);
    }
    if constexpr (ADD) {                               // (ADD kernels have a single panel)
      sadd[set][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsAdd, (int)aoffs[q], kt * 64, 0));
    }
  };
  auto gload_b = [&](int set, int q, int kt) {
    sbr[set][q] = __builtin_amdgcn_raw_buffer_load_b128(rsW, lane16, bsoff[q] + kt * SUBT, 0);
  };
  // Split of one staged A chunk in NS micro-steps of at most 2 VALU instructions (the last one: the three LDS writes).  A gap
  // between two 32-cycle MFMAs hides about 24 cycles of other issue; the first version used 7 steps of 4 VALU + waits, every
  // such gap overflowed by ~15 cycles and idle gaps cannot win that back: ~500 cycles per stage (tools/x6_stages.py).
  //   NP == 3 (bf16): 12 micro-steps (+1 for the ADD sum);  NP == 2 (fp16): scale, hi, lo low halves, lo high halves, writes = 5 (+1).
  constexpr int NS = (NP == 3 ? 12 : NP == 4 ? 7 : (PATHS_X6_MIX16 ? 5 : 6)) + (ADD ? 1 : 0);
  float tf[NA][2];
  auto a_step = [&](int set, int q, int st0, int buf) __attribute__((always_inline)) {
    f32x4& v = sa[set][q];
    const int st = ADD ? st0 - 1 : st0;
    if constexpr (ADD) {
      if (st0 == 0) v += sadd[set][q];                 // the fp32 sum the reference materialises (Y = X + h1)
    }
    if constexpr (NP == 3) {
      if (st == 0) { hi[q][0] = pk_bf16(v[0], v[1]); hi[q][1] = pk_bf16(v[2], v[3]); }
      if (st == 1) { tf[q][0] = bf_lo(hi[q][0]); tf[q][1] = bf_hi(hi[q][0]); }
      if (st == 2) { v[0] -= tf[q][0]; v[1] -= tf[q][1]; }
      if (st == 3) { tf[q][0] = bf_lo(hi[q][1]); tf[q][1] = bf_hi(hi[q][1]); }
      if (st == 4) { v[2] -= tf[q][0]; v[3] -= tf[q][1]; }
      if (st == 5) { mid[q][0] = pk_bf16(v[0], v[1]); mid[q][1] = pk_bf16(v[2], v[3]); }
      if (st == 6) { tf[q][0] = bf_lo(mid[q][0]); tf[q][1] = bf_hi(mid[q][0]); }
      if (st == 7) { v[0] -= tf[q][0]; v[1] -= tf[q][1]; }
      if (st == 8) { tf[q][0] = bf_lo(mid[q][1]); tf[q][1] = bf_hi(mid[q][1]); }
      if (st == 9) { v[2] -= tf[q][0]; v[3] -= tf[q][1]; }
      if (st == 10) { lo[q][0] = pk_bf16(v[0], v[1]); lo[q][1] = pk_bf16(v[2], v[3]); }
      if (st == 11) {
        char* d = smem + buf * STAGE + awr[q];
        *reinterpret_cast<u32x2*>(d) = u32x2{hi[q][0], hi[q][1]};
        *reinterpret_cast<u32x2*>(d + FRAG) = u32x2{mid[q][0], mid[q][1]};
        *reinterpret_cast<u32x2*>(d + 2 * FRAG) = u32x2{lo[q][0], lo[q][1]};
      }
    } else if constexpr (NP == 4) {                     // hi | mid of the three-plane split: 16 significant bits, no scale
      if (st == 0) { hi[q][0] = pk_bf16(v[0], v[1]); hi[q][1] = pk_bf16(v[2], v[3]); }
      if (st == 1) { tf[q][0] = bf_lo(hi[q][0]); tf[q][1] = bf_hi(hi[q][0]); }
      if (st == 2) { v[0] -= tf[q][0]; v[1] -= tf[q][1]; }
      if (st == 3) { tf[q][0] = bf_lo(hi[q][1]); tf[q][1] = bf_hi(hi[q][1]); }
      if (st == 4) { v[2] -= tf[q][0]; v[3] -= tf[q][1]; }
      if (st == 5) { mid[q][0] = pk_bf16(v[0], v[1]); mid[q][1] = pk_bf16(v[2], v[3]); }
      if (st == 6) {
        char* d = smem + buf * STAGE + awr[q];
        *reinterpret_cast<u32x2*>(d) = u32x2{hi[q][0], hi[q][1]};
        *reinterpret_cast<u32x2*>(d + FRAG) = u32x2{mid[q][0], mid[q][1]};
      }
    } else {
#ifndef PATHS_X6_WHATIF_NOSPLIT
      if (st == 0) v *= g.a_scale;                      // power of two: exact
#endif
      if (st == 1) { hi[q][0] = pk_f16(v[0], v[1]); hi[q][1] = pk_f16(v[2], v[3]); }
      // lo plane: residual AND its rounding to fp16 in one instruction per value (v_fma_mixlo_f16 / v_fma_mixhi_f16 write one half
      // of the packed register each; round 4: was 2 x v_fma_mix_f32 + v_cvt_pk per pair).  The two halves of a register go to
      // different micro-steps: a half-register write directly in front of the other half's costs a wait state.
#if defined(PATHS_X6_WHATIF_NOSPLIT)   // diagnostic build, WRONG results: what a pre-split resident image could save at most (no scale, no lo plane work)
      if (st == 2) { lo[q][0] = hi[q][0]; lo[q][1] = hi[q][1]; }
      if (st == 4) {
#elif PATHS_X6_MIX16
      if (st == 2) {
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo[q][0]) : "v"(hi[q][0]), "v"(v[0]));
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo[q][1]) : "v"(hi[q][1]), "v"(v[2]));
      }
      if (st == 3) {
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo[q][0]) : "v"(hi[q][0]), "v"(v[1]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo[q][1]) : "v"(hi[q][1]), "v"(v[3]));
      }
      if (st == 4) {
#else   // (round-3 form, kept for A/B builds: -DPATHS_X6_MIX16=0)
      if (st == 2) f16_pair_residuals(hi[q][0], v[0], v[1], tf[q][0], tf[q][1]);
      if (st == 3) { float r2, r3; f16_pair_residuals(hi[q][1], v[2], v[3], r2, r3); v[2] = r2; v[3] = r3; }
      if (st == 4) { lo[q][0] = pk_f16(tf[q][0], tf[q][1]); lo[q][1] = pk_f16(v[2], v[3]); }
      if (st == 5) {
#endif
        char* d = smem + buf * STAGE + awr[q];
        *reinterpret_cast<u32x2*>(d) = u32x2{hi[q][0], hi[q][1]};
        *reinterpret_cast<u32x2*>(d + FRAG) = u32x2{lo[q][0], lo[q][1]};
      }
    }
  };
  auto swrite_b = [&](int set, int q, int buf) { *reinterpret_cast<u32x4*>(smem + buf * STAGE + bwr[q]) = sbr[set][q]; };

  // accumulators start from the epilogue's initial value (zero, or the once-per-parent partial pre-activation)
  f32x16 acc[WTM][WTN];
  epi.template init<WTM, WTN>(acc, m0 + wm * WTM * 32, n0 + wn * WTN * 32, lane, g.M);

  const char* sA = smem + (wm * WTM) * SUBT + lane * 16;
  const char* sB = smem + (SA + wn * WTN) * SUBT + lane * 16;
  u32x4 fa[2][PL], fb[2][WTN][PL];                     // 8 x 16-bit per lane and plane
  auto read_a = [&](int buf, int i, int slot, int p) { fa[slot][p] = *reinterpret_cast<const u32x4*>(sA + buf * STAGE + i * SUBT + p * FRAG); };
  auto read_b = [&](int buf, int j, int slot, int p) { fb[slot][j][p] = *reinterpret_cast<const u32x4*>(sB + buf * STAGE + j * SUBT + p * FRAG); };
  constexpr int RG = WTN * NPROD;                      // MFMA gaps per accumulator row
  auto one_mfma = [&](int gq, int sb) __attribute__((always_inline)) {
    const int i = gq / RG, j = (gq % RG) / NPROD, t = gq % NPROD, sl = i & 1;
    if constexpr (NP == 3) {
      constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};   // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[sl][PA_[t]]), __builtin_bit_cast(bf16x8, fb[sb][j][PB_[t]]), acc[i][j], 0, 0, 0);
    } else if constexpr (NP == 4) {
      constexpr int PA_[3] = {1, 0, 0}, PB_[3] = {0, 1, 0};                     // mid*hi, hi*mid, hi*hi
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[sl][PA_[t]]), __builtin_bit_cast(bf16x8, fb[sb][j][PB_[t]]), acc[i][j], 0, 0, 0);
    } else {
      constexpr int PA_[3] = {1, 0, 0}, PB_[3] = {0, 1, 0};                     // lo*hi, hi*lo, hi*hi
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[sl][PA_[t]]), __builtin_bit_cast(f16x8, fb[sb][j][PB_[t]]), acc[i][j], 0, 0, 0);
    }
  };
  // Staging slots, one MFMA gap each, SPG micro-steps to a gap (the smallest SPG that fits before the barrier).
  //   PF == 1: A chunk q owns GPC consecutive gaps (its micro-steps; its reload with stage kt+2 shares the last one, so
  //            the load has a full stage), then W piece q one gap (LDS write + reload).
  //   PF == 2 (128-row tiles: only one accumulator row precedes the barrier): all micro-steps packed SPG to a gap, then the
  //            W writes - everything that must precede the barrier - then the NA + NB reloads, which need not.
  constexpr int AG = (WTM - 1) * RG;
  constexpr auto fits = [](int spg) constexpr {
    return PF == 1 ? ((NS + spg - 1) / spg) * NA + NB <= AG : (NS * NA + spg - 1) / spg + NB <= AG;
  };
  constexpr int SPG = fits(1) ? 1 : fits(2) ? 2 : fits(3) ? 3 : 4;
  static_assert(fits(SPG), "staging does not fit before the barrier");
  constexpr int GPC = (NS + SPG - 1) / SPG;            // PF == 1: gaps per A chunk
  constexpr int GA2 = (NS * NA + SPG - 1) / SPG;       // PF == 2: gaps of all A micro-steps
  constexpr int TS = PF == 1 ? GPC * NA + NB : GA2 + NB + NA + NB;
  static_assert(TS <= WTM * RG, "staging does not fit in a stage");
  constexpr int AR0 = RG / 2;                          // gaps AR0 .. AR0+2 of row i: fragment reads of A row i+1
  auto staging_slot = [&](auto sc, int kt, auto bufc, auto m1c, auto m2c) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value, buf = decltype(bufc)::value;
    constexpr bool more1 = decltype(m1c)::value, more2 = decltype(m2c)::value;
    constexpr int set = PF == 2 ? (buf ^ 1) : 0;       // register set holding stage kt+1 (reloaded with stage kt+1+PF)
    if constexpr (PF == 1) {
      if constexpr (s < GPC * NA) {
        constexpr int q = s / GPC, g0 = s % GPC;
        if constexpr (more1) {
          static_for<g0 * SPG, (g0 + 1) * SPG < NS ? (g0 + 1) * SPG : NS>([&](auto mc) __attribute__((always_inline)) {
            a_step(set, q, decltype(mc)::value, buf ^ 1);
          });
        }
        if constexpr (more2 && g0 == GPC - 1) gload_a(set, q, kt + 1 + PF);
      } else if constexpr (s < TS) {
        if constexpr (more1) swrite_b(set, s - GPC * NA, buf ^ 1);
        if constexpr (more2) gload_b(set, s - GPC * NA, kt + 1 + PF);
      }
    } else {
      if constexpr (s < GA2) {
        if constexpr (more1) {
          static_for<s * SPG, (s + 1) * SPG < NS * NA ? (s + 1) * SPG : NS * NA>([&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value;
            a_step(set, m / NS, m % NS, buf ^ 1);
          });
        }
      } else if constexpr (s < GA2 + NB) {
        if constexpr (more1) swrite_b(set, s - GA2, buf ^ 1);
      } else if constexpr (s < GA2 + NB + NA) {
        if constexpr (more2) gload_a(set, s - GA2 - NB, kt + 1 + PF);
      } else if constexpr (s < TS) {
        if constexpr (more2) gload_b(set, s - GA2 - NB - NA, kt + 1 + PF);
      }
    }
  };
  auto stage_body = [&](int kt, auto bufc, auto m1c, auto m2c) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value, sb = buf;
    constexpr bool more1 = decltype(m1c)::value, more2 = decltype(m2c)::value;
    constexpr int set = PF == 2 ? (buf ^ 1) : 0;       // register set holding stage kt+1 (reloaded with stage kt+1+PF)
#ifdef PATHS_X6_DEBUG
    if (kt == 10) dbg_s[0] = __builtin_amdgcn_s_memtime();
#endif
    static_for<0, AG>([&](auto gc) __attribute__((always_inline)) {
      constexpr int gq = decltype(gc)::value, i = gq / RG, gr = gq % RG;
      one_mfma(gq, sb);
      if constexpr (gr >= AR0 && gr < AR0 + PL) read_a(buf, i + 1, (i + 1) & 1, gr - AR0);
      staging_slot(std::integral_constant<int, gq>{}, kt, bufc, m1c, m2c);
      __builtin_amdgcn_sched_barrier(0);
    });
#ifdef PATHS_X6_DEBUG
    if (kt == 10) dbg_s[1] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef PATHS_X6_DEBUG
    if (kt == 10) dbg_s[2] = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_sched_barrier(0);
    static_for<AG, WTM * RG>([&](auto gc) __attribute__((always_inline)) {
      constexpr int gq = decltype(gc)::value, gr = gq % RG;
      one_mfma(gq, sb);
      if constexpr (more1) {
        static_for<2 * gr, 2 * gr + 2>([&](auto fc) __attribute__((always_inline)) {
          constexpr int f = decltype(fc)::value;
          if constexpr (f < PL) read_a(buf ^ 1, 0, 0, f);
          else if constexpr (f < PL + PL * WTN) read_b(buf ^ 1, (f - PL) / PL, sb ^ 1, (f - PL) % PL);
        });
      }
      staging_slot(std::integral_constant<int, gq>{}, kt, bufc, m1c, m2c);   // (reload slots of the PF == 2 layout)
      __builtin_amdgcn_sched_barrier(0);
    });
#ifdef PATHS_X6_DEBUG
    if (kt == 10) dbg_s[3] = __builtin_amdgcn_s_memtime();
#endif
  };
  constexpr std::integral_constant<int, 0> I0{};
  constexpr std::integral_constant<int, 1> I1{};
  constexpr std::true_type T{};
  constexpr std::false_type F{};

  // prologue: stage 0 -> LDS buffer 0, stages 1 .. PF -> register sets
#pragma unroll
  for (int q = 0; q < NA; ++q) gload_a(0, q, 0);
#pragma unroll
  for (int q = 0; q < NB; ++q) gload_b(0, q, 0);
#pragma unroll
  for (int q = 0; q < NA; ++q)
#pragma unroll
    for (int st = 0; st < NS; ++st) a_step(0, q, st, 0);
#pragma unroll
  for (int q = 0; q < NB; ++q) swrite_b(0, q, 0);
#pragma unroll
  for (int sidx = 1; sidx <= PF; ++sidx) {
#pragma unroll
    for (int q = 0; q < NA; ++q) gload_a(PF == 2 ? (sidx & 1) : 0, q, sidx);
#pragma unroll
    for (int q = 0; q < NB; ++q) gload_b(PF == 2 ? (sidx & 1) : 0, q, sidx);
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < PL; ++p) read_a(0, 0, 0, p);
#pragma unroll
  for (int j = 0; j < WTN; ++j)
#pragma unroll
    for (int p = 0; p < PL; ++p) read_b(0, j, 0, p);
#ifdef PATHS_X6_DEBUG
  dbg_t1 = __builtin_amdgcn_s_memtime();
#endif
  for (int kt = 0; kt < nk - 2 * PF; kt += 2) {
    __builtin_amdgcn_sched_barrier(0);
    stage_body(kt, I0, T, T);
    stage_body(kt + 1, I1, T, T);
  }
  if constexpr (PF == 2) {
    stage_body(nk - 4, I0, T, T);
    stage_body(nk - 3, I1, T, F);
  }
  stage_body(nk - 2, I0, T, F);
  stage_body(nk - 1, I1, F, F);
#ifdef PATHS_X6_DEBUG
  dbg_t2 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();     // epilogues may reuse LDS
  epi.template run<WTM, WTN, 2, 2>(acc, m0 + wm * WTM * 32, n0 + wn * WTN * 32, lane, wm, wn, g.M, reinterpret_cast<float*>(smem));
#ifdef PATHS_X6_DEBUG
  if (g.dbg) {
    __builtin_amdgcn_s_waitcnt(0);
    const uint64_t t3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      uint64_t* d = g.dbg + 9 * ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave);
      d[0] = dbg_t1 - dbg_t0; d[1] = dbg_t2 - dbg_t1; d[2] = t3 - dbg_t2; d[3] = r3 - dbg_r0; d[4] = dbg_r0; d[5] = r3;
      d[6] = dbg_s[1] - dbg_s[0]; d[7] = dbg_s[2] - dbg_s[1]; d[8] = dbg_s[3] - dbg_s[2];   // stage 10: pre-barrier, barrier wait, post-barrier
    }
  }
#endif
}

// fp32 [N, K] (row stride ldw) -> split tiled image, rows >= N zero.  NP == 2: values are multiplied by wscale (a power of two) first.
template <int NP>
__global__ void x6_pack_kernel(const float* __restrict__ w, int64_t ldw, uint16_t* __restrict__ out, int N, int Npad, int K, float wscale) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)Npad * K) return;
  const int n = (int)(i / K), k = (int)(i % K);
  const float v = n < N ? w[(int64_t)n * ldw + k] * wscale : 0.f;
  uint16_t* o = out + ((int64_t)(n >> 5) * (K >> 4) + (k >> 4)) * (subt<NP>() / 2) + ((k >> 3) & 1) * 256 + (n & 31) * 8 + (k & 7);
  if constexpr (NP == 3) {
    const __bf16 h = (__bf16)v; const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1; const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    o[0] = __builtin_bit_cast(uint16_t, h); o[FRAG / 2] = __builtin_bit_cast(uint16_t, m); o[FRAG] = __builtin_bit_cast(uint16_t, l);
  } else if constexpr (NP == 4) {
    const __bf16 h = (__bf16)v; const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    o[0] = __builtin_bit_cast(uint16_t, h); o[FRAG / 2] = __builtin_bit_cast(uint16_t, m);
  } else {
    const _Float16 h = (_Float16)v; const float r1 = v - (float)h;
    const _Float16 l = (_Float16)r1;
    o[0] = __builtin_bit_cast(uint16_t, h); o[FRAG / 2] = __builtin_bit_cast(uint16_t, l);
  }
}

// The same image from the TRANSPOSED source: element (n, k) = w[k * ldw + n] (w is [Kvalid, N] row-major; k >= Kvalid and n >= N are zero).
// One thread per (n, eight consecutive k): eight reads coalesced across the lanes' n, one 16-byte write per plane.  The dX products of the
// backward (dX = dY W) take W^T as their weight: this replaces a transpose launch + a pack launch per product by one.
template <int NP>
__global__ void x6_pack_t_kernel(const float* __restrict__ w, int64_t ldw, uint16_t* __restrict__ out, int N, int Npad, int K, int Kvalid) {
  static_assert(NP == 3 || NP == 4, "bf16 planes (scale 1)");
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)Npad * (K >> 3)) return;
  const int n = (int)(i % Npad), k0 = (int)(i / Npad) * 8;
  uint16_t h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + j;
    const float v = (n < N && k < Kvalid) ? w[(int64_t)k * ldw + n] : 0.f;
    const __bf16 hh = (__bf16)v; const float r1 = v - (float)hh;
    const __bf16 mm = (__bf16)r1; const float r2 = r1 - (float)mm;
    const __bf16 ll = (__bf16)r2;
    h[j] = __builtin_bit_cast(uint16_t, hh); m[j] = __builtin_bit_cast(uint16_t, mm); l[j] = __builtin_bit_cast(uint16_t, ll);
  }
  uint16_t* o = out + ((int64_t)(n >> 5) * (K >> 4) + (k0 >> 4)) * (subt<NP>() / 2) + ((k0 >> 3) & 1) * 256 + (n & 31) * 8;
  typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
  *reinterpret_cast<u16x8*>(o) = u16x8{h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
  *reinterpret_cast<u16x8*>(o + FRAG / 2) = u16x8{m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]};
  if constexpr (NP == 3) *reinterpret_cast<u16x8*>(o + FRAG) = u16x8{l[0], l[1], l[2], l[3], l[4], l[5], l[6], l[7]};
}

template <int NP, int WTM, int WTN, int PF, bool ADD, bool ROWS, class Epi, int OCC = 1>
int launch_x6_np(const X6Operands& g, int Npad, const Epi& epi, hipStream_t stream, const char* name) {
  constexpr int BM = WTM * 64, BN = WTN * 64;
  constexpr size_t lds = 2ull * (2 * WTM + 2 * WTN) * subt<NP>();
  const int K = g.K0 + g.K1;
  PATHS_REQUIRE(g.M > 0, "%s: M must be > 0", name);
  PATHS_REQUIRE(g.K0 > 0 && g.K0 % 16 == 0 && g.K1 % 16 == 0 && K % 32 == 0 && K >= 128, "%s: K panels (%d,%d): multiples of 16, total a multiple of 32, >= 128", name, g.K0, g.K1);
  PATHS_REQUIRE(Npad % BN == 0, "%s: packed N (%d) must be a multiple of %d", name, Npad, BN);
  PATHS_REQUIRE(g.lda0 % 4 == 0 && g.lda1 % 4 == 0, "%s: leading dims must be multiples of 4 floats", name);
  PATHS_REQUIRE(((uintptr_t)g.A0 % 16 == 0) && ((uintptr_t)g.A1 % 16 == 0) && ((uintptr_t)g.Wt % 16 == 0), "%s: operands must be 16-byte aligned", name);
  PATHS_REQUIRE(ROWS == (g.A0rows != nullptr) && (!ROWS || g.K1 == 0), "%s: row-pointer form is single-panel", name);
  PATHS_REQUIRE((int64_t)g.M * (g.lda0 > g.lda1 ? (g.lda0 > g.ldadd ? g.lda0 : g.ldadd) : (g.lda1 > g.ldadd ? g.lda1 : g.ldadd)) * 4 < (int64_t)1 << 32, "%s: A panel larger than 4 GiB", name);
  auto kern = gemm_x6_kernel<NP, WTM, WTN, PF, ADD, ROWS, Epi, OCC>;
  PATHS_LDS_OPT_IN(kern, lds, name);
  PATHS_REQUIRE(g.ksplit == 1 || (g.ksplit > 1 && g.K1 == 0 && g.K0 % (32 * g.ksplit) == 0 && g.K0 / g.ksplit >= 128), "%s: split-K needs a single panel and k windows that are multiples of 32, >= 128", name);
  dim3 grid(Npad / BN, (g.M + BM - 1) / BM, g.ksplit);
#ifdef PATHS_X6_DEBUG
  X6Operands gd = g; gd.dbg = g_x6_dbg;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, gd, epi);
#else
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, g, epi);
#endif
  PATHS_LAUNCH_CHECK(name);
  return PATHS_OK;
}

// ---- split-K in two launches.  A GEMM whose tile grid covers half the chip (importance/proj: M/128 x 1 blocks) runs its k loop
// as ksplit x as many blocks, each storing its RAW accumulators (EpiRaw: one [4][64 lanes][4] slab per 32x32 tile, the register
// layout itself, so stores and loads are 16 bytes per lane and fully coalesced); x6_finish_kernel then sums the ksplit slabs
// into registers and runs the real epilogue on 64-row blocks (2x the blocks again).  No inter-block waiting anywhere.
struct EpiRaw {
  float* ws; int64_t zstride; int ntn;       // ws [ksplit][M_pad/32][ntn = N_pad/32][1024] floats
  template <int WTM, int WTN>
  __device__ __forceinline__ void init(f32x16 (&acc)[WTM][WTN], int, int, int, int) const {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  template <int WTM, int WTN, int WGM, int WGN>
  __device__ __forceinline__ void run(f32x16 (&acc)[WTM][WTN], int row0, int col0, int lane, int, int, int, float*) const {
    float* base = ws + (int64_t)blockIdx.z * zstride;
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        f32x4* t = reinterpret_cast<f32x4*>(base + ((int64_t)((row0 >> 5) + i) * ntn + (col0 >> 5) + j) * 1024) + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
#ifdef PATHS_RAW_NT
          __builtin_nontemporal_store(v, t + 64 * q);     // experiment: streaming stores leave no dirty lines for the kernel-end write-back
#else
          t[64 * q] = v;
#endif
        }
      }
  }
};

template <int NZ, class Epi>
__global__ void __launch_bounds__(256)
x6_finish_kernel(const float* __restrict__ ws, int64_t zstride, int ntn, int M, const int64_t* __restrict__ num_ims, int rows_per_slide, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];       // 1 KiB used; the launcher pads the request (one workgroup per CU)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * 64;
  if (block_all_padding(num_ims, rows_per_slide, m0, 64, M)) return;
  const int row0 = m0 + wm * 32, col0 = wn * 128;
  f32x16 acc[1][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4* t = reinterpret_cast<const f32x4*>(ws + ((int64_t)(row0 >> 5) * ntn + (col0 >> 5) + j) * 1024) + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = t[64 * q];
#pragma unroll
      for (int z = 1; z < NZ; ++z) v += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(t + 64 * q) + z * zstride);
      acc[0][j][4 * q] = v[0]; acc[0][j][4 * q + 1] = v[1]; acc[0][j][4 * q + 2] = v[2]; acc[0][j][4 * q + 3] = v[3];
    }
  }
  epi.template run<1, 4, 2, 2>(acc, row0, col0, lane, wm, wn, M, reinterpret_cast<float*>(smem));
}

// planes = 3: bf16 x6; planes = 2: fp16 x3 (operands pre-scaled by powers of two, undone through Epi::acc_scale)
template <int WTM, int WTN, int PF, bool ADD, class Epi>
int launch_x6(int planes, const X6Operands& g, int Npad, const Epi& epi, hipStream_t stream, const char* name) {
  if (g.A0rows != nullptr) {       // row-pointer form: the default (two-plane) split only
    if (planes == 2) return launch_x6_np<2, WTM, WTN, PF, ADD, true>(g, Npad, epi, stream, name);
    return paths_set_error(PATHS_EUNSUPPORTED, "%s: row pointers need planes = 2", name);
  }
  if constexpr (std::is_same<Epi, EpiBias>::value && !ADD) {      // two bf16 planes: the plain NT products only (training's dX GEMMs)
    if (planes == 4) return launch_x6_np<4, WTM, WTN, PF, ADD, false>(g, Npad, epi, stream, name);
  }
  if (planes == 3) return launch_x6_np<3, WTM, WTN, PF, ADD, false>(g, Npad, epi, stream, name);
  if (planes == 2) return launch_x6_np<2, WTM, WTN, PF, ADD, false>(g, Npad, epi, stream, name);
  return paths_set_error(PATHS_EINVAL, "%s: planes must be 3 (bf16 x6) or 2 (fp16 x3), got %d", name, planes);
}

#ifndef PATHS_H_OCC
#define PATHS_H_OCC 3
#endif
constexpr int H_OCC = PATHS_H_OCC;
#ifndef PATHS_H_TRAIN_OCC
#define PATHS_H_TRAIN_OCC 2
#endif
constexpr int H_TRAIN_OCC = PATHS_H_TRAIN_OCC;
static const bool H_TRAIN_SMALL = getenv("PATHS_H_TRAIN_SMALL") == nullptr || atoi(getenv("PATHS_H_TRAIN_SMALL")) != 0;   // A/B switch
static const bool FIN_XCD_ORDER = getenv("PATHS_FIN_XCD_ORDER") == nullptr || atoi(getenv("PATHS_FIN_XCD_ORDER")) != 0;   // A/B switch
static const bool IP_TILE128 = getenv("PATHS_IP_TILE128") != nullptr && atoi(getenv("PATHS_IP_TILE128")) != 0;   // measured: 69 us vs 58 (split-K)
static const bool O_RAW = getenv("PATHS_O_RAW") == nullptr || atoi(getenv("PATHS_O_RAW")) != 0;
static const bool H_SMALL_TILES = getenv("PATHS_H_SMALL_TILES") == nullptr || atoi(getenv("PATHS_H_SMALL_TILES")) != 0;

inline int plane_count(int planes) { return planes == 4 ? 2 : planes; }      // API mode 4 = two bf16 planes
inline int64_t group_stride(int planes, int Kpacked) { return (int64_t)(Kpacked / 16) * plane_count(planes) * FRAG; }
inline bool pow2(float x) { int e; return x > 0.f && frexpf(x, &e) == 0.5f; }

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

#if defined(PATHS_X6_DEBUG) && (PATHS_X6_PART == 0 || PATHS_X6_PART == 1)
// development hook (tools/x6_stages.py, debug build only): buffer of 9 uint64 per wave, or NULL
void paths_x6_debug_buffer(uint64_t* p) { g_x6_dbg = p; }
#endif

#if PATHS_X6_PART == 0 || PATHS_X6_PART == 1
// bytes of the packed image of an [Npad, K] weight: planes x 2 bytes per element
int64_t paths_x6_packed_bytes(int Npad, int K, int planes) { return (int64_t)Npad * K * 2 * plane_count(planes); }

// w [N, K] fp32 (row stride ldw) -> out (paths_x6_packed_bytes bytes); Npad % 32 == 0, K % 16 == 0.
// planes 3: exact bf16 hi|mid|lo (w_scale must be 1); planes 2: fp16 hi|lo of w * w_scale (w_scale a power of two chosen by
// the caller so that max|w| * w_scale < 65504; the same w_scale is passed to every kernel that consumes the image).
int paths_x6_pack_weights(const float* w, int64_t ldw, void* out, int N, int Npad, int K, int planes, float w_scale, hipStream_t stream) {
  PATHS_REQUIRE(N > 0 && Npad >= N && Npad % 32 == 0 && K % 16 == 0, "x6_pack_weights: bad shape N=%d Npad=%d K=%d", N, Npad, K);
  PATHS_REQUIRE((uintptr_t)out % 16 == 0, "x6_pack_weights: out must be 16-byte aligned");
  PATHS_REQUIRE(((planes == 3 || planes == 4) && w_scale == 1.0f) || (planes == 2 && pow2(w_scale)), "x6_pack_weights: planes 3 / 4 (bf16) with w_scale 1, or planes 2 with a power-of-two w_scale");
  const int64_t n = (int64_t)Npad * K;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (planes == 4) hipLaunchKernelGGL(x6_pack_kernel<4>, grid, dim3(256), 0, stream, w, ldw, reinterpret_cast<uint16_t*>(out), N, Npad, K, 1.0f);
  else if (planes == 3) hipLaunchKernelGGL(x6_pack_kernel<3>, grid, dim3(256), 0, stream, w, ldw, reinterpret_cast<uint16_t*>(out), N, Npad, K, 1.0f);
  else hipLaunchKernelGGL(x6_pack_kernel<2>, grid, dim3(256), 0, stream, w, ldw, reinterpret_cast<uint16_t*>(out), N, Npad, K, w_scale);
  PATHS_LAUNCH_CHECK("x6_pack_weights");
  return PATHS_OK;
}

// The image of W^T from W: w is [Kvalid, N] fp32 (row stride ldw), the packed weight is its transpose [N, K] with zero rows n >= N and
// zero columns k >= Kvalid (K % 16 == 0); planes 3 or 4 (bf16, scale 1).
int paths_x6_pack_weights_t(const float* w, int64_t ldw, void* out, int N, int Npad, int K, int Kvalid, int planes, hipStream_t stream) {
  PATHS_REQUIRE(N > 0 && Npad >= N && Npad % 32 == 0 && K % 16 == 0 && Kvalid > 0 && Kvalid <= K, "x6_pack_weights_t: bad shape N=%d Npad=%d K=%d Kvalid=%d", N, Npad, K, Kvalid);
  PATHS_REQUIRE(w != nullptr && out != nullptr && (uintptr_t)out % 16 == 0, "x6_pack_weights_t: null / unaligned operand");
  PATHS_REQUIRE(planes == 3 || planes == 4, "x6_pack_weights_t: planes 3 or 4 (bf16)");
  const int64_t n = (int64_t)Npad * (K / 8);
  const dim3 grid((unsigned)((n + 255) / 256));
  if (planes == 4) hipLaunchKernelGGL(x6_pack_t_kernel<4>, grid, dim3(256), 0, stream, w, ldw, reinterpret_cast<uint16_t*>(out), N, Npad, K, Kvalid);
  else hipLaunchKernelGGL(x6_pack_t_kernel<3>, grid, dim3(256), 0, stream, w, ldw, reinterpret_cast<uint16_t*>(out), N, Npad, K, Kvalid);
  PATHS_LAUNCH_CHECK("x6_pack_weights_t");
  return PATHS_OK;
}

// paths_lstm_cell with the gate / mem_to_out weights given as split images (planes, scales as in paths_x6_pack_weights;
// a_scale: power of two applied to the fp32 activations before the fp16 split - |activation| * a_scale must stay < 65504):
//   w_gates_x6 = pack([3Hc + D, 2D] packed gate rows, see paths_lstm_cell), w_mem_x6 = pack([D, Hc])
int paths_lstm_cell_x6(const float* x, int64_t ldx, const int64_t* x_rows, const float* h0, int64_t ldh0, const float* c0, int64_t ldc0,
                       const void* w_gates_x6, const float* b_gates, const void* w_mem_x6, const float* b_mem,
                       float* state_out, int64_t ldso, float* y, int64_t ldy, float* ws_o, float* save_frm, float* save_tc,
                       const float* hp, const int* hp_row, int M, int D, int Hc, const int64_t* num_ims, int rows_per_slide,
                       int phases, int planes, float wg_scale, float wm_scale, float a_scale, hipStream_t stream) {
  PATHS_REQUIRE(D % 256 == 0 && Hc % 64 == 0, "lstm_cell_x6: D (%d) must be a multiple of 256 and Hc (%d) of 64", D, Hc);
  PATHS_REQUIRE(hp != nullptr || (h0 == nullptr) == (c0 == nullptr), "lstm_cell_x6: h0 and c0 must both be given or both be null");
  PATHS_REQUIRE((hp == nullptr) == (hp_row == nullptr) && (hp == nullptr || h0 == nullptr),
                "lstm_cell_x6: hp/hp_row come together and replace h0 (the h half of the gate GEMM was done per parent)");
  PATHS_REQUIRE(planes == 3 || (pow2(wg_scale) && pow2(wm_scale) && pow2(a_scale)), "lstm_cell_x6: scales must be powers of two");
  PATHS_REQUIRE(x_rows == nullptr || (h0 == nullptr && y == nullptr && save_tc == nullptr), "lstm_cell_x6: x_rows (rows addressed in place) excludes h0, y and the training saves");
  if (planes == 3) wg_scale = wm_scale = a_scale = 1.0f;
  const char* wg = reinterpret_cast<const char*>(w_gates_x6);
  const int64_t gs = group_stride(planes, 2 * D);
  const float sg = 1.0f / (wg_scale * a_scale), sm = 1.0f / (wm_scale * a_scale);
  X6Operands g{x, ldx, D, x_rows, h0, h0 ? ldh0 : 0, h0 ? D : 0, nullptr, 0, wg, gs, M, num_ims, rows_per_slide, a_scale};
  // Few rows (M <= 8192: K = 1024 x 8 slides, small batches, drop-in calls): 256-row blocks put ceil(M / 256) x 4 <= 128 workgroups on
  // 256 CUs - half the chip idles through the two biggest launches of a level.  128-row blocks (the parent GEMM's tile family) make it
  // one full round of workgroups with half the work each; above 8192 rows they would spill into a second round and lose (round 2:
  // 165 against 280 TFLOP/s per workgroup pair).  Default split (two fp16 planes) only; PATHS_LSTM_SMALL_TILES=0 turns it off.
  static const bool small_on = getenv("PATHS_LSTM_SMALL_TILES") == nullptr || atoi(getenv("PATHS_LSTM_SMALL_TILES")) != 0;
  const bool small_m = small_on && planes == 2 && M <= 8192 && (h0 == nullptr);
  if (phases & 1) {   // c-part: N = 3Hc, block 256 x 192
    EpiLstmC e{b_gates, c0, ldc0, state_out + D, ldso, save_frm, (int64_t)3 * Hc, hp, (int64_t)3 * Hc + D, hp_row, sg};
    int rc;
    if (small_m) rc = x_rows ? launch_x6_np<2, 2, 3, 2, false, true>(g, 3 * Hc, e, stream, "lstm_cell_x6(c, 128-row tiles)")
                             : launch_x6_np<2, 2, 3, 2, false, false>(g, 3 * Hc, e, stream, "lstm_cell_x6(c, 128-row tiles)");
    else rc = launch_x6<4, 3, 1, false>(planes, g, 3 * Hc, e, stream, "lstm_cell_x6(c)");
    if (rc) return rc;
  }
  // inference (no Y, no training saves): the gate travels to phase 4 as raw pre-activations in the accumulator layout (EpiLstmORaw)
  const bool o_raw = y == nullptr && save_tc == nullptr && save_frm == nullptr && O_RAW;
  if (phases & 2) {   // o gate: N = D, block 256 x 256
    X6Operands go = g;
    go.Wt = wg + (int64_t)(3 * Hc / 32) * gs;
    int rc;
    if (o_raw) {
      EpiLstmORaw e{b_gates + 3 * Hc, ws_o, D, D, hp, (int64_t)3 * Hc + D, hp_row, 3 * Hc, sg};
      if (small_m) rc = x_rows ? launch_x6_np<2, 2, 4, 2, false, true>(go, D, e, stream, "lstm_cell_x6(o, raw, 128-row tiles)")
                               : launch_x6_np<2, 2, 4, 2, false, false>(go, D, e, stream, "lstm_cell_x6(o, raw, 128-row tiles)");
      else rc = launch_x6<4, 4, 1, false>(planes, go, D, e, stream, "lstm_cell_x6(o, raw)");
    } else {
      EpiLstmO e{b_gates + 3 * Hc, ws_o, D, D, hp, (int64_t)3 * Hc + D, hp_row, 3 * Hc, sg};
      rc = launch_x6<4, 4, 1, false>(planes, go, D, e, stream, "lstm_cell_x6(o)");
    }
    if (rc) return rc;
  }
  if (phases & 4) {   // h1 = o * tanh(Wc c1 + bc), Y = X + h1
    X6Operands gh{state_out + D, ldso, Hc, nullptr, nullptr, 0, 0, nullptr, 0, reinterpret_cast<const char*>(w_mem_x6), group_stride(planes, Hc), M, num_ims, rows_per_slide, a_scale};
    int rc;
    PATHS_REQUIRE(save_tc == nullptr || y != nullptr, "lstm_cell_x6: save_tc (training) needs y");
    if (save_tc != nullptr) {
      EpiLstmH<true, true> e{b_mem, ws_o, D, x, ldx, state_out, ldso, y, ldy, D, save_tc, sm};
      // (round 5: the training form on the inference form's 128 x 128 tiles at H_TRAIN_OCC waves per SIMD: K = 256 is 16 k16 stages
      // of MFMA against ~250 MB of epilogue traffic - several workgroups per CU hide each other's epilogue loads / stores)
      if (planes == 2 && H_SMALL_TILES && H_TRAIN_SMALL) rc = launch_x6_np<2, 2, 2, 2, false, false, decltype(e), H_TRAIN_OCC>(gh, D, e, stream, "lstm_cell_x6(h, save, 128x128)");
      else rc = launch_x6<4, 4, 1, false>(planes, gh, D, e, stream, "lstm_cell_x6(h, save)");
    } else if (y != nullptr) {
      EpiLstmH<true, false> e{b_mem, ws_o, D, x, ldx, state_out, ldso, y, ldy, D, nullptr, sm};
      rc = launch_x6<4, 4, 1, false>(planes, gh, D, e, stream, "lstm_cell_x6(h)");
    } else if (o_raw) {
      EpiLstmH<false, false, true> e{b_mem, ws_o, D, x, ldx, state_out, ldso, nullptr, 0, D, nullptr, sm};
      // inference: 16 k16 stages of MFMA against 120 MB of epilogue traffic -> 128 x 128 tiles, several workgroups per CU
      if (planes == 2 && H_SMALL_TILES) rc = launch_x6_np<2, 2, 2, 2, false, false, decltype(e), H_OCC>(gh, D, e, stream, "lstm_cell_x6(h, raw o, 128x128)");
      else rc = launch_x6<4, 4, 1, false>(planes, gh, D, e, stream, "lstm_cell_x6(h, raw o)");
    } else {
      EpiLstmH<false, false> e{b_mem, ws_o, D, x, ldx, state_out, ldso, nullptr, 0, D, nullptr, sm};
      if (planes == 2 && H_SMALL_TILES) rc = launch_x6_np<2, 2, 2, 2, false, false, decltype(e), H_OCC>(gh, D, e, stream, "lstm_cell_x6(h, no y, 128x128)");
      else rc = launch_x6<4, 4, 1, false>(planes, gh, D, e, stream, "lstm_cell_x6(h, no y)");
    }
    if (rc) return rc;
  }
  return PATHS_OK;
}

#endif
#if PATHS_X6_PART == 0 || PATHS_X6_PART == 2
// paths_importance_proj with w_ip_x6 = pack([256, D], rows interleaved as paths_importance_proj documents).  y_add (optional): the
// GEMM input is y + y_add, summed in fp32 while staging - the caller passes (x, h1) and never materialises Y = X + h1
// (paths_lstm_cell_x6 with y = NULL)
// bytes of the optional split-K workspace of paths_importance_proj_x6 (two k halves of raw [M_pad, 256] accumulators)
int64_t paths_importance_proj_x6_workspace(int M) { return 2ll * ((M + 127) / 128 * 4) * 8 * 1024 * 4; }

int paths_importance_proj_x6(const float* y, int64_t ldy, const int64_t* y_rows, const float* y_add, int64_t ldya, const void* w_ip_x6, const float* b1, const float* w2, const float* b2,
                             const float* bp, const float* special, const float* div_term, const float* pe_table, int pe_rows, const int64_t* locs,
                             const int64_t* num_ims, int rows_per_slide, int patch_size, int pe_mode, int imp_mul,
                             float* importance, float* tokens, float* save_hid, float* save_pproj, int M, int D, int Hi, int d,
                             int skip_padding, int planes, float w_scale, float a_scale, float* splitk_ws, hipStream_t stream) {
  PATHS_REQUIRE(Hi == 128 && d == 128, "importance_proj_x6: this build supports importance_mlp_hidden_dim=128, trans_dim=128 (got %d, %d)", Hi, d);
  PATHS_REQUIRE(pe_mode == 1 || pe_mode == 2, "importance_proj_x6: pe_mode must be 1 (1d) or 2 (2d)");
  PATHS_REQUIRE(pe_mode == 1 || locs != nullptr, "importance_proj_x6: 2d positional encoding needs locs");
  PATHS_REQUIRE(num_ims != nullptr && rows_per_slide > 0 && M % rows_per_slide == 0, "importance_proj_x6: bad slide layout");
  PATHS_REQUIRE(b1 != nullptr && w2 != nullptr && b2 != nullptr, "importance_proj_x6: b1, w2 and b2 (device scalar) are required");
  PATHS_REQUIRE(y_add == nullptr || (ldya % 4 == 0 && (uintptr_t)y_add % 16 == 0), "importance_proj_x6: y_add must be 16-byte aligned with ldya %% 4 == 0");
  PATHS_REQUIRE(planes == 3 || (pow2(w_scale) && pow2(a_scale)), "importance_proj_x6: scales must be powers of two");
  if (planes == 3) w_scale = a_scale = 1.0f;
  const float sc = 1.0f / (w_scale * a_scale);
  X6Operands g{y, ldy, D, y_rows, nullptr, 0, 0, y_add, ldya, reinterpret_cast<const char*>(w_ip_x6), group_stride(planes, D), M, skip_padding ? num_ims : nullptr, rows_per_slide, a_scale};
  PATHS_REQUIRE((save_hid == nullptr) == (save_pproj == nullptr), "importance_proj_x6: save_hid and save_pproj come together");
  PATHS_REQUIRE(pe_table == nullptr || pe_rows > 0, "importance_proj_x6: pe_rows must be > 0 with a pe_table");
  auto go = [&](auto epi) {
    decltype(epi) e{b1, w2, b2, bp, special, div_term, locs, num_ims, rows_per_slide, patch_size, pe_mode, imp_mul, importance, tokens,
                    save_hid, save_pproj, pe_table, pe_table ? pe_rows : 0, sc};
    {
      // two launches (default split): the GEMM stores RAW accumulators, the epilogue runs on 64-row blocks.  Default: split-K, two k
      // halves of 128 x 256 tiles.  PATHS_IP_TILE128=1 (experiment, slower, Y = X + h1 form only): 128 x 128 tiles over the full k at two
      // waves per SIMD - the x + h1 operand is then staged twice as often.  Round 5: also the training form (a stored Y, the SAVE
      // epilogues): one launch of M / 128 blocks fills half the chip there too (115 us against ~60).
      if (splitk_ws != nullptr && planes == 2 && (y_add != nullptr || y_rows == nullptr) && D % 64 == 0 && D >= 256) {
        const int mt = (M + 127) / 128 * 4;                         // 32-row tiles, padded to the GEMM's 128-row blocks
        const int64_t zstride = (int64_t)mt * 8 * 1024;
        int rc;
        const bool tile128 = IP_TILE128 && y_add != nullptr;
        if (tile128) {
          EpiRaw raw{splitk_ws, zstride, 8};
          rc = y_rows ? launch_x6_np<2, 2, 2, 2, true, true, EpiRaw, 2>(g, 256, raw, stream, "importance_proj_x6(raw, 128x128)")
                      : launch_x6_np<2, 2, 2, 2, true, false, EpiRaw, 2>(g, 256, raw, stream, "importance_proj_x6(raw, 128x128)");
        } else {
          X6Operands gs = g; gs.ksplit = 2;
          EpiRaw raw{splitk_ws, zstride, 8};
          rc = y_add == nullptr ? launch_x6_np<2, 2, 4, 2, false, false>(gs, 256, raw, stream, "importance_proj_x6(split-k, stored y)")
               : y_rows ? launch_x6_np<2, 2, 4, 2, true, true>(gs, 256, raw, stream, "importance_proj_x6(split-k)")
                        : launch_x6_np<2, 2, 4, 2, true, false>(gs, 256, raw, stream, "importance_proj_x6(split-k)");
        }
        if (rc != PATHS_OK) return rc;
        // the dispatcher packs a CU to its limit before it moves on: ask for LDS that spreads the blocks over all CUs
        const int fblk = (M + 63) / 64, fdepth = (fblk + 255) / 256;
        const int flds = fdepth == 1 ? 84 * 1024 : fdepth == 2 ? 54 * 1024 : fdepth == 3 ? 41 * 1024 : 1024;
        PATHS_LDS_OPT_IN((x6_finish_kernel<2, decltype(epi)>), 84 * 1024, "importance_proj_x6(finish)");
        PATHS_LDS_OPT_IN((x6_finish_kernel<1, decltype(epi)>), 84 * 1024, "importance_proj_x6(finish)");
        if (tile128)
          PATHS_LAUNCH_STOP((x6_finish_kernel<1, decltype(epi)>), dim3(fblk), dim3(256), flds, stream, splitk_ws, zstride, 8, M,
                            skip_padding ? num_ims : nullptr, rows_per_slide, e);
        else
          PATHS_LAUNCH_STOP((x6_finish_kernel<2, decltype(epi)>), dim3(fblk), dim3(256), flds, stream, splitk_ws, zstride, 8, M,
                            skip_padding ? num_ims : nullptr, rows_per_slide, e);
        PATHS_LAUNCH_CHECK("importance_proj_x6(finish)");
        return PATHS_OK;
      }
    }
    if (y_add != nullptr) return launch_x6<2, 4, 2, true>(planes, g, 256, e, stream, "importance_proj_x6(sum)");
    return launch_x6<2, 4, 2, false>(planes, g, 256, e, stream, "importance_proj_x6");
  };
  if (pe_table != nullptr) return save_hid ? go(EpiImpProj<true, true>{}) : go(EpiImpProj<true, false>{});
  return save_hid ? go(EpiImpProj<false, true>{}) : go(EpiImpProj<false, false>{});
}


// The split-K importance / projection GEMM and its FUSED finish (finish_qkv.h): the default inference form of
// paths_importance_proj_x6 + paths_token_layer_ws(do_qkv only) for trans_dim 128 / 4 heads / importance hidden 128 (reference
// model/paths.py:95-98,119-124, model/aggregator.py:37-65 and the first decoder layer's self_attn in_proj, aggregator.py:70-72).
// TOKEN ORDER of this form: patch i of a slide is token i and the SPECIAL token sits at index num_ims[b], right behind the valid
// patches (the reference prepends it, aggregator.py:62-64; self-attention with a key mask is invariant under that permutation and only
// the special token's output row is read) - so a 64-row block of the GEMM result IS a 64-token tile of the attention's operand
// images, with no shift by one.  Consumers: paths_attention_h3_img / paths_token_layer_ws (position-agnostic) and
// paths_token0_tail_ws with special_last = 1.  N % 64 == 0.
//   phases bit 1: the split-K GEMM over (y | y_rows) + y_add into splitk_ws (paths_importance_proj_x6_workspace(B * N) bytes);
//          bit 2: the importance-only finish (alpha -> importance [B, N]);
//          bit 8: (instead of bit 2) the importance finish AND the top-K of every slide in one launch: keep_idx [B, ldk] (score descending,
//                 index ascending; keep < 0: all, original order), keep_count [B], optionally kept_rows [B, ldk] = addresses of
//                 row_base[b, keep_idx[b, i], :] (row stride row_ld floats, N rows per slide; zero_row beyond the count) - the outputs of
//                 paths_topk_rows; counters: 2 B int32 words, zero on entry, left zero; status (optional): bit 4 on a timed-out arrival wait;
//          bit 4: the tokens + in_proj finish: importance (computed, or read back when alpha_from_importance), tokens [B, N + 1, 128]
//                 and the q | k | v operand images of paths_attention_h3_img in qkv_images (paths_attention_x6_workspace(B, N + 1, 4, 32, 2)).
// Bits 2 and 4 are stop-event capable launches; they may be issued by separate calls on different streams (the caller orders them
// behind bit 1).  pe_table is required (paths_pe_table; positions are clamped to its rows).  w_qkv: paths_tlayer_pack_ws part 1 image with scale s_wqkv.
int paths_importance_qkv_x6(const float* y, int64_t ldy, const int64_t* y_rows, const float* y_add, int64_t ldya, const void* w_ip_x6,
                            const float* b1, const float* w2, const float* b2, const float* bp, const float* special,
                            const float* pe_table, int pe_rows, const int64_t* locs, const int64_t* num_ims, int B, int N,
                            int patch_size, int pe_mode, int imp_mul, float* importance, float* tokens, int D, int skip_padding,
                            float w_scale, float a_scale, float* splitk_ws, const void* w_qkv, const float* bqkv, float s_wqkv,
                            float qscale, void* qkv_images, int phases, int alpha_from_importance,
                            int keep, int* keep_idx, int64_t ldk, int* keep_count, const float* row_base, int64_t row_ld, int64_t* kept_rows,
                            const float* zero_row, int* counters, int* status, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && N > 0 && N % 64 == 0 && D % 64 == 0 && D >= 256, "importance_qkv_x6: bad shape B=%d N=%d (a multiple of 64) D=%d", B, N, D);
  PATHS_REQUIRE(pe_mode == 1 || pe_mode == 2, "importance_qkv_x6: pe_mode must be 1 (1d) or 2 (2d)");
  PATHS_REQUIRE(pe_table != nullptr && pe_rows > 0 && (pe_mode == 1 || locs != nullptr), "importance_qkv_x6: needs the positional-encoding table (and locs in 2d mode)");
  PATHS_REQUIRE(num_ims != nullptr && splitk_ws != nullptr && (uintptr_t)splitk_ws % 16 == 0, "importance_qkv_x6: num_ims and a 16-byte aligned workspace are required");
  PATHS_REQUIRE(phases > 0 && (phases & ~15) == 0 && (phases & 10) != 10, "importance_qkv_x6: phases is a mask of 1 (GEMM), 2 (importance finish) or 8 (importance + top-K finish), 4 (tokens + in_proj finish)");
  PATHS_REQUIRE(pow2(w_scale) && pow2(a_scale), "importance_qkv_x6: scales must be powers of two");
  const int64_t M64 = (int64_t)B * N;
  PATHS_REQUIRE(M64 < (1ll << 24), "importance_qkv_x6: B * N must stay below 2^24 rows");
  const int M = (int)M64, T = N + 1, Tp = (T + 63) / 64 * 64;
  const int mt = (M + 127) / 128 * 4;                                // 32-row tiles, padded to the GEMM's 128-row blocks
  const int64_t zstride = (int64_t)mt * 8 * 1024;
  if (phases & 1) {
    PATHS_REQUIRE(w_ip_x6 != nullptr && y_add != nullptr && (y != nullptr) != (y_rows != nullptr), "importance_qkv_x6: the GEMM takes (y or y_rows) + y_add");
    PATHS_REQUIRE(ldya % 4 == 0 && (uintptr_t)y_add % 16 == 0, "importance_qkv_x6: y_add must be 16-byte aligned with ldya %% 4 == 0");
    X6Operands g{y, ldy, D, y_rows, nullptr, 0, 0, y_add, ldya, reinterpret_cast<const char*>(w_ip_x6), group_stride(2, D), M, skip_padding ? num_ims : nullptr, N, a_scale};
    g.ksplit = 2;
    EpiRaw raw{splitk_ws, zstride, 8};
    const int rc = y_rows ? launch_x6_np<2, 2, 4, 2, true, true>(g, 256, raw, stream, "importance_qkv_x6(split-k)")
                          : launch_x6_np<2, 2, 4, 2, true, false>(g, 256, raw, stream, "importance_qkv_x6(split-k)");
    if (rc != PATHS_OK) return rc;
  }
  if (phases & 14) {
    PATHS_REQUIRE(b1 && w2 && b2 && importance, "importance_qkv_x6: b1, w2, b2 (device scalar) and importance are required");
    FinQkvParams f{splitk_ws, zstride, 2, b1, w2, b2, bp, special, pe_table, pe_rows, locs, num_ims, N, T, Tp, B, patch_size, pe_mode, imp_mul,
                   skip_padding, 1.0f / (w_scale * a_scale), alpha_from_importance, importance, tokens, w_qkv, bqkv, 1.0f / s_wqkv, qscale, qkv_images,
                   (B % 8 == 0 && (M / 128) % 8 == 0 && M % 128 == 0 && FIN_XCD_ORDER) ? 1 : 0};
    if (phases & 2) {
      const int rc = paths_launch_finish_importance(f, stream);
      if (rc != PATHS_OK) return rc;
    }
    if (phases & 8) {
      PATHS_REQUIRE(keep_idx && keep_count && counters && ldk > 0 && ldk <= N && N <= 8192 && (kept_rows == nullptr || (row_base && zero_row)),
                    "importance_qkv_x6: the importance + top-K finish needs keep_idx [B, ldk <= N], keep_count, counters (2 B zeroed int32), N <= 8192");
      FinTopkParams tk{keep, keep_idx, ldk, keep_count, row_base, row_ld, kept_rows, zero_row, counters, status};
      const int rc = paths_launch_finish_importance_topk(f, tk, stream);
      if (rc != PATHS_OK) return rc;
    }
    if (phases & 4) {
      PATHS_REQUIRE(bp && special && tokens && w_qkv && bqkv && qkv_images && pow2(s_wqkv), "importance_qkv_x6: null operand (tokens + in_proj finish)");
      PATHS_REQUIRE(((uintptr_t)tokens | (uintptr_t)w_qkv | (uintptr_t)qkv_images) % 16 == 0, "importance_qkv_x6: buffers must be 16-byte aligned");
      return paths_launch_finish_qkv(f, stream);
    }
  }
  return PATHS_OK;
}

#endif
#if PATHS_X6_PART == 0 || PATHS_X6_PART == 3
// out[M,N] (+)= maskop(act(A[M,K] * W[N,K]^T + b)) + residual with W given as the split image of an [Npad, Kpacked]
// weight; k0 selects the column window [k0, k0 + K) of it (k0 % 16 == 0), Npad % 256 == 0.
int paths_gemm_nt_x6(const float* a, int64_t lda, const void* w_x6, int Kpacked, int k0, const float* b, float* out, int64_t ldo,
                     int M, int N, int Npad, int K, int act, const float* residual, int64_t ldr, const float* mask,
                     int64_t ldm, int accumulate, int planes, float w_scale, float a_scale, hipStream_t stream) {
  PATHS_REQUIRE(k0 % 16 == 0 && k0 >= 0 && k0 + K <= Kpacked, "gemm_nt_x6: bad k window");
  PATHS_REQUIRE(planes == 3 || planes == 4 || (planes == 2 && pow2(w_scale) && pow2(a_scale)), "gemm_nt_x6: planes 3 / 4 (bf16), or planes 2 with power-of-two scales");
  if (planes != 2) w_scale = a_scale = 1.0f;
  X6Operands g{a, lda, K, nullptr, nullptr, 0, 0, nullptr, 0, reinterpret_cast<const char*>(w_x6) + (int64_t)(k0 / 16) * plane_count(planes) * FRAG, group_stride(planes, Kpacked), M, nullptr, 0, a_scale};
  EpiBias e{b, out, ldo, N, act, residual, ldr, mask, ldm, accumulate, 1.0f / (w_scale * a_scale)};
  if (Npad % 256 != 0) {           // 128 / 384 output columns (the transformer layers' d = 128 products): 128 x 128 tiles, 2 workgroups per CU
    if (planes == 4) return launch_x6_np<4, 2, 2, 2, false, false, EpiBias, 2>(g, Npad, e, stream, "gemm_nt_x6(n128, bf16 x 2)");
    if (planes == 3) return launch_x6_np<3, 2, 2, 2, false, false, EpiBias, 2>(g, Npad, e, stream, "gemm_nt_x6(n128)");
    return launch_x6_np<2, 2, 2, 2, false, false, EpiBias, 2>(g, Npad, e, stream, "gemm_nt_x6(n128)");
  }
  return launch_x6<2, 4, 2, false>(planes, g, Npad, e, stream, "gemm_nt_x6");
}

// paths_gemm_nt_x6 without bias / activation, with the A operand given as row ADDRESSES (planes = 2): out[M, Npad] = A W^T where
// row m of A is the K floats at a_rows[m] (16-byte aligned; e.g. kept parents' h rows inside the level's state tensor)
int paths_gemm_rows_nt_x6(const int64_t* a_rows, const void* w_x6, int Kpacked, int k0, float* out, int64_t ldo,
                          int M, int Npad, int K, int planes, float w_scale, float a_scale, hipStream_t stream) {
  PATHS_REQUIRE(a_rows != nullptr && planes == 2 && pow2(w_scale) && pow2(a_scale), "gemm_rows_nt_x6: row addresses need planes = 2 with power-of-two scales");
  PATHS_REQUIRE(k0 % 16 == 0 && k0 >= 0 && k0 + K <= Kpacked, "gemm_rows_nt_x6: bad k window");
  X6Operands g{nullptr, K, K, a_rows, nullptr, 0, 0, nullptr, 0,
               reinterpret_cast<const char*>(w_x6) + (int64_t)(k0 / 16) * planes * FRAG, group_stride(planes, Kpacked), M, nullptr, 0, a_scale};
  EpiBias e{nullptr, out, ldo, Npad, 0, nullptr, 0, nullptr, 0, 0, 1.0f / (w_scale * a_scale)};
  return launch_x6<2, 4, 2, false>(planes, g, Npad, e, stream, "gemm_rows_nt_x6");
}

// out[M, N] = act((A + A_add) W^T + b): paths_gemm_nt_x6 (planes = 2) with the GEMM input summed in fp32 while it is staged (the
// reference's Y = X + h1, model/paths.py:89-91, never materialised) and A given either as a matrix (a, lda) or as row ADDRESSES
// (a_rows: feature rows read in place in the resident grids); num_ims (optional): whole 128-row tiles of padding are skipped.
// The importance / projection products of aggregator geometries other than the fused 128 / 128 one (ops.importance_proj_generic_add).
int paths_gemm_add_nt_x6(const float* a, int64_t lda, const int64_t* a_rows, const float* a_add, int64_t ld_add, const void* w_x6, int Kpacked,
                         const float* b, float* out, int64_t ldo, int M, int N, int Npad, int K, int act, const int64_t* num_ims,
                         int rows_per_slide, float w_scale, float a_scale, hipStream_t stream) {
  PATHS_REQUIRE((a != nullptr) != (a_rows != nullptr), "gemm_add_nt_x6: exactly one of a / a_rows");
  PATHS_REQUIRE(a_add != nullptr && ld_add % 4 == 0 && (uintptr_t)a_add % 16 == 0, "gemm_add_nt_x6: a_add must be 16-byte aligned with ld_add %% 4 == 0");
  PATHS_REQUIRE(pow2(w_scale) && pow2(a_scale) && K == Kpacked && (Npad % 256 == 0 || Npad % 192 == 0),
                "gemm_add_nt_x6: power-of-two scales, whole-K image, Npad a multiple of 256 or of 192");
  X6Operands g{a, lda, K, a_rows, nullptr, 0, 0, a_add, ld_add, reinterpret_cast<const char*>(w_x6), group_stride(2, Kpacked), M, num_ims, rows_per_slide, a_scale};
  EpiBias e{b, out, ldo, N, act, nullptr, 0, nullptr, 0, 0, 1.0f / (w_scale * a_scale)};
  if (Npad % 256 != 0) {
    // 128 x 192 tiles: a width between two multiples of 256 pays for the columns it has (the [W1 ; Wp] product of trans_dim 192 is 320
    // columns: two tiles of 192 instead of two of 256 - 61 -> 48 us)
    return a_rows ? launch_x6_np<2, 2, 3, 2, true, true>(g, Npad, e, stream, "gemm_add_nt_x6(rows, 192-wide tiles)")
                  : launch_x6_np<2, 2, 3, 2, true, false>(g, Npad, e, stream, "gemm_add_nt_x6(192-wide tiles)");
  }
  return a_rows ? launch_x6_np<2, 2, 4, 2, true, true>(g, Npad, e, stream, "gemm_add_nt_x6(rows)")
                : launch_x6_np<2, 2, 4, 2, true, false>(g, Npad, e, stream, "gemm_add_nt_x6");
}

#endif

}  // extern "C"
