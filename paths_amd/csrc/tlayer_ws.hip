// Weight-stationary token-row chain of one post-LN nn.TransformerDecoderLayer (+ the next layer's QKV projection) on the fp16
// matrix cores with fp32 accuracy (two-plane operand split, see gemm_x6.hip / tlayer_h3.hip).
//
// Same math and call site as paths_token_layer_h3 (reference model/aggregator.py:25-33, 70-72):
//   x = norm1(x + out_proj(attn)) ; x = norm2(x + multihead_attn.out_proj.bias) ; x = norm3(x + linear2(relu(linear1(x))))
//   q, k, v = in_proj(x) of the NEXT layer, q pre-scaled for the exp2 softmax
//
// What is different from tlayer_h3.hip (where every wave kept 16 tokens in registers and streamed ALL weights through LDS: each
// wave re-read every 32-KiB weight chunk from LDS for 48 MFMAs, 24 dependent chunk steps, MFMA pipe 11 % busy):
//   * a workgroup = 4 waves = 64 tokens; wave w owns the output features [w N/4, (w+1) N/4) of every product for ALL 64 tokens
//     (products are still computed transposed, Y^T[out][token] = W[out][:] X^T[:][token]);
//   * the weights of a wave's slice are read by that wave only, so they never touch LDS: they stream global (L2) -> registers
//     through a prefetch ring, as ready-made MFMA fragments (1 KiB per load instruction, fully coalesced);
//   * the ACTIVATIONS are what the waves share: each product's result is split into fp16 hi | lo planes in registers and written
//     to LDS as the B-operand fragments of the next product (the accumulators of two 16-feature tiles in k-slot order (g, j) <->
//     feature 4g + (j&3) + 16 (j>>2) ARE one k32 block of that operand), one barrier per product;
//   * LayerNorm statistics cross the four waves through LDS as (mean, M2) pairs merged with Chan's formula (one barrier each);
//   * 96-192 MFMAs per wave between barriers instead of 48, 10 barriers per 64 tokens instead of 24 per 128.
// q / k images use the k-slot dim order (a permutation of the contraction index of q.k, harmless as long as q and k agree);
// v is computed with the operands swapped (tokens as MFMA rows), which yields the V^T fragments of attn_x6.hip without shuffles.
#include "common.h"
#include "finish_qkv.h"
#include "gemm_epi.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int FRAG = 1024;         // bytes of one 16-row x 32-k fragment of one plane
constexpr int NW = 4;              // waves per workgroup
#ifndef PATHS_WS_TT
#define PATHS_WS_TT 4
#endif
constexpr int TT = PATHS_WS_TT;    // 16-token tiles per workgroup (4: 64 tokens, one workgroup per CU; 2: 32 tokens, two per CU)
constexpr int TOK = 16 * TT;
constexpr int NFF = 4;             // feed-forward hidden chunks of DM features (dim_feedforward = 4 DM, reference aggregator.py:29)
#ifndef PATHS_WS_NPF
#define PATHS_WS_NPF 6
#endif
constexpr int NPF = PATHS_WS_NPF;  // weight prefetch ring: k32 steps in flight per wave (6: inside a level the weights come from beyond the L2 - it is
                                   // invalidated at every launch boundary; row chain 24.1 -> 22.9 us in the recursion, 4 = 8 in a hot loop)

template <int DM> struct Geo {
  static_assert(DM % 64 == 0, "trans_dim must be a multiple of 64");
  static constexpr int KB = DM / 32;                      // k32 blocks of a DM-wide activation
  static constexpr int OT = DM / 64;                      // 16-feature output tiles per wave of an N = DM product
  static constexpr int STEP = OT * 2 * FRAG;              // weight bytes one wave consumes per k32 step
  static constexpr int WAVE_UNIT = KB * STEP;             // one wave's slice of a unit (N = DM outputs x K = DM inputs)
  static constexpr int UNIT = NW * WAVE_UNIT;             // = DM * DM * 4 bytes
  static constexpr int ACT = KB * TT * 2 * FRAG;          // activation image of 64 tokens: [kb][tt][plane][64 lanes][16 B]
  static constexpr int N_POST = 1 + 2 * NFF, N_QKV = 3;   // units: Wo | (W1 rows c DM.., W2[:, c DM..]) x 4   and   Wq | Wk | Wv
};

__device__ __forceinline__ uint32_t pk_f16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// 8 fp32 -> hi | lo planes of 8 fp16 (22 significant bits)
__device__ __forceinline__ void split8h(const float (&x)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const uint32_t h = pk_f16(a, b);
    float ra, rb;
    f16_pair_residuals(h, a, b, ra, rb);
    hi[i] = h; lo[i] = pk_f16(ra, rb);
  }
}
__device__ __forceinline__ void split4h(const float (&x)[4], u32x2& hi, u32x2& lo) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const uint32_t h = pk_f16(a, b);
    float ra, rb;
    f16_pair_residuals(h, a, b, ra, rb);
    hi[i] = h; lo[i] = pk_f16(ra, rb);
  }
}
__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 ldg_u32x4(const char* p) {
  typedef const u32x4 __attribute__((address_space(1))) * gptr;
  return *reinterpret_cast<gptr>(reinterpret_cast<uintptr_t>(p));
}

// x + (x of lane ^ 16) and x + (x of lane ^ 32) by v_permlane16/32_swap (VALU latency instead of the LDS round trip of
// ds_bpermute).  v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of the second,
// v_permlane32_swap the upper half of the first with the lower half of the second: given the same value in both, the two results
// add up to the pair sum in every lane.  Inline asm: through __builtin_amdgcn_permlane16_swap hipcc (ROCm 7.2) added result 0 to
// itself here (v_add v, r0, r0: wrong sums).  s_nop 1 = the two wait states between a VALU write of an operand and the swap.
__device__ __forceinline__ float sum_xor16(float x) {
#ifdef PATHS_WS_LN_SHFL
  return x + __shfl_xor(x, 16);
#endif
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float sum_xor32(float x) {
#ifdef PATHS_WS_LN_SHFL
  return x + __shfl_xor(x, 32);
#endif
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

#ifdef PATHS_WS_STAMPS
#define WS_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (p.stamps && tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WS_STAMP(i) do { } while (0)
#endif

struct WsParams {
  const float* x_in;            // [B][T][DM] residual stream (POST) / in_proj input (QKV only)
  const float* attn;            // POST: attention output, fp32 token-major [B][T][DM] (when attn_img is null)
  const char* attn_img;         // POST: or its fragment image [B][Tp/64][kb = head][tt][plane][64 lanes][16 B] (attn_x6.hip)
  float* x_out;                 // POST: [B][T][DM]
  const char* w_post;           // 9 units (paths_tlayer_pack_ws part 0 of THIS layer)
  const char* w_qkv;            // 3 units (part 1 of the NEXT layer)
  const float *bo, *ln1g, *ln1b, *cab, *ln2g, *ln2b, *b1, *b2, *ln3g, *ln3b, *bqkv;
  float inv_wo, inv_w1, inv_w2, inv_wqkv;      // 1 / (power-of-two scale of the packed tensor)
  char* qkv_img;                // QKV: the two-plane operand images of attn_x6_kernel<2>, Q | K | V
  float* qkv_rows; int64_t ld_qkv;   // QKV, ROWS form: q | k | v as fp32 token-major rows [B*T][ld_qkv] (q unscaled: the shape-generic attention kernels)
  const int64_t* num_ims;
  int T, Tp, B, skip_padding;
  float qscale, eps;
  int* zero_words; int n_zero;  // optional: words zeroed by block (0, 0) (the arrival counters of the token-0 tail that follows)
#ifdef PATHS_WS_STAMPS
  unsigned long long* stamps;   // diagnostic build only: 16 s_memtime stamps per workgroup
#endif
  FinQkvParams fin;             // FIN kernels only (finish_qkv.h): the input rows are BUILT here from the raw importance / projection GEMM result
};

constexpr int TOK_LD = 132;     // FIN: row stride (floats) of the fp32 token tile in LDS (16-byte aligned rows, rows 4 banks apart)

// FIN prologue, part 1 (all 4 waves = 2 x 2 over 64 token slots x 256 raw columns, the layout of x6_finish_kernel / EpiImpProj in
// gemm_x6.hip / gemm_epi.h; token slot = patch row, special token at slot num_ims[b]: finish_qkv.h): sum the k-half slabs, importance logits -> alpha, tokens -> global + LDS (sTok).
// Packed GEMM columns: [W1[0:64] ; Wp[0:64] ; W1[64:128] ; Wp[64:128]]: wave column half wn owns hidden units / token channels 64 wn ..
// Which (slide, token tile) a finish workgroup owns.  The split-K GEMM that wrote the slabs runs its 128-row blocks in an XCD-aware
// order (gemm_x6.hip): with M / 128 a multiple of 8, XCD x computes - and leaves in ITS L2 - the rows [x M / 8, (x + 1) M / 8).  The
// hardware places workgroup h of a launch on XCD h % 8, so with `xcd_order` (B % 8 == 0: whole slides per XCD) workgroup h takes the
// (h / 8)-th tile of the slides of XCD h % 8 and reads its slabs from the L2 they were written to instead of across the fabric.
// Otherwise: the natural (tile, slide) grid.
__device__ __forceinline__ void fin_tile_of_workgroup(const FinQkvParams& f, int& b, int& t0) {
  if (!f.xcd_order) return;
  const int h = blockIdx.y * gridDim.x + blockIdx.x, x = h & 7, k = h >> 3;
  const int tiles = f.Tp / TOK, spx = f.B >> 3;           // tiles per slide, slides per XCD
  b = x * spx + k / tiles;
  t0 = (k % tiles) * TOK;
}

#ifdef PATHS_WS_STAMPS
#define FIN_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (stamps && tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FIN_STAMP(i) do { } while (0)
#endif
__device__ __forceinline__ void fin_tokens(const FinQkvParams& f, int b, int t0, int tid, float* sTok, float* sAlpha, unsigned long long* stamps) {
  constexpr int d = 128, NZ = 2;                       // (two k halves: the only split the GEMM is launched with)
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int u0 = 64 * wn;
  const int nim = (int)f.num_ims[b];
  float* const sA = sAlpha + 2 * TOK;                  // [TOK] alpha of every slot of the tile
  // loads that do not depend on the GEMM result first: their round trips (position -> table row are two dependent ones) run under the
  // 128-KB slab stream below
  float bpv[2], spv[2], b1v[2], w2v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int c = u0 + 32 * j + (lane & 31); bpv[j] = f.bp[c]; spv[j] = f.special[c]; b1v[j] = f.b1[c]; w2v[j] = f.w2[c]; }
  // Load order (loads return in order; inside a level all of them come from beyond the L2): positions, then the first pair of slab
  // tiles - ISSUED before the positions are waited for (position -> table row -> table value are two dependent round trips: they run
  // under the slab stream, not in front of it), then the table rows, then the sums.
  int64_t lp[16];
  if (f.pe_mode == 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {                     // (clamped indices: always legal)
      const int sl = t0 + 32 * wm + c32_row(r, lane);
      lp[r] = f.locs[2 * ((int64_t)b * f.N + min(sl, f.N - 1)) + wn];
    }
  }
  const bool has_rows = t0 < f.N;                      // (the extra tile - slot N, the special token of a full slide - has no GEMM rows behind it)
  const int64_t trow = (((int64_t)b * f.N + min(t0, f.N - 1)) >> 5) + wm;
  // hidden-unit tiles (j = 0, 1) only when alpha is computed here; with alpha read back (an importance-only finish ran) only the
  // projection half of the slabs is touched (64 KB per workgroup instead of 128)
  f32x4 v[2][4][NZ];
  auto issue = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4* t = reinterpret_cast<const f32x4*>(f.ws + (trow * 8 + 4 * wn + j0 + j) * 1024) + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int z = 0; z < NZ; ++z) v[j][q][z] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(t + 64 * q) + z * f.zstride);
    }
  };
  float acc[4][16];
  auto reduce = [&](auto j0_) __attribute__((always_inline)) {
    constexpr int j0 = decltype(j0_)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 sum = v[j][q][0] + v[j][q][1];
        acc[j0 + j][4 * q] = sum[0]; acc[j0 + j][4 * q + 1] = sum[1]; acc[j0 + j][4 * q + 2] = sum[2]; acc[j0 + j][4 * q + 3] = sum[3];
      }
  };
  if (has_rows) issue(f.alpha_from_importance ? 2 : 0);
  __builtin_amdgcn_sched_barrier(0);
  const float ps_inv = 1.0f / (float)f.patch_size;
  int tp[16];
  if (f.pe_mode == 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px24 = (int)min(max(lp[r], (int64_t)0), (int64_t)((1 << 24) - 1));
      tp[r] = min(paths_epi::div_u24(px24, f.patch_size, ps_inv), f.pe_rows - 1);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) tp[r] = min(min(t0 + 32 * wm + c32_row(r, lane), f.N - 1), f.pe_rows - 1);
  }
  float pev[16][2];                                    // positional-encoding values: in flight while the slabs land and the logits are reduced
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = u0 + 32 * j + (lane & 31);
      pev[r][j] = f.pe_mode == 2 ? f.pe_table[(int64_t)tp[r] * (d / 2) + (c & (d / 2 - 1))] : f.pe_table[(int64_t)tp[r] * d + c];
    }
  __builtin_amdgcn_sched_barrier(0);
  if (!has_rows) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  } else if (f.alpha_from_importance) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    reduce(std::integral_constant<int, 2>{});
  } else {
    reduce(std::integral_constant<int, 0>{});
    issue(2);
    reduce(std::integral_constant<int, 2>{});
  }
  FIN_STAMP(2);
  if (!f.alpha_from_importance) {
    // ---- partial importance logits over this wave's 64 hidden units, summed over the 32 lanes of each half-wave by the halving
    // butterfly of EpiImpProj (gemm_epi.h): afterwards lane l holds the total of row index rho(l) = bits 4..1 of l
    float part[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float bb = b1v[j], w = w2v[j];
#pragma unroll
      for (int r = 0; r < 16; ++r) part[r] += fmaxf(fmaf(acc[j][r], f.acc_scale, bb), 0.f) * w;
    }
    float p8[8], p4[4], p2[2], p1;
    {
      const bool up = (lane & 16) != 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float keep = up ? part[8 + k] : part[k], send = up ? part[k] : part[8 + k]; p8[k] = keep + __shfl_xor(send, 16); }
    }
    {
      const bool up = (lane & 8) != 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float keep = up ? p8[4 + k] : p8[k], send = up ? p8[k] : p8[4 + k]; p4[k] = keep + __shfl_xor(send, 8); }
    }
    {
      const bool up = (lane & 4) != 0;
#pragma unroll
      for (int k = 0; k < 2; ++k) { const float keep = up ? p4[2 + k] : p4[k], send = up ? p4[k] : p4[2 + k]; p2[k] = keep + __shfl_xor(send, 4); }
    }
    {
      const bool up = (lane & 2) != 0;
      const float keep = up ? p2[1] : p2[0], send = up ? p2[0] : p2[1];
      p1 = keep + __shfl_xor(send, 2);
    }
    p1 += __shfl_xor(p1, 1);
    const int rho = (lane >> 1) & 15;
    if ((lane & 1) == 0) sAlpha[wn * TOK + 32 * wm + c32_row(rho, lane)] = p1;
  }
  __syncthreads();
  FIN_STAMP(3);
  // ---- alpha of the tile's 64 slots, once each (64 threads; one coalesced store of the importance row piece)
  if (tid < TOK) {
    const int sl = t0 + tid;
    float a = 0.f;
    if (sl < nim) a = f.alpha_from_importance ? f.importance[(int64_t)b * f.N + sl] : sigmoid_acc((sAlpha[tid] + sAlpha[TOK + tid]) + *f.b2);
    if (!f.alpha_from_importance && sl < f.N) f.importance[(int64_t)b * f.N + sl] = a;
    sA[tid] = a;
  }
  __syncthreads();
  // ---- tokens of this wave's 64 channels -> LDS (fp32 rows; the in_proj's operand image is built from them)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int lt = 32 * wm + c32_row(r, lane), sl = t0 + lt;
    const bool valid = sl < nim;
    const float av = f.imp_mul ? sA[lt] : 1.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = u0 + 32 * j + (lane & 31);
      // padded slots: the projection of a padded row may be anything (its LSTM tile may have been skipped): select, do not multiply
      const float pj = valid ? acc[2 + j][r] * f.acc_scale : 0.f;
      sTok[lt * TOK_LD + c] = sl == nim ? spv[j] : (sl < f.T ? av * pj + bpv[j] + pev[r][j] : 0.f);
    }
  }
  FIN_STAMP(4);
}

// the tile's fp32 token rows from LDS to tokens [B, T, 128]: whole 512-byte rows, 16 bytes per lane (call after a barrier behind fin_tokens)
__device__ __forceinline__ void fin_store_tokens(const FinQkvParams& f, int b, int t0, int tid, const float* sTok) {
#pragma unroll
  for (int i = 0; i < TOK * 32 / (64 * NW); ++i) {
    const int idx = tid + i * 64 * NW, lt = idx >> 5, c4 = idx & 31, sl = t0 + lt;
    if (sl < f.T) *reinterpret_cast<f32x4*>(f.tokens + ((int64_t)b * f.T + sl) * 128 + 4 * c4) = *reinterpret_cast<const f32x4*>(sTok + lt * TOK_LD + 4 * c4);
  }
}

// One workgroup = 64 tokens of one slide.  Lane (ql = lane & 15, g = lane >> 4) of wave w holds, for token tile tt and output
// tile ot, the features 16 (OT w + ot) + 4 g + r (r = 0..3) of token 16 tt + ql  -  the C layout of v_mfma_f32_16x16x32_f16 with the
// weights as A (rows = output features) and the activations as B (columns = tokens).
template <int DM, bool POST, bool QKV, bool ROWS = false, bool FIN = false>
__global__ void __launch_bounds__(64 * NW, TT == 2 ? 2 : 1)
tlayer_ws_kernel(WsParams p) {
  static_assert(!FIN || (DM == 128 && !POST && QKV && !ROWS && TT == 4), "FIN: the fused finish of the importance / projection GEMM (trans_dim 128, in_proj only)");
  using G = Geo<DM>;
  constexpr int KB = G::KB, OT = G::OT;
  constexpr int POST_STEPS = POST ? G::N_POST * KB : 0, NSTEPS = POST_STEPS + (QKV ? G::N_QKV * KB : 0);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sAct = smem;                                                        // [ACT]
  char* const sHid = smem + G::ACT;                                               // POST: [2][ACT]
  float* const sStat = reinterpret_cast<float*>(smem + G::ACT + (POST ? 2 * G::ACT : 0));   // [2][NW][TOK][2]
  float* const sVec = sStat + 2 * NW * TOK * 2;                                   // POST: 9 DM + 4 DM floats, then QKV: 3 DM
  float* const sB1 = sVec + 9 * DM;
  float* const sBqkv = sVec + (POST ? 13 * DM : 0);

  int b = blockIdx.y, t0 = blockIdx.x * TOK;
  if constexpr (FIN) fin_tile_of_workgroup(p.fin, b, t0);
  const int tid = threadIdx.x, lane = tid & 63, ql = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (POST && p.zero_words != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid < p.n_zero) p.zero_words[tid] = 0;
  const int len = p.num_ims ? min((int)p.num_ims[b] + 1, p.T) : p.T;               // valid tokens: special token + patches
  if (p.skip_padding && t0 >= len) return;
  const int fbase = 16 * OT * wave + 4 * g;                                       // this lane's first feature (tile 0)
  WS_STAMP(0);

  // ---- weight stream of this wave: [unit][wave][kb][ot][plane][64 lanes][16 B]; a k32 step = OT x 2 loads of 1 KiB
  u32x4 wr[NPF][OT][2];
  auto wload = [&](auto S_) __attribute__((always_inline)) {
    constexpr int S = decltype(S_)::value;
    if constexpr (S < NSTEPS) {
      constexpr bool in_post = S < POST_STEPS;
      constexpr int s = in_post ? S : S - POST_STEPS;
      const char* src = (in_post ? p.w_post : p.w_qkv) + (int64_t)(s / KB) * G::UNIT + wave * G::WAVE_UNIT + (s % KB) * G::STEP + lane * 16;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wr[S % NPF][ot][pl] = ldg_u32x4(src + (ot * 2 + pl) * FRAG);
    }
  };

  // ---- prologue: the first activation image, the head of the weight stream, the small vectors, the residual rows.  Every load is
  // ISSUED before the first one is waited for, in the order their data is needed (loads return in order): written as "load, store to
  // LDS" per vector, the prologue was four dependent memory round trips - the bias / LayerNorm vectors drained the queue in front
  // of the attention-output image - and inside a level every one of them comes from beyond the L2 (5.3 us of a 23-us kernel).
  int tokc[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) tokc[tt] = min(t0 + 16 * tt + ql, p.T - 1);
  constexpr int NI = G::ACT / (16 * 64 * NW);          // 16-byte pieces of the image per thread
  constexpr int NB1 = (4 * DM + 64 * NW - 1) / (64 * NW), NBQ = (3 * DM + 64 * NW - 1) / (64 * NW);
  static_assert(DM <= 64 * NW, "one thread per feature for the small vectors");
  const bool img_in = POST && p.attn_img != nullptr;
  u32x4 oi[POST ? NI : 1];
  if (img_in) {
    // the attention kernel writes one image per 64-TOKEN GROUP, [kb][4 tiles][plane]; this workgroup takes tiles tt0 .. tt0 + TT - 1
    constexpr int ACT64 = KB * 4 * 2 * FRAG, ROWB = TT * 2 * FRAG;                 // bytes per group / per k32 block of this workgroup
    const char* src = p.attn_img + ((int64_t)b * (p.Tp / 64) + (t0 / 64)) * ACT64 + ((t0 >> 4) & 3) * 2 * FRAG;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int piece = tid + i * 64 * NW, kb = piece / (ROWB / 16), r = piece % (ROWB / 16);
      oi[POST ? i : 0] = ldg_u32x4(src + (int64_t)kb * (4 * 2 * FRAG) + r * 16);
    }
  }
  static_for<0, NPF - 1>([&](auto s) { wload(s); });
  float v9[POST ? 9 : 1], vb1[POST ? NB1 : 1], vbq[QKV ? NBQ : 1];
  if constexpr (POST) {
    const float* const vecs[9] = {p.bo, p.ln1g, p.ln1b, p.cab, p.ln2g, p.ln2b, p.b2, p.ln3g, p.ln3b};
#pragma unroll
    for (int j = 0; j < 9; ++j) v9[j] = tid < DM ? vecs[j][tid] : 0.f;
#pragma unroll
    for (int q = 0; q < NB1; ++q) vb1[q] = tid + q * 64 * NW < 4 * DM ? p.b1[tid + q * 64 * NW] : 0.f;
  }
  if constexpr (QKV) {
#pragma unroll
    for (int q = 0; q < NBQ; ++q) vbq[q] = tid + q * 64 * NW < 3 * DM ? p.bqkv[tid + q * 64 * NW] : 0.f;
  }
  f32x4 xr[OT][TT];                 // POST: the residual stream of this lane's (feature, token) set
  if constexpr (POST) {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) xr[ot][tt] = ldg_f32x4(p.x_in + ((int64_t)b * p.T + tokc[tt]) * DM + fbase + 16 * ot);
  }
  // ... and only now the LDS stores, in arrival order
  if (img_in) {
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(sAct + (tid + i * 64 * NW) * 16) = oi[POST ? i : 0];
  }

  // first B-operand image: the attention output as an image (above) or as rows, or the input rows themselves (QKV only)
  if (img_in) {
  } else if constexpr (FIN) {
    // the input rows are built HERE from the raw result of the importance / projection GEMM (finish_qkv.h); the weight loads and
    // the bias vector issued above land under it
    float* const sTok = sBqkv + 3 * DM;
#ifdef PATHS_WS_STAMPS
    fin_tokens(p.fin, b, t0, tid, sTok, sTok + TOK * TOK_LD, p.stamps);
#else
    fin_tokens(p.fin, b, t0, tid, sTok, sTok + TOK * TOK_LD, nullptr);
#endif
    __syncthreads();
    fin_store_tokens(p.fin, b, t0, tid, sTok);
    for (int kb = wave; kb < KB; kb += NW) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const float* row = sTok + (16 * tt + ql) * TOK_LD + 32 * kb + 4 * g;
        const f32x4 a = *reinterpret_cast<const f32x4*>(row), c = *reinterpret_cast<const f32x4*>(row + 16);
        const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        u32x4 hi, lo;
        split8h(v, hi, lo);
        *reinterpret_cast<u32x4*>(sAct + ((kb * TT + tt) * 2) * FRAG + lane * 16) = hi;
        *reinterpret_cast<u32x4*>(sAct + ((kb * TT + tt) * 2 + 1) * FRAG + lane * 16) = lo;
      }
    }
  } else {
    const float* src = POST ? p.attn : p.x_in;
    for (int kb = wave; kb < KB; kb += NW) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const float* row = src + ((int64_t)b * p.T + tokc[tt]) * DM + 32 * kb + 4 * g;
        const f32x4 a = ldg_f32x4(row), c = ldg_f32x4(row + 16);
        const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        u32x4 hi, lo;
        split8h(v, hi, lo);
        *reinterpret_cast<u32x4*>(sAct + ((kb * TT + tt) * 2) * FRAG + lane * 16) = hi;
        *reinterpret_cast<u32x4*>(sAct + ((kb * TT + tt) * 2 + 1) * FRAG + lane * 16) = lo;
      }
    }
  }
  // the small vectors (first read after the first product / in the in_proj epilogue): behind every other load of the prologue
  if constexpr (POST) {
    if (tid < DM) {
#pragma unroll
      for (int j = 0; j < 9; ++j) sVec[j * DM + tid] = v9[j];
    }
#pragma unroll
    for (int q = 0; q < NB1; ++q)
      if (tid + q * 64 * NW < 4 * DM) sB1[tid + q * 64 * NW] = vb1[q];
  }
  if constexpr (QKV) {
#pragma unroll
    for (int q = 0; q < NBQ; ++q)
      if (tid + q * 64 * NW < 3 * DM) sBqkv[tid + q * 64 * NW] = vbq[q];
  }
  __syncthreads();
  WS_STAMP(1);

  // acc[ot][tt] += W_unit[this wave's rows 16 ot ..][all DM k] . X[k][token tile tt]   (SWAP: the operands exchanged, rows = tokens)
  // The B fragments of k32 step kb + 1 are read from LDS while the MFMAs of step kb run (two register sets; a step's reads were
  // otherwise exposed: ~300 of ~700 cycles per step).  PRE: block 0 is already in bf[0] (the previous unit fetched it, NEXT).
  u32x4 bf[2][TT][2];
  auto ldsb = [&](auto slot_, const char* sB, int kb) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_)::value;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) bf[slot][tt][pl] = *reinterpret_cast<const u32x4*>(sB + ((kb * TT + tt) * 2 + pl) * FRAG + lane * 16);
  };
  auto unit = [&](auto S0_, auto swap_, auto pre_, f32x4 (&acc)[OT][TT], const char* sB, const char* sNext) __attribute__((always_inline)) {
    constexpr int S0 = decltype(S0_)::value;
    constexpr bool SWAP = decltype(swap_)::value, PRE = decltype(pre_)::value;
    static_assert(KB % 2 == 0, "fragment double buffer: even number of k32 steps per unit");
    if constexpr (!PRE) ldsb(std::integral_constant<int, 0>{}, sB, 0);
    static_for<0, KB>([&](auto kb_) {
      constexpr int kb = decltype(kb_)::value, S = S0 + kb, slot = S % NPF, cur = kb & 1;
      wload(std::integral_constant<int, S + NPF - 1>{});
      if constexpr (kb + 1 < KB) ldsb(std::integral_constant<int, cur ^ 1>{}, sB, kb + 1);
      else if (sNext != nullptr) ldsb(std::integral_constant<int, 0>{}, sNext, 0);
      __builtin_amdgcn_sched_barrier(0);      // (hipcc otherwise sinks the fragment reads down to their first use: next step's head)
      // hi*hi + hi*lo + lo*hi, smallest first
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
          acc[ot][tt] = SWAP ? mfma_f16(bf[cur][tt][0], wr[slot][ot][1], acc[ot][tt]) : mfma_f16(wr[slot][ot][1], bf[cur][tt][0], acc[ot][tt]);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
          acc[ot][tt] = SWAP ? mfma_f16(bf[cur][tt][1], wr[slot][ot][0], acc[ot][tt]) : mfma_f16(wr[slot][ot][0], bf[cur][tt][1], acc[ot][tt]);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
          acc[ot][tt] = SWAP ? mfma_f16(bf[cur][tt][0], wr[slot][ot][0], acc[ot][tt]) : mfma_f16(wr[slot][ot][0], bf[cur][tt][0], acc[ot][tt]);
    });
  };
  constexpr std::false_type NO{};
  constexpr std::true_type YES{};
  auto zero = [&](f32x4 (&a)[OT][TT]) {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) a[ot][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // this wave's features of all 64 tokens -> fp16 hi | lo fragments of the k32 blocks they belong to (tile pair = one block)
  auto put_act = [&](const f32x4 (&x)[OT][TT], char* img) __attribute__((always_inline)) {
    if constexpr (OT % 2 == 0) {
#pragma unroll
      for (int op = 0; op < OT / 2; ++op) {
        const int blk = (OT / 2) * wave + op;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const float v[8] = {x[2 * op][tt][0], x[2 * op][tt][1], x[2 * op][tt][2], x[2 * op][tt][3],
                              x[2 * op + 1][tt][0], x[2 * op + 1][tt][1], x[2 * op + 1][tt][2], x[2 * op + 1][tt][3]};
          u32x4 hi, lo;
          split8h(v, hi, lo);
          *reinterpret_cast<u32x4*>(img + ((blk * TT + tt) * 2) * FRAG + lane * 16) = hi;
          *reinterpret_cast<u32x4*>(img + ((blk * TT + tt) * 2 + 1) * FRAG + lane * 16) = lo;
        }
      }
    } else {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const int gt = OT * wave + ot, blk = gt >> 1, half = gt & 1;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const float v[4] = {x[ot][tt][0], x[ot][tt][1], x[ot][tt][2], x[ot][tt][3]};
          u32x2 hi, lo;
          split4h(v, hi, lo);
          *reinterpret_cast<u32x2*>(img + ((blk * TT + tt) * 2) * FRAG + lane * 16 + 8 * half) = hi;
          *reinterpret_cast<u32x2*>(img + ((blk * TT + tt) * 2 + 1) * FRAG + lane * 16 + 8 * half) = lo;
        }
      }
    }
  };
  // LayerNorm over the DM features of every token: the wave's DM/4 features give (mean, M2) per token, the four waves' pairs
  // cross through LDS and are merged with Chan's formula (two-pass accuracy, one barrier)
  auto layernorm = [&](f32x4 (&x)[OT][TT], const float* gamma, const float* beta, int sbuf) __attribute__((always_inline)) {
    float* st = sStat + sbuf * NW * TOK * 2;
    constexpr float inv_w = 1.0f / (DM / NW);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      float s = 0.f;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) s += (x[ot][tt][0] + x[ot][tt][1]) + (x[ot][tt][2] + x[ot][tt][3]);
      s = sum_xor32(sum_xor16(s));
      const float mw = s * inv_w;
      float v = 0.f;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float c = x[ot][tt][r] - mw; v += c * c; }
      v = sum_xor32(sum_xor16(v));
      if (g == 0) *reinterpret_cast<f32x2*>(st + (wave * TOK + 16 * tt + ql) * 2) = f32x2{mw, v};
    }
    __syncthreads();
    f32x4 gm[OT], bt[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      gm[ot] = *reinterpret_cast<const f32x4*>(gamma + fbase + 16 * ot);
      bt[ot] = *reinterpret_cast<const f32x4*>(beta + fbase + 16 * ot);
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      f32x2 ms[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) ms[w] = *reinterpret_cast<const f32x2*>(st + (w * TOK + 16 * tt + ql) * 2);
      const float mean = ((ms[0][0] + ms[1][0]) + (ms[2][0] + ms[3][0])) * (1.0f / NW);
      float m2 = (ms[0][1] + ms[1][1]) + (ms[2][1] + ms[3][1]);
      float dev = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { const float dlt = ms[w][0] - mean; dev += dlt * dlt; }
      m2 += dev * (float)(DM / NW);
      const float rstd = 1.0f / sqrtf(m2 * (1.0f / DM) + p.eps);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[ot][tt][r] = (x[ot][tt][r] - mean) * rstd * gm[ot][r] + bt[ot][r];
    }
  };

  if constexpr (POST) {
    // ---- out_proj(attn) + residual -> norm1 -> + cross-attention bias -> norm2
    {
      f32x4 acc[OT][TT];
      zero(acc);
      unit(std::integral_constant<int, 0>{}, NO, NO, acc, sAct, nullptr);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const f32x4 bo = *reinterpret_cast<const f32x4*>(sVec + fbase + 16 * ot);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) xr[ot][tt] = xr[ot][tt] + (acc[ot][tt] * p.inv_wo + bo);
      }
    }
    WS_STAMP(2);
    layernorm(xr, sVec + DM, sVec + 2 * DM, 0);
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const f32x4 cab = *reinterpret_cast<const f32x4*>(sVec + 3 * DM + fbase + 16 * ot);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) xr[ot][tt] = xr[ot][tt] + cab;
    }
    layernorm(xr, sVec + 4 * DM, sVec + 5 * DM, 1);
    WS_STAMP(3);
    put_act(xr, sAct);                      // (every wave has finished reading the attention image: two barriers ago)
    __syncthreads();
    WS_STAMP(4);

    // ---- feed-forward in 4 hidden chunks of DM: hid = relu(W1_c x + b1_c) -> LDS -> y += W2[:, c] hid.  linear2 of chunk c and
    // linear1 of chunk c + 1 run back to back (no barrier between them): one 8-step pipelined sequence per chunk
    f32x4 y[OT][TT], hid[OT][TT];
    zero(y);
    auto put_hidden = [&](auto c_) __attribute__((always_inline)) {
      constexpr int c = decltype(c_)::value;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sB1 + c * DM + fbase + 16 * ot);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) hid[ot][tt][r] = fmaxf(fmaf(hid[ot][tt][r], p.inv_w1, b1[r]), 0.f);
      }
      put_act(hid, sHid + (c & 1) * G::ACT);      // (chunk c-2's readers all passed the barrier of chunk c-1)
      __syncthreads();
    };
    zero(hid);
    unit(std::integral_constant<int, KB>{}, NO, NO, hid, sAct, nullptr);
    put_hidden(std::integral_constant<int, 0>{});
    static_for<0, NFF>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      unit(std::integral_constant<int, (2 + 2 * c) * KB>{}, NO, NO, y, sHid + (c & 1) * G::ACT, c + 1 < NFF ? sAct : nullptr);
      if constexpr (c + 1 < NFF) {
        zero(hid);
        unit(std::integral_constant<int, (3 + 2 * c) * KB>{}, NO, YES, hid, sAct, nullptr);
        put_hidden(std::integral_constant<int, c + 1>{});
      }
      WS_STAMP(5 + c);
    });
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(sVec + 6 * DM + fbase + 16 * ot);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) xr[ot][tt] = xr[ot][tt] + (y[ot][tt] * p.inv_w2 + b2);
    }
    layernorm(xr, sVec + 7 * DM, sVec + 8 * DM, 0);
    WS_STAMP(9);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
      if (t0 + 16 * tt + ql < p.T) {
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
          *reinterpret_cast<f32x4*>(p.x_out + ((int64_t)b * p.T + t0 + 16 * tt + ql) * DM + fbase + 16 * ot) = xr[ot][tt];
      }
    if constexpr (QKV) {
      put_act(xr, sAct);                    // (the last reader of the x1 image, linear1 of chunk 3, is two barriers back)
      __syncthreads();
    }
    WS_STAMP(10);
  }

  if constexpr (QKV && ROWS) {
    // ---- in_proj as fp32 rows (any DM % 64 == 0, any head count): q | k | v = x W^T + b, token-major, unscaled - what the
    // shape-generic attention kernels read (csrc/generic.hip, attn_h3_any.hip); three units, features on the accumulator rows
    static_for<0, 3>([&](auto which_) {
      constexpr int which = decltype(which_)::value;
      f32x4 acc[OT][TT];
      zero(acc);
      if constexpr (which == 0) unit(std::integral_constant<int, POST_STEPS>{}, NO, NO, acc, sAct, sAct);
      else if constexpr (which == 1) unit(std::integral_constant<int, POST_STEPS + KB>{}, NO, YES, acc, sAct, sAct);
      else unit(std::integral_constant<int, POST_STEPS + 2 * KB>{}, NO, YES, acc, sAct, nullptr);
      f32x4 bb[OT];
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) bb[ot] = *reinterpret_cast<const f32x4*>(sBqkv + which * DM + fbase + 16 * ot);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int tok = t0 + 16 * tt + ql;
        if (tok < p.T) {
#pragma unroll
          for (int ot = 0; ot < OT; ++ot)
            *reinterpret_cast<f32x4*>(p.qkv_rows + ((int64_t)b * p.T + tok) * p.ld_qkv + which * DM + fbase + 16 * ot) = acc[ot][tt] * p.inv_wqkv + bb[ot];
        }
      }
    });
  } else if constexpr (QKV) {
    static_assert(OT == 2 || OT == 3, "in_proj images: head_dim 32 or 48 (wave = head)");
    // ---- in_proj: wave w = head w (head_dim 16 OT).  q, k: features on the accumulator rows; v: operands swapped, tokens on the rows.
    // Images in the layout csrc/attn_h3_any.hip documents (= csrc/attn_x6.hip's at head_dim 32): NKB k-blocks of 32 score dims per
    // 16-token tile (head_dim 48: the second block holds dims 32..47 and 16 zeros - the same positions in Q and K), OT dv tiles per
    // 32-key group.  The order of the dims inside a k-block is this kernel's accumulator order, identically for Q and K.
    constexpr int NKB = (OT + 1) / 2;
    const int64_t img_qk = (int64_t)p.B * NW * p.Tp * (NKB * 128);   // bytes of the Q (or K) image: 32 NKB dims x 2 planes x 2 B per token and head
    char* const hbase = p.qkv_img + ((int64_t)b * NW + wave) * p.Tp * (NKB * 128);
    char* const vbase = p.qkv_img + 2 * img_qk + ((int64_t)b * NW + wave) * p.Tp * (OT * 64);
    static_for<0, 2>([&](auto which_) {
      constexpr int which = decltype(which_)::value;
      f32x4 acc[OT][TT];
      zero(acc);
      if constexpr (which == 0) unit(std::integral_constant<int, POST_STEPS>{}, NO, NO, acc, sAct, sAct);
      else unit(std::integral_constant<int, POST_STEPS + KB>{}, NO, YES, acc, sAct, sAct);
      WS_STAMP(11 + 2 * which);
      const float sc = which == 0 ? p.qscale : 1.0f;
      f32x4 bb[OT];
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) bb[ot] = *reinterpret_cast<const f32x4*>(sBqkv + which * DM + fbase + 16 * ot);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int tok = t0 + 16 * tt + ql;
        const bool live = which == 0 ? tok < p.T : tok < len;    // masked keys: K = 0 (V = 0 below: 0 * garbage would poison O)
#pragma unroll
        for (int kk = 0; kk < NKB; ++kk) {
          float v[8];
#pragma unroll
          for (int o2 = 0; o2 < 2; ++o2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int ot = 2 * kk + o2;
              v[4 * o2 + r] = (ot < OT && live) ? (acc[ot < OT ? ot : 0][tt][r] * p.inv_wqkv + bb[ot < OT ? ot : 0][r]) * sc : 0.f;
            }
          u32x4 hi, lo;
          split8h(v, hi, lo);
          char* dst = hbase + which * img_qk + ((int64_t)((t0 >> 4) + tt) * NKB + kk) * 2 * FRAG + lane * 16;   // [16-token tile][k-block][plane][lane]
          *reinterpret_cast<u32x4*>(dst) = hi;
          *reinterpret_cast<u32x4*>(dst + FRAG) = lo;
        }
      }
      WS_STAMP(12 + 2 * which);
    });
    {
      f32x4 acc[OT][TT];                    // [dv tile][token tile]: column = dim 16 ot + ql, rows = tokens 16 tt + 4 g + r
      zero(acc);
      unit(std::integral_constant<int, POST_STEPS + 2 * KB>{}, YES, YES, acc, sAct, nullptr);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const float bv = sBqkv[2 * DM + 16 * OT * wave + 16 * ot + ql];
#pragma unroll
        for (int u = 0; u < TT / 2; ++u) {  // 32-key group: k-slot (g, j) <-> key 4 g + (j & 3) + 16 (j >> 2)
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int tok = t0 + 32 * u + 16 * (j >> 2) + 4 * g + (j & 3);
            v[j] = tok < len ? acc[ot][2 * u + (j >> 2)][j & 3] * p.inv_wqkv + bv : 0.f;
          }
          u32x4 hi, lo;
          split8h(v, hi, lo);
          char* dst = vbase + (int64_t)(((t0 >> 5) + u) * OT + ot) * 2 * FRAG + lane * 16;   // [32 keys][dv tile][plane][lane]
          *reinterpret_cast<u32x4*>(dst) = hi;
          *reinterpret_cast<u32x4*>(dst + FRAG) = lo;
        }
      }
    }
  }
  WS_STAMP(15);
}

// One fragment-pair chunk per workgroup: unit `blockIdx.x`, all of its [wave][kb][ot][plane] fragments.
struct WsPackJob { const float* w; int ldw; int row0; int k0; float scale; };
struct WsPackJobs { WsPackJob j[12]; };

template <int DM>
__global__ void __launch_bounds__(256)
tlayer_pack_ws_kernel(WsPackJobs jobs, char* __restrict__ out) {
  using G = Geo<DM>;
  const WsPackJob jb = jobs.j[blockIdx.x];
  char* dst = out + (int64_t)blockIdx.x * G::UNIT;
  for (int piece = threadIdx.x; piece < G::UNIT / 16; piece += 256) {
    const int lane = piece & 63, f = piece >> 6;                // fragment f = ((wave * KB + kb) * OT + ot) * 2 + plane
    const int plane = f & 1, ot = (f >> 1) % G::OT, kb = ((f >> 1) / G::OT) % G::KB, wave = (f >> 1) / (G::OT * G::KB);
    const int ql = lane & 15, g = lane >> 4;
    const float* src = jb.w + (int64_t)(jb.row0 + 16 * (G::OT * wave + ot) + ql) * jb.ldw + jb.k0 + 32 * kb;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[4 * g + (j & 3) + 16 * (j >> 2)] * jb.scale;
    u32x4 hi, lo;
    split8h(v, hi, lo);
    *reinterpret_cast<u32x4*>(dst + (int64_t)f * FRAG + lane * 16) = plane == 0 ? hi : lo;
  }
}

template <int DM>
size_t ws_lds_bytes(bool post, bool qkv) {
  return Geo<DM>::ACT + (post ? 2 * Geo<DM>::ACT : 0) + 2 * NW * TOK * 2 * sizeof(float) + ((post ? 13 * DM : 0) + (qkv ? 3 * DM : 0)) * sizeof(float);
}

template <int DM, bool POST, bool QKV, bool ROWS = false>
int launch_ws(const WsParams& p, hipStream_t stream) {
  const size_t lds = ws_lds_bytes<DM>(POST, QKV);
  PATHS_LDS_OPT_IN((tlayer_ws_kernel<DM, POST, QKV, ROWS>), 160 * 1024, "token_layer_ws");
  // (> 80 KiB per workgroup: one workgroup per CU, so a grid of ~one workgroup per CU spreads over the whole chip)
  const size_t ask = TT == 2 ? lds : (lds > 84 * 1024 ? lds : 84 * 1024);
  hipLaunchKernelGGL((tlayer_ws_kernel<DM, POST, QKV, ROWS>), dim3((p.T + TOK - 1) / TOK, p.B), dim3(64 * NW), ask, stream, p);
  PATHS_LAUNCH_CHECK("token_layer_ws");
  return PATHS_OK;
}

// Partial importance logits of a 64-slot tile -> sAlpha [2][TOK] (the two column halves' sums; the caller adds them after a barrier):
// the hidden-unit columns of the raw GEMM result, the same summation tree as fin_tokens / EpiImpProj (bit-identical alpha whichever
// kernel computes it).  All 256 threads.
__device__ __forceinline__ void fin_alpha_partials(const FinQkvParams& f, int b, int t0, int tid, float* sAlpha) {
  constexpr int NZ = 2;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  float part[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) part[r] = 0.f;
  const int64_t trow = (((int64_t)b * f.N + t0) >> 5) + wm;
  f32x4 v[2][4][NZ];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const f32x4* t = reinterpret_cast<const f32x4*>(f.ws + (trow * 8 + 4 * wn + j) * 1024) + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int z = 0; z < NZ; ++z) v[j][q][z] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(t + 64 * q) + z * f.zstride);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = 64 * wn + 32 * j + (lane & 31);
    const float bb = f.b1[u], w = f.w2[u];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 sum = v[j][q][0] + v[j][q][1];
#pragma unroll
      for (int e = 0; e < 4; ++e) part[4 * q + e] += fmaxf(fmaf(sum[e], f.acc_scale, bb), 0.f) * w;
    }
  }
  float p8[8], p4[4], p2[2], p1;
  {
    const bool up = (lane & 16) != 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float keep = up ? part[8 + k] : part[k], send = up ? part[k] : part[8 + k]; p8[k] = keep + __shfl_xor(send, 16); }
  }
  {
    const bool up = (lane & 8) != 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float keep = up ? p8[4 + k] : p8[k], send = up ? p8[k] : p8[4 + k]; p4[k] = keep + __shfl_xor(send, 8); }
  }
  {
    const bool up = (lane & 4) != 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) { const float keep = up ? p4[2 + k] : p4[k], send = up ? p4[k] : p4[2 + k]; p2[k] = keep + __shfl_xor(send, 4); }
  }
  {
    const bool up = (lane & 2) != 0;
    const float keep = up ? p2[1] : p2[0], send = up ? p2[0] : p2[1];
    p1 = keep + __shfl_xor(send, 2);
  }
  p1 += __shfl_xor(p1, 1);
  if ((lane & 1) == 0) sAlpha[wn * TOK + 32 * wm + c32_row((lane >> 1) & 15, lane)] = p1;
}

// The importance half of the finish alone (phase 2 of paths_importance_qkv_x6): alpha of every token slot of a 64-slot tile from the
// hidden-unit columns of the raw GEMM result; what the top-K waits for when the tokens / in_proj finish runs on another stream.
__global__ void __launch_bounds__(256)
finish_importance_kernel(FinQkvParams f) {
  __shared__ float sAlpha[2 * TOK];
  int b = blockIdx.y, t0 = blockIdx.x * TOK;
  fin_tile_of_workgroup(f, b, t0);
  const int tid = threadIdx.x;
  const int nim = (int)f.num_ims[b];
  if (t0 >= f.N || (f.skip_padding && t0 >= nim)) return;
  fin_alpha_partials(f, b, t0, tid, sAlpha);
  __syncthreads();
  if (tid < TOK) {
    const int sl = t0 + tid;
    if (sl < f.N) f.importance[(int64_t)b * f.N + sl] = sl < nim ? sigmoid_acc((sAlpha[tid] + sAlpha[TOK + tid]) + *f.b2) : 0.f;
  }
}

// ---- importance finish AND top-K in one launch (phase 8 of paths_importance_qkv_x6; reference model/paths.py:95 +
// data_utils/slide.py:294-301).  The two were tiny latency-bound kernels back to back on the recursion's critical path (8 + 12 us
// and two launch boundaries for ~0.3 us of arithmetic).  A workgroup owns 64 slots of one slide in both halves: it finishes their
// alpha, publishes them (write-through stores, drained, then ONE release add on the slide's arrival counter), waits until the
// slide's ceil(num_ims / 64) workgroups have arrived (bounded spin; the waited-for workgroups are part of this launch and small
// enough - 20 KB of LDS - to be resident together even beside other kernels), then ranks its 64 elements against all scores of the
// slide exactly as topk_rank_kernel (csrc/select.hip) does: 64-bit keys (score descending, index ascending), rank = number of
// smaller keys, kept iff rank < count, output position = rank.  The last workgroup of a slide to leave zeroes its two counters.
__device__ __forceinline__ unsigned long long fin_topk_key(float score, int idx) {
  uint32_t u = __float_as_uint(score);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // monotone float -> uint
  return ((unsigned long long)(~u) << 32) | (uint32_t)idx;
}

__global__ void __launch_bounds__(256)
finish_importance_topk_kernel(FinQkvParams f, FinTopkParams k) {
  extern __shared__ __attribute__((aligned(16))) char smem_tk[];
  float* const sAlpha = reinterpret_cast<float*>(smem_tk);                                   // [2][TOK]
  int* const part = reinterpret_cast<int*>(smem_tk + 2 * TOK * sizeof(float));              // [4][64] partial counts
  int* const sFlag = part + 256;
  unsigned long long* const keys = reinterpret_cast<unsigned long long*>(smem_tk + 2 * TOK * sizeof(float) + 260 * sizeof(int));   // (1552 bytes in: 16-byte aligned)
  const int b = blockIdx.y, t0 = blockIdx.x * TOK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = (int)f.num_ims[b];
  const int count = k.keep < 0 ? n : min(n, k.keep);
  const int i = t0 + lane;                              // this lane's element in the ranking
  if (blockIdx.x == 0 && tid == 0) k.keep_count[b] = count;
  auto row_addr = [&](int idx) { return (int64_t)reinterpret_cast<uintptr_t>(k.row_base + ((int64_t)b * f.N + idx) * k.row_ld); };
  // entries [count, ldk) of the row table point at the zero row (every workgroup covers its own 64 positions)
  if (k.kept_rows && wave == 0 && i >= count && i < k.ldk) k.kept_rows[(int64_t)b * k.ldk + i] = (int64_t)reinterpret_cast<uintptr_t>(k.zero_row);
  if (t0 >= f.N || t0 >= n) return;                     // no valid slot here (workgroup-uniform; the importance buffer is zero there)
  // ---- alpha of this tile
  fin_alpha_partials(f, b, t0, tid, sAlpha);
  __syncthreads();
  if (tid < TOK) {
    const int sl = t0 + tid;
    if (sl < f.N) __hip_atomic_store(f.importance + (int64_t)b * f.N + sl, sl < n ? sigmoid_acc((sAlpha[tid] + sAlpha[TOK + tid]) + *f.b2) : 0.f,
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (k.keep < 0) {                                     // keep all, original order (slide.py:294 not taken): nothing to wait for
    if (wave == 0 && i < n) {
      k.keep_idx[(int64_t)b * k.ldk + i] = i;
      if (k.kept_rows) k.kept_rows[(int64_t)b * k.ldk + i] = row_addr(i);
    }
    return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- arrival of the slide's workgroups
  const int need = (n + TOK - 1) / TOK;
  int* const cnt = k.counters + 2 * b;
  if (tid == 0) {
    __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0, ok = 1;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) {                        // give up loudly rather than hang: this slide's selection is garbage
        if (k.status) atomicOr(k.status, 4);
        ok = 0;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *sFlag = ok;
  }
  __syncthreads();
  // ---- all n scores of the slide -> keys in LDS (agent-scope loads: other workgroups of this launch wrote them)
  const int np = (n + 7) & ~7;
  const float* s = f.importance + (int64_t)b * f.N;
  for (int j = tid; j < np; j += 256) keys[j] = j < n ? fin_topk_key(__hip_atomic_load(s + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), j) : ~0ull;
  __syncthreads();
  const unsigned long long mine = i < n ? keys[i] : 0ull;
  const int pairs = np >> 1, q0 = (pairs * wave) >> 2, q1 = (pairs * (wave + 1)) >> 2;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2* kp = reinterpret_cast<const u64x2*>(keys);
  int c = 0, q = q0;
  for (; q + 4 <= q1; q += 4) {
    const u64x2 a = kp[q], cc = kp[q + 1], d = kp[q + 2], e = kp[q + 3];
    c += (a[0] < mine) + (a[1] < mine) + (cc[0] < mine) + (cc[1] < mine) + (d[0] < mine) + (d[1] < mine) + (e[0] < mine) + (e[1] < mine);
  }
  for (; q < q1; ++q) { const u64x2 a = kp[q]; c += (a[0] < mine) + (a[1] < mine); }
  part[wave * 64 + lane] = c;
  __syncthreads();
  if (wave == 0 && i < n) {
    const int rank = part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane];
    if (rank < count) {
      k.keep_idx[(int64_t)b * k.ldk + rank] = i;
      if (k.kept_rows) k.kept_rows[(int64_t)b * k.ldk + rank] = row_addr(i);
    }
  }
  // ---- the last workgroup of the slide to get here leaves the counters zero for the next launch
  if (tid == 0 && __hip_atomic_fetch_add(cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == need - 1) {
    __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace

#ifdef PATHS_WS_STAMPS
static unsigned long long* g_ws_stamps = nullptr;
extern "C" void paths_ws_stamp_buffer(unsigned long long* p) { g_ws_stamps = p; }     // development hook (tools/ws_time.py)
#endif

// finish_qkv.h: phases bit 2 = the importance-only finish, bit 4 = tokens + in_proj images (alpha computed there unless
// alpha_from_importance); both stop-event capable (the recursion's forks ride on them)
int paths_launch_finish_importance(const FinQkvParams& f, hipStream_t stream) {
  PATHS_LAUNCH_STOP(finish_importance_kernel, dim3(f.Tp / TOK, f.B), dim3(256), 0, stream, f);
  PATHS_LAUNCH_CHECK("importance_qkv_x6(importance finish)");
  return PATHS_OK;
}

int paths_launch_finish_importance_topk(const FinQkvParams& f, const FinTopkParams& k, hipStream_t stream) {
  const size_t lds = 2 * TOK * sizeof(float) + 260 * sizeof(int) + (size_t)(((f.N + 7) & ~7) + 2) * 8;
  PATHS_LDS_OPT_IN(finish_importance_topk_kernel, 2 * TOK * sizeof(float) + 260 * sizeof(int) + (8192 + 2) * 8, "importance_qkv_x6(importance + top-K finish)");
  // the grid covers every slot of the slide AND every position of the kept-row table (ldk <= N)
  PATHS_LAUNCH_STOP(finish_importance_topk_kernel, dim3((f.N + TOK - 1) / TOK, f.B), dim3(256), lds, stream, f, k);
  PATHS_LAUNCH_CHECK("importance_qkv_x6(importance + top-K finish)");
  return PATHS_OK;
}

int paths_launch_finish_qkv(const FinQkvParams& f, hipStream_t stream) {
  static_assert(TT == 4, "the fused finish owns 64-token tiles");
  WsParams p{};
  p.bqkv = f.bqkv; p.inv_wo = p.inv_w1 = p.inv_w2 = 1.0f; p.inv_wqkv = f.inv_wqkv;
  p.w_qkv = reinterpret_cast<const char*>(f.w_qkv);
  p.qkv_img = reinterpret_cast<char*>(f.qkv_img);
  p.num_ims = f.num_ims; p.T = f.T; p.Tp = f.Tp; p.B = f.B; p.skip_padding = f.skip_padding; p.qscale = f.qscale; p.eps = 0.f;
  p.fin = f;
#ifdef PATHS_WS_STAMPS
  p.stamps = g_ws_stamps;
#endif
  const size_t lds = ws_lds_bytes<128>(false, true) + (size_t)(TOK * TOK_LD + 3 * TOK) * sizeof(float);
  PATHS_LDS_OPT_IN((tlayer_ws_kernel<128, false, true, false, true>), 160 * 1024, "importance_qkv_x6(finish)");
  // (> 80 KiB per workgroup: one workgroup per CU - a grid of ~one workgroup per CU spreads over the whole chip)
  const size_t ask = lds > 84 * 1024 ? lds : 84 * 1024;
  PATHS_LAUNCH_STOP((tlayer_ws_kernel<128, false, true, false, true>), dim3(f.Tp / TOK, f.B), dim3(64 * NW), ask, stream, p);
  PATHS_LAUNCH_CHECK("importance_qkv_x6(finish)");
  return PATHS_OK;
}


extern "C" {

// bytes of the weight images of paths_token_layer_ws: part 0 = (Wo, W1, W2) of one layer (9 units), part 1 = in_proj (3 units)
int64_t paths_tlayer_ws_image_bytes(int part, int d) {
  return (int64_t)(part == 0 ? 9 : 3) * d * d * 4;
}

// Pack one part: part 0 = (wo [d,d], w1 [4d,d], w2 [d,4d]) scaled by the powers of two s_a, s_b, s_c; part 1 = wqkv [3d,d] by s_a.
int paths_tlayer_pack_ws(int part, const float* wa, const float* wb, const float* wc, float s_a, float s_b, float s_c, void* out, int d,
                         hipStream_t stream) {
  PATHS_REQUIRE(d == 128 || d == 192, "tlayer_pack_ws: trans_dim 128 or 192 (got %d)", d);
  PATHS_REQUIRE((part == 0 && wa && wb && wc) || (part == 1 && wa), "tlayer_pack_ws: bad arguments");
  PATHS_REQUIRE(out != nullptr && (uintptr_t)out % 16 == 0, "tlayer_pack_ws: out must be 16-byte aligned");
  WsPackJobs jobs;
  int n = 0;
  if (part == 0) {
    jobs.j[n++] = WsPackJob{wa, d, 0, 0, s_a};
    for (int c = 0; c < NFF; ++c) {
      jobs.j[n++] = WsPackJob{wb, d, d * c, 0, s_b};
      jobs.j[n++] = WsPackJob{wc, 4 * d, 0, d * c, s_c};
    }
  } else {
    for (int c = 0; c < 3; ++c) jobs.j[n++] = WsPackJob{wa, d, d * c, 0, s_a};
  }
  if (d == 192) hipLaunchKernelGGL(tlayer_pack_ws_kernel<192>, dim3(n), dim3(256), 0, stream, jobs, reinterpret_cast<char*>(out));
  else hipLaunchKernelGGL(tlayer_pack_ws_kernel<128>, dim3(n), dim3(256), 0, stream, jobs, reinterpret_cast<char*>(out));
  PATHS_LAUNCH_CHECK("tlayer_pack_ws");
  return PATHS_OK;
}

// The decoder layer's token-row chain (reference model/aggregator.py:25-33, 70-72 = torch's post-LN TransformerDecoderLayer after
// the self-attention) and / or the next layer's in_proj, weight-stationary form.  w_post / w_qkv: paths_tlayer_pack_ws images with
// scales s_*.  attn: fp32 [B,T,d], or attn_img: the fragment image paths_attention_h3_img wrote.  qkv_images: a
// paths_attention_x6_workspace(B, T, 4, 32, 2) buffer (the operand images of paths_attention_x6 / _h3_img).  zero_words
// (optional): n_zero <= 256 int32 words set to 0 (the arrival counters of paths_token0_tail_ws, which follows on the same stream).
int paths_token_layer_ws(const float* x_in, const float* attn, const void* attn_img, float* x_out, const void* w_post, const void* w_qkv,
                         const float* bo, const float* ln1g, const float* ln1b, const float* cab, const float* ln2g, const float* ln2b,
                         const float* b1, const float* b2, const float* ln3g, const float* ln3b, const float* bqkv,
                         float s_wo, float s_w1, float s_w2, float s_wqkv, void* qkv_images, const int64_t* num_ims,
                         int B, int T, int d, int H, int do_post, int do_qkv, int skip_padding, float qscale, float eps,
                         int* zero_words, int n_zero, hipStream_t stream) {
  // trans_dim 192 (the reference's dataclass default, config.py:30): the post-attention chain only, attention output as fp32 rows (its
  // in_proj and attention run on the shape-generic kernels, whose operands are token-major fp32)
  // ... or the in_proj only, writing the head_dim-48 operand images of paths_attention_h3_any_img (4 heads: wave = head)
  PATHS_REQUIRE((d == 128 && H == 4) || (d == 192 && do_post && !do_qkv && attn != nullptr && attn_img == nullptr) || (d == 192 && H == 4 && !do_post && do_qkv),
                "token_layer_ws: trans_dim 128 / 4 heads, or trans_dim 192 with do_post only and fp32 attention rows, or 192 / 4 heads with do_qkv only (got %d, %d)", d, H);
  PATHS_REQUIRE(B > 0 && T > 0 && (do_post || do_qkv), "token_layer_ws: nothing to do");
  PATHS_REQUIRE(!skip_padding || num_ims, "token_layer_ws: skip_padding needs num_ims");
  PATHS_REQUIRE(x_in && (!do_post || ((attn || attn_img) && x_out && w_post && bo && ln1g && ln1b && cab && ln2g && ln2b && b1 && b2 && ln3g && ln3b)),
                "token_layer_ws: null operand (post)");
  PATHS_REQUIRE(!do_qkv || (w_qkv && bqkv && qkv_images), "token_layer_ws: null operand (in_proj)");
  PATHS_REQUIRE(n_zero >= 0 && n_zero <= 256 && (n_zero == 0 || (zero_words && do_post)), "token_layer_ws: zero_words needs do_post, n_zero <= 256");
  PATHS_REQUIRE(((uintptr_t)x_in | (uintptr_t)attn | (uintptr_t)attn_img | (uintptr_t)x_out | (uintptr_t)w_post | (uintptr_t)w_qkv | (uintptr_t)qkv_images) % 16 == 0,
                "token_layer_ws: buffers must be 16-byte aligned");
  WsParams p{x_in, attn, reinterpret_cast<const char*>(attn_img), x_out, reinterpret_cast<const char*>(w_post), reinterpret_cast<const char*>(w_qkv),
             bo, ln1g, ln1b, cab, ln2g, ln2b, b1, b2, ln3g, ln3b, bqkv,
             do_post ? 1.0f / s_wo : 1.0f, do_post ? 1.0f / s_w1 : 1.0f, do_post ? 1.0f / s_w2 : 1.0f, do_qkv ? 1.0f / s_wqkv : 1.0f,
             reinterpret_cast<char*>(qkv_images), nullptr, 0, num_ims, T, (T + 63) / 64 * 64, B, skip_padding, qscale, eps, zero_words, n_zero
#ifdef PATHS_WS_STAMPS
             , g_ws_stamps
#endif
  };
  if (d == 192) return do_qkv ? launch_ws<192, false, true>(p, stream) : launch_ws<192, true, false>(p, stream);
  if (do_post && do_qkv) return launch_ws<128, true, true>(p, stream);
  if (do_post) return launch_ws<128, true, false>(p, stream);
  return launch_ws<128, false, true>(p, stream);
}

// The same kernel for trans_dim 192 (the reference's dataclass default, config.py:30; any head count) with the in_proj result written
// as fp32 token-major rows qkv_rows [B*T][ld_qkv] = [q | k | v] (q UNscaled) - the operand of the shape-generic attention kernels -
// instead of the head_dim-32 fragment images: do_post and / or do_qkv as above, attention output as fp32 rows.
int paths_token_layer_ws_rows(const float* x_in, const float* attn, float* x_out, const void* w_post, const void* w_qkv,
                              const float* bo, const float* ln1g, const float* ln1b, const float* cab, const float* ln2g, const float* ln2b,
                              const float* b1, const float* b2, const float* ln3g, const float* ln3b, const float* bqkv,
                              float s_wo, float s_w1, float s_w2, float s_wqkv, float* qkv_rows, int64_t ld_qkv, const int64_t* num_ims,
                              int B, int T, int d, int do_post, int do_qkv, int skip_padding, float eps, hipStream_t stream) {
  PATHS_REQUIRE(d == 192, "token_layer_ws_rows: this build instantiates trans_dim 192 (got %d)", d);
  PATHS_REQUIRE(B > 0 && T > 0 && (do_post || do_qkv), "token_layer_ws_rows: nothing to do");
  PATHS_REQUIRE(!skip_padding || num_ims, "token_layer_ws_rows: skip_padding needs num_ims");
  PATHS_REQUIRE(x_in && (!do_post || (attn && x_out && w_post && bo && ln1g && ln1b && cab && ln2g && ln2b && b1 && b2 && ln3g && ln3b)),
                "token_layer_ws_rows: null operand (post)");
  PATHS_REQUIRE(!do_qkv || (w_qkv && bqkv && qkv_rows && ld_qkv >= 3 * d && ld_qkv % 4 == 0), "token_layer_ws_rows: null operand / bad ld (in_proj)");
  PATHS_REQUIRE(((uintptr_t)x_in | (uintptr_t)attn | (uintptr_t)x_out | (uintptr_t)w_post | (uintptr_t)w_qkv | (uintptr_t)qkv_rows) % 16 == 0,
                "token_layer_ws_rows: buffers must be 16-byte aligned");
  WsParams p{x_in, attn, nullptr, x_out, reinterpret_cast<const char*>(w_post), reinterpret_cast<const char*>(w_qkv),
             bo, ln1g, ln1b, cab, ln2g, ln2b, b1, b2, ln3g, ln3b, bqkv,
             do_post ? 1.0f / s_wo : 1.0f, do_post ? 1.0f / s_w1 : 1.0f, do_post ? 1.0f / s_w2 : 1.0f, do_qkv ? 1.0f / s_wqkv : 1.0f,
             nullptr, qkv_rows, ld_qkv, num_ims, T, (T + 63) / 64 * 64, B, skip_padding, 1.0f, eps, nullptr, 0
#ifdef PATHS_WS_STAMPS
             , g_ws_stamps
#endif
  };
  if (do_post && do_qkv) return launch_ws<192, true, true, true>(p, stream);
  if (do_post) return launch_ws<192, true, false>(p, stream);
  return launch_ws<192, false, true, true>(p, stream);
}

}  // extern "C"
