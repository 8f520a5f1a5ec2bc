// fp32-exact tiled GEMM on the CDNA4 f32-input matrix cores (v_mfma_f32_32x32x2_f32) with the fused
// epilogues of the PATHS selection chain.
//
//   C[M, N] = [A0 | A1][M, K0+K1] * Bt[N, K0+K1]^T          (Bt = nn.Linear weight layout, K contiguous)
//
// Replaces (reference file:line):
//   * LSTMCell.forward gates            model/interface.py:49-56   -> EpiLstmC (f,r,m -> c1), EpiLstmO (o gate)
//   * h1 = o * tanh(Wc c1 + bc), Y=X+h1  model/interface.py:56, model/paths.py:89-91 -> EpiLstmH
//   * importance MLP + sigmoid + mask   model/paths.py:95, utils.py:106-115 }
//   * Z = Y*alpha ; proj_in ; 2-D / 1-D positional encoding ; special token   } -> EpiImpProj
//       model/paths.py:96-98,119-124, model/aggregator.py:37-65, utils.py:16-23,47-67
//
// Design (MI355X): 64-lane waves each own WTM x WTN tiles of 32x32 fp32 accumulators; A and Bt k-tiles
// (BK = 32) are staged global -> registers -> LDS (row stride 36 floats: conflict-free ds_read_b128 for the
// 16-lane groups of gfx950), double-buffered with one barrier per k-tile and the next tile's global loads
// in flight under the MFMAs.  The MFMA's two k-slots per step are fed with a PERMUTED k order
// (lane half h takes k = 8q+4h+e) so that every lane fetches 4 steps of operand with ONE 16-byte LDS read;
// A and B use the same permutation, the sum over k is unchanged.  The "concat(x, h)" of the reference is
// never materialised: the k loop walks panel A0 (features) then A1 (previous h state, strided view).
#include "common.h"
#include "gemm_epi.h"

namespace {
using namespace paths_epi;

constexpr int BK = 32;
constexpr int LDK = BK + 4;  // padded LDS row stride (floats)

struct GemmOperands {
  const float* A0; int64_t lda0; int K0;
  const float* A1; int64_t lda1; int K1;
  const float* Bt; int64_t ldb;   // [Npad rows, K0+K1]
  int M;
  const int64_t* num_ims;         // optional padding skip
  int rows_per_slide;
};

template <int WTM, int WTN, int WGM, int WGN, class Epi>
__global__ void __launch_bounds__(64 * WGM * WGN)
gemm_f32_kernel(GemmOperands g, Epi epi) {
  constexpr int BM = WTM * 32 * WGM, BN = WTN * 32 * WGN, NT = 64 * WGM * WGN;
  constexpr int RPP = NT / 8;                 // rows staged per pass (8 threads x float4 = one 32-float row)
  constexpr int PA = BM / RPP, PB = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/threads mismatch");
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][(BM+BN)*LDK]

  // XCD-aware tile order (speed only, any placement is correct): workgroups are dealt round-robin over the 8 XCDs,
  // so give XCD x (= linear id % 8) a contiguous run of the tile sequence; inside that run tiles advance
  // column-block fastest within groups of GM row-blocks, so the ~64 workgroups resident on one XCD cover a
  // GM x (N/BN) patch and re-use each A and W k-tile from that XCD's private L2.
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

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int c4 = tid & 7, r0 = tid >> 3;

  const float* a0p[PA]; const float* a1p[PA]; const float* bp[PB];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    int row = min(m0 + r0 + p * RPP, g.M - 1);
    a0p[p] = g.A0 + (int64_t)row * g.lda0 + 4 * c4;
    a1p[p] = g.A1 ? g.A1 + (int64_t)row * g.lda1 + 4 * c4 : a0p[p];   // never dereferenced when K1 == 0; keeps the select in the global address space
  }
#pragma unroll
  for (int p = 0; p < PB; ++p) bp[p] = g.Bt + (int64_t)(n0 + r0 + p * RPP) * g.ldb + 4 * c4;

  f32x4 ra[PA], rb[PB];
  // One staged row-chunk at a time, addressed by a compile-time index (everything below is fully unrolled).
  // Panel select is pointer arithmetic, not a branch: the k-tile body must stay ONE basic block.
  auto gload_one = [&](int idx, int kt) {
    const int k = kt * BK;
    if (idx < PA) {
      const bool first = k < g.K0;
      ra[idx] = ldg_f32x4((first ? a0p[idx] : a1p[idx]) + (first ? k : k - g.K0));
    } else {
      rb[idx - PA] = ldg_f32x4(bp[idx - PA] + k);
    }
  };
  auto swrite_one = [&](int idx, int buf) {
    float* s = smem + buf * (BM + BN) * LDK;
    if (idx < PA) *reinterpret_cast<f32x4*>(s + (r0 + idx * RPP) * LDK + 4 * c4) = ra[idx];
    else *reinterpret_cast<f32x4*>(s + (BM + r0 + (idx - PA) * RPP) * LDK + 4 * c4) = rb[idx - PA];
  };

  // accumulators start from the epilogue's initial value (zero, or the once-per-parent partial pre-activation): issued
  // here, the loads overlap the first tile's staging instead of sitting in the epilogue
  f32x16 acc[WTM][WTN];
  epi.template init<WTM, WTN>(acc, m0 + wm * WTM * 32, n0 + wn * WTN * 32, lane, g.M);

  const int nk = (g.K0 + g.K1) / BK;
  const int fragoff = (lane & 31) * LDK + 4 * (lane >> 5);
  const float* sAbase = smem + (wm * WTM * 32) * LDK + fragoff;
  const float* sBbase = smem + (BM + wn * WTN * 32) * LDK + fragoff;
  constexpr int NF = WTM + WTN;           // fragments (16-byte LDS reads) per k-group
  constexpr int NG = PA + PB;             // staged chunks per thread per k-tile
  static_assert(NG <= 12, "staging schedule assumes <= 12 chunks per thread");
  f32x4 fa[2][WTM], fb[2][WTN];
  auto read_frag = [&](int buf, int q, int slot, int f) {
    if (f < WTM) fa[slot][f] = *reinterpret_cast<const f32x4*>(sAbase + buf * (BM + BN) * LDK + 8 * q + f * 32 * LDK);
    else fb[slot][f - WTM] = *reinterpret_cast<const f32x4*>(sBbase + buf * (BM + BN) * LDK + 8 * q + (f - WTM) * 32 * LDK);
  };
  auto mfma_step = [&](int slot, int e) {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) acc[i][j] = mfma32(fa[slot][i][e], fb[slot][j][e], acc[i][j]);
  };
  // A wave issues IN ORDER and an fp32 MFMA holds the matrix pipe for 64 cycles, so everything that is not an MFMA
  // must sit between MFMAs, not in a cluster at the tile boundary (hipcc's default placement left ~2,100 non-MFMA
  // cycles per 4,096-cycle k-tile).  A k-tile is 16 pinned groups; group gq = (k-group q = gq/4, step e = gq%4):
  //     WTM*WTN MFMAs | fragment reads of k-group q+1 | ONE global load of tile kt+1 (early groups) or ONE LDS
  //     write of it (late groups, >= 6 groups = 1,500+ cycles after its load)
  // Group 15 is rotated past the barrier so that its MFMAs cover the first fragment reads of the next tile.
  auto group = [&](int gq, int buf, int kt, bool stage) {
    const int q = gq >> 2, e = gq & 3, slot = q & 1;
    mfma_step(slot, e);
    if (q < 3) {
#pragma unroll
      for (int f = e; f < NF; f += 4) read_frag(buf, q + 1, slot ^ 1, f);
    }
    if (stage) {
      if (gq < NG) gload_one(gq, kt + 1);
      if (gq >= 16 - NG) swrite_one(gq - (16 - NG), buf ^ 1);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

#pragma unroll
  for (int idx = 0; idx < NG; ++idx) gload_one(idx, 0);
#pragma unroll
  for (int idx = 0; idx < NG; ++idx) swrite_one(idx, 0);
  __syncthreads();
#pragma unroll
  for (int f = 0; f < NF; ++f) read_frag(0, 0, 0, f);
  int buf = 0;
  for (int kt = 0; kt < nk - 1; ++kt) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int gq = 0; gq < 15; ++gq) group(gq, buf, kt, true);
    // group 15 = last MFMA step of this tile; its LDS write goes BEFORE the barrier, its MFMAs after it
    swrite_one(NG - 1, buf ^ 1);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    buf ^= 1;
#pragma unroll
    for (int f = 0; f < NF; ++f) read_frag(buf, 0, 0, f);
    mfma_step(1, 3);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int gq = 0; gq < 16; ++gq) group(gq, buf, 0, false);
  __syncthreads();     // epilogues may reuse LDS
  epi.template run<WTM, WTN, WGM, WGN>(acc, m0 + wm * WTM * 32, n0 + wn * WTN * 32, lane, wm, wn, g.M, smem);
}

template <int WTM, int WTN, int WGM, int WGN, class Epi>
int launch_gemm(const GemmOperands& g, int Npad, const Epi& epi, hipStream_t stream, const char* name) {
  constexpr int BM = WTM * 32 * WGM, BN = WTN * 32 * WGN;
  constexpr size_t lds = 2ull * (BM + BN) * LDK * sizeof(float);
  PATHS_REQUIRE(g.M > 0, "%s: M must be > 0", name);
  PATHS_REQUIRE(g.K0 > 0 && g.K0 % BK == 0 && g.K1 % BK == 0, "%s: K panels (%d,%d) must be multiples of %d", name, g.K0, g.K1, BK);
  PATHS_REQUIRE(Npad % BN == 0, "%s: packed N (%d) must be a multiple of %d", name, Npad, BN);
  PATHS_REQUIRE(g.lda0 % 4 == 0 && g.lda1 % 4 == 0 && g.ldb % 4 == 0, "%s: leading dims must be multiples of 4 floats", name);
  PATHS_REQUIRE(((uintptr_t)g.A0 % 16 == 0) && ((uintptr_t)g.A1 % 16 == 0) && ((uintptr_t)g.Bt % 16 == 0), "%s: operands must be 16-byte aligned", name);
  auto kern = gemm_f32_kernel<WTM, WTN, WGM, WGN, Epi>;
  PATHS_LDS_OPT_IN(kern, lds, name);
  dim3 grid(Npad / BN, (g.M + BM - 1) / BM);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), lds, stream, g, epi);
  PATHS_LAUNCH_CHECK(name);
  return PATHS_OK;
}

// PE table: out[pos][c] = c odd ? cosf(pos * div[c >> 1]) : sinf(pos * div[c >> 1]),  c < W (W = d/2 for "2d", d for "1d"):
// exactly the expressions of EpiImpProj, evaluated once per (position, channel) instead of once per token element.
__global__ void pe_table_kernel(const float* __restrict__ div_term, int W, int rows, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * W) return;
  const int pos = i / W, c = i % W;
  const float ang = (float)pos * div_term[c >> 1];
  out[i] = (c & 1) ? cosf(ang) : sinf(ang);
}

// out[M, N] (+)= maskop(act(a[M, K] w[N, K]^T + bias)) + residual for M <= SMALL_M rows (EpiBias semantics).  The 128 x 128 tile
// kernel spent 23 us on an 8-row product (61 launches per training step: the last layer's token-0 row chain, forward, recompute and
// backward); here one wave owns one output column: 16-byte coalesced reads of its weight row, the few rows of `a` re-read through
// L1, one wave reduction per (row, column).
constexpr int SMALL_M = 16;
__global__ void __launch_bounds__(256)
gemm_nt_small_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ w, int64_t ldw, const float* __restrict__ bias,
                     float* __restrict__ out, int64_t ldo, int M, int N, int K, int act, const float* __restrict__ residual, int64_t ldr,
                     const float* __restrict__ mask, int64_t ldm, int accumulate) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[SMALL_M];
#pragma unroll
  for (int m = 0; m < SMALL_M; ++m) acc[m] = 0.f;
  for (int k0 = 4 * lane; k0 < K; k0 += 256) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (int64_t)n * ldw + k0);
#pragma unroll
    for (int m = 0; m < SMALL_M; ++m) {
      if (m < M) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + (int64_t)m * lda + k0);
        acc[m] = fmaf(av[0], wv[0], fmaf(av[1], wv[1], fmaf(av[2], wv[2], fmaf(av[3], wv[3], acc[m]))));
      }
    }
  }
#pragma unroll
  for (int m = 0; m < SMALL_M; ++m) {
    if (m < M) {
      float v = acc[m];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
      if (lane == 0) {
        v += bias ? bias[n] : 0.f;
        if (act == 1) v = fmaxf(v, 0.f);
        if (mask && !(mask[(int64_t)m * ldm + n] > 0.f)) v = 0.f;
        if (residual) v += residual[(int64_t)m * ldr + n];
        float* o = out + (int64_t)m * ldo + n;
        *o = accumulate ? *o + v : v;
      }
    }
  }
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

// Packed LSTM gate weights: rows [0,768) = c-part in 96-row groups [f(32)|r(32)|m(32)] per j-block,
// rows [768, 768+D) = out_select_gate.  See paths_amd/_pack.py.
int paths_lstm_cell(const float* x, int64_t ldx, const float* h0, int64_t ldh0, const float* c0, int64_t ldc0,
                    const float* w_gates /*[3Hc+D, 2D] packed*/, const float* b_gates /*[3Hc+D] packed*/,
                    const float* w_mem /*[D, Hc]*/, const float* b_mem /*[D]*/,
                    float* state_out /*[M, D+Hc]: h1 | c1*/, int64_t ldso, float* y /*[M,D]*/, int64_t ldy,
                    float* ws_o /*[M,D] workspace*/, float* save_frm /*[M,3Hc] or null*/, float* save_tc /*[M,D] or null*/,
                    const float* hp /*[*, 3Hc+D] or null*/, const int* hp_row /*[M] or null*/,
                    int M, int D, int Hc, const int64_t* num_ims, int rows_per_slide, int phases, hipStream_t stream) {
  PATHS_REQUIRE(D % 128 == 0 && Hc % 64 == 0, "lstm_cell: D (%d) must be a multiple of 128 and Hc (%d) of 64", D, Hc);
  PATHS_REQUIRE(hp != nullptr || (h0 == nullptr) == (c0 == nullptr), "lstm_cell: h0 and c0 must both be given or both be null");
  PATHS_REQUIRE((hp == nullptr) == (hp_row == nullptr) && (hp == nullptr || h0 == nullptr),
                "lstm_cell: hp/hp_row come together and replace h0 (the h half of the gate GEMM was done per parent)");
  const int Ktot = 2 * D;
  GemmOperands g{x, ldx, D, h0, ldh0, h0 ? D : 0, w_gates, Ktot, M, num_ims, rows_per_slide};
  // phases: bit0 = memory-cell GEMM (f,r,m -> c1), bit1 = output-gate GEMM, bit2 = mem_to_out GEMM (+ residual).
  // Callers normally pass 7; the bench brackets single phases with events.
  // (1) c-part: N = 3Hc, wave tile 64x96, block 128x192
  if (phases & 1) {
    EpiLstmC e{b_gates, c0, ldc0, state_out + D, ldso, save_frm, (int64_t)3 * Hc, hp, (int64_t)3 * Hc + D, hp_row};
    int rc = launch_gemm<2, 3, 2, 2>(g, 3 * Hc, e, stream, "lstm_cell(c)");
    if (rc) return rc;
  }
  // (2) o gate: N = D
  if (phases & 2) {
    GemmOperands go = g;
    go.Bt = w_gates + (int64_t)3 * Hc * Ktot;
    EpiLstmO e{b_gates + 3 * Hc, ws_o, D, D, hp, (int64_t)3 * Hc + D, hp_row, 3 * Hc};
    int rc = launch_gemm<2, 2, 2, 2>(go, D, e, stream, "lstm_cell(o)");
    if (rc) return rc;
  }
  // (3) h1 = o * tanh(Wc c1 + bc), Y = X + h1
  if (phases & 4) {
    GemmOperands gh{state_out + D, ldso, Hc, nullptr, 0, 0, w_mem, Hc, M, num_ims, rows_per_slide};
    PATHS_REQUIRE(y != nullptr, "lstm_cell: y is required (only paths_lstm_cell_x6 can skip it)");
    int rc;
    if (save_tc != nullptr) {
      EpiLstmH<true, true> e{b_mem, ws_o, D, x, ldx, state_out, ldso, y, ldy, D, save_tc};
      rc = launch_gemm<2, 2, 2, 2>(gh, D, e, stream, "lstm_cell(h, save)");
    } else {
      EpiLstmH<true, false> e{b_mem, ws_o, D, x, ldx, state_out, ldso, y, ldy, D, nullptr};
      rc = launch_gemm<2, 2, 2, 2>(gh, D, e, stream, "lstm_cell(h)");
    }
    if (rc) return rc;
  }
  return PATHS_OK;
}

int paths_importance_proj(const float* y, int64_t ldy, const float* w_ip /*[256, D]: W1 ; Wp*/,
                          const float* b1, const float* w2, const float* b2, const float* bp, const float* special,
                          const float* div_term, const float* pe_table, int pe_rows, const int64_t* locs, const int64_t* num_ims,
                          int rows_per_slide, int patch_size, int pe_mode, int imp_mul,
                          float* importance, float* tokens, float* save_hid, float* save_pproj, int M, int D, int Hi, int d,
                          int skip_padding, hipStream_t stream) {
  PATHS_REQUIRE(Hi == 128 && d == 128, "importance_proj: this build supports importance_mlp_hidden_dim=128, trans_dim=128 (got %d, %d)", Hi, d);
  PATHS_REQUIRE(pe_mode == 1 || pe_mode == 2, "importance_proj: pe_mode must be 1 (1d) or 2 (2d)");
  PATHS_REQUIRE(pe_mode == 1 || locs != nullptr, "importance_proj: 2d positional encoding needs locs");
  PATHS_REQUIRE(num_ims != nullptr && rows_per_slide > 0 && M % rows_per_slide == 0, "importance_proj: bad slide layout");
  PATHS_REQUIRE(b1 != nullptr && w2 != nullptr && b2 != nullptr, "importance_proj: b1, w2 and b2 (device scalar) are required");
  GemmOperands g{y, ldy, D, nullptr, 0, 0, w_ip, D, M, skip_padding ? num_ims : nullptr, rows_per_slide};
  PATHS_REQUIRE((save_hid == nullptr) == (save_pproj == nullptr), "importance_proj: save_hid and save_pproj come together");
  PATHS_REQUIRE(pe_table == nullptr || pe_rows > 0, "importance_proj: pe_rows must be > 0 with a pe_table");
  auto go = [&](auto epi) {
    decltype(epi) e{b1, w2, b2, bp, special, div_term, locs, num_ims, rows_per_slide, patch_size, pe_mode, imp_mul, importance, tokens,
                    save_hid, save_pproj, pe_table, pe_table ? pe_rows : 0};
    return launch_gemm<1, 4, 2, 2>(g, 256, e, stream, "importance_proj");
  };
  if (pe_table != nullptr) return save_hid ? go(EpiImpProj<true, true>{}) : go(EpiImpProj<true, false>{});
  return save_hid ? go(EpiImpProj<false, true>{}) : go(EpiImpProj<false, false>{});
}

// positional-encoding table for paths_importance_proj(_x6): out [rows, d/2] (pe_mode 2) or [rows, d] (pe_mode 1)
int paths_pe_table(const float* div_term, int pe_mode, int d, int rows, float* out, hipStream_t stream) {
  PATHS_REQUIRE((pe_mode == 1 || pe_mode == 2) && d > 0 && d % 4 == 0 && rows > 0, "pe_table: bad arguments");
  const int W = pe_mode == 2 ? d / 2 : d;
  hipLaunchKernelGGL(pe_table_kernel, dim3((rows * W + 255) / 256), dim3(256), 0, stream, div_term, W, rows, out);
  PATHS_LAUNCH_CHECK("pe_table");
  return PATHS_OK;
}

// out[M,N] = act(A[M,K] * W[N,K]^T + b).  W rows must be padded (zero rows) to a multiple of 128.
int paths_linear_f32(const float* a, int64_t lda, const float* w, const float* b, float* out, int64_t ldo,
                     int M, int N, int Npad, int K, int act, hipStream_t stream) {
  GemmOperands g{a, lda, K, nullptr, 0, 0, w, K, M, nullptr, 0};
  EpiBias e{b, out, ldo, N, act, nullptr, 0, nullptr, 0, 0};
  return launch_gemm<2, 2, 2, 2>(g, Npad, e, stream, "linear_f32");
}

// out[M,N] (+)= maskop(act(A[M,K] * W[N,K]^T + b)) + residual      (W rows zero-padded to Npad, a multiple of 128)
// The backward pass calls it with W = a transposed weight copy: dX = dY * W  ==  dY * (W^T)^T.
int paths_gemm_nt_f32(const float* a, int64_t lda, const float* w, int64_t ldw, const float* b, float* out, int64_t ldo,
                      int M, int N, int Npad, int K, int act, const float* residual, int64_t ldr, const float* mask,
                      int64_t ldm, int accumulate, hipStream_t stream) {
  PATHS_REQUIRE(ldw >= K, "gemm_nt: ldw < K");
  if (M <= SMALL_M && K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && ((uintptr_t)a | (uintptr_t)w) % 16 == 0) {
    // a handful of rows (the last decoder layer's token-0 chain: M = slides per batch): one wave per output column
    hipLaunchKernelGGL(gemm_nt_small_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, a, lda, w, ldw, b, out, ldo, M, N, K, act, residual, ldr,
                       mask, ldm, accumulate);
    PATHS_LAUNCH_CHECK("gemm_nt_f32(small M)");
    return PATHS_OK;
  }
  GemmOperands g{a, lda, K, nullptr, 0, 0, w, ldw, M, nullptr, 0};
  EpiBias e{b, out, ldo, N, act, residual, ldr, mask, ldm, accumulate};
  return launch_gemm<2, 2, 2, 2>(g, Npad, e, stream, "gemm_nt_f32");
}

}  // extern "C"
