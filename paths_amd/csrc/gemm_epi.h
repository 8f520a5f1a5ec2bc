// Fused epilogues of the PATHS selection chain, shared by the f32-MFMA GEMM (gemm_f32.hip) and the split-bf16 GEMM
// (gemm_x6.hip).  Both kernels hand an epilogue the same thing: per wave WTM x WTN accumulator tiles of 32x32 fp32 in the
// v_mfma 32x32 C layout.
#pragma once
#include "common.h"

namespace paths_epi {

// ------------------------------------------------------------------------------------------------
// Epilogues.  acc[i][j][r] is C[row0 + 32 i + c32_row(r, lane)][col0 + 32 j + (lane & 31)].
// ------------------------------------------------------------------------------------------------

// Epilogue structure (all of them): per 32x32 tile FIRST issue every load the tile needs (clamped row index: always
// a legal address, no branch), THEN compute, THEN store under a row predicate.  Written naively (load -> math ->
// store per element inside `if (row < M)`) hipcc emits one exec-masked branch + one s_waitcnt vmcnt(0) per element:
// 64-128 dependent L2 round trips per thread, which made the K=256 GEMM spend 2/3 of its time in its epilogue.

// Addressing.  An epilogue touches each of its arrays once per accumulator element; done naively that is a 64-bit
// multiply-add, a bounds compare and an exec-masked branch PER ELEMENT (45 instructions per sigmoid output, measured), and
// with one wave per SIMD (gemm_x6) nothing hides it.  So: (1) the row-bounds test is made once per workgroup
// (FULL = the whole BM-row block is inside M: every block but the last), (2) element addresses are a wave-uniform 64-bit
// base (tile origin + the element's row offset, scalar arithmetic) plus ONE per-lane 32-bit offset computed once.
struct TileView {
  char* base; int64_t ldb; uint32_t lane_off;
  __device__ TileView(const float* p, int64_t ld, int lane)
      : base(reinterpret_cast<char*>(const_cast<float*>(p))), ldb(ld * 4), lane_off((uint32_t)((4 * (lane >> 5)) * ld + (lane & 31)) * 4u) {}
  // element r of the 32x32 tile whose top-left corner is (trow, tcol); trow / tcol wave-uniform
  __device__ float* at(int trow, int tcol, int r) const {
    char* tile = base + (int64_t)trow * ldb + (int64_t)tcol * 4;                              // uniform: one per (array, tile)
    const uint32_t off = lane_off + (uint32_t)((r & 3) + 8 * (r >> 2)) * (uint32_t)ldb;        // per lane, 32-bit
    return reinterpret_cast<float*>(tile + off);
  }
  // the two halves of at(), for epilogues that hoist the tile origin out of their element loop themselves
  __device__ char* tile(int trow, int tcol) const { return base + (int64_t)trow * ldb + (int64_t)tcol * 4; }
  __device__ float* elem(char* tile_origin, int r) const {
    return reinterpret_cast<float*>(tile_origin + (lane_off + (uint32_t)((r & 3) + 8 * (r >> 2)) * (uint32_t)ldb));
  }
  // the same element with the row clamped to M-1 (partial blocks: always a legal address)
  __device__ float* at_clamped(int trow, int tcol, int r, int lane, int M) const {
    const int row = min(trow + c32_row(r, lane), M - 1);
    return reinterpret_cast<float*>(base + (int64_t)row * ldb + (int64_t)(tcol + (lane & 31)) * 4);
  }
};
template <bool FULL>
__device__ __forceinline__ float* tile_elem(const TileView& v, int trow, int tcol, int r, int lane, int M) {
  if constexpr (FULL) return v.at(trow, tcol, r);
  else return v.at_clamped(trow, tcol, r, lane, M);
}
template <bool FULL>
__device__ __forceinline__ float* tile_elem_o(const TileView& v, char* origin, int trow, int tcol, int r, int lane, int M) {
  if constexpr (FULL) return v.elem(origin, r);
  else return v.at_clamped(trow, tcol, r, lane, M);
}
template <bool FULL>
__device__ __forceinline__ bool tile_row_ok(int trow, int r, int lane, int M) {
  if constexpr (FULL) return true;
  else return trow + c32_row(r, lane) < M;
}

// c1 = c0 * sigmoid(f) + sigmoid(r) * tanh(m); packed columns per wave = [f(32) | r(32) | m(32)] of one j-block.
struct EpiLstmC {
  const float* bias;     // packed like the weight rows
  const float* c0; int64_t ldc0;   // nullptr at depth 0 (c0 = 0)
  float* c1; int64_t ldc1;         // state_out + D
  float* frm; int64_t ldfrm;       // optional (training): post-activation f|r|m in the packed column order
  const float* hp; int64_t ldhp; const int* hp_row;   // optional: once-per-parent partial pre-activations h_parent Wh^T
  float acc_scale = 1.0f;          // accumulators hold (true value) / acc_scale (fp16-split GEMM: operands are pre-scaled by powers of two)
  template <int WTM, int WTN>
  __device__ __forceinline__ void init(f32x16 (&acc)[WTM][WTN], int row0, int col0, int lane, int M) const {
    const int jj = lane & 31;
    const float seed = 1.0f / acc_scale;
    if (hp == nullptr) {
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; acc[i][2][r] = 0.f; }
      return;
    }
    // siblings share the parent's h: its half of the gate GEMM was done once per parent.  All parent indices first (one
    // round trip for the lot), then the gathers; rows without a parent (hp_row < 0) read row 0 and select zero.
    int pr[WTM][16];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[i][r] = hp_row[min(row0 + 32 * i + c32_row(r, lane), M - 1)];
    float t[2][3][16];                  // two batches in flight, like EpiLstmO::init
    auto issue = [&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* ph = hp + (int64_t)max(pr[i][r], 0) * ldhp + col0 + jj;
        t[i & 1][0][r] = ph[0]; t[i & 1][1][r] = ph[32]; t[i & 1][2][r] = ph[64];
      }
    };
    issue(std::integral_constant<int, 0>{});
    static_for<0, WTM>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (i + 1 < WTM) issue(std::integral_constant<int, i + 1>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool has = pr[i][r] >= 0;
        acc[i][0][r] = has ? t[i & 1][0][r] * seed : 0.f; acc[i][1][r] = has ? t[i & 1][1][r] * seed : 0.f; acc[i][2][r] = has ? t[i & 1][2][r] * seed : 0.f;
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  }
  template <bool FULL, bool WITH_FRM, int WTM, int WTN>
  __device__ __forceinline__ void run_impl(f32x16 (&acc)[WTM][WTN], int row0, int col0, int lane, int M) const {
    const int jj = lane & 31;
    const int tcol = (col0 / 96) * 32;             // memory-unit block of this wave's f|r|m triple
    const float bf = bias[col0 + jj], br = bias[col0 + 32 + jj], bm = bias[col0 + 64 + jj];
    const TileView c0v(c0, ldc0, lane), c1v(c1, ldc1, lane), fv(frm, ldfrm, lane);
    static_for<0, WTM>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const int trow = row0 + 32 * i;
      char* o0 = c0v.tile(trow, tcol); char* o1 = c1v.tile(trow, tcol); char* of = fv.tile(trow, col0);
      float cp[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) cp[r] = c0 ? *tile_elem_o<FULL>(c0v, o0, trow, tcol, r, lane, M) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float f = sigmoid_acc(fmaf(acc[i][0][r], acc_scale, bf));
        const float rg = sigmoid_acc(fmaf(acc[i][1][r], acc_scale, br));
        const float mp = tanh_acc(fmaf(acc[i][2][r], acc_scale, bm));
        const float v = cp[r] * f + rg * mp;
        if (tile_row_ok<FULL>(trow, r, lane, M)) {
          *tile_elem_o<FULL>(c1v, o1, trow, tcol, r, lane, M) = v;
          if constexpr (WITH_FRM) {
            float* fr = tile_elem_o<FULL>(fv, of, trow, col0, r, lane, M);
            fr[0] = f; fr[32] = rg; fr[64] = mp;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep one row-tile's loads in flight at a time (register pressure at WTM = 4)
    });
  }
  template <int WTM, int WTN, int WGM, int WGN>
  __device__ __forceinline__ void run(f32x16 (&acc)[WTM][WTN], int row0, int col0, int lane, int, int, int M, float*) const {
    static_assert(WTN == 3, "LSTM c epilogue wants f|r|m tiles");
    const bool full = row0 + 32 * WTM <= M;
    if (frm == nullptr) {
      if (full) run_impl<true, false>(acc, row0, col0, lane, M);
      else run_impl<false, false>(acc, row0, col0, lane, M);
    } else {
      if (full) run_impl<true, true>(acc, row0, col0, lane, M);
      else run_impl<false, true>(acc, row0, col0, lane, M);
    }
  }
};

// o = sigmoid(acc + b).
// RAW (inference): the gate is only ever read back by the mem_to_out epilogue (EpiLstmH) on the same 32x32 tiles and lanes, so it
// is stored (a) as the PRE-activation a = acc + b (the sigmoid moves into that HBM-bound epilogue, whose VALU has slack; with one
// wave per SIMD here nothing hid its ~18 instructions per element) and (b) in the accumulator layout itself,
// [tile row][tile col][4][64 lanes][4] (one 16-byte store per lane and quarter tile instead of sixteen 4-byte stores).  `o` is then
// a scratch of ceil(M / 256) * 256 x N floats, `ldo` = N.
template <bool RAW>
struct EpiLstmO_ {
  const float* bias; float* o; int64_t ldo; int N;   // N % 32 == 0
  const float* hp; int64_t ldhp; const int* hp_row; int hp_col0;   // optional parent partials (columns hp_col0 + col)
  float acc_scale = 1.0f;          // see EpiLstmC
  template <int WTM, int WTN>
  __device__ __forceinline__ void init(f32x16 (&acc)[WTM][WTN], int row0, int col0, int lane, int M) const {
    const float seed = 1.0f / acc_scale;
    if (hp == nullptr) {
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      return;
    }
    int pr[WTM][16];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[i][r] = hp_row[min(row0 + 32 * i + c32_row(r, lane), M - 1)];
    // one row-tile (16 x WTN gathers per lane) per batch, two batches in flight: batch i+1 is issued BEFORE batch i is consumed,
    // so the L2 round trips of consecutive batches overlap (left to itself hipcc emitted load -> s_waitcnt vmcnt(0) ->
    // v_accvgpr_write per element: 256 serial L2 round trips; one batch at a time still paid four full round trips)
    float t[2][WTN][16];
    auto issue = [&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* ph = hp + (int64_t)max(pr[i][r], 0) * ldhp + hp_col0 + (lane & 31);
#pragma unroll
        for (int j = 0; j < WTN; ++j) t[i & 1][j][r] = ph[min(col0 + 32 * j, N - 32)];
      }
    };
    issue(std::integral_constant<int, 0>{});
    static_for<0, WTM>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (i + 1 < WTM) issue(std::integral_constant<int, i + 1>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j][r] = pr[i][r] >= 0 ? t[i & 1][j][r] * seed : 0.f;
    });
    __builtin_amdgcn_sched_barrier(0);
  }
  template <bool FULL, int WTM, int WTN>
  __device__ __forceinline__ void run_impl(f32x16 (&acc)[WTM][WTN], int row0, int col0, int lane, int M) const {
    const TileView ov(o, ldo, lane);
    static_for<0, WTM * WTN>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value, i = t % WTM, j = t / WTM;
      const int trow = row0 + 32 * i, tcol = col0 + 32 * j;
      if (tcol < N) {                              // wave-uniform (zero-padded weight rows beyond N)
        const float b = bias[tcol + (lane & 31)];
        char* origin = ov.tile(trow, tcol);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = sigmoid_acc(fmaf(acc[i][j][r], acc_scale, b));
          if (tile_row_ok<FULL>(trow, r, lane, M)) *tile_elem_o<FULL>(ov, origin, trow, tcol, r, lane, M) = v;
        }
      }
    });
  }
  template <int WTM, int WTN, int WGM, int WGN>
  __device__ __forceinline__ void run(f32x16 (&acc)[WTM][WTN], int row0, int col0, int lane, int, int, int M, float*) const {
    if constexpr (RAW) {
      const int ntn = N >> 5;
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        const int tcol = col0 + 32 * j;
        if (tcol < N) {
          const float b = bias[tcol + (lane & 31)];
#pragma unroll
          for (int i = 0; i < WTM; ++i) {
            f32x4* t = reinterpret_cast<f32x4*>(o + ((int64_t)((row0 >> 5) + i) * ntn + (tcol >> 5)) * 1024) + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              t[64 * q] = f32x4{fmaf(acc[i][j][4 * q], acc_scale, b), fmaf(acc[i][j][4 * q + 1], acc_scale, b),
                                fmaf(acc[i][j][4 * q + 2], acc_scale, b), fmaf(acc[i][j][4 * q + 3], acc_scale, b)};
          }
        }
      }
    } else {
      if (row0 + 32 * WTM <= M) run_impl<true>(acc, row0, col0, lane, M);
      else run_impl<false>(acc, row0, col0, lane, M);
    }
  }
};
typedef EpiLstmO_<false> EpiLstmO;
typedef EpiLstmO_<true> EpiLstmORaw;

// h1 = o * tanh(acc + bc) ; Y = X + h1 (WITH_Y; otherwise Y is never materialised: see paths_importance_proj_x6's y_add)
// RAW_O: `o` holds the gate's PRE-activations in the accumulator layout (EpiLstmORaw): four 16-byte loads per tile, sigmoid here.
template <bool WITH_Y, bool WITH_TC = false, bool RAW_O = false>
struct EpiLstmH {
  template <int WTM, int WTN>
  __device__ __forceinline__ void init(f32x16 (&acc)[WTM][WTN], int, int, int, int) const {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  const float* bias; const float* o; int64_t ldo; const float* x; int64_t ldx;
  float* h1; int64_t ldh; float* y; int64_t ldy; int N;      // N % 32 == 0; y optional (Y = X + h1 not materialised)
  float* tc_out;                   // optional (training): tanh(Wc c1 + bc), [M, N]
  float acc_scale = 1.0f;          // see EpiLstmC
  template <bool FULL, int WTM, int WTN>
  __device__ __forceinline__ void run_impl(f32x16 (&acc)[WTM][WTN], int row0, int col0, int lane, int M) const {
    // tile-pipelined: the loads of tiles t+1 .. t+PD are in flight while tile t is computed and stored (PD+1 register
    // buffers).  With one wave per SIMD nothing else hides the ~2 us a load takes under load: at PD = 1 this epilogue
    // moved 2.4 TB/s, i.e. it was bound by (bytes in flight) / latency, not by HBM.
    constexpr int NT = WTM * WTN, PD = (NT >= 4 && !WITH_Y) ? 3 : 1, NB = PD + 1;
    const TileView ovw(o, ldo, lane), xvw(x, ldx, lane), hv(h1, ldh, lane), yv(y, ldy, lane), tv(tc_out, N, lane);
    float ov[NB][16], xv[WITH_Y ? NB : 1][16];
    auto load = [&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value, i = t % WTM, j = t / WTM, s = t % NB;
      const int trow = row0 + 32 * i, tcol = min(col0 + 32 * j, N - 32);
      char* to = ovw.tile(trow, tcol); char* tx = xvw.tile(trow, tcol);
      if constexpr (RAW_O) {
        const f32x4* t4 = reinterpret_cast<const f32x4*>(o + ((int64_t)(trow >> 5) * (N >> 5) + (tcol >> 5)) * 1024) + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = t4[64 * q];
          ov[s][4 * q] = v[0]; ov[s][4 * q + 1] = v[1]; ov[s][4 * q + 2] = v[2]; ov[s][4 * q + 3] = v[3];
        }
      }
      static_for<0, 16>([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        if constexpr (FULL) {
          if constexpr (!RAW_O) ov[s][r] = *ovw.elem(to, r);
          if constexpr (WITH_Y) xv[s][r] = *xvw.elem(tx, r);
        } else {
          if constexpr (!RAW_O) ov[s][r] = *ovw.at_clamped(trow, tcol, r, lane, M);
          if constexpr (WITH_Y) xv[s][r] = *xvw.at_clamped(trow, tcol, r, lane, M);
        }
      });
    };
    static_for<0, (PD < NT ? PD : NT)>([&](auto tc) __attribute__((always_inline)) { load(tc); });
    static_for<0, NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value, i = t % WTM, j = t / WTM, s = t % NB;
      if constexpr (t + PD < NT) load(std::integral_constant<int, t + PD>{});
      const int trow = row0 + 32 * i, tcol = col0 + 32 * j;
      if (tcol < N) {
        const float b = bias[tcol + (lane & 31)];
        char* th = hv.tile(trow, tcol); char* ty = yv.tile(trow, tcol); char* tt = tv.tile(trow, tcol);
        static_for<0, 16>([&](auto rc) __attribute__((always_inline)) {
          constexpr int r = decltype(rc)::value;
          const float tcv = tanh_acc(fmaf(acc[i][j][r], acc_scale, b));
          const float h = (RAW_O ? sigmoid_acc(ov[s][r]) : ov[s][r]) * tcv;
          if constexpr (FULL) {
            if constexpr (WITH_TC) *tv.elem(tt, r) = tcv;
            *hv.elem(th, r) = h;
            if constexpr (WITH_Y) *yv.elem(ty, r) = xv[s][r] + h;
          } else if (tile_row_ok<false>(trow, r, lane, M)) {
            if constexpr (WITH_TC) *tv.at_clamped(trow, tcol, r, lane, M) = tcv;
            *hv.at_clamped(trow, tcol, r, lane, M) = h;
            if constexpr (WITH_Y) *yv.at_clamped(trow, tcol, r, lane, M) = xv[s][r] + h;
          }
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  template <int WTM, int WTN, int WGM, int WGN>
  __device__ __forceinline__ void run(f32x16 (&acc)[WTM][WTN], int row0, int col0, int lane, int, int, int M, float*) const {
    if (row0 + 32 * WTM <= M) run_impl<true>(acc, row0, col0, lane, M);
    else run_impl<false>(acc, row0, col0, lane, M);
  }
};

// Generic linear epilogue (forward of the non-LSTM variant, every dX = dY W of the backward pass):
//   v = acc + bias ; act 1: relu ; mask: v = mask > 0 ? v : 0 (relu backward) ; v += residual ; accumulate: v += out
struct EpiBias {
  template <int WTM, int WTN>
  __device__ __forceinline__ void init(f32x16 (&acc)[WTM][WTN], int, int, int, int) const {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  const float* bias; float* out; int64_t ldo; int N; int act;
  const float* residual; int64_t ldr; const float* mask; int64_t ldm; int accumulate;
  float acc_scale = 1.0f;          // see EpiLstmC
  template <bool FULL, int WTM, int WTN>
  __device__ __forceinline__ void run_impl(f32x16 (&acc)[WTM][WTN], int row0, int col0, int lane, int M) const {
    constexpr int NT = WTM * WTN;
    const TileView outv(out, ldo, lane), resv(residual, ldr, lane), maskv(mask, ldm, lane);
    float rv[2][16], mv[2][16], ov[2][16];
    const bool any_load = residual != nullptr || mask != nullptr || accumulate != 0;
    auto load = [&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value, i = t % WTM, j = t / WTM, s = t & 1;
      const int trow = row0 + 32 * i, tcol = col0 + 32 * j;
      const int lcol = min(tcol + (lane & 31), N - 1) - (lane & 31);     // keeps partial last column tiles in bounds
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        rv[s][r] = residual ? *tile_elem<FULL>(resv, trow, lcol, r, lane, M) : 0.f;
        mv[s][r] = mask ? *tile_elem<FULL>(maskv, trow, lcol, r, lane, M) : 1.f;
        ov[s][r] = accumulate ? *tile_elem<FULL>(outv, trow, lcol, r, lane, M) : 0.f;
      }
    };
    if (any_load) load(std::integral_constant<int, 0>{});
    static_for<0, NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value, i = t % WTM, j = t / WTM, s = t & 1;
      if constexpr (t + 1 < NT) {
        if (any_load) load(std::integral_constant<int, t + 1>{});
      }
      const int trow = row0 + 32 * i, tcol = col0 + 32 * j;
      const int colr = tcol + (lane & 31);
      const float b = bias ? bias[min(colr, N - 1)] : 0.f;
      if (tcol < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = fmaf(acc[i][j][r], acc_scale, b);
          if (act == 1) v = fmaxf(v, 0.f);
          if (any_load) {
            if (!(mv[s][r] > 0.f)) v = 0.f;
            v += rv[s][r] + ov[s][r];
          }
          if (tile_row_ok<FULL>(trow, r, lane, M) && (tcol + 32 <= N || colr < N)) *tile_elem<FULL>(outv, trow, tcol, r, lane, M) = v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  template <int WTM, int WTN, int WGM, int WGN>
  __device__ __forceinline__ void run(f32x16 (&acc)[WTM][WTN], int row0, int col0, int lane, int, int, int M, float*) const {
    if (row0 + 32 * WTM <= M) run_impl<true>(acc, row0, col0, lane, M);
    else run_impl<false>(acc, row0, col0, lane, M);
  }
};

// Exact x / d for 0 <= x < 2^24, d > 0 (rd = 1.0f / d): a float estimate that is off by at most one, then one fix-up.
// (A 64-bit integer division is ~150 instructions on this ISA and the token epilogue needed two per row.)
__device__ __forceinline__ int div_u24(int x, int d, float rd) {
  int q = (int)((float)x * rd);
  const int r = x - q * d;
  q += (r >= d) - (r < 0);
  return q;
}
__device__ __forceinline__ int div_pos(int64_t x, int d, float rd) {      // pixel coordinate / patch_size
  return (x >= 0 && x < (1 << 24)) ? div_u24((int)x, d, rd) : (int)(x / d);
}

// Packed weight rows (ops.pack_level "w_ip_fwd") = [W1[0:64] ; Wp[0:64] ; W1[64:128] ; Wp[64:128]]: the block covers all 256
// columns and BOTH column halves (wn = 0, 1) own 64 importance hidden units and 64 projected token channels, so the two
// epilogue phases (importance logit, tokens) are shared by all waves instead of running one after the other on half of them.
//   alpha = valid ? sigmoid(w2 . relu(acc + b1) + b2) : 0            (importance_mode "mul": token = alpha*acc + bp + PE)
template <bool PE_TAB, bool SAVE = false>    // SAVE: training keeps relu(Y W1^T + b1) and Y Wp^T
struct EpiImpProj {
  template <int WTM, int WTN>
  __device__ __forceinline__ void init(f32x16 (&acc)[WTM][WTN], int, int, int, int) const {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  const float* b1; const float* w2; const float* b2;   // b2: device pointer to the scalar bias of the importance MLP's last layer
  const float* bp;                 // proj_in bias [d]
  const float* special;            // special token [d]
  const float* div_term;           // 2d: [d/4] ; 1d: [d/2]   (host: torch.exp(arange * -ln(1e4)/d), utils.py:18,56)
  const int64_t* locs;             // [M,2] pixel coords (2d mode)
  const int64_t* num_ims;          // [B]
  int rows_per_slide;              // N
  int patch_size;
  int pe_mode;                     // 2 = "2d", 1 = "1d"
  int imp_mul;                     // importance_mode == "mul"
  float* importance;               // [M]
  float* tokens;                   // [B, N+1, d]
  float* hid_out;                  // optional (training): relu(Y W1^T + b1) [M,128]
  float* pproj_out;                // optional (training): Y Wp^T (before alpha / bias / PE) [M,128]
  const float* pe_table; int pe_rows;   // PE_TAB: paths_pe_table output (same sinf/cosf values, read instead of recomputed)
  float acc_scale = 1.0f;               // see EpiLstmC
  template <int WTM, int WTN, int WGM, int WGN>
  __device__ __forceinline__ void run(f32x16 (&acc)[WTM][WTN], int row0, int col0, int lane, int wm, int wn, int M, float* smem) const {
    static_assert(WTN == 4 && WGN == 2, "imp/proj epilogue layout: one block spans the 128 hidden + 128 projected columns");
    constexpr int d = 128, RB = WGM * WTM * 32;    // rows per block
    const float rps_inv = 1.0f / (float)rows_per_slide, ps_inv = 1.0f / (float)patch_size;
    const bool small_rows = M < (1 << 24);
    const int u0 = 64 * wn;                        // this wave's hidden units u0 .. u0+63 (tiles 0,1) and channels u0 .. u0+63 (tiles 2,3)
    float* alpha_p = smem;                         // [2][RB] partial w2 . relu(..) sums of the two column halves (LDS is free here)
    // ---- phase A (all waves): partial importance logits over this wave's 64 hidden units
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
      float part[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int u = u0 + 32 * j + (lane & 31);
        const float b = b1[u], w = w2[u];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float hv = fmaxf(fmaf(acc[i][j][r], acc_scale, b), 0.f);
          part[r] += hv * w;
          const int row = row0 + 32 * i + c32_row(r, lane);
          if constexpr (SAVE) { if (row < M) hid_out[(int64_t)row * 128 + u] = hv; }
        }
      }
      // Sum over the 32 lanes (= hidden units) of each half-wave, 16 rows at once.  Halving butterfly: at distance 16 a lane
      // keeps 8 rows and hands the other 8 to its partner, then 4, 2, 1: 8+4+2+1+1 = 16 cross-lane moves instead of 16 x 5.
      // Afterwards lane l holds the total of row index rho(l) = bits 4..1 of l (both lanes of a pair hold the same value).
      float p8[8], p4[4], p2[2], p1;
      {
        const bool up = (lane & 16) != 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float keep = up ? part[8 + k] : part[k], send = up ? part[k] : part[8 + k];
          p8[k] = keep + __shfl_xor(send, 16);
        }
      }
      {
        const bool up = (lane & 8) != 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float keep = up ? p8[4 + k] : p8[k], send = up ? p8[k] : p8[4 + k];
          p4[k] = keep + __shfl_xor(send, 8);
        }
      }
      {
        const bool up = (lane & 4) != 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float keep = up ? p4[2 + k] : p4[k], send = up ? p4[k] : p4[2 + k];
          p2[k] = keep + __shfl_xor(send, 4);
        }
      }
      {
        const bool up = (lane & 2) != 0;
        const float keep = up ? p2[1] : p2[0], send = up ? p2[0] : p2[1];
        p1 = keep + __shfl_xor(send, 2);
      }
      p1 += __shfl_xor(p1, 1);
      const int rho = (lane >> 1) & 15;               // bit4 -> r bit 3, bit3 -> r bit 2, bit2 -> r bit 1, bit1 -> r bit 0
      if ((lane & 1) == 0) alpha_p[wn * RB + wm * WTM * 32 + 32 * i + c32_row(rho, lane)] = p1;
    }
    __syncthreads();
    // ---- phase B (all waves): alpha of every row (both column halves compute the same value), tokens of this wave's 64 channels
    const float b2v = *b2;
    float bpv[2], dtv[2], spv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = u0 + 32 * j + (lane & 31);
      bpv[j] = bp[c];
      spv[j] = special[c];
      dtv[j] = pe_mode == 2 ? div_term[(c & (d / 2 - 1)) >> 1] : div_term[c >> 1];
    }
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
      int64_t lp[16];                              // 2-D mode: this wave's coordinate (x for channels < d/2, y for the others)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowc = min(row0 + 32 * i + c32_row(r, lane), M - 1);
        lp[r] = pe_mode == 2 ? locs[2 * (int64_t)rowc + wn] : 0;
      }
      // slide / row-in-slide of this lane's rows: ONE division for the tile's first row, the others follow by counting (a
      // 32-row tile crosses at most one slide boundary when rows_per_slide >= 32; smaller slides take the division per row)
      const int trow0 = min(row0 + 32 * i, M - 1);
      const int b0 = small_rows ? div_u24(trow0, rows_per_slide, rps_inv) : trow0 / rows_per_slide, idx0 = trow0 - b0 * rows_per_slide;
      const int nslides = M / rows_per_slide;
      const int nim0 = (int)num_ims[b0], nim1 = (int)num_ims[min(b0 + 1, nslides - 1)];
      int tokrow[16]; float av[16]; bool first[16]; float pev[16][2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lrow = wm * WTM * 32 + 32 * i + c32_row(r, lane);
        const int rowr = row0 + 32 * i + c32_row(r, lane);
        int b, idx, nim;
        if (rows_per_slide >= 32) {
          const int off = min(rowr, M - 1) - trow0;
          const bool next = idx0 + off >= rows_per_slide;
          b = b0 + (next ? 1 : 0); idx = idx0 + off - (next ? rows_per_slide : 0); nim = next ? nim1 : nim0;
        } else {
          const int row = min(rowr, M - 1);
          b = small_rows ? div_u24(row, rows_per_slide, rps_inv) : row / rows_per_slide; idx = row - b * rows_per_slide; nim = (int)num_ims[b];
        }
        float a = 0.f;
        if (idx < nim) a = sigmoid_acc((alpha_p[lrow] + alpha_p[RB + lrow]) + b2v);
        if (wn == 0 && (lane & 31) == 0 && rowr < M) importance[rowr] = a;
        av[r] = imp_mul ? a : 1.f;
        tokrow[r] = b * (rows_per_slide + 1) + idx + 1;                   // token row (row 0 of a slide = special token)
        first[r] = idx == 0;                                              // this row also emits its slide's special token
        if constexpr (PE_TAB) {
          // the caller guarantees positions < pe_rows (paths_amd passes the level's grid size); clamped for memory safety,
          // which also makes the division branch-free (no 64-bit fallback)
          const int px24 = (int)min(max(lp[r], (int64_t)0), (int64_t)((1 << 24) - 1));
          const int ipos = pe_mode == 2 ? div_u24(px24, patch_size, ps_inv) : idx;
          const int tp = min(ipos, pe_rows - 1);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c = u0 + 32 * j + (lane & 31);
            pev[r][j] = pe_mode == 2 ? pe_table[(int64_t)tp * (d / 2) + (c & (d / 2 - 1))] : pe_table[(int64_t)tp * d + c];
          }
        } else {
          const int ipos = pe_mode == 2 ? div_pos(lp[r], patch_size, ps_inv) : idx;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c = u0 + 32 * j + (lane & 31);
            const float ang = (float)ipos * dtv[j];
            pev[r][j] = (c & 1) ? cosf(ang) : sinf(ang);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + 32 * i + c32_row(r, lane);
        if (row < M) {
          float* trow = tokens + (int64_t)tokrow[r] * d;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c = u0 + 32 * j + (lane & 31);
            const float pj = acc[i][2 + j][r] * acc_scale;
            trow[c] = av[r] * pj + bpv[j] + pev[r][j];
            if constexpr (SAVE) pproj_out[(int64_t)row * 128 + c] = pj;
            if (first[r]) trow[c - d] = spv[j];
          }
        }
      }
    }
  }
};

}  // namespace paths_epi
