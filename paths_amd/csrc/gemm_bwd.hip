// Backward-pass GEMM building blocks (fp32-exact MFMA, gfx950).
//
//   paths_gemm_tn_f32   C[N1,N2] (+)= A[M,N1]^T * [B0 | B1][M,N2]      weight gradients dW = dY^T X (reduction over rows)
//   paths_colsum_f32    out[N]   (+)= sum_m A[m,N]                     bias gradients
//   paths_transpose_f32 out[C,R]   = in[R,C]^T                        W^T copies so that dX = dY W runs on the NT kernel
//
// These replace what autograd does for nn.Linear inside the reference's train step (train.py:65 loss.backward()):
// aten::mm(dY^T, X), aten::sum(dY, 0), aten::mm(dY, W).
//
// TN mapping: both operands are row-major with the REDUCTION index (row m) as the slow dimension, so a k-tile is 32
// consecutive rows staged as-is ([m][n] in LDS, 16-byte coalesced loads along n); the MFMA A/B fragments
// (A[i=n1][k=m], B[k=m][j=n2]) are then 4-byte LDS reads with lanes running along n: conflict-free, four times the
// read instructions of the NT kernel, still far from the LDS limit for 64-cycle fp32 MFMAs.  The reduction over M is
// split across gridDim.z; every split writes its own fp32 slab and a second kernel adds the slabs in a fixed order
// (deterministic: needed for N-rank == 1-rank gradient parity, no float atomics).
#include "common.h"
#include "reduce.h"

namespace {

constexpr int TK = 32;                 // rows of the reduction dimension per k-tile
constexpr int TLD = 128;               // LDS row stride (floats) of a [TK][128] tile

struct TnOperands {
  const float* A; int64_t lda;         // [M, N1]
  const float* B0; int64_t ldb0; int NB0;   // [M, NB0]  columns [0, NB0) of the logical B
  const float* B1; int64_t ldb1;       // [M, N2-NB0] columns [NB0, N2)  (may be null)
  int M, N1, N2;
  float* slabs;                        // [splits][N1][N2]  (or the output itself when there is one split)
  int64_t ld_out;                      // row stride of a slab / of the output
  int rows_per_split;                  // multiple of TK
};

__global__ void __launch_bounds__(256)
gemm_tn_kernel(TnOperands g) {
  __shared__ __attribute__((aligned(16))) float sA[2][TK * TLD];
  __shared__ __attribute__((aligned(16))) float sB[2][TK * TLD];
  const int n1_0 = blockIdx.y * 128, n2_0 = blockIdx.x * 128;
  const int m_begin = blockIdx.z * g.rows_per_split;
  const int m_end = min(g.M, m_begin + g.rows_per_split);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // B column block comes from one panel (NB0 is a multiple of 128)
  const bool second = g.B1 != nullptr && n2_0 >= g.NB0;
  const float* Bp = second ? g.B1 + (n2_0 - g.NB0) : g.B0 + n2_0;
  const int64_t ldb = second ? g.ldb1 : g.ldb0;
  const float* Ap = g.A + n1_0;
  // ragged edge tiles (N1 / N2 not multiples of 128): columns past the end are read from the tile's first column instead (their
  // products only reach accumulator rows / columns the epilogue never stores)
  const int bcols = (second ? g.N2 : g.NB0) - n2_0;

  // staging: 32 rows x 32 float4 per operand -> 4 float4 per thread per operand (8 chunks per k-tile)
  const int c4 = tid & 31, r0 = tid >> 5;          // r0 in [0,8): rows r0 + 8p
  const int ca = n1_0 + 4 * c4 < g.N1 ? 4 * c4 : 0, cb = 4 * c4 < bcols ? 4 * c4 : 0;
  f32x4 ra[4], rb[4];
  // one chunk (compile-time index after unrolling); rows past the end are read clamped and zeroed by a select, not a
  // branch: the k-tile body must stay one basic block so its instruction order can be pinned (see gemm_f32.hip)
  auto gload_one = [&](int idx, int m0) {
    const int p = idx & 3;
    const int m = m0 + r0 + 8 * p;
    const int mc = min(m, m_end - 1);
    if (idx < 4) ra[p] = ldg_f32x4(Ap + (int64_t)mc * g.lda + ca);
    else rb[p] = ldg_f32x4(Bp + (int64_t)mc * ldb + cb);
  };
  // the zeroing of rows past the end happens HERE (8 groups after the load was issued): touching the loaded value
  // earlier would put an s_waitcnt vmcnt(0) right behind every load
  auto swrite_one = [&](int idx, int buf, int m0) {
    const int p = idx & 3;
    const bool ok = m0 + r0 + 8 * p < m_end;
    const f32x4 z{0.f, 0.f, 0.f, 0.f};
    if (idx < 4) *reinterpret_cast<f32x4*>(&sA[buf][(r0 + 8 * p) * TLD + 4 * c4]) = ok ? ra[p] : z;
    else *reinterpret_cast<f32x4*>(&sB[buf][(r0 + 8 * p) * TLD + 4 * c4]) = ok ? rb[p] : z;
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (m_end - m_begin + TK - 1) / TK;
  const int h = lane >> 5, li = lane & 31;
  if (nk > 0) {
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) gload_one(idx, m_begin);
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) swrite_one(idx, 0, m_begin);
  }
  __syncthreads();
  int buf = 0;
  float fa[2][2], fb[2][2];          // fragments of MFMA step s: [slot][tile]
  auto read_step = [&](int bufi, int s_, int slot) {
    const float* a = &sA[bufi][(2 * s_ + h) * TLD + wm * 64 + li];
    const float* b = &sB[bufi][(2 * s_ + h) * TLD + wn * 64 + li];
    fa[slot][0] = a[0]; fa[slot][1] = a[32]; fb[slot][0] = b[0]; fb[slot][1] = b[32];
  };
  auto mfma_step = [&](int slot) {
    acc[0][0] = mfma32(fa[slot][0], fb[slot][0], acc[0][0]);
    acc[0][1] = mfma32(fa[slot][0], fb[slot][1], acc[0][1]);
    acc[1][0] = mfma32(fa[slot][1], fb[slot][0], acc[1][0]);
    acc[1][1] = mfma32(fa[slot][1], fb[slot][1], acc[1][1]);
  };
  if (nk > 0) read_step(0, 0, 0);
  // 16 pinned groups per k-tile: 4 MFMAs | fragment reads of the next step | one global load (groups 0-7) or one LDS
  // write (groups 8-15) of the next tile; the last group's MFMAs sit behind the barrier and cover the next tile's first reads
  for (int kt = 0; kt < nk - 1; ++kt) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s_ = 0; s_ < 15; ++s_) {
      mfma_step(s_ & 1);
      read_step(buf, s_ + 1, (s_ + 1) & 1);
      if (s_ < 8) gload_one(s_, m_begin + (kt + 1) * TK);
      else swrite_one(s_ - 8, buf ^ 1, m_begin + (kt + 1) * TK);
      __builtin_amdgcn_sched_barrier(0);
    }
    swrite_one(7, buf ^ 1, m_begin + (kt + 1) * TK);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    buf ^= 1;
    read_step(buf, 0, 0);
    mfma_step(1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (nk > 0) {
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) {
      mfma_step(s_ & 1);
      if (s_ < 15) read_step(buf, s_ + 1, (s_ + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float* out = g.slabs + (int64_t)blockIdx.z * g.N1 * g.ld_out;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n2_0 + wn * 64 + 32 * j + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n1_0 + wm * 64 + 32 * i + c32_row(r, lane);
        if (row < g.N1 && col < g.N2) out[(int64_t)row * g.ld_out + col] = acc[i][j][r];
      }
    }
}

// out[i] (+)= sum_s slabs[s][i]   (fixed order: deterministic)
__global__ void __launch_bounds__(256)
reduce_slabs_kernel(const float* __restrict__ slabs, int splits, int64_t n, float* __restrict__ out, int64_t ldo, int ncols,
                    int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // fixed summation order (deterministic) with 8 independent partial sums so that 8 loads are in flight per thread:
  // a one-workgroup launch of this kernel with 256 splits was a chain of 256 dependent L2 round trips (23 us)
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

// The same sum for short outputs (bias / LayerNorm-affine gradients: n = 128 .. 1792): with one thread per output a 128-column
// reduction was ONE workgroup walking 256 slabs (6.9 us, ~190 calls per training step).  Here a workgroup owns 64 outputs and
// four thread groups each sum a quarter of the slabs (interleaved, 8 loads in flight), joined through LDS in a fixed order.
__global__ void __launch_bounds__(256)
reduce_slabs_small_kernel(const float* __restrict__ slabs, int splits, int n, float* __restrict__ out, int accumulate) {
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    int k = q;
    for (; k + 28 < splits; k += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] += slabs[(int64_t)(k + 4 * u) * n + i];
    }
    for (int u = 0; k < splits; k += 4, ++u) p[u & 7] += slabs[(int64_t)k * n + i];
  }
  part[q][c] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  __syncthreads();
  if (q == 0 && i < n) {
    const float s = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
    out[i] = accumulate ? out[i] + s : s;
  }
}

// partial column sums: block (x = column chunk of 256, y = row split)
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float* __restrict__ a, int64_t lda, int M, int N, int rows_per_split, float* __restrict__ slabs) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int m0 = blockIdx.y * rows_per_split, m1 = min(M, m0 + rows_per_split);
  if (col >= N) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int m = m0;
  for (; m + 3 < m1; m += 4) {
    s0 += a[(int64_t)m * lda + col]; s1 += a[(int64_t)(m + 1) * lda + col];
    s2 += a[(int64_t)(m + 2) * lda + col]; s3 += a[(int64_t)(m + 3) * lda + col];
  }
  for (; m < m1; ++m) s0 += a[(int64_t)m * lda + col];
  slabs[(int64_t)blockIdx.y * N + col] = (s0 + s1) + (s2 + s3);
}

// partial column sums, 16-byte form (N, lda multiples of 4, base 16-byte aligned): block = CGB column groups of 4 floats
// x (256 / CGB) row lanes; every thread keeps its loads of up to 8 rows in flight, the row lanes are summed through LDS in
// a fixed order (deterministic).  8 MB of gradients per call stream at HBM rate instead of one 4-byte column per thread.
template <int CGB>
__global__ void __launch_bounds__(256)
colsum_partial4_kernel(const float* __restrict__ a, int64_t lda, int M, int N, int rows_per_split, float* __restrict__ slabs) {
  constexpr int RL = 256 / CGB;
  __shared__ f32x4 part[RL][CGB];
  const int cgl = threadIdx.x % CGB, rl = threadIdx.x / CGB;
  const int cg = blockIdx.x * CGB + cgl;
  const int m0 = blockIdx.y * rows_per_split, m1 = min(M, m0 + rows_per_split);
  f32x4 s0{0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (4 * cg < N) {
    const f32x4* base = reinterpret_cast<const f32x4*>(a + 4 * cg);
    const int64_t ld4 = lda / 4;
    int m = m0 + rl;
    for (; m + 3 * RL < m1; m += 4 * RL) {
      const f32x4 v0 = base[(int64_t)m * ld4], v1 = base[(int64_t)(m + RL) * ld4];
      const f32x4 v2 = base[(int64_t)(m + 2 * RL) * ld4], v3 = base[(int64_t)(m + 3 * RL) * ld4];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; m < m1; m += RL) s0 += base[(int64_t)m * ld4];
  }
  part[rl][cgl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && 4 * cg < N) {
    f32x4 t = part[0][cgl];
#pragma unroll
    for (int r = 1; r < RL; ++r) t += part[r][cgl];
    *reinterpret_cast<f32x4*>(slabs + (int64_t)blockIdx.y * N + 4 * cg) = t;
  }
}

__global__ void __launch_bounds__(256)
transpose_kernel(const float* __restrict__ in, int64_t ldi, int R, int C, float* __restrict__ out, int64_t ldo) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? in[(int64_t)r * ldi + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < C && r < R) out[(int64_t)c * ldo + r] = tile[tx][ty + 8 * k];
  }
}

}  // namespace

extern "C" {

// workspace floats needed by paths_gemm_tn_f32 for a given shape / split count
int64_t paths_gemm_tn_workspace(int N1, int N2, int splits) { return (int64_t)splits * N1 * N2; }

int paths_gemm_tn_f32(const float* a, int64_t lda, const float* b0, int64_t ldb0, int nb0, const float* b1, int64_t ldb1,
                      float* out, int64_t ldo, int M, int N1, int N2, int splits, int accumulate, float* workspace,
                      hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && N1 > 0 && N2 > 0 && splits > 0 && a && b0 && out && workspace, "gemm_tn: bad arguments");
  PATHS_REQUIRE(N1 % 4 == 0 && N2 % 4 == 0 && lda % 4 == 0 && ldb0 % 4 == 0 && (b1 == nullptr || ldb1 % 4 == 0), "gemm_tn: dims must be multiples of 4");
  PATHS_REQUIRE(b1 == nullptr || (nb0 % 128 == 0 && nb0 > 0 && nb0 < N2), "gemm_tn: panel split must be a multiple of 128");
  PATHS_REQUIRE(((uintptr_t)a | (uintptr_t)b0 | (uintptr_t)b1) % 16 == 0, "gemm_tn: operands must be 16-byte aligned");
  int rps = (M + splits - 1) / splits;
  rps = (rps + TK - 1) / TK * TK;
  if (splits == 1 && !accumulate) {      // enough tiles to fill the chip: write the result in place, no slab pass
    TnOperands g1{a, lda, b0, ldb0, b1 ? nb0 : N2, b1, ldb1, M, N1, N2, out, ldo, rps};
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((N2 + 127) / 128, (N1 + 127) / 128, 1), dim3(256), 0, stream, g1);
    PATHS_LAUNCH_CHECK("gemm_tn");
    return PATHS_OK;
  }
  TnOperands g{a, lda, b0, ldb0, b1 ? nb0 : N2, b1, ldb1, M, N1, N2, workspace, (int64_t)N2, rps};
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((N2 + 127) / 128, (N1 + 127) / 128, splits), dim3(256), 0, stream, g);
  PATHS_LAUNCH_CHECK("gemm_tn");
  const int64_t n = (int64_t)N1 * N2;
  if (const int rd = paths_reduce_try_defer(workspace, splits, n, out, ldo, N2, accumulate, 0, stream)) return rd == PATHS_DEFERRED ? PATHS_OK : rd;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, workspace, splits, n, out, ldo, N2, accumulate);
  PATHS_LAUNCH_CHECK("gemm_tn(reduce)");
  return PATHS_OK;
}

int paths_colsum_f32(const float* a, int64_t lda, int M, int N, float* out, int splits, int accumulate, float* workspace,
                     hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && N > 0 && splits > 0 && a && out && workspace, "colsum: bad arguments");
  const int rps = (M + splits - 1) / splits;
  if (N % 4 == 0 && lda % 4 == 0 && reinterpret_cast<uintptr_t>(a) % 16 == 0 && reinterpret_cast<uintptr_t>(workspace) % 16 == 0) {
    if (N <= 128) hipLaunchKernelGGL(colsum_partial4_kernel<32>, dim3((N / 4 + 31) / 32, splits), dim3(256), 0, stream, a, lda, M, N, rps, workspace);
    else hipLaunchKernelGGL(colsum_partial4_kernel<64>, dim3((N / 4 + 63) / 64, splits), dim3(256), 0, stream, a, lda, M, N, rps, workspace);
  } else {
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 255) / 256, splits), dim3(256), 0, stream, a, lda, M, N, rps, workspace);
  }
  PATHS_LAUNCH_CHECK("colsum");
  if (const int rd = paths_reduce_try_defer(workspace, splits, N, out, N, N, accumulate, 1, stream)) return rd == PATHS_DEFERRED ? PATHS_OK : rd;
  hipLaunchKernelGGL(reduce_slabs_small_kernel, dim3((N + 63) / 64), dim3(256), 0, stream, workspace, splits, N, out, accumulate);
  PATHS_LAUNCH_CHECK("colsum(reduce)");
  return PATHS_OK;
}

// out[n] (+)= sum over `splits` slabs of n floats each, fixed order (the second half of paths_colsum_f32, for producers that write
// their own slabs: paths_layernorm_bwd_sums)
int paths_reduce_slabs_f32(const float* slabs, int splits, int n, float* out, int accumulate, hipStream_t stream) {
  PATHS_REQUIRE(slabs && out && splits > 0 && n > 0, "reduce_slabs: bad arguments");
  if (const int rd = paths_reduce_try_defer(slabs, splits, n, out, n, n, accumulate, 1, stream)) return rd == PATHS_DEFERRED ? PATHS_OK : rd;
  hipLaunchKernelGGL(reduce_slabs_small_kernel, dim3((n + 63) / 64), dim3(256), 0, stream, slabs, splits, n, out, accumulate);
  PATHS_LAUNCH_CHECK("reduce_slabs");
  return PATHS_OK;
}

int paths_transpose_f32(const float* in, int64_t ldi, int R, int C, float* out, int64_t ldo, hipStream_t stream) {
  PATHS_REQUIRE(R > 0 && C > 0 && in && out && ldi >= C && ldo >= R, "transpose: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, stream, in, ldi, R, C, out, ldo);
  PATHS_LAUNCH_CHECK("transpose");
  return PATHS_OK;
}

}  // extern "C"
