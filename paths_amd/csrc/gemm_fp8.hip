// e4m3 GEMM on the CDNA4 block-scaled matrix-core instruction (v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales: 64 k per
// instruction at twice the bf16 rate): the low-precision variant BASELINE.json configs[4] names for the STRESS geometry (one level,
// 8192 patches x 1536 features; "fp8 (e4m3, per-tensor scale) on K3-K5": proj / attention / FFN products of the aggregator,
// reference model/aggregator.py:25-33,70-72).  NOT a parity path: an e4m3 operand carries 4 significant bits, so the logits move
// by ~1e-3 (the north star's bar is 1e-4); `bench.py --mode stress --fp8` prints the measured distance next to the speed, and
// tests/test_gpu_parity.py pins the error band.  Opt-in only (ops.AGG_FP8 / PATHS_AGG_FP8=1).
//
//   out[M, N] = act( (A[M,K] sa) (W[N,K] sw)^T / (sa sw) + bias ) (+ residual)
//
// A: fp32 activations quantised to an e4m3 image in one pass of their own (paths_fp8_quantize: x * sa -> saturate at +-448 -> e4m3,
// sa = 448 / max|A| from paths_fp8_scale, a device scalar: no host sync; quantising while the GEMM stages A was measured first and
// is bound by the fp32 bytes every one of the N / 256 column tiles pulls through L2: 245-550 TFLOP/s); W: e4m3 image [Npad, K] +
// device scalar sw from paths_fp8_pack_weight.
// One workgroup = 4 waves (2 x 2) = a 256 x 256 output tile, every wave 128 x 128 = 4 x 4 MFMA tiles (256 accumulator registers),
// k in stages of 64 bytes: double-buffered LDS with 80-byte rows (conflict-free 16-byte fragment reads), the next stage's global
// loads in flight under the 16 MFMAs of the current one.  The fragment of lane (r = l & 31, h = l >> 5) is 32 consecutive k bytes
// of row r at k offset 32 h, for A and B alike: whatever k the hardware assigns to a fragment slot, the same slot of A and of B
// hold the same k, which is all a dot product needs.
#include <type_traits>

#include "common.h"

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int ROWB = 80;                       // bytes per LDS row (64 + 16 pad)
constexpr int TILE_B = BM * ROWB;              // 20 KiB per operand tile
constexpr int SCALE_ONE = 0x7F7F7F7F;          // E8M0 127 = 2^0 in every byte

__device__ __forceinline__ uint32_t fp8x4(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
  int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (uint32_t)v;
}

// ---- max |x| over a [M, K] fp32 matrix (row stride ld) as IEEE bits in *bits (non-negative floats order like unsigned ints)
// num_ims != null: row r is token r % rows_per_slide of slide r / rows_per_slide and counts only if that token is valid (special token +
// num_ims patches): rows of padded tokens may hold anything (tiles that are all padding are skipped by their producers)
__global__ void __launch_bounds__(256)
absmax_kernel(const float* __restrict__ x, int64_t ld, int64_t M, int K, unsigned int* __restrict__ bits, const int64_t* __restrict__ num_ims,
              int rows_per_slide) {
  float m = 0.f;
  const int k4 = K / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M * k4; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / k4;
    const int c = (int)(i - r * k4);
    if (num_ims != nullptr && (r % rows_per_slide) > num_ims[r / rows_per_slide]) continue;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ld + 4 * c);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(bits, __float_as_uint(m));
}
__global__ void scale_from_bits_kernel(unsigned int* __restrict__ bits, float* __restrict__ scale) {
  const float m = __uint_as_float(*bits);
  *scale = (m > 0.f && m < INFINITY) ? 448.0f / m : 1.0f;
  *bits = 0u;                                  // ready for the next use of the scratch word
}

// ---- W8[n][k] = e4m3(W[n][k] * sw), rows n >= N zero
__global__ void __launch_bounds__(256)
pack_weight_kernel(const float* __restrict__ w, int64_t ldw, int N, int Npad, int K, const float* __restrict__ scale, uint8_t* __restrict__ w8) {
  const float s = *scale;
  const int k4 = K / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)Npad * k4; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / k4;
    const int c = (int)(i - n * k4);
    uint32_t v = 0u;
    if (n < N) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(w + n * ldw + 4 * c);
      v = fp8x4(x[0] * s, x[1] * s, x[2] * s, x[3] * s);
    }
    *reinterpret_cast<uint32_t*>(w8 + n * K + 4 * c) = v;
  }
}

struct Fp8Gemm {
  const uint8_t* A8;                 // [Mpad, K] e4m3 (rows >= M zero)
  const uint8_t* W8;                 // [Npad, K]
  const float* a_scale; const float* w_scale;
  const float* bias;                 // [N] or null
  const float* residual; int64_t ldr;
  float* out; int64_t ldo;
  int M, N, K, act;
  int MT, NT;                        // tiles along M / N
  uint8_t* out8; const float* out_scale;   // optional e4m3 output image [Mpad, N] = e4m3(result * *out_scale) instead of fp32 `out`
  unsigned int* out_absmax;          // optional: max |result| as float bits (atomicMax), for calibrating out_scale
};

__global__ void __launch_bounds__(256)
gemm_fp8_kernel(Fp8Gemm g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [2 buffers][A tile | B tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (workgroup b runs on XCD b % 8, each with its own L2): XCD x owns the row panels mt = x + 8 q, so an A panel
  // is fetched by ONE L2; inside an XCD 32 consecutive workgroups are 4 row panels x 8 column tiles (1.5 MB of A + 3 MB of W at
  // K = 1536): the W tiles are shared by 4 workgroups and stay in that L2 while the group of panels walks across N
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int grp = jx / (4 * g.NT), rem = jx - grp * 4 * g.NT;
  const int mt = xcd + 8 * (4 * grp + (rem & 3)), nt = rem >> 2;
  if (mt >= g.MT) return;
  const int m0 = mt * BM, n0 = nt * BN;
  // staging: a 256 x 64-byte tile per operand and stage = 4 x 16-byte loads per thread (4 lanes per 64-byte row piece)
  const int sr = tid >> 2, sq = tid & 3;
  u32x4 ra[2][4], rb[2][4];                  // two register sets: the loads of stage k + 2 are issued while stage k computes
  auto gload = [&](int kt, auto set) {
    constexpr int S = decltype(set)::value;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      ra[S][p] = *reinterpret_cast<const u32x4*>(g.A8 + (int64_t)(m0 + sr + 64 * p) * g.K + kt * BK + 16 * sq);
      rb[S][p] = *reinterpret_cast<const u32x4*>(g.W8 + (int64_t)(n0 + sr + 64 * p) * g.K + kt * BK + 16 * sq);
    }
  };
  auto swrite = [&](int buf, auto set) {
    constexpr int S = decltype(set)::value;
    char* sA = smem + buf * 2 * TILE_B;
    char* sB = sA + TILE_B;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = sr + 64 * p;
      *reinterpret_cast<u32x4*>(sA + row * ROWB + 16 * sq) = ra[S][p];
      *reinterpret_cast<u32x4*>(sB + row * ROWB + 16 * sq) = rb[S][p];
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / BK;
  gload(0, S0{});
  gload(1, S1{});
  swrite(0, S0{});
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  auto frag = [&](const char* base) -> i32x8 {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(base);
    const u32x4 hi = *reinterpret_cast<const u32x4*>(base + 16);
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  // One stage, as ONE basic block with the phase order pinned by hand -
  // hipcc otherwise waits for the next stage's global loads and writes them to LDS BEFORE the MFMAs of this stage (a full memory
  // round trip exposed per 64 k: 0.7 PFLOP/s).  Stage j's operands travel in register set j & 1.  Stage k (LDS buffer k & 1): global
  // loads of stage k + 2 into set k & 1 (free: stage k went to LDS one stage ago; they have two stages to land) -> fragments ->
  // 16 MFMAs (the A fragments of rows 2, 3 are read under the MFMAs of row 0) -> LDS write of stage k + 1 from set (k + 1) & 1
  // (loaded during stage k - 1) -> barrier
  // Every stage runs the same code (K is a multiple of 128: an even number of stages, no peeled variants whose accumulator
  // registers would have to be reconciled through scratch): past the end the loads re-read the last stage and the LDS write lands
  // in a buffer nobody reads.
  auto stage = [&](int kt, auto set) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    const int buf = kt & 1;
    gload(min(kt + 2, nk - 1), set);             // set k & 1 is free: stage k went to LDS during stage k - 1
    const char* sA = smem + buf * 2 * TILE_B + (wm * 128 + fr) * ROWB + 32 * fh;
    const char* sB = smem + buf * 2 * TILE_B + TILE_B + (wn * 128 + fr) * ROWB + 32 * fh;
    i32x8 bf[4], af[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = frag(sB + j * 32 * ROWB);
    af[0] = frag(sA);
    af[1] = frag(sA + 32 * ROWB);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[0], bf[j], acc[0][j], 0, 0, 0, SCALE_ONE, 0, SCALE_ONE);
    af[2] = frag(sA + 2 * 32 * ROWB);
    af[3] = frag(sA + 3 * 32 * ROWB);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 1; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], bf[j], acc[i][j], 0, 0, 0, SCALE_ONE, 0, SCALE_ONE);
    __builtin_amdgcn_sched_barrier(0);
    swrite(buf ^ 1, std::integral_constant<int, SET ^ 1>{});
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    stage(kt, S0{});
    stage(kt + 1, S1{});
  }

  // epilogue: C/D map of the 32 x 32 forms: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  Rows past M are
  // read (residual) from row M - 1 and never stored: the loads of a tile go out together instead of one branch each.
  const float inv = 1.0f / (*g.a_scale * *g.w_scale);
  const float oscale = g.out8 ? *g.out_scale : 1.0f;
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + wn * 128 + 32 * j + fr;
    const int colc = min(col, g.N - 1);
    const float bv = g.bias ? g.bias[colc] : 0.f;
    const float* rp = g.residual ? g.residual + colc : nullptr;
    float* op = g.out + col;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int rbase = m0 + wm * 128 + 32 * i + 4 * fh + 16 * half;       // rows rbase + (r & 3) + 8 (r >> 2), r = 0 .. 7
        float res[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) res[r] = rp ? rp[(int64_t)min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1) * g.ldr] : 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          float v = acc[i][j][8 * half + r] * inv + bv;
          if (g.act == 1) v = fmaxf(v, 0.f);
          v += res[r];
          if (row < g.M && col < g.N) {
            amax = fmaxf(amax, fabsf(v));
            if (g.out8) {
              const float q = fminf(fmaxf(v * oscale, -448.f), 448.f);
              g.out8[(int64_t)row * g.N + col] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(q, 0.f, 0, false) & 0xff);
            } else {
              op[(int64_t)row * g.ldo] = v;
            }
          }
        }
      }
  }
  if (g.out_absmax) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0) atomicMax(g.out_absmax, __float_as_uint(amax));
  }
}

}  // namespace

extern "C" {

// *scale = 448 / max|x| over the [M, K] fp32 matrix (row stride ld); scratch: one zero-initialised uint32 on the device (left zero);
// num_ims / rows_per_slide (optional): token-major activations [slides * rows_per_slide, K] - only valid token rows count
int paths_fp8_scale(const float* x, int64_t ld, int64_t M, int K, float* scale, unsigned int* scratch, const int64_t* num_ims,
                    int rows_per_slide, hipStream_t stream) {
  PATHS_REQUIRE(x && scale && scratch && M > 0 && K > 0 && K % 4 == 0 && ld % 4 == 0 && (uintptr_t)x % 16 == 0, "fp8_scale: bad arguments");
  PATHS_REQUIRE(num_ims == nullptr || (rows_per_slide > 0 && M % rows_per_slide == 0), "fp8_scale: M must be slides x rows_per_slide with num_ims");
  const int64_t n4 = M * (K / 4);
  const unsigned blocks = (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, stream, x, ld, M, K, scratch, num_ims, rows_per_slide);
  PATHS_LAUNCH_CHECK("fp8_scale(absmax)");
  hipLaunchKernelGGL(scale_from_bits_kernel, dim3(1), dim3(1), 0, stream, scratch, scale);
  PATHS_LAUNCH_CHECK("fp8_scale");
  return PATHS_OK;
}

// e4m3 image of a weight matrix W [N, K] (row stride ldw): w8 [Npad, K] bytes (Npad = N rounded up to 256, zero rows), *scale = 448 / max|W|
int paths_fp8_pack_weight(const float* w, int64_t ldw, int N, int K, uint8_t* w8, float* scale, unsigned int* scratch, hipStream_t stream) {
  PATHS_REQUIRE(w && w8 && scale && scratch && N > 0 && K > 0 && K % 64 == 0 && ldw % 4 == 0, "fp8_pack_weight: K must be a multiple of 64 (got %d)", K);
  const int rc = paths_fp8_scale(w, ldw, N, K, scale, scratch, nullptr, 0, stream);
  if (rc != PATHS_OK) return rc;
  const int Npad = (N + BN - 1) / BN * BN;
  const int64_t n4 = (int64_t)Npad * (K / 4);
  const unsigned blocks = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, stream, w, ldw, N, Npad, K, scale, w8);
  PATHS_LAUNCH_CHECK("fp8_pack_weight");
  return PATHS_OK;
}

// x8 [ceil(M/256)*256, K] = e4m3(x * *scale) of an fp32 [M, K] matrix (row stride ld), zero rows behind M (the GEMM reads whole tiles)
int paths_fp8_quantize(const float* x, int64_t ld, int M, int K, const float* scale, uint8_t* x8, hipStream_t stream) {
  PATHS_REQUIRE(x && x8 && scale && M > 0 && K > 0 && K % 4 == 0 && ld % 4 == 0 && (uintptr_t)x % 16 == 0, "fp8_quantize: bad arguments");
  const int Mpad = (M + BM - 1) / BM * BM;
  const int64_t n4 = (int64_t)Mpad * (K / 4);
  const unsigned blocks = (unsigned)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, stream, x, ld, M, Mpad, K, scale, x8);
  PATHS_LAUNCH_CHECK("fp8_quantize");
  return PATHS_OK;
}

// out[M, N] (ldo) = act(A W^T + bias) (+ residual (ldr)) with e4m3 operands: a8 = paths_fp8_quantize image of A [M, K] with *a_scale,
// w8 / *w_scale from paths_fp8_pack_weight; act 0 = none, 1 = relu.  K % 128 == 0.
int paths_gemm_nt_fp8(const uint8_t* a8, const uint8_t* w8, const float* a_scale, const float* w_scale, const float* bias,
                      float* out, int64_t ldo, int M, int N, int K, int act, const float* residual, int64_t ldr, hipStream_t stream) {
  PATHS_REQUIRE(a8 && w8 && a_scale && w_scale && out && M > 0 && N > 0 && K > 0, "gemm_nt_fp8: bad arguments");
  PATHS_REQUIRE(K % (2 * BK) == 0 && ((uintptr_t)a8 | (uintptr_t)w8) % 16 == 0, "gemm_nt_fp8: K (%d) must be a multiple of 128, operands 16-byte aligned", K);
  PATHS_REQUIRE(act == 0 || act == 1, "gemm_nt_fp8: act must be 0 or 1");
  PATHS_LDS_OPT_IN(gemm_fp8_kernel, 4 * TILE_B, "gemm_nt_fp8");
  const int MT = (M + BM - 1) / BM, NT = (N + BN - 1) / BN;
  Fp8Gemm g{a8, w8, a_scale, w_scale, bias, residual, ldr, out, ldo, M, N, K, act, MT, NT, nullptr, nullptr, nullptr};
  const int groups = ((MT + 7) / 8 + 3) / 4;            // groups of 4 row panels per XCD
  hipLaunchKernelGGL(gemm_fp8_kernel, dim3(8 * groups * 4 * NT), dim3(256), 4 * TILE_B, stream, g);
  PATHS_LAUNCH_CHECK("gemm_nt_fp8");
  return PATHS_OK;
}

// The same product handed on in e4m3: out8 [ceil(M/256)*256, N] = e4m3(act(A W^T + bias) * *out_scale) (the A image of the next GEMM:
// rows >= M must be zero on entry and are not written), with *out_scale a CALIBRATED per-tensor scale (448 / max|result| of an
// earlier call: out_absmax, if given, receives max|result| of this call as float bits through atomicMax, so the caller can
// calibrate and check).  Saves the fp32 round trip of a wide intermediate (the feed-forward's hidden layer: 1.6 GB at the stress shape).
int paths_gemm_nt_fp8_out8(const uint8_t* a8, const uint8_t* w8, const float* a_scale, const float* w_scale, const float* bias,
                           uint8_t* out8, const float* out_scale, unsigned int* out_absmax, int M, int N, int K, int act,
                           hipStream_t stream) {
  PATHS_REQUIRE(a8 && w8 && a_scale && w_scale && out8 && out_scale && M > 0 && N > 0 && K > 0, "gemm_nt_fp8_out8: bad arguments");
  PATHS_REQUIRE(K % (2 * BK) == 0 && ((uintptr_t)a8 | (uintptr_t)w8) % 16 == 0, "gemm_nt_fp8_out8: K (%d) must be a multiple of 128, operands 16-byte aligned", K);
  PATHS_REQUIRE(act == 0 || act == 1, "gemm_nt_fp8_out8: act must be 0 or 1");
  PATHS_LDS_OPT_IN(gemm_fp8_kernel, 4 * TILE_B, "gemm_nt_fp8_out8");
  const int MT = (M + BM - 1) / BM, NT = (N + BN - 1) / BN;
  Fp8Gemm g{a8, w8, a_scale, w_scale, bias, nullptr, 0, nullptr, 0, M, N, K, act, MT, NT, out8, out_scale, out_absmax};
  const int groups = ((MT + 7) / 8 + 3) / 4;
  hipLaunchKernelGGL(gemm_fp8_kernel, dim3(8 * groups * 4 * NT), dim3(256), 4 * TILE_B, stream, g);
  PATHS_LAUNCH_CHECK("gemm_nt_fp8_out8");
  return PATHS_OK;
}

}  // extern "C"
