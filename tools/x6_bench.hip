// Feasibility microbenchmark: fp32-accurate GEMM on the bf16 matrix cores by 3-way operand splitting ("bf16x6").
//
//   x = hi + mid + lo exactly (three bf16, 8 significant bits each = the 24 of fp32);
//   x*w ~= hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid      (dropped terms <= 2^-25 |x w|)
//
// 6 v_mfma_f32_32x32x16_bf16 per fp32 MAC-block instead of 8 v_mfma_f32_32x32x2_f32 at 1/16 the rate:
// 2.67x the fp32-MFMA peak.  Prints time, fp32-equivalent TFLOP/s and the error against an fp64 host reference
// next to the error of a plain fp32 FMA chain.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/x6_bench.hip -o tools/_bin/x6_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <type_traits>

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h;
}
__global__ void fill_kernel(float* p, int64_t n, uint32_t seed, float scale) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = (((fmix32((uint32_t)i * 0x9E3779B1u + seed) >> 8) * (1.0f / 8388608.0f)) - 1.0f) * scale;
}

// split layout: [rows][K/16][3 planes][16] bf16
__global__ void split_kernel(const float* __restrict__ x, int64_t ldx, __bf16* __restrict__ out, int rows, int K) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;   // one thread per (row, k)
  if (i >= (int64_t)rows * K) return;
  int r = i / K, k = i % K;
  float v = x[(int64_t)r * ldx + k];
  __bf16 h = (__bf16)v; float r1 = v - (float)h;
  __bf16 m = (__bf16)r1; float r2 = r1 - (float)m;
  __bf16 l = (__bf16)r2;
  __bf16* o = out + (int64_t)r * 3 * K + (k >> 4) * 48 + (k & 15);
  o[0] = h; o[16] = m; o[32] = l;
}

constexpr int ROWB = 112;   // LDS bytes per staged row: 96 data (3 planes x 16 bf16) + 16 pad -> 7r mod 16 distinct: conflict-free b128

template <int WTM, int WTN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_x6_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ W, float* __restrict__ C, int M, int N, int K, uint64_t*) {
  constexpr int BM = WTM * 64, BN = WTN * 64;          // 2 x 2 waves
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int PA = BM / 128, PB = BN / 128;          // staging passes: 256 threads = 128 rows x 2 half-rows (48 B each)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int nbx = gridDim.x;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  (void)nbx;
  const int srow = tid >> 1, shalf = tid & 1;
  const int64_t rowbytes = (int64_t)K * 6;
  const char* gp[PA + PB];
#pragma unroll
  for (int p = 0; p < PA; ++p) gp[p] = (const char*)A + (int64_t)min(m0 + p * 128 + srow, M - 1) * rowbytes + shalf * 48;
#pragma unroll
  for (int p = 0; p < PB; ++p) gp[PA + p] = (const char*)W + (int64_t)min(n0 + p * 128 + srow, N - 1) * rowbytes + shalf * 48;
  u32x4 st[PA + PB][3];
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < PA + PB; ++p)
#pragma unroll
      for (int c = 0; c < 3; ++c) st[p][c] = *reinterpret_cast<const u32x4*>(gp[p] + (int64_t)kt * 96 + c * 16);
  };
  auto swrite = [&](int buf) {
    char* s = smem + buf * STAGE;
#pragma unroll
    for (int p = 0; p < PA + PB; ++p)
#pragma unroll
      for (int c = 0; c < 3; ++c) *reinterpret_cast<u32x4*>(s + (p * 128 + srow) * ROWB + shalf * 48 + c * 16) = st[p][c];
  };
  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fragoff = (lane & 31) * ROWB + (lane >> 5) * 16;
  const char* sA = smem + (wm * WTM * 32) * ROWB + fragoff;
  const char* sB = smem + (BM + wn * WTN * 32) * ROWB + fragoff;
  const int nk = K / 16;
  gload(0);
  swrite(0);
  __syncthreads();
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
    bf16x8 fb[WTN][3];
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[j][p] = *reinterpret_cast<const bf16x8*>(sB + buf * STAGE + j * 32 * ROWB + p * 32);
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
      bf16x8 fa[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[p] = *reinterpret_cast<const bf16x8*>(sA + buf * STAGE + i * 32 * ROWB + p * 32);
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[j][0], c, 0, 0, 0);   // lo*hi
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j][2], c, 0, 0, 0);   // hi*lo
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[j][1], c, 0, 0, 0);   // mid*mid
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[j][0], c, 0, 0, 0);   // mid*hi
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j][1], c, 0, 0, 0);   // hi*mid
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j][0], c, 0, 0, 0);   // hi*hi
        acc[i][j] = c;
      }
    }
    if (kt + 1 < nk) swrite(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  const int row0 = m0 + wm * WTM * 32, col0 = n0 + wn * WTN * 32;
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = col0 + 32 * j + (lane & 31);
        if (row < M && col < N) C[(int64_t)row * N + col] = acc[i][j][r];
      }
}


// ---- v1: pinned software pipeline (1 wave / SIMD, 512 registers): see the schedule comment inside
typedef const u32x4 __attribute__((address_space(1))) * gptr16;
__device__ __forceinline__ u32x4 ldg16(const char* p) { return *reinterpret_cast<gptr16>(reinterpret_cast<uintptr_t>(p)); }

template <int WTM, int WTN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_x6p_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ W, float* __restrict__ C, int M, int N, int K, uint64_t* dbg) {
  constexpr int BM = WTM * 64, BN = WTN * 64;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int PA = BM / 128, PB = BN / 128, NP = PA + PB;
  constexpr int NL = 3 * NP;                       // 16-byte chunks staged per thread per k16 stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  // XCD-aware order: XCD x (= linear id % 8) owns a contiguous run of tiles, column-block fastest within 8 row-blocks
  const int nbx = gridDim.x, nby = gridDim.y, nblk = nbx * nby;
  int lin = blockIdx.y * nbx + blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = lin & 7, j = lin >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  constexpr int GM = 8;
  const int per_group = GM * nbx;
  const int grp = lin / per_group, in_grp = lin - grp * per_group;
  const int rows_in_grp = min(GM, nby - grp * GM);
  const int by = grp * GM + in_grp % rows_in_grp, bx = in_grp / rows_in_grp;
  const int m0 = by * BM, n0 = bx * BN;

  const int srow = tid >> 1, shalf = tid & 1;
  const int64_t rowbytes = (int64_t)K * 6;
  const char* gp[NP];
#pragma unroll
  for (int p = 0; p < PA; ++p) gp[p] = (const char*)A + (int64_t)min(m0 + p * 128 + srow, M - 1) * rowbytes + shalf * 48;
#pragma unroll
  for (int p = 0; p < PB; ++p) gp[PA + p] = (const char*)W + (int64_t)min(n0 + p * 128 + srow, N - 1) * rowbytes + shalf * 48;
  u32x4 st[NL];
  auto gload_one = [&](int idx, int kt) { st[idx] = ldg16(gp[idx / 3] + (int64_t)kt * 96 + (idx % 3) * 16); };
  auto swrite_one = [&](int idx, int buf) {
    *reinterpret_cast<u32x4*>(smem + buf * STAGE + ((idx / 3) * 128 + srow) * ROWB + shalf * 48 + (idx % 3) * 16) = st[idx];
  };
  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fragoff = (lane & 31) * ROWB + (lane >> 5) * 16;
  const char* sA = smem + (wm * WTM * 32) * ROWB + fragoff;
  const char* sB = smem + (BM + wn * WTN * 32) * ROWB + fragoff;
  bf16x8 fa[2][3], fb[2][WTN][3];
  auto read_a = [&](int buf, int i, int slot, int p) { fa[slot][p] = *reinterpret_cast<const bf16x8*>(sA + buf * STAGE + i * 32 * ROWB + p * 32); };
  auto read_b = [&](int buf, int j, int slot, int p) { fb[slot][j][p] = *reinterpret_cast<const bf16x8*>(sB + buf * STAGE + j * 32 * ROWB + p * 32); };
  // One k16 stage = WTM*WTN*6 MFMAs (32 cycles each), issued strictly back to back; every other instruction is pinned
  // into one of the gaps between them, at most two LDS operations per gap (4 waves x 2 x 1 KiB / 32 cyc = the LDS
  // array's 256 B/clk):
  //   gaps [0, NL)                    : the NL global loads of stage kt+1, one per gap
  //   row i, gaps 12..14 of the row   : the 3 fragment reads of A row i+1 (other register slot)
  //   row WTM-2, every 2nd gap (or 1) : the NL LDS writes of stage kt+1 (>= (WTM-2) rows after their loads)
  //   barrier after row WTM-2, then row WTM-1 covers the fragment reads of stage kt+1 (A row 0, all of B; 2 per gap)
  constexpr int RG = WTN * 6;                // gaps per row
  constexpr int WSTEP = (RG >= 2 * NL) ? 2 : 1;
  static_assert(RG >= NL, "LDS writes must fit in one row of gaps");
  auto one_mfma = [&](int g, int sb) {
    const int i = g / RG, j = (g % RG) / 6, t = g % 6, sa = i & 1;
    constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};   // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sa][PA_[t]], fb[sb][j][PB_[t]], acc[i][j], 0, 0, 0);
  };
  // stage kt: st[] holds stage kt+1 (loaded one full stage ago) -> written to the other LDS buffer in the early gaps,
  // each register refilled right away with its chunk of stage kt+2.
  auto stage_body = [&](int kt, int buf, int sb, bool more1, bool more2) {
#pragma unroll
    for (int g = 0; g < (WTM - 1) * RG; ++g) {
      one_mfma(g, sb);
      const int i = g / RG, gr = g % RG;
      if (gr >= 12 && gr < 15) read_a(buf, i + 1, (i + 1) & 1, gr - 12);
      if (more1 && g % 2 == 0 && g / 2 < NL) swrite_one(g / 2, buf ^ 1);
      if (more2 && g % 2 == 1 && g / 2 < NL) gload_one(g / 2, kt + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = (WTM - 1) * RG; g < WTM * RG; ++g) {
      one_mfma(g, sb);
      const int gr = g % RG;
      if (more1) {
#pragma unroll
        for (int f = 2 * gr; f < 2 * gr + 2; ++f) {
          if (f < 3) read_a(buf ^ 1, 0, 0, f);
          else if (f < 3 + 3 * WTN) read_b(buf ^ 1, (f - 3) / 3, sb ^ 1, (f - 3) % 3);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  static_assert(2 * NL <= (WTM - 1) * RG, "staging does not fit before the barrier");
  static_assert(WTM % 2 == 0, "A fragment slots alternate per row: row WTM-1 must use slot 1 while slot 0 is refilled");
  const int nk = K / 16;   // even (K % 32 == 0)
#pragma unroll
  for (int idx = 0; idx < NL; ++idx) gload_one(idx, 0);
#pragma unroll
  for (int idx = 0; idx < NL; ++idx) swrite_one(idx, 0);
#pragma unroll
  for (int idx = 0; idx < NL; ++idx) gload_one(idx, 1);
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 3; ++p) read_a(0, 0, 0, p);
#pragma unroll
  for (int j = 0; j < WTN; ++j)
#pragma unroll
    for (int p = 0; p < 3; ++p) read_b(0, j, 0, p);
  uint64_t t0 = 0, r0 = 0;
  if (dbg) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int kt = 0; kt < nk - 2; kt += 2) {
    __builtin_amdgcn_sched_barrier(0);
    stage_body(kt, 0, 0, true, true);
    stage_body(kt + 1, 1, 1, true, true);
  }
  stage_body(nk - 2, 0, 0, true, false);
  stage_body(nk - 1, 1, 1, false, false);
  if (dbg) {
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { uint64_t* d = dbg + 2 * ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave); d[0] = t1 - t0; d[1] = r1 - r0; }
  }
  const int row0 = m0 + wm * WTM * 32, col0 = n0 + wn * WTN * 32;
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = col0 + 32 * j + (lane & 31);
        if (row < M && col < N) C[(int64_t)row * N + col] = acc[i][j][r];
      }
}


// ---- v3: A stays fp32 in HBM and is split in registers on its way to LDS; W is pre-split in the MFMA-native tiled
// image [n/32][k/16][plane 3][k-half 2][n%32][8 bf16] (a 32-row x 16-k x 1-plane fragment = 1 KiB, lane l owns bytes [16l,16l+16)),
// so every W load instruction is 1 KiB contiguous and every fragment read is conflict-free.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

__global__ void tile_split_kernel(const float* __restrict__ w, int64_t ldw, __bf16* __restrict__ out, int N, int K) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * K) return;
  int n = i / K, k = i % K;
  float v = w[(int64_t)n * ldw + k];
  __bf16 h = (__bf16)v; float r1 = v - (float)h;
  __bf16 m = (__bf16)r1; float r2 = r1 - (float)m;
  __bf16 l = (__bf16)r2;
  __bf16* o = out + ((int64_t)(n >> 5) * (K >> 4) + (k >> 4)) * 1536 + ((k >> 3) & 1) * 256 + (n & 31) * 8 + (k & 7);
  o[0] = h; o[512] = m; o[1024] = l;
}

template <int WTM, int WTN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_x6f_kernel(const float* __restrict__ A, int64_t lda, const __bf16* __restrict__ Wt, float* __restrict__ C, int M, int N, int K, uint64_t* dbg) {
  constexpr int BM = WTM * 64, BN = WTN * 64;
  constexpr int SA = 2 * WTM, SB = 2 * WTN;            // 32-row sub-tiles per block
  constexpr int STAGE = (SA + SB) * 3072;
  constexpr int NA = BM / 64;                          // fp32 A chunks (4 floats) per thread per k16 stage
  constexpr int NPB = SB * 3, NB = (NPB + 3) / 4;      // 1-KiB W pieces per stage, per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int nbx = gridDim.x, nby = gridDim.y, nblk = nbx * nby;
  int lin = blockIdx.y * nbx + blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = lin & 7, j = lin >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  constexpr int GM = 8;
  const int per_group = GM * nbx;
  const int grp = lin / per_group, in_grp = lin - grp * per_group;
  const int rows_in_grp = min(GM, nby - grp * GM);
  const int by = grp * GM + in_grp % rows_in_grp, bx = in_grp / rows_in_grp;
  const int m0 = by * BM, n0 = bx * BN;

  // ---- staging addresses
  const int arow = tid >> 2, ac = tid & 3;
  const float* ap[NA]; int awr[NA];
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const int row = p * 64 + arow;
    ap[p] = A + (int64_t)min(m0 + row, M - 1) * lda + 4 * ac;
    awr[p] = (row >> 5) * 3072 + (ac >> 1) * 512 + (row & 31) * 16 + (ac & 1) * 8;
  }
  const char* bp[NB]; int bwr[NB];
  const int64_t wsub = (int64_t)(K >> 4) * 3072;       // bytes between 32-row groups of W
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int pc = wave + 4 * i;
    if (pc >= NPB) pc -= 4;                            // duplicate of this wave's previous piece (same data, same slot)
    const int sub = pc / 3, pl = pc % 3;
    bp[i] = (const char*)Wt + ((int64_t)(n0 >> 5) + sub) * wsub + pl * 1024 + lane * 16;
    bwr[i] = SA * 3072 + sub * 3072 + pl * 1024 + lane * 16;
  }
  f32x4 sa[NA]; u32x4 sbr[NB];
  uint32_t hi[NA][2], mid[NA][2], lo[NA][2];
  auto gload_a = [&](int q, int kt) { sa[q] = *reinterpret_cast<const f32x4 __attribute__((address_space(1)))*>(reinterpret_cast<uintptr_t>(ap[q] + kt * 16)); };
  auto gload_b = [&](int q, int kt) { sbr[q] = ldg16(bp[q] + (int64_t)kt * 3072); };
  // split of one staged A chunk in 7 small steps (each <= 4 VALU ops: they ride in MFMA gaps)
  auto a_step = [&](int q, int st, int buf) {
    f32x4& v = sa[q];
    if (st == 0) { hi[q][0] = pk_bf16(v[0], v[1]); hi[q][1] = pk_bf16(v[2], v[3]); }
    if (st == 1) { v[0] -= bf_lo(hi[q][0]); v[1] -= bf_hi(hi[q][0]); }
    if (st == 2) { v[2] -= bf_lo(hi[q][1]); v[3] -= bf_hi(hi[q][1]); }
    if (st == 3) { mid[q][0] = pk_bf16(v[0], v[1]); mid[q][1] = pk_bf16(v[2], v[3]); }
    if (st == 4) { v[0] -= bf_lo(mid[q][0]); v[1] -= bf_hi(mid[q][0]); }
    if (st == 5) { v[2] -= bf_lo(mid[q][1]); v[3] -= bf_hi(mid[q][1]); }
    if (st == 6) {
      lo[q][0] = pk_bf16(v[0], v[1]); lo[q][1] = pk_bf16(v[2], v[3]);
      char* d = smem + buf * STAGE + awr[q];
      *reinterpret_cast<u32x2*>(d) = u32x2{hi[q][0], hi[q][1]};
      *reinterpret_cast<u32x2*>(d + 1024) = u32x2{mid[q][0], mid[q][1]};
      *reinterpret_cast<u32x2*>(d + 2048) = u32x2{lo[q][0], lo[q][1]};
    }
  };
  auto swrite_b = [&](int q, int buf) { *reinterpret_cast<u32x4*>(smem + buf * STAGE + bwr[q]) = sbr[q]; };

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const char* sA = smem + (wm * WTM) * 3072 + lane * 16;
  const char* sB = smem + (SA + wn * WTN) * 3072 + lane * 16;
  bf16x8 fa[2][3], fb[2][WTN][3];
  auto read_a = [&](int buf, int i, int slot, int p) { fa[slot][p] = *reinterpret_cast<const bf16x8*>(sA + buf * STAGE + i * 3072 + p * 1024); };
  auto read_b = [&](int buf, int j, int slot, int p) { fb[slot][j][p] = *reinterpret_cast<const bf16x8*>(sB + buf * STAGE + j * 3072 + p * 1024); };
  constexpr int RG = WTN * 6;
  auto one_mfma = [&](int g, int sb) {
    const int i = g / RG, j = (g % RG) / 6, t = g % 6, sl = i & 1;
    constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sl][PA_[t]], fb[sb][j][PB_[t]], acc[i][j], 0, 0, 0);
  };
  // gaps before the barrier: [0, 8 NA): A chunk q = g/8 -> steps 0..6 then its reload; [8 NA, 8 NA + 2 NB): W pieces (write, reload)
  constexpr int GA = 8 * NA, GB = 2 * NB;
  static_assert(GA + GB <= (WTM - 1) * RG, "staging does not fit before the barrier");
  auto stage_body = [&](int kt, auto bufc, auto m1c, auto m2c) {
    constexpr int buf = decltype(bufc)::value, sb = buf;
    constexpr bool more1 = decltype(m1c)::value, more2 = decltype(m2c)::value;
    static_for<0, (WTM - 1) * RG>([&](auto gc) {
      constexpr int g = decltype(gc)::value, i = g / RG, gr = g % RG;
      one_mfma(g, sb);
      if constexpr (gr >= 12 && gr < 15) read_a(buf, i + 1, (i + 1) & 1, gr - 12);
      if constexpr (g < GA) {
        if constexpr (more1 && g % 8 < 7) a_step(g / 8, g % 8, buf ^ 1);
        if constexpr (more2 && g % 8 == 7) gload_a(g / 8, kt + 2);
      } else if constexpr (g < GA + GB) {
        constexpr int q = (g - GA) / 2;
        if constexpr (more1 && (g - GA) % 2 == 0) swrite_b(q, buf ^ 1);
        if constexpr (more2 && (g - GA) % 2 == 1) gload_b(q, kt + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    static_for<(WTM - 1) * RG, WTM * RG>([&](auto gc) {
      constexpr int g = decltype(gc)::value, gr = g % RG;
      one_mfma(g, sb);
      if constexpr (more1) {
        static_for<2 * gr, 2 * gr + 2>([&](auto fc) {
          constexpr int f = decltype(fc)::value;
          if constexpr (f < 3) read_a(buf ^ 1, 0, 0, f);
          else if constexpr (f < 3 + 3 * WTN) read_b(buf ^ 1, (f - 3) / 3, sb ^ 1, (f - 3) % 3);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  static_assert(WTM % 2 == 0, "A fragment slots alternate per row");
  constexpr std::integral_constant<int, 0> I0{}; constexpr std::integral_constant<int, 1> I1{};
  constexpr std::true_type T{}; constexpr std::false_type F{};
  const int nk = K / 16;
#pragma unroll
  for (int q = 0; q < NA; ++q) gload_a(q, 0);
#pragma unroll
  for (int q = 0; q < NB; ++q) gload_b(q, 0);
#pragma unroll
  for (int q = 0; q < NA; ++q)
#pragma unroll
    for (int st = 0; st < 7; ++st) a_step(q, st, 0);
#pragma unroll
  for (int q = 0; q < NB; ++q) swrite_b(q, 0);
#pragma unroll
  for (int q = 0; q < NA; ++q) gload_a(q, 1);
#pragma unroll
  for (int q = 0; q < NB; ++q) gload_b(q, 1);
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 3; ++p) read_a(0, 0, 0, p);
#pragma unroll
  for (int j = 0; j < WTN; ++j)
#pragma unroll
    for (int p = 0; p < 3; ++p) read_b(0, j, 0, p);
  uint64_t t0 = 0, r0 = 0;
  if (dbg) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int kt = 0; kt < nk - 2; kt += 2) {
    __builtin_amdgcn_sched_barrier(0);
    stage_body(kt, I0, T, T);
    stage_body(kt + 1, I1, T, T);
  }
  stage_body(nk - 2, I0, T, F);
  stage_body(nk - 1, I1, F, F);
  if (dbg) {
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { uint64_t* d = dbg + 2 * ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave); d[0] = t1 - t0; d[1] = r1 - r0; }
  }
  const int row0 = m0 + wm * WTM * 32, col0 = n0 + wn * WTN * 32;
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = col0 + 32 * j + (lane & 31);
        if (row < M && col < N) C[(int64_t)row * N + col] = acc[i][j][r];
      }
}

// plain fp32 FMA chain for sampled outputs (what the f32-input MFMA computes, bitwise, per the microarch guide)
__global__ void ref32_kernel(const float* A, const float* W, const int* rows, const int* cols, float* out, int ns, int K) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const float* a = A + (int64_t)rows[i] * K; const float* w = W + (int64_t)cols[i] * K;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s = fmaf(a[k], w[k], s);
  out[i] = s;
}

template <int WTM, int WTN>
float run3(const float* A, const __bf16* Wt, float* C, int M, int N, int K, int iters) {
  constexpr int BM = WTM * 64, BN = WTN * 64;
  size_t lds = 2ull * (2 * WTM + 2 * WTN) * 3072;
  auto kern = gemm_x6f_kernel<WTM, WTN>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, A, (int64_t)K, Wt, C, M, N, K, nullptr);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, A, (int64_t)K, Wt, C, M, N, K, nullptr);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int nw = grid.x * grid.y * 4;
  uint64_t* dbg; CK(hipMalloc(&dbg, nw * 16)); CK(hipMemset(dbg, 0, nw * 16));
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, A, (int64_t)K, Wt, C, M, N, K, nullptr);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, A, (int64_t)K, Wt, C, M, N, K, dbg);
  CK(hipDeviceSynchronize());
  std::vector<uint64_t> h(nw * 2); CK(hipMemcpy(h.data(), dbg, nw * 16, hipMemcpyDeviceToHost));
  std::vector<double> cyc, ghz;
  for (int i = 0; i < nw; ++i) if (h[2 * i + 1]) { cyc.push_back((double)h[2 * i]); ghz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0)); }
  std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
  printf("   main loop: median %.0f cycles (%.0f per k16 stage; MFMA floor %d), in-kernel clock median %.2f GHz\n", cyc[cyc.size() / 2],
         cyc[cyc.size() / 2] / (K / 16), WTM * WTN * 6 * 32, ghz[ghz.size() / 2]);
  CK(hipFree(dbg));
  return ms / iters * 1e3f;
}

template <class Kern>
void launch(Kern kern, dim3 grid, size_t lds, const __bf16* As, const __bf16* Ws, float* C, int M, int N, int K, uint64_t* dbg);
template <int WTM, int WTN, int V>
float run(const __bf16* As, const __bf16* Ws, float* C, int M, int N, int K, int iters) {
  constexpr int BM = WTM * 64, BN = WTN * 64;
  size_t lds = 2ull * (BM + BN) * ROWB;
  auto kern = V == 0 ? gemm_x6_kernel<WTM, WTN> : gemm_x6p_kernel<WTM, WTN>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch(kern, grid, lds, As, Ws, C, M, N, K, nullptr);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch(kern, grid, lds, As, Ws, C, M, N, K, nullptr);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  if (V == 1) {   // in-kernel clock and loop cycles (diagnostic launch after ~iters back-to-back launches)
    const int nw = grid.x * grid.y * 4;
    uint64_t* dbg; CK(hipMalloc(&dbg, nw * 16)); CK(hipMemset(dbg, 0, nw * 16));
    for (int i = 0; i < 50; ++i) launch(kern, grid, lds, As, Ws, C, M, N, K, nullptr);
    launch(kern, grid, lds, As, Ws, C, M, N, K, dbg);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> h(nw * 2); CK(hipMemcpy(h.data(), dbg, nw * 16, hipMemcpyDeviceToHost));
    std::vector<double> cyc, ghz;
    for (int i = 0; i < nw; ++i) if (h[2 * i + 1]) { cyc.push_back((double)h[2 * i]); ghz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0)); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    printf("   main loop: median %.0f cycles (%.0f per k16 stage; MFMA floor %d), in-kernel clock median %.2f GHz\n", cyc[cyc.size() / 2],
           cyc[cyc.size() / 2] / (K / 16), WTM * WTN * 6 * 32, ghz[ghz.size() / 2]);
    CK(hipFree(dbg));
  }
  return ms / iters * 1e3f;
}

template <class Kern>
void launch(Kern kern, dim3 grid, size_t lds, const __bf16* As, const __bf16* Ws, float* C, int M, int N, int K, uint64_t* dbg) {
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, As, Ws, C, M, N, K, dbg);
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
  float *A, *W, *C; __bf16 *As, *Ws;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
  CK(hipMalloc(&As, (size_t)M * K * 6)); CK(hipMalloc(&Ws, (size_t)N * K * 6));
  fill_kernel<<<((int64_t)M * K + 255) / 256, 256>>>(A, (int64_t)M * K, 1u, 1.7320508f);
  fill_kernel<<<((int64_t)N * K + 255) / 256, 256>>>(W, (int64_t)N * K, 77u, 0.03125f);
  split_kernel<<<((int64_t)M * K + 255) / 256, 256>>>(A, K, As, M, K);
  split_kernel<<<((int64_t)N * K + 255) / 256, 256>>>(W, K, Ws, N, K);
  __bf16* Wt; CK(hipMalloc(&Wt, (size_t)N * K * 6));
  tile_split_kernel<<<((int64_t)N * K + 255) / 256, 256>>>(W, K, Wt, N, K);
  CK(hipDeviceSynchronize());
  const double gflop = 2.0 * M * N * K * 1e-9;
  float us;
  us = run<4, 4, 0>(As, Ws, C, M, N, K, 20); printf("x6 v0 256x256 (4 waves 4x4): %.1f us  %.1f TF fp32-equivalent\n", us, gflop / us * 1e3);
  us = run<4, 2, 1>(As, Ws, C, M, N, K, 20); printf("x6 v1 256x128 (4 waves 4x2): %.1f us  %.1f TF fp32-equivalent\n", us, gflop / us * 1e3);
  us = run<4, 4, 1>(As, Ws, C, M, N, K, 20); printf("x6 v1 256x256 (4 waves 4x4): %.1f us  %.1f TF fp32-equivalent\n", us, gflop / us * 1e3);
  CK(hipMemset(C, 0, (size_t)M * N * 4));
  if (N % 192 == 0) { us = run3<4, 3>(A, Wt, C, M, N, K, 20); printf("x6 v3 256x192 (fp32 A split in-flight, tiled W): %.1f us  %.1f TF fp32-equivalent\n", us, gflop / us * 1e3); }
  if (N % 256 == 0) us = run3<4, 4>(A, Wt, C, M, N, K, 20); printf("x6 v3 256x256 (fp32 A split in-flight, tiled W): %.1f us  %.1f TF fp32-equivalent\n", us, gflop / us * 1e3);
  // accuracy on sampled outputs: error of the split-bf16 product and of a plain fp32 FMA chain against fp64
  const int ns = 4096;
  std::vector<int> hr(ns), hc(ns);
  uint32_t s = 12345u;
  auto nxt = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  for (int i = 0; i < ns; ++i) { hr[i] = nxt() % M; hc[i] = nxt() % N; }
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hC((size_t)M * N), h32(ns);
  CK(hipMemcpy(hA.data(), A, hA.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hW.data(), W, hW.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
  int *dr, *dc; float* d32;
  CK(hipMalloc(&dr, ns * 4)); CK(hipMalloc(&dc, ns * 4)); CK(hipMalloc(&d32, ns * 4));
  CK(hipMemcpy(dr, hr.data(), ns * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc.data(), ns * 4, hipMemcpyHostToDevice));
  ref32_kernel<<<(ns + 63) / 64, 64>>>(A, W, dr, dc, d32, ns, K);
  CK(hipMemcpy(h32.data(), d32, ns * 4, hipMemcpyDeviceToHost));
  double e6 = 0, e32 = 0, r6 = 0, r32 = 0, scale = 0;
  for (int i = 0; i < ns; ++i) {
    double ref = 0, sabs = 0;
    for (int k = 0; k < K; ++k) { double p = (double)hA[(size_t)hr[i] * K + k] * hW[(size_t)hc[i] * K + k]; ref += p; sabs += fabs(p); }
    double d6 = fabs(hC[(size_t)hr[i] * N + hc[i]] - ref), d32e = fabs(h32[i] - ref);
    e6 = fmax(e6, d6); e32 = fmax(e32, d32e); r6 += d6 * d6; r32 += d32e * d32e; scale += sabs;
  }
  scale /= ns;
  printf("error vs fp64 over %d samples (mean sum|a w| = %.3f):\n  bf16x6 MFMA : max %.3e  rms %.3e  (max/scale %.2e)\n  fp32 chain  : max %.3e  rms %.3e  (max/scale %.2e)\n",
         ns, scale, e6, sqrt(r6 / ns), e6 / scale, e32, sqrt(r32 / ns), e32 / scale);
  return 0;
}
