// Device-resident patch selection: importance top-K, 4-child expansion, bounds/background filter, stable
// compaction and the row gathers that build the next magnification level's batch.
//
// Replaces the per-slide host loop of the reference (utils.py:248-260 -> data_utils/slide.py:277-331 and
// data_utils/dataset.py:206-235): .cpu() sync, torch.topk, torch.cat of 4 child blocks, boolean filters,
// host-RAM gather of child features, re-padding.  Nothing here synchronises with the host.
//
// All of it is HBM-bound integer / copy work: rows are 4 KB (features) and 5 KB (LSTM state) and move as
// coalesced 16-byte-per-lane copies; the per-slide bookkeeping (sort, scan) lives in LDS.
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

// ------------------------------------------------------------------------------------------------
// top-K: indices of the `keep` largest scores, ordered by (score descending, index ascending).
// torch.topk (slide.py:298) leaves tie order unspecified; this rule equals it whenever scores are distinct.
//
// Selection by RANK COUNTING instead of a sort: with the 64-bit key (~monotone(score) << 32 | index) every element's output
// position is simply the number of keys below its own, rank_i = #{j : key_j < key_i} (keys are unique), and it is kept iff
// rank_i < count.  That is O(n^2) compares, but they are independent: 64 elements per workgroup x n/64 workgroups per slide put
// the whole chip on a top-K that a one-workgroup-per-slide bitonic sort (66 dependent stages at n = 2048, 21 us) ran on 8 CUs.
// Each workgroup builds all n keys of its slide in LDS (n <= 8192: 64 KiB); lane l of every wave owns element 64 blockIdx.x + l,
// the four waves count over a quarter of the keys each (key pairs read by LDS broadcast), partial counts meet in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int TOPK_MAX = 8192;

__device__ __forceinline__ unsigned long long topk_key(float score, int idx) {
  uint32_t u = __float_as_uint(score);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // monotone float -> uint
  return ((unsigned long long)(~u) << 32) | (uint32_t)idx;
}

__global__ void __launch_bounds__(256)
topk_rank_kernel(const float* __restrict__ scores, int64_t ld, const int64_t* __restrict__ num_ims, int keep,
                 int* __restrict__ keep_idx, int64_t ldk, int* __restrict__ keep_count,
                 const float* __restrict__ row_base, int64_t row_ld, int64_t slide_rows,        // optional: kept_rows[b, i] = address of
                 int64_t* __restrict__ kept_rows, const float* __restrict__ zero_row) {          // row_base[b, keep_idx[b, i], :] (zero_row beyond count)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = (int)num_ims[b];
  const int count = keep < 0 ? n : min(n, keep);
  const int i = blockIdx.x * 64 + lane;                 // this lane's element
  if (blockIdx.x == 0 && tid == 0) keep_count[b] = count;
  auto row_addr = [&](int idx) { return (int64_t)reinterpret_cast<uintptr_t>(row_base + ((int64_t)b * slide_rows + idx) * row_ld); };
  // entries [count, ldk) of the row table point at the zero row (every workgroup covers its own 64 positions)
  if (kept_rows && wave == 0 && i >= count && i < ldk) kept_rows[(int64_t)b * ldk + i] = (int64_t)reinterpret_cast<uintptr_t>(zero_row);
  if (blockIdx.x * 64 >= n) return;                     // nothing valid here (workgroup-uniform)
  if (keep < 0) {                                       // keep all, original order (slide.py:294 not taken)
    if (wave == 0 && i < n) {
      keep_idx[(int64_t)b * ldk + i] = i;
      if (kept_rows) kept_rows[(int64_t)b * ldk + i] = row_addr(i);
    }
    return;
  }
  const int np = (n + 7) & ~7;                          // keys beyond n: all ones (never below a real key)
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
  int* part = reinterpret_cast<int*>(smem + (size_t)((np + 1) & ~1) * 8);      // [4][64] partial counts
  const float* s = scores + (int64_t)b * ld;
  for (int j = tid; j < np; j += 256) keys[j] = j < n ? topk_key(s[j], j) : ~0ull;
  __syncthreads();
  const unsigned long long mine = i < n ? keys[i] : 0ull;
  // quarter of the key range per wave, in units of key PAIRS (16-byte broadcast reads)
  const int pairs = np >> 1, q0 = (pairs * wave) >> 2, q1 = (pairs * (wave + 1)) >> 2;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2* kp = reinterpret_cast<const u64x2*>(keys);
  int cnt = 0;
  int q = q0;
  for (; q + 4 <= q1; q += 4) {
    const u64x2 a = kp[q], c = kp[q + 1], d = kp[q + 2], e = kp[q + 3];
    cnt += (a[0] < mine) + (a[1] < mine) + (c[0] < mine) + (c[1] < mine) + (d[0] < mine) + (d[1] < mine) + (e[0] < mine) + (e[1] < mine);
  }
  for (; q < q1; ++q) { const u64x2 a = kp[q]; cnt += (a[0] < mine) + (a[1] < mine); }
  part[wave * 64 + lane] = cnt;
  __syncthreads();
  if (wave == 0 && i < n) {
    const int rank = part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane];
    if (rank < count) {
      keep_idx[(int64_t)b * ldk + rank] = i;
      if (kept_rows) kept_rows[(int64_t)b * ldk + rank] = row_addr(i);
    }
  }
}

static int launch_topk(const float* scores, int64_t ld, const int64_t* num_ims, int B, int n_max, int keep, int* keep_idx, int64_t ldk,
                       int* keep_count, const float* row_base, int64_t row_ld, int64_t slide_rows, int64_t* kept_rows,
                       const float* zero_row, hipStream_t stream) {
  const size_t lds = (size_t)(((n_max + 7) & ~7) + 2) * 8 + 4 * 64 * sizeof(int);
  PATHS_LDS_OPT_IN(topk_rank_kernel, (TOPK_MAX + 2) * 8 + 1024, "topk");
  // x covers every element AND every position of the kept-row table (ldk may exceed n_max only through padding; both <= TOPK_MAX)
  const int cover = (int)(ldk > n_max ? ldk : n_max);
  PATHS_LAUNCH_STOP(topk_rank_kernel, dim3((cover + 63) / 64, B), dim3(256), lds, stream, scores, ld, num_ims, keep, keep_idx, ldk,
                    keep_count, row_base, row_ld, slide_rows, kept_rows, zero_row);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// expansion: kept patch (x,y) -> blocks (2x,2y) | (2x,2y+1) | (2x+1,2y) | (2x+1,2y+1), each in top-K
// order (slide.py:307-315); keep child iff in bounds and tissue (slide.py:320-325); stable compaction.
// ------------------------------------------------------------------------------------------------
#ifndef PATHS_EXPAND_THREADS
#define PATHS_EXPAND_THREADS 512           // threads of the one workgroup per slide (a multiple of 64, <= 1024).  Alone the kernel is fastest at
                                           // 1024 (11.7 us against 15.1 / 23.5 at 512 / 256), but a 16-wave workgroup has to wait for a CU with four free
                                           // wave slots per SIMD beside the aggregator's kernels: in the recursion 512 gives 3,591-3,624 slides/s against
                                           // 3,527-3,535 (sustained 3,717-3,731 against 3,609-3,616), 256 gives 3,589-3,618
#endif
constexpr int EXP_NT = PATHS_EXPAND_THREADS, EXP_NW = EXP_NT / 64;
__global__ void __launch_bounds__(EXP_NT)
expand_kernel(const int* __restrict__ keep_idx, int64_t ldk, const int* __restrict__ keep_count,
              const int64_t* __restrict__ locs, int64_t n_cur, int patch_size,
              const int* __restrict__ next_x, const int* __restrict__ next_y,          // [B] next-level grid dims
              const int64_t* __restrict__ mask_ptrs,                                     // [B] -> uint8 [X*Y], 1 = tissue
              int64_t n_next,
              int64_t* __restrict__ num_out, int64_t* __restrict__ locs_out, int64_t* __restrict__ parent_out,
              int* __restrict__ src_row, int* __restrict__ src_cell, int* __restrict__ status,
              int* __restrict__ child_pos /*[B, 4*ldk] or null: output row of every candidate child, -1 if dropped*/,
              int* __restrict__ hp_row /*[B, n_next] or null: row of the kept-parent table (b*ldk + i) of every child*/) {
  __shared__ int part[2 * EXP_NW];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int count = keep_count[b];
  const int total = 4 * count;
  const int X = next_x[b], Y = next_y[b];
  const uint8_t* mask = reinterpret_cast<const uint8_t*>(mask_ptrs[b]);
  const int per = (total + EXP_NT - 1) / EXP_NT;
  const int c0 = tid * per, c1 = min(c0 + per, total);

  auto child = [&](int c, int& cx, int& cy, int& i) -> bool {
    const int blk = c / count;
    i = c - blk * count;
    const int prow = keep_idx[(int64_t)b * ldk + i];
    const int64_t lx = locs[((int64_t)b * n_cur + prow) * 2], ly = locs[((int64_t)b * n_cur + prow) * 2 + 1];
    const bool small = (uint64_t)(lx | ly) < 0x80000000ull;       // (pixel coordinates: a 32-bit division is a fraction of the 64-bit one)
    const int64_t px = small ? (int64_t)((int)lx / patch_size) : lx / patch_size;
    const int64_t py = small ? (int64_t)((int)ly / patch_size) : ly / patch_size;
    const int64_t x = 2 * px + (blk >> 1), y = 2 * py + (blk & 1);
    cx = (int)x; cy = (int)y;
    if (x >= X || y >= Y) return false;
    return mask[(int64_t)x * Y + y] != 0;
  };

  // the first candidates of this thread are evaluated ONCE (each evaluation is a chain of four dependent loads: kept index -> its
  // location -> the tissue bit; at keep = 512 a thread owns two candidates) and kept in registers for the write pass below
  constexpr int CACHE = 4;
  int ccx[CACHE], ccy[CACHE], cci[CACHE];
  bool ckeep[CACHE];
  int mine = 0;
  {
    // ... in three stages, so that the CACHE chains of a thread run side by side (all kept indices, then all locations, then all bits)
    int blk[CACHE], prow[CACHE];
    bool on[CACHE], inb[CACHE];
    int64_t lx[CACHE], ly[CACHE];
#pragma unroll
    for (int u = 0; u < CACHE; ++u) {
      on[u] = c0 + u < c1;
      const int c = on[u] ? c0 + u : 0;
      blk[u] = count > 0 ? c / count : 0;
      cci[u] = c - blk[u] * count;
      prow[u] = on[u] ? keep_idx[(int64_t)b * ldk + cci[u]] : 0;
    }
#pragma unroll
    for (int u = 0; u < CACHE; ++u) {
      lx[u] = on[u] ? locs[((int64_t)b * n_cur + prow[u]) * 2] : 0;
      ly[u] = on[u] ? locs[((int64_t)b * n_cur + prow[u]) * 2 + 1] : 0;
    }
#pragma unroll
    for (int u = 0; u < CACHE; ++u) {
      const bool small = (uint64_t)(lx[u] | ly[u]) < 0x80000000ull;
      const int64_t px = small ? (int64_t)((int)lx[u] / patch_size) : lx[u] / patch_size;
      const int64_t py = small ? (int64_t)((int)ly[u] / patch_size) : ly[u] / patch_size;
      const int64_t x = 2 * px + (blk[u] >> 1), y = 2 * py + (blk[u] & 1);
      ccx[u] = (int)x; ccy[u] = (int)y;
      inb[u] = on[u] && x < X && y < Y;
    }
#pragma unroll
    for (int u = 0; u < CACHE; ++u) {
      ckeep[u] = inb[u] && mask[(int64_t)ccx[u] * Y + ccy[u]] != 0;
      mine += ckeep[u] ? 1 : 0;
    }
  }
  for (int c = c0 + CACHE; c < c1; ++c) { int cx, cy, i; mine += child(c, cx, cy, i) ? 1 : 0; }
  // inclusive scan over the per-thread partials: shuffles inside a wave, the wave totals through LDS (two barriers instead of the
  // twenty of a Hillis-Steele scan over LDS)
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if ((tid & 63) >= off) incl += v;
  }
  if ((tid & 63) == 63) part[tid >> 6] = incl;
  __syncthreads();
  if (tid < EXP_NW) {
    int t = part[tid];
#pragma unroll
    for (int off = 1; off < EXP_NW; off <<= 1) {
      const int v = __shfl_up(t, off);
      if (tid >= off) t += v;
    }
    part[EXP_NW + tid] = t;                             // inclusive totals of waves 0 .. tid
  }
  __syncthreads();
  int pos = incl - mine + ((tid >> 6) ? part[EXP_NW + (tid >> 6) - 1] : 0);
  const int n_out = part[2 * EXP_NW - 1];
  if (tid == 0) {
    num_out[b] = n_out;
    if (n_out == 0) atomicOr(status, 1);               // reference falls back to "all cells" (slide.py:336-352)
    if (n_out > n_next) atomicOr(status, 2);
  }
  if (n_out > n_next) return;
  for (int c = c0; c < c1; ++c) {
    int cx, cy, i;
    bool keepc;
    const int u = c - c0;
    if (u < CACHE) {
      keepc = false;
#pragma unroll
      for (int v = 0; v < CACHE; ++v)
        if (v == u) { keepc = ckeep[v]; cx = ccx[v]; cy = ccy[v]; i = cci[v]; }
    } else {
      keepc = child(c, cx, cy, i);
    }
    if (child_pos) child_pos[(int64_t)b * 4 * ldk + c] = keepc ? pos : -1;
    if (keepc) {
      const int64_t o = (int64_t)b * n_next + pos;
      locs_out[2 * o] = (int64_t)cx * patch_size;
      locs_out[2 * o + 1] = (int64_t)cy * patch_size;
      parent_out[o] = i;
      src_row[o] = keep_idx[(int64_t)b * ldk + i];
      src_cell[o] = cx * Y + cy;
      if (hp_row) hp_row[o] = (int)((int64_t)b * ldk + i);
      ++pos;
    }
  }
  // padding tail: zero the bookkeeping rows like collate_fn's zero padding (dataset.py:216-218)
  for (int j = n_out + tid; j < n_next; j += EXP_NT) {
    const int64_t o = (int64_t)b * n_next + j;
    locs_out[2 * o] = 0; locs_out[2 * o + 1] = 0; parent_out[o] = 0; src_row[o] = -1; src_cell[o] = -1;
    if (hp_row) hp_row[o] = -1;
  }
}


// Rare fallback (reference data_utils/slide.py:336-352): a slide whose kept patches have NO tissue children
// continues with every tissue cell of the next grid (or every cell if the grid has no tissue at all), zero patch
// context, parent_inds = cell index.  Only slides with num_out[b] == 0 are touched.
__global__ void __launch_bounds__(1024)
fallback_all_cells_kernel(const int* __restrict__ next_x, const int* __restrict__ next_y, const int64_t* __restrict__ mask_ptrs,
                          int patch_size, int64_t n_next, int64_t* __restrict__ num_out, int64_t* __restrict__ locs_out,
                          int64_t* __restrict__ parent_out, int* __restrict__ src_row, int* __restrict__ src_cell,
                          int* __restrict__ status, int* __restrict__ hp_row) {
  __shared__ int part[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (num_out[b] != 0) return;
  const int X = next_x[b], Y = next_y[b];
  const int total = X * Y;
  const uint8_t* mask = reinterpret_cast<const uint8_t*>(mask_ptrs[b]);
  const int per = (total + 1023) / 1024;
  const int c0 = min(tid * per, total), c1 = min(c0 + per, total);
  int mine = 0;
  for (int c = c0; c < c1; ++c) mine += mask[c] != 0;
  part[tid] = mine;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  const int tissue = part[1023];
  const bool all = tissue == 0;                       // "hope that there will be tissue later" (slide.py:344-347)
  const int n_out = all ? total : tissue;
  int pos = all ? c0 : part[tid] - mine;
  __syncthreads();
  if (tid == 0) {
    num_out[b] = n_out;
    if (n_out > n_next) atomicOr(status, 2);
  }
  if (n_out > n_next) return;
  for (int c = c0; c < c1; ++c) {
    if (all || mask[c]) {
      const int64_t o = (int64_t)b * n_next + pos;
      locs_out[2 * o] = (int64_t)(c / Y) * patch_size;
      locs_out[2 * o + 1] = (int64_t)(c % Y) * patch_size;
      parent_out[o] = c;
      src_row[o] = -1;
      src_cell[o] = c;
      if (hp_row) hp_row[o] = -1;
      ++pos;
    }
  }
}

// One WAVE per output row (four rows per workgroup): features from the next-level grid, LSTM state (h|c) from the kept parent.
// (One 256-thread workgroup per row was 16,384 workgroups for 16 MB in the default path - only the 1-KiB memory-cell part of the
// state is copied there, 64 of the 256 threads had anything to do: 14.6 us.)
constexpr int GATHER_ROWS = 4;
__global__ void __launch_bounds__(256)
gather_kernel(const int64_t* __restrict__ grid_ptrs, const int* __restrict__ src_cell, int D,
              const float* __restrict__ state_cur /*already offset to the first copied column*/, int64_t n_cur, int64_t ld_state_cur,
              const int* __restrict__ src_row, int Dp /*columns copied*/, const int64_t* __restrict__ num_out, int64_t n_next,
              float* __restrict__ fts_out, float* __restrict__ state_out, int zero_pad,
              int64_t* __restrict__ row_ptrs, const float* __restrict__ zero_row) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * GATHER_ROWS + (threadIdx.x >> 6);
  if (j >= n_next) return;
  const int64_t o = (int64_t)b * n_next + j;
  f32x4* fo = fts_out ? reinterpret_cast<f32x4*>(fts_out + o * D) : nullptr;
  f32x4* so = state_out ? reinterpret_cast<f32x4*>(state_out + o * Dp) : nullptr;
  if (j < num_out[b]) {
    const f32x4* fi = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(grid_ptrs[b]) + (int64_t)src_cell[o] * D);
    if (row_ptrs && lane == 0) row_ptrs[o] = (int64_t)reinterpret_cast<uintptr_t>(fi);     // consumers read the row where it lives
    if (fo) for (int i = lane; i < D / 4; i += 64) fo[i] = fi[i];
    if (so) {
      const int sr = src_row[o];
      if (sr >= 0) {
        const f32x4* si = reinterpret_cast<const f32x4*>(state_cur + ((int64_t)b * n_cur + sr) * ld_state_cur);
        for (int i = lane; i < Dp / 4; i += 64) so[i] = si[i];
      } else {                                   // fallback rows (slide.py:338): fresh zero context
        const f32x4 z{0.f, 0.f, 0.f, 0.f};
        for (int i = lane; i < Dp / 4; i += 64) so[i] = z;
      }
    }
  } else {
    if (row_ptrs && lane == 0) row_ptrs[o] = (int64_t)reinterpret_cast<uintptr_t>(zero_row);   // padding rows: a row of zeros
    if (zero_pad) {
      const f32x4 z{0.f, 0.f, 0.f, 0.f};
      if (fo) for (int i = lane; i < D / 4; i += 64) fo[i] = z;
      if (so) for (int i = lane; i < Dp / 4; i += 64) so[i] = z;
    }
  }
}



// Kept parents' h rows -> compact table [B*ldk, D] (rows beyond keep_count are zero): the A operand of the
// once-per-parent half of the next level's LSTM gate GEMM.
__global__ void __launch_bounds__(256)
gather_kept_rows_kernel(const float* __restrict__ src, int64_t n_cur, int64_t ld_src, const int* __restrict__ keep_idx, int64_t ldk,
                        const int* __restrict__ keep_count, int D, float* __restrict__ out) {
  const int b = blockIdx.y, i = blockIdx.x, tid = threadIdx.x;
  f32x4* o = reinterpret_cast<f32x4*>(out + ((int64_t)b * ldk + i) * D);
  if (i < keep_count[b]) {
    const f32x4* r = reinterpret_cast<const f32x4*>(src + ((int64_t)b * n_cur + keep_idx[(int64_t)b * ldk + i]) * ld_src);
    for (int c = tid; c < D / 4; c += 256) o[c] = r[c];
  } else {
    const f32x4 z{0.f, 0.f, 0.f, 0.f};
    for (int c = tid; c < D / 4; c += 256) o[c] = z;
  }
}

// Backward of the parent-state gather (reference data_utils/slide.py:318 `ctx_patch = cat((ctx_patch,)*4)` + filter):
// a kept parent receives the sum of the gradients of its (up to 4) surviving children, in block order (deterministic).
__global__ void __launch_bounds__(256)
gather_bwd_kernel(const int* __restrict__ keep_idx, int64_t ldk, const int* __restrict__ keep_count,
                  const int* __restrict__ child_pos, const float* __restrict__ d_next, int64_t n_next, int Dp,
                  float* __restrict__ d_cur, int64_t n_cur) {
  const int b = blockIdx.y, i = blockIdx.x, tid = threadIdx.x;
  const int count = keep_count[b];
  if (i >= count) return;
  const int row = keep_idx[(int64_t)b * ldk + i];
  int pos[4];
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) pos[blk] = child_pos[(int64_t)b * 4 * ldk + blk * count + i];
  f32x4* out = reinterpret_cast<f32x4*>(d_cur + ((int64_t)b * n_cur + row) * Dp);
  for (int c = tid; c < Dp / 4; c += 256) {
    f32x4 s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
      if (pos[blk] >= 0) s += reinterpret_cast<const f32x4*>(d_next + ((int64_t)b * n_next + pos[blk]) * Dp)[c];
    out[c] = s;
  }
}

// The same sum with separate row strides, an optional compact destination (keep_idx == nullptr: kept slot i -> row i) and a column
// window: dst[b, row(i), 0:width] (+)= sum over the surviving children of kept parent i of src[b, child, 0:width].  The training
// step's once-per-parent form: dHP = sum of the children's gate-pre-activation gradients (compact table), dc_parent = sum of dc0.
__global__ void __launch_bounds__(256)
sibling_sum_kernel(const int* __restrict__ keep_idx, int64_t ldk, const int* __restrict__ keep_count, const int* __restrict__ child_pos,
                   const float* __restrict__ src, int64_t n_src, int64_t ld_src, int width, float* __restrict__ dst, int64_t n_dst,
                   int64_t ld_dst) {
  const int b = blockIdx.y, i = blockIdx.x, tid = threadIdx.x;
  const int count = keep_count[b];
  if (i >= count) return;
  const int row = keep_idx ? keep_idx[(int64_t)b * ldk + i] : i;
  int pos[4];
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) pos[blk] = child_pos[(int64_t)b * 4 * ldk + blk * count + i];
  f32x4* out = reinterpret_cast<f32x4*>(dst + ((int64_t)b * n_dst + row) * ld_dst);
  for (int c = tid; c < width / 4; c += 256) {
    f32x4 s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
      if (pos[blk] >= 0) s += reinterpret_cast<const f32x4*>(src + ((int64_t)b * n_src + pos[blk]) * ld_src)[c];
    out[c] = s;
  }
}

// dst[b, keep_idx[b, i], 0:width] = src[b, i, 0:width] for i < keep_count[b] (the kept parents' h gradient back into the level's state gradient)
__global__ void __launch_bounds__(256)
scatter_kept_rows_kernel(const float* __restrict__ src, int64_t ldk, int64_t ld_src, const int* __restrict__ keep_idx,
                         const int* __restrict__ keep_count, float* __restrict__ dst, int64_t n_dst, int64_t ld_dst, int width) {
  const int b = blockIdx.y, i = blockIdx.x, tid = threadIdx.x;
  if (i >= keep_count[b]) return;
  const int row = keep_idx[(int64_t)b * ldk + i];
  const f32x4* in = reinterpret_cast<const f32x4*>(src + ((int64_t)b * ldk + i) * ld_src);
  f32x4* out = reinterpret_cast<f32x4*>(dst + ((int64_t)b * n_dst + row) * ld_dst);
  for (int c = tid; c < width / 4; c += 256) out[c] = in[c];
}

// Level 0: every grid cell, row-major, no background filter (slide.py:257-269).
__global__ void __launch_bounds__(256)
level0_kernel(const int64_t* __restrict__ grid_ptrs, const int* __restrict__ gx, const int* __restrict__ gy, int D,
              int patch_size, int64_t n0, float* __restrict__ fts, int64_t* __restrict__ locs,
              int64_t* __restrict__ parent, int64_t* __restrict__ num_ims, int zero_pad,
              int64_t* __restrict__ row_ptrs, const float* __restrict__ zero_row) {
  const int b = blockIdx.y;
  const int64_t j = blockIdx.x;
  const int X = gx[b], Y = gy[b];
  const int64_t n = (int64_t)X * Y;
  const int64_t o = (int64_t)b * n0 + j;
  const int tid = threadIdx.x;
  f32x4* fo = fts ? reinterpret_cast<f32x4*>(fts + o * D) : nullptr;
  if (j == 0 && tid == 0) num_ims[b] = n;
  if (j < n) {
    const f32x4* fi = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(grid_ptrs[b]) + j * D);
    if (fo) for (int i = tid; i < D / 4; i += 256) fo[i] = fi[i];
    if (tid == 0) {
      locs[2 * o] = (j / Y) * patch_size; locs[2 * o + 1] = (j % Y) * patch_size; parent[o] = j;
      if (row_ptrs) row_ptrs[o] = (int64_t)reinterpret_cast<uintptr_t>(fi);
    }
  } else {
    if (fo && zero_pad) { const f32x4 z{0.f, 0.f, 0.f, 0.f}; for (int i = tid; i < D / 4; i += 256) fo[i] = z; }
    if (tid == 0) {
      locs[2 * o] = 0; locs[2 * o + 1] = 0; parent[o] = 0;
      if (row_ptrs) row_ptrs[o] = (int64_t)reinterpret_cast<uintptr_t>(zero_row);
    }
  }
}

// tissue mask: 1 iff the row's fp32 sum != 0 (slide.py:324 `fts[...].sum(dim=1) != 0`); one wave per cell.
__global__ void __launch_bounds__(256)
tissue_mask_kernel(const float* __restrict__ grid, int64_t cells, int D, uint8_t* __restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int64_t cell = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= cells) return;
  const f32x4* row = reinterpret_cast<const f32x4*>(grid + cell * D);
  float s = 0.f;
  for (int i = lane; i < D / 4; i += 64) { const f32x4 v = row[i]; s += (v[0] + v[1]) + (v[2] + v[3]); }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
  if (lane == 0) mask[cell] = s != 0.f ? 1 : 0;
}


// The same pass + max|x| of the whole grid (fp16-split range contract of the default GEMM mode: |x| * a_scale must stay a finite
// fp16).  absmax_bits holds the fp32 BIT PATTERN of the running maximum: non-negative floats order like their bit patterns and a
// NaN's pattern is above +inf's, so a NaN anywhere reads back as "not finite".  The atomic is only issued by waves that beat the
// value they last saw, i.e. a handful of times per grid.
__global__ void __launch_bounds__(256)
tissue_mask_absmax_kernel(const float* __restrict__ grid, int64_t cells, int D, uint8_t* __restrict__ mask, unsigned* __restrict__ absmax_bits) {
  const int lane = threadIdx.x & 63;
  const int64_t cell = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= cells) return;
  const f32x4* row = reinterpret_cast<const f32x4*>(grid + cell * D);
  float s = 0.f;
  unsigned m = 0;
  for (int i = lane; i < D / 4; i += 64) {
    const f32x4 v = row[i];
    s += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) m = max(m, __float_as_uint(v[e]) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { s += __shfl_xor(s, o); m = max(m, (unsigned)__shfl_xor((int)m, o)); }
  if (lane == 0) {
    if (mask) mask[cell] = s != 0.f ? 1 : 0;
    if (m > __hip_atomic_load(absmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(absmax_bits, m);
  }
}

// Z[row] = alpha[row] * X[row] (+ H[row] on valid rows): the non-LSTM hierarchical-context update
// (reference model/paths.py:96-109: Z = Y * alpha; Z += apply_to_non_padded(hctx_mlp, prev Z)).
__global__ void __launch_bounds__(256)
scale_add_rows_kernel(const float* __restrict__ x, const float* __restrict__ alpha, const float* __restrict__ h,
                      const int64_t* __restrict__ num_ims, int rows_per_slide, int D, int64_t M, int use_alpha,
                      float* __restrict__ z) {
  const int64_t row = blockIdx.x;
  if (row >= M) return;
  const int b = (int)(row / rows_per_slide), idx = (int)(row % rows_per_slide);
  const bool valid = idx < (int)num_ims[b];
  const float a = use_alpha ? alpha[row] : 1.0f;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * D);
  const f32x4* hr = h ? reinterpret_cast<const f32x4*>(h + row * D) : nullptr;
  f32x4* zr = reinterpret_cast<f32x4*>(z + row * D);
  for (int i = threadIdx.x; i < D / 4; i += 256) {
    f32x4 v = xr[i] * a;
    if (hr && valid) v += hr[i];
    zr[i] = v;
  }
}

// Synthetic grid (paths_amd/synthetic.py): one thread per 4 channels.
__global__ void __launch_bounds__(256)
synth_grid_kernel(float* __restrict__ grid, int X, int Y, int D, uint32_t k2, int level, unsigned long long bg_thr) {
  const int64_t cell = blockIdx.x;
  const int x = (int)(cell / Y), y = (int)(cell % Y);
  const uint32_t k3 = fmix32(k2 + (uint32_t)x * 0xC2B2AE3Du);
  const uint32_t ck = fmix32(k3 + (uint32_t)y * 0x27D4EB2Fu);
  const bool bg = level >= 1 && (unsigned long long)fmix32(ck ^ 0xB6B6B6B6u) < bg_thr;
  for (int c4 = threadIdx.x; c4 < D / 4; c4 += 256) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t u = fmix32(ck + (uint32_t)(4 * c4 + e) * 0x165667B1u + 0x9E3779B9u);
      const float f = ((float)(u >> 8) * 0x1p-23f - 1.0f) * 1.7320508075688772f;
      v[e] = bg ? 0.f : f;
    }
    *reinterpret_cast<f32x4*>(grid + cell * D + 4 * c4) = v;
  }
}

}  // namespace

extern "C" {

int paths_topk(const float* scores, int64_t ld, const int64_t* num_ims, int B, int n_max, int keep,
               int* keep_idx, int64_t ldk, int* keep_count, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && n_max > 0 && n_max <= TOPK_MAX, "topk: n_max (%d) must be in [1, %d]", n_max, TOPK_MAX);
  PATHS_REQUIRE(keep == -1 || keep > 0, "topk: keep must be -1 (all) or > 0");
  PATHS_REQUIRE(ldk >= (keep < 0 ? n_max : (keep < n_max ? keep : n_max)), "topk: keep_idx row too short");
  PATHS_REQUIRE(ldk <= TOPK_MAX, "topk: keep_idx row longer than %d", TOPK_MAX);
  launch_topk(scores, ld, num_ims, B, n_max, keep, keep_idx, ldk, keep_count, nullptr, 0, 0, nullptr, nullptr, stream);
  PATHS_LAUNCH_CHECK("topk");
  return PATHS_OK;
}

// paths_topk + the ADDRESS of every kept row of a row-major [B, slide_rows, row_ld] table (kept_rows [B, ldk]; entries beyond a
// slide's keep_count point at zero_row): the once-per-parent GEMM reads the kept parents' h rows through them, in place.
int paths_topk_rows(const float* scores, int64_t ld, const int64_t* num_ims, int B, int n_max, int keep,
                    int* keep_idx, int64_t ldk, int* keep_count, const float* row_base, int64_t row_ld, int64_t slide_rows,
                    int64_t* kept_rows, const float* zero_row, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && n_max > 0 && n_max <= TOPK_MAX, "topk_rows: n_max (%d) must be in [1, %d]", n_max, TOPK_MAX);
  PATHS_REQUIRE(keep == -1 || keep > 0, "topk_rows: keep must be -1 (all) or > 0");
  PATHS_REQUIRE(ldk >= (keep < 0 ? n_max : (keep < n_max ? keep : n_max)), "topk_rows: keep_idx row too short");
  PATHS_REQUIRE(row_base && kept_rows && zero_row && row_ld > 0 && slide_rows >= n_max, "topk_rows: row table arguments");
  PATHS_REQUIRE(ldk <= TOPK_MAX, "topk_rows: keep_idx row longer than %d", TOPK_MAX);
  launch_topk(scores, ld, num_ims, B, n_max, keep, keep_idx, ldk, keep_count, row_base, row_ld, slide_rows, kept_rows, zero_row, stream);
  PATHS_LAUNCH_CHECK("topk_rows");
  return PATHS_OK;
}

int paths_expand_children(const int* keep_idx, int64_t ldk, const int* keep_count, const int64_t* locs, int64_t n_cur,
                          int patch_size, const int* next_x, const int* next_y, const int64_t* mask_ptrs, int B,
                          int64_t n_next, int64_t* num_out, int64_t* locs_out, int64_t* parent_out, int* src_row,
                          int* src_cell, int* status, int* child_pos, int* hp_row, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && n_cur > 0 && n_next > 0 && patch_size > 0, "expand_children: bad shape");
  PATHS_REQUIRE(4 * ldk <= (int64_t)1 << 30, "expand_children: too many candidates");
  hipLaunchKernelGGL(expand_kernel, dim3(B), dim3(EXP_NT), 0, stream, keep_idx, ldk, keep_count, locs, n_cur, patch_size,
                     next_x, next_y, mask_ptrs, n_next, num_out, locs_out, parent_out, src_row, src_cell, status, child_pos, hp_row);
  PATHS_LAUNCH_CHECK("expand_children");
  return PATHS_OK;
}

int paths_fallback_all_cells(const int* next_x, const int* next_y, const int64_t* mask_ptrs, int patch_size, int B,
                             int64_t n_next, int64_t* num_out, int64_t* locs_out, int64_t* parent_out, int* src_row,
                             int* src_cell, int* status, int* hp_row, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && n_next > 0 && patch_size > 0, "fallback_all_cells: bad shape");
  hipLaunchKernelGGL(fallback_all_cells_kernel, dim3(B), dim3(1024), 0, stream, next_x, next_y, mask_ptrs, patch_size, n_next,
                     num_out, locs_out, parent_out, src_row, src_cell, status, hp_row);
  PATHS_LAUNCH_CHECK("fallback_all_cells");
  return PATHS_OK;
}

int paths_gather_rows(const int64_t* grid_ptrs, const int* src_cell, int D, const float* state_cur, int64_t n_cur,
                      int64_t ld_state_cur, const int* src_row, int Dp, const int64_t* num_out, int B, int64_t n_next,
                      float* fts_out, float* state_out, int zero_pad, int64_t* row_ptrs, const float* zero_row,
                      hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && n_next > 0 && D % 4 == 0 && Dp % 4 == 0 && ld_state_cur % 4 == 0, "gather_rows: bad shape");
  PATHS_REQUIRE(fts_out != nullptr || row_ptrs != nullptr, "gather_rows: features must go somewhere (a copy or row pointers)");
  PATHS_REQUIRE(row_ptrs == nullptr || zero_row != nullptr, "gather_rows: row_ptrs needs a zero row for padding");
  PATHS_REQUIRE((state_cur == nullptr) == (state_out == nullptr), "gather_rows: state in/out must both be given or null");
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n_next + GATHER_ROWS - 1) / GATHER_ROWS), B), dim3(256), 0, stream, grid_ptrs, src_cell, D, state_cur,
                     n_cur, ld_state_cur, src_row, Dp, num_out, n_next, fts_out, state_out, zero_pad, row_ptrs, zero_row);
  PATHS_LAUNCH_CHECK("gather_rows");
  return PATHS_OK;
}

// d_cur [B, n_cur, Dp] must be zero-initialised (rows that were not kept receive no gradient).
int paths_gather_kept_rows(const float* src, int64_t n_cur, int64_t ld_src, const int* keep_idx, int64_t ldk, const int* keep_count,
                           int D, int B, float* out, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && ldk > 0 && D % 4 == 0 && ld_src % 4 == 0 && src && keep_idx && keep_count && out, "gather_kept_rows: bad arguments");
  hipLaunchKernelGGL(gather_kept_rows_kernel, dim3((unsigned)ldk, B), dim3(256), 0, stream, src, n_cur, ld_src, keep_idx, ldk, keep_count, D, out);
  PATHS_LAUNCH_CHECK("gather_kept_rows");
  return PATHS_OK;
}

int paths_gather_rows_bwd(const int* keep_idx, int64_t ldk, const int* keep_count, const int* child_pos, const float* d_next,
                          int64_t n_next, int Dp, float* d_cur, int64_t n_cur, int B, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && ldk > 0 && Dp % 4 == 0 && keep_idx && keep_count && child_pos && d_next && d_cur, "gather_rows_bwd: bad arguments");
  hipLaunchKernelGGL(gather_bwd_kernel, dim3((unsigned)ldk, B), dim3(256), 0, stream, keep_idx, ldk, keep_count, child_pos, d_next, n_next, Dp, d_cur, n_cur);
  PATHS_LAUNCH_CHECK("gather_rows_bwd");
  return PATHS_OK;
}

int paths_sibling_sum(const int* keep_idx, int64_t ldk, const int* keep_count, const int* child_pos, const float* src, int64_t n_src,
                      int64_t ld_src, int width, float* dst, int64_t n_dst, int64_t ld_dst, int B, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && ldk > 0 && width > 0 && width % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && keep_count && child_pos && src && dst,
                "sibling_sum: bad arguments");
  PATHS_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "sibling_sum: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(sibling_sum_kernel, dim3((unsigned)ldk, B), dim3(256), 0, stream, keep_idx, ldk, keep_count, child_pos, src, n_src, ld_src,
                     width, dst, n_dst, ld_dst);
  PATHS_LAUNCH_CHECK("sibling_sum");
  return PATHS_OK;
}

int paths_scatter_kept_rows(const float* src, int64_t ldk, int64_t ld_src, const int* keep_idx, const int* keep_count, float* dst,
                            int64_t n_dst, int64_t ld_dst, int width, int B, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && ldk > 0 && width > 0 && width % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && keep_idx && keep_count && src && dst,
                "scatter_kept_rows: bad arguments");
  PATHS_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "scatter_kept_rows: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(scatter_kept_rows_kernel, dim3((unsigned)ldk, B), dim3(256), 0, stream, src, ldk, ld_src, keep_idx, keep_count, dst, n_dst,
                     ld_dst, width);
  PATHS_LAUNCH_CHECK("scatter_kept_rows");
  return PATHS_OK;
}

int paths_level0_batch(const int64_t* grid_ptrs, const int* gx, const int* gy, int B, int D, int patch_size, int64_t n0,
                       float* fts, int64_t* locs, int64_t* parent, int64_t* num_ims, int zero_pad, int64_t* row_ptrs,
                       const float* zero_row, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && n0 > 0 && D % 4 == 0, "level0_batch: bad shape");
  PATHS_REQUIRE((fts != nullptr || row_ptrs != nullptr) && (row_ptrs == nullptr || zero_row != nullptr), "level0_batch: features need a destination (copy or row pointers + zero row)");
  hipLaunchKernelGGL(level0_kernel, dim3((unsigned)n0, B), dim3(256), 0, stream, grid_ptrs, gx, gy, D, patch_size, n0,
                     fts, locs, parent, num_ims, zero_pad, row_ptrs, zero_row);
  PATHS_LAUNCH_CHECK("level0_batch");
  return PATHS_OK;
}

int paths_scale_add_rows(const float* x, const float* alpha, const float* h, const int64_t* num_ims, int rows_per_slide,
                         int D, int64_t M, int use_alpha, float* z, hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && D % 4 == 0 && rows_per_slide > 0 && num_ims && x && z && (alpha || !use_alpha), "scale_add_rows: bad arguments");
  hipLaunchKernelGGL(scale_add_rows_kernel, dim3((unsigned)M), dim3(256), 0, stream, x, alpha, h, num_ims, rows_per_slide, D, M, use_alpha, z);
  PATHS_LAUNCH_CHECK("scale_add_rows");
  return PATHS_OK;
}

int paths_tissue_mask(const float* grid, int64_t cells, int D, uint8_t* mask, hipStream_t stream) {
  PATHS_REQUIRE(cells > 0 && D % 4 == 0, "tissue_mask: bad shape");
  hipLaunchKernelGGL(tissue_mask_kernel, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, stream, grid, cells, D, mask);
  PATHS_LAUNCH_CHECK("tissue_mask");
  return PATHS_OK;
}

int paths_tissue_mask_absmax(const float* grid, int64_t cells, int D, uint8_t* mask, uint32_t* absmax_bits, hipStream_t stream) {
  PATHS_REQUIRE(cells > 0 && D % 4 == 0 && absmax_bits != nullptr, "tissue_mask_absmax: bad arguments");
  hipLaunchKernelGGL(tissue_mask_absmax_kernel, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, stream, grid, cells, D, mask, absmax_bits);
  PATHS_LAUNCH_CHECK("tissue_mask_absmax");
  return PATHS_OK;
}

int paths_synth_grid(float* grid, int X, int Y, int D, uint32_t slide_level_key, int level, uint64_t bg_threshold, hipStream_t stream) {
  PATHS_REQUIRE(X > 0 && Y > 0 && D % 4 == 0 && (int64_t)X * Y < ((int64_t)1 << 31), "synth_grid: bad shape");
  hipLaunchKernelGGL(synth_grid_kernel, dim3((unsigned)((int64_t)X * Y)), dim3(256), 0, stream, grid, X, Y, D, slide_level_key, level, (unsigned long long)bg_threshold);
  PATHS_LAUNCH_CHECK("synth_grid");
  return PATHS_OK;
}

}  // extern "C"
