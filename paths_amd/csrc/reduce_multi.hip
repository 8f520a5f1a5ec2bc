// Deferred slab reductions (round 4).  Every split-M product of the backward - weight gradients (gemm_tn_f32 / gemm_tn_x6), bias
// gradients (colsum), LayerNorm affine gradients - ends in "out (+)= sum over slabs, fixed order".  As its own launch per product that
// was ~170 launches of 4-10 us per training step (profiles/r04u_kernel_stats_train_k2048_b8.csv: reduce_slabs_small 126 x 3.8 us,
// reduce_slabs_x6 44 x 9.6 us) on a device-bound step.  Between paths_defer_reductions(1) and paths_flush_reductions() the producers
// only REGISTER their reduction; the flush runs all of them in one launch per 32 entries (the table travels by value in the kernel
// arguments: nothing to keep alive on the host).  Same per-element summation order as the single launches -> bit-identical results.
// The caller keeps the slab workspaces alive until the flush and flushes before anything reads an output (paths_amd/backward.py:
// deferred_reductions); an entry whose output overlaps a pending one flushes first (accumulate chains stay ordered).
#include "common.h"
#include "reduce.h"
#include <vector>

namespace {

constexpr int MAX_ENTRIES = 32;
struct Entry {
  const float* slabs;
  float* out;
  int64_t n, ldo;
  int splits, ncols, accumulate, first_block;      // first_block: prefix sum of the entries' block counts; accumulate bit 1: the short-output order
};
struct Table {
  Entry e[MAX_ENTRIES];
  int count;
};

__global__ void __launch_bounds__(256)
reduce_multi_kernel(const Table t) {
  __shared__ float part[4][64];
  int k = 0;
  while (k + 1 < t.count && (int)blockIdx.x >= t.e[k + 1].first_block) ++k;
  const Entry& en = t.e[k];
  const int blk = blockIdx.x - en.first_block;
  const float* __restrict__ slabs = en.slabs;
  const int splits = en.splits;
  const int64_t n = en.n;
  if (en.accumulate & 2) {                         // (reduce_slabs_small_kernel's order: 64 outputs per workgroup, four interleaved quarters of the slabs joined in LDS)
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blk * 64 + c;
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < n) {
      int s = q;
      for (; s + 28 < splits; s += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] += slabs[(int64_t)(s + 4 * u) * n + i];
      }
      for (int u = 0; s < splits; s += 4, ++u) p[u & 7] += slabs[(int64_t)s * n + i];
    }
    part[q][c] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    __syncthreads();
    if (q == 0 && i < n) {
      const float s = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
      float* o = en.out + (int64_t)(i / en.ncols) * en.ldo + i % en.ncols;
      *o = (en.accumulate & 1) ? *o + s : s;
    }
    return;
  }
  const int64_t i = (int64_t)blk * 256 + threadIdx.x;      // (reduce_slabs_kernel's order: 8 interleaved partial sums)
  if (i >= n) return;
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= splits; s += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] += slabs[(int64_t)(s + u) * n + i];
  }
  for (; s < splits; ++s) p[s & 7] += slabs[(int64_t)s * n + i];
  const float sum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  float* o = en.out + (i / en.ncols) * en.ldo + i % en.ncols;
  *o = (en.accumulate & 1) ? *o + sum : sum;
}

struct Pending { Entry e; hipStream_t stream; };
thread_local std::vector<Pending> g_pending;
thread_local int g_defer = 0;

inline bool overlaps(const float* a0, int64_t an, const float* b0, int64_t bn) { return a0 < b0 + bn && b0 < a0 + an; }
inline int64_t out_span(const Entry& e) { return (e.n / e.ncols - 1) * e.ldo + e.ncols; }

int flush(hipStream_t stream) {
  size_t at = 0;
  while (at < g_pending.size()) {
    Table t;
    int blocks = 0;
    t.count = 0;
    for (; at < g_pending.size() && t.count < MAX_ENTRIES; ++at) {
      Entry e = g_pending[at].e;
      e.first_block = blocks;
      blocks += (e.accumulate & 2) ? (int)((e.n + 63) / 64) : (int)((e.n + 255) / 256);
      t.e[t.count++] = e;
    }
    hipLaunchKernelGGL(reduce_multi_kernel, dim3(blocks), dim3(256), 0, stream, t);
    PATHS_LAUNCH_CHECK("reduce_multi");
  }
  g_pending.clear();
  return PATHS_OK;
}

}  // namespace

// Called by the producers in place of their own reduce launch; returns PATHS_DEFERRED when the reduction was registered.
int paths_reduce_try_defer(const float* slabs, int splits, int64_t n, float* out, int64_t ldo, int ncols, int accumulate, int short_order, hipStream_t stream) {
  if (!g_defer) return 0;
  Entry e{slabs, out, n, ldo, splits, ncols, (accumulate ? 1 : 0) | (short_order ? 2 : 0), 0};
  for (const Pending& p : g_pending) {
    // an output that overlaps a pending output (accumulate chains) or pending slabs, slabs that overlap a pending output, or another stream
    if (p.stream != stream || overlaps(out, out_span(e), p.e.out, out_span(p.e)) || overlaps(out, out_span(e), p.e.slabs, (int64_t)p.e.splits * p.e.n) ||
        overlaps(slabs, (int64_t)splits * n, p.e.out, out_span(p.e))) {
      const int rc = flush(p.stream);
      if (rc != PATHS_OK) return rc;
      break;
    }
  }
  g_pending.push_back(Pending{e, stream});
  return PATHS_DEFERRED;
}

extern "C" {

// on != 0: reductions of the split-M backward products are registered instead of launched, until paths_flush_reductions.  Returns
// the previous setting.  Turning it off does NOT flush (call paths_flush_reductions first); per host thread.
int paths_defer_reductions(int on) {
  const int prev = g_defer;
  g_defer = on != 0;
  return prev;
}

// Launch every registered reduction on `stream` (one launch per 32 entries).  Returns the number of entries that were pending through
// *n_entries (may be NULL).
int paths_flush_reductions(int* n_entries, hipStream_t stream) {
  if (n_entries) *n_entries = (int)g_pending.size();
  return flush(stream);
}

}  // extern "C"
