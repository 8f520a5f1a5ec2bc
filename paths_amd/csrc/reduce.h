// Deferred slab reductions: see reduce_multi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
constexpr int PATHS_DEFERRED = 1000;     // paths_reduce_try_defer: the reduction was registered (0: deferral is off, launch it now)
int paths_reduce_try_defer(const float* slabs, int splits, int64_t n, float* out, int64_t ldo, int ncols, int accumulate, int short_order, hipStream_t stream);
// short_order: sum in reduce_slabs_small_kernel's order (colsum / LayerNorm sums) instead of reduce_slabs_kernel's - the deferred result is bit-identical to the launch it replaces
