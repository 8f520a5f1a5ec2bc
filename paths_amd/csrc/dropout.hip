// Row-wise dropout kernels of the training path (masks regenerated from a counter-based hash, csrc/dropout.h).
#include "common.h"
#include "dropout.h"

namespace {

// out[r, c] = (resid ? resid[r, c] : 0) + (vec ? vec[c] : x[r, c]) * mask(r * N + c) / (1 - p)        N % 4 == 0
//   x, vec: exactly one is given.  x / out / resid may alias (element-wise).
__global__ void __launch_bounds__(256)
dropout_rows_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ vec, const float* __restrict__ resid, int64_t ldr,
                    float* __restrict__ out, int64_t ldo, int64_t M, int N, DropSite site) {
  const int n4 = N >> 2;
  const int64_t total = M * n4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / n4;
    const int c = (int)(i - r * n4) * 4;
    const f32x4 v = vec ? *reinterpret_cast<const f32x4*>(vec + c) : *reinterpret_cast<const f32x4*>(x + r * ldx + c);
    f32x4 o = resid ? *reinterpret_cast<const f32x4*>(resid + r * ldr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    const uint64_t idx = (uint64_t)r * (uint64_t)N + (uint64_t)c;
    float m[4];                                           // (N % 4 == 0: idx is even, two hashes for the four elements)
    drop_mult2(site, idx, m[0], m[1]);
    drop_mult2(site, idx + 2, m[2], m[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] += v[e] * m[e];
    *reinterpret_cast<f32x4*>(out + r * ldo + c) = o;
  }
}

// mask[i] = 1 (kept) / 0 (dropped) for i in [0, n): tests and debugging only
__global__ void __launch_bounds__(256)
dropout_mask_kernel(float* __restrict__ mask, int64_t n, DropSite site) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    mask[i] = drop_mult(site, (uint64_t)i) != 0.f ? 1.f : 0.f;
}

}  // namespace

DropSite paths_make_drop_site(uint64_t key, float p) {
  DropSite s;
  s.key_lo = (uint32_t)key; s.key_hi = (uint32_t)(key >> 32);
  const double t = (double)p * 65536.0;                     // 16-bit threshold: one 32-bit hash serves two elements (dropout.h)
  s.thr = p <= 0.f ? 0u : (t >= 65535.0 ? 65535u : (t < 1.0 ? 1u : (uint32_t)(t + 0.5)));
  s.scale = s.thr == 0u ? 1.0f : (float)(1.0 / (1.0 - (double)s.thr / 65536.0));
  return s;
}

extern "C" {

int paths_dropout_rows(const float* x, int64_t ldx, const float* vec, const float* resid, int64_t ldr, float* out, int64_t ldo,
                       int64_t M, int N, uint64_t key, float p, hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && out != nullptr && ((x != nullptr) != (vec != nullptr)), "dropout_rows: bad arguments");
  PATHS_REQUIRE(p >= 0.f && p < 1.f, "dropout_rows: p must be in [0, 1)");
  PATHS_REQUIRE(ldo % 4 == 0 && (x == nullptr || ldx % 4 == 0) && (resid == nullptr || ldr % 4 == 0), "dropout_rows: leading dimensions must be multiples of 4");
  const int64_t total = M * (N / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(dropout_rows_kernel, dim3(blocks), dim3(256), 0, stream, x, ldx, vec, resid, ldr, out, ldo, M, N, paths_make_drop_site(key, p));
  PATHS_LAUNCH_CHECK("dropout_rows");
  return PATHS_OK;
}

int paths_dropout_mask(float* mask, int64_t n, uint64_t key, float p, hipStream_t stream) {
  PATHS_REQUIRE(n > 0 && mask != nullptr && p >= 0.f && p < 1.f, "dropout_mask: bad arguments");
  const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, stream, mask, n, paths_make_drop_site(key, p));
  PATHS_LAUNCH_CHECK("dropout_mask");
  return PATHS_OK;
}

}  // extern "C"
