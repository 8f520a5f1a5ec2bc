// Multi-tensor AdamW in ONE launch (reference train.py:49-50: torch.optim.AdamW; the training step of BASELINE configs[3]).
//
// torch's default ("foreach") AdamW runs ~12 multi-tensor launches per step over ~300 parameter tensors and costs 3-4 ms of HOST
// time per step (list handling in Python / ATen); the training step here is host ~ device.  This kernel applies the same update to
// every tensor of a list from a device-resident pointer table, and it follows the foreach implementation's OPERATION ORDER AND
// ROUNDING step by step (torch/optim/adam.py:_multi_tensor_adam, non-capturable, decoupled weight decay) - every intermediate of
// that sequence is an fp32 tensor there, so it is rounded to fp32 here at the same places:
//     p  = p * (1 - lr wd)
//     m  = m + w (g - m)                    w = 1 - beta1      (torch.lerp, weight < 0.5 form)
//     v  = v * beta2
//     v  = v + (1 - beta2) (g g)            (addcmul)
//     s  = sqrt(v);  s = s / sqrt(1 - beta2^t);  s = s + eps
//     p  = p + step_size (m / s)            step_size = -lr / (1 - beta1^t)     (addcdiv)
// so that a training trajectory is the one torch's own optimizer produces bit for bit (tests/test_gpu_backward.py::
// test_hip_adamw_is_bitwise_torch_foreach; fixture G10 = the reference's epoch loop).  The fused torch variant (fused=True) rounds
// differently and was dropped in round 3 for that reason.  `flavor` selects where a multiply-add is ONE fused operation (bit 0: lerp,
// bit 1: addcmul, bit 2: addcdiv) - what the ATen kernels of the installed torch build do was determined by that test (default 7).
#include "common.h"

namespace {

struct AdamwParams {
  const int64_t* table;      // [n][4] device addresses: param, grad, exp_avg, exp_avg_sq
  const int32_t* blocks;     // [nblocks][2]: tensor index, first element
  const int64_t* numel;      // [n]
  const float* step_size;    // [n]  -lr / (1 - beta1^t) per tensor
  const float* bc2_sqrt;     // [n]  sqrt(1 - beta2^t) per tensor
  float wd_scale, lerp_w, beta2, value, eps;
  int has_wd, flavor;
};

constexpr int CHUNK = 4096;  // elements per workgroup (256 threads x 4 x float4)

__device__ __forceinline__ float adamw_one(float& p, float g, float& m, float& v, const AdamwParams& a, float step_size, float bc2) {
#pragma clang fp contract(off)
  if (a.has_wd) p = p * a.wd_scale;
  const float diff = g - m;
  m = (a.flavor & 1) ? fmaf(a.lerp_w, diff, m) : m + a.lerp_w * diff;
  v = v * a.beta2;
  const float gg = g * g;
  v = (a.flavor & 2) ? fmaf(a.value, gg, v) : v + a.value * gg;
  float s = sqrtf(v);
  s = s / bc2;
  s = s + a.eps;
  const float q = m / s;
  p = (a.flavor & 4) ? fmaf(step_size, q, p) : p + step_size * q;
  return p;
}

__global__ void __launch_bounds__(256) adamw_multi_kernel(AdamwParams a) {
  const int t = a.blocks[2 * blockIdx.x], e0 = a.blocks[2 * blockIdx.x + 1];
  const int64_t n = a.numel[t];
  float* P = reinterpret_cast<float*>(a.table[4 * t]);
  const float* G = reinterpret_cast<const float*>(a.table[4 * t + 1]);
  float* M = reinterpret_cast<float*>(a.table[4 * t + 2]);
  float* V = reinterpret_cast<float*>(a.table[4 * t + 3]);
  const float ss = a.step_size[t], bc2 = a.bc2_sqrt[t];
  const bool vec = (((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = e0 + (int64_t)(r * 256 + threadIdx.x) * 4;
    if (i >= n) break;
    if (vec && i + 4 <= n) {
      f32x4 p = *reinterpret_cast<f32x4*>(P + i), m = *reinterpret_cast<f32x4*>(M + i), v = *reinterpret_cast<f32x4*>(V + i);
      const f32x4 g = *reinterpret_cast<const f32x4*>(G + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) { float pk = p[k], mk = m[k], vk = v[k]; adamw_one(pk, g[k], mk, vk, a, ss, bc2); p[k] = pk; m[k] = mk; v[k] = vk; }
      *reinterpret_cast<f32x4*>(P + i) = p; *reinterpret_cast<f32x4*>(M + i) = m; *reinterpret_cast<f32x4*>(V + i) = v;
    } else {
      for (int64_t j = i; j < n && j < i + 4; ++j) { float pk = P[j], mk = M[j], vk = V[j]; adamw_one(pk, G[j], mk, vk, a, ss, bc2); P[j] = pk; M[j] = mk; V[j] = vk; }
    }
  }
}

}  // namespace

extern "C" {

// elements of a tensor that one workgroup of paths_adamw_multi updates (the caller builds the block map with it)
int paths_adamw_chunk() { return CHUNK; }

// One AdamW step over n tensors (fp32, contiguous).  table [n][4] int64 device addresses (param, grad, exp_avg, exp_avg_sq), numel [n]
// int64, blocks [nblocks][2] int32 (tensor index, first element; CHUNK elements per block), step_size / bc2_sqrt [n] fp32 - all in
// DEVICE memory.  wd_scale = 1 - lr * weight_decay (has_wd = 0 skips the decay), lerp_w = 1 - beta1, value = 1 - beta2.
int paths_adamw_multi(const int64_t* table, const int64_t* numel, const int32_t* blocks, int nblocks, const float* step_size,
                      const float* bc2_sqrt, float wd_scale, int has_wd, float lerp_w, float beta2, float value, float eps, int flavor,
                      hipStream_t stream) {
  PATHS_REQUIRE(table && numel && blocks && step_size && bc2_sqrt && nblocks > 0, "adamw_multi: null table / empty block map");
  PATHS_REQUIRE(lerp_w > 0.f && lerp_w < 0.5f, "adamw_multi: 1 - beta1 must be in (0, 0.5) (torch.lerp's small-weight form); got %g", (double)lerp_w);
  AdamwParams a{table, blocks, numel, step_size, bc2_sqrt, wd_scale, lerp_w, beta2, value, eps, has_wd, flavor};
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(nblocks), dim3(256), 0, stream, a);
  PATHS_LAUNCH_CHECK("adamw_multi");
  return PATHS_OK;
}

}  // extern "C"
