// Measured roofline denominators on the box ("spec peaks must be replaced by a measured back-to-back-MFMA and a stream-copy
// microbenchmark"): prints ONE JSON object.
//   mfma_f16_32x32x16 / mfma_f16_16x16x32: back-to-back issue, operands in registers (random fp16 data, not zeros: the chip holds a
//     lower clock on random operands, guide "DVFS give-back"), 4 independent accumulators, one and two waves per SIMD on all CUs
//   copy / read: float4 streams over 1 GiB (far beyond the 256 MiB Infinity Cache)
// build: hipcc -O3 --offload-arch=gfx950 tools/peaks.hip -o tools/_bin/peaks
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
__device__ __forceinline__ f16x8 rnd_frag(uint32_t seed) {
  f16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (_Float16)(((fmix32(seed * 8u + i) >> 8) * (1.0f / 8388608.0f)) - 1.0f);
  return v;
}

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  f16x8 a0 = rnd_frag(s), a1 = rnd_frag(s + 77777u), b0 = rnd_frag(s + 1234567u), b1 = rnd_frag(s + 7654321u);
  if constexpr (SHAPE == 0) {
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = f32x16{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c[0], 0, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c[1], 0, 0, 0);
        c[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c[2], 0, 0, 0);
        c[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c[3], 0, 0, 0);
      }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) t += c[i][j];
    out[s] = t;
  } else {
    f32x4 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = f32x4{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c[0], 0, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c[1], 0, 0, 0);
        c[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c[2], 0, 0, 0);
        c[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c[3], 0, 0, 0);
      }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) t += c[i][j];
    out[s] = t;
  }
}

__global__ void __launch_bounds__(256) copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) read_kernel(const f32x4* __restrict__ src, float* __restrict__ out, int64_t n) {
  f32x4 acc = {0, 0, 0, 0};
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += src[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

template <class F> float time_us(F&& launch, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* out; CK(hipMalloc(&out, 64 << 20));
  const int iters = 4096;                                    // x 16 MFMAs per iteration
  printf("{\"device\": \"%s\", \"cus\": %d", prop.gcnArchName, cus);
  for (int shape = 0; shape < 2; ++shape)
    for (int wps = 1; wps <= 2; ++wps) {
      const int blocks = cus * wps;                          // 256 threads = 4 waves = one per SIMD
      auto launch = [&] { if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
                          else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, out, iters); };
      for (int w = 0; w < 20; ++w) launch();                 // settle the clock under load
      const float us = time_us(launch, 20);
      const double flop = (double)blocks * 4 * iters * 16 * 2.0 * (shape == 0 ? 32.0 * 32 * 16 : 16.0 * 16 * 32);
      printf(", \"mfma_f16_%s_%dwps_tflops\": %.1f", shape == 0 ? "32x32x16" : "16x16x32", wps, flop / us * 1e-6);
    }
  const int64_t n = (1ll << 30) / 16;
  f32x4 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
  CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 2, n * 16));
  const float cus_ = time_us([&] { hipLaunchKernelGGL(copy_kernel, dim3(cus * 8), dim3(256), 0, 0, a, b, n); }, 10);
  const float rus_ = time_us([&] { hipLaunchKernelGGL(read_kernel, dim3(cus * 8), dim3(256), 0, 0, a, out, n); }, 10);
  printf(", \"copy_1gib_gbs\": %.0f, \"copy_note\": \"read + write bytes / time\", \"read_1gib_gbs\": %.0f}\n", 2.0 * n * 16 / cus_ * 1e-3, 1.0 * n * 16 / rus_ * 1e-3);
  return 0;
}
