// What one SIMD of gfx950 overlaps: in-kernel cycle counts (s_memtime) of MFMA / VALU instruction mixes at one and two waves per SIMD.
// Decides the shape of the attention step (round 4).  build: hipcc -O3 --offload-arch=gfx950 tools/simd_probe.hip -o tools/_bin/simd_probe
// Unit of work "U" = 12 x v_mfma_f32_32x32x16_f16 = 24 x v_mfma_f32_16x16x32_f16 (one score or PV product of a 64-key attention step),
// VALU load "W" = 72 x v_fma_f32 (or 32 v_exp + 40 v_fma).  Every instruction is asm volatile: the order below IS the issue order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define M32(c) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define M16(c) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

enum { K_M32, K_M16, K_M16DEP, K_V, K_EV, K_M32_V, K_M16_V, K_M32_EV, K_M16_EV, K_M32_THEN_V, K_M16_THEN_V, K_M32_THEN_EV, K_M16_THEN_EV, K_LOCK32_V, K_LOCK32_EV, K_LOCK16_EV, K_LOCK32_ATTN, K_FREE32_ATTN, K_LOCK32_ATTN_VLD, K_LOCK32_ATTN_VDMA, K_LOCK32_ATTN_MDMA, K_LOCK32_ATTN_MLD, K_MDMA_WAIT, K_MDMA_STREAM, K_MDMA_STREAM_WAIT, K_COUNT };
static const char* NAMES[] = {"12xM32", "24xM16 (8 accumulators)", "24xM16 (chains of 3)", "72 fma", "32 exp + 40 fma",
  "12x(M32 + 6 fma)", "24x(M16 + 3 fma)", "12x(M32 + 2-3 exp + 3-4 fma)", "24x(M16 + 1-2 exp + 1-2 fma)",
  "12xM32 then 72 fma", "24xM16 then 72 fma", "12xM32 then 32 exp + 40 fma", "24xM16 then 32 exp + 40 fma",
  "LOCKED 12xM32 | 72 fma", "LOCKED 12xM32 | 32 exp + 40 fma", "LOCKED 24xM16 | 32 exp + 40 fma", "LOCKED 24xM32 | 32 exp + 80 fma", "free   24xM32 then 32 exp + 80 fma",
  "LOCKED 24xM32 | 4 loads + V", "LOCKED 24xM32 | 4 LDS-DMA + V", "LOCKED 4 LDS-DMA + 24xM32 | V", "LOCKED 4 loads + 24xM32 | V",
  "LOCKED 4 LDS-DMA + 24xM32 | V, vmcnt(4) after V", "LOCKED 4 LDS-DMA (streaming source) + 24xM32 | V", "LOCKED 4 LDS-DMA (streaming) + 24xM32 | V, vmcnt(4)"};
// LOCKED: waves 0-3 run [matrix segment, barrier, vector segment, barrier], waves 4-7 [vector, barrier, matrix, barrier]: on every SIMD one
// wave is in its matrix segment while the other is in its vector segment (2 waves per SIMD only).

template <int MODE>
__global__ void __launch_bounds__(512) probe(unsigned long long* out, int iters, const char* src) {
  __shared__ char lds_sink[16384];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 ld[4] = {};
  if (iters < 0) lds_sink[threadIdx.x] = 1;
  const uint32_t s = blockIdx.x * 512u + threadIdx.x;
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(((s * 2654435761u + i * 40503u) >> 9 & 0xffff) * (1.0f / 32768.0f) - 1.0f); b[i] = (_Float16)(((s * 2246822519u + i * 3266489917u) >> 9 & 0xffff) * (1.0f / 32768.0f) - 1.0f); }
  f32x16 C[4]; f32x4 c[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) C[i] = f32x16{0};
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
  float x[8], y = 0.999f, z = 1e-3f;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 0.01f * (float)(threadIdx.x + i);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == K_M32) {
#pragma unroll
      for (int i = 0; i < 12; ++i) M32(C[i & 3]);
    } else if constexpr (MODE == K_M16) {
#pragma unroll
      for (int i = 0; i < 24; ++i) M16(c[i & 7]);
    } else if constexpr (MODE == K_M16DEP) {
#pragma unroll
      for (int i = 0; i < 24; ++i) M16(c[i / 3]);
    } else if constexpr (MODE == K_V) {
#pragma unroll
      for (int i = 0; i < 72; ++i) FMA(x[i & 7]);
    } else if constexpr (MODE == K_EV) {
#pragma unroll
      for (int i = 0; i < 72; ++i) { if (i % 9 < 4) EXP(x[i & 7]); else FMA(x[i & 7]); }
    } else if constexpr (MODE == K_M32_V) {
#pragma unroll
      for (int i = 0; i < 12; ++i) { M32(C[i & 3]);
#pragma unroll
        for (int j = 0; j < 6; ++j) FMA(x[(6 * i + j) & 7]); }
    } else if constexpr (MODE == K_M16_V) {
#pragma unroll
      for (int i = 0; i < 24; ++i) { M16(c[i & 7]);
#pragma unroll
        for (int j = 0; j < 3; ++j) FMA(x[(3 * i + j) & 7]); }
    } else if constexpr (MODE == K_M32_EV) {
#pragma unroll
      for (int i = 0; i < 12; ++i) { M32(C[i & 3]);
#pragma unroll
        for (int j = 0; j < 6; ++j) { const int k = 6 * i + j; if (k % 9 < 4) EXP(x[k & 7]); else FMA(x[k & 7]); } }
    } else if constexpr (MODE == K_M16_EV) {
#pragma unroll
      for (int i = 0; i < 24; ++i) { M16(c[i & 7]);
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int k = 3 * i + j; if (k % 9 < 4) EXP(x[k & 7]); else FMA(x[k & 7]); } }
    } else if constexpr (MODE == K_M32_THEN_V || MODE == K_M32_THEN_EV) {
#pragma unroll
      for (int i = 0; i < 12; ++i) M32(C[i & 3]);
#pragma unroll
      for (int i = 0; i < 72; ++i) { if (MODE == K_M32_THEN_EV && i % 9 < 4) EXP(x[i & 7]); else FMA(x[i & 7]); }
    } else if constexpr (MODE == K_M16_THEN_V || MODE == K_M16_THEN_EV) {
#pragma unroll
      for (int i = 0; i < 24; ++i) M16(c[i & 7]);
#pragma unroll
      for (int i = 0; i < 72; ++i) { if (MODE == K_M16_THEN_EV && i % 9 < 4) EXP(x[i & 7]); else FMA(x[i & 7]); }
    } else {
      constexpr int NV = (MODE == K_LOCK32_ATTN || MODE == K_FREE32_ATTN || MODE >= K_LOCK32_ATTN_VLD) ? 112 : 72;
      // staging variants: waves 4-7 (threadIdx >= 256) own the 4 x 1 KiB per wave of a step, issued in their V or in their M segment
      auto stage = [&]() __attribute__((always_inline)) {
        const char* g = src + ((size_t)blockIdx.x * 64 + ((MODE == K_MDMA_STREAM || MODE == K_MDMA_STREAM_WAIT) ? (it & 63) : (it & 15))) * 16384 + (threadIdx.x & 255) * 16;
        if constexpr (MODE == K_LOCK32_ATTN_VLD || MODE == K_LOCK32_ATTN_MLD) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { ld[i] = *reinterpret_cast<const u32x4*>(g + 4096 * i); asm volatile("" : "+v"(ld[i]) :: "memory"); }
        } else if constexpr (MODE == K_LOCK32_ATTN_VDMA || MODE >= K_LOCK32_ATTN_MDMA && MODE != K_LOCK32_ATTN_MLD) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g + 4096 * i), "s"((unsigned)__builtin_amdgcn_readfirstlane(((threadIdx.x >> 6) & 3) * 1024 + 4096 * i)) : "memory");
          }
        }
      };
      auto mseg = [&]() __attribute__((always_inline)) {
        if constexpr (MODE == K_LOCK16_EV) {
#pragma unroll
          for (int i = 0; i < 24; ++i) M16(c[i & 7]);
        } else {
#pragma unroll
          for (int i = 0; i < (NV == 112 ? 24 : 12); ++i) M32(C[i & 3]);
        }
      };
      auto vseg = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NV; ++i) { if (MODE != K_LOCK32_V && (NV == 112 ? i % 7 < 2 : i % 9 < 4)) EXP(x[i & 7]); else FMA(x[i & 7]); }
      };
      if constexpr (MODE == K_FREE32_ATTN) { mseg(); vseg(); }
      else if (threadIdx.x < 256) { mseg(); __builtin_amdgcn_s_barrier(); vseg(); __builtin_amdgcn_s_barrier(); }
      else {
        if constexpr (MODE == K_LOCK32_ATTN_VLD || MODE == K_LOCK32_ATTN_VDMA) stage();
        vseg();
        if constexpr (MODE == K_MDMA_WAIT || MODE == K_MDMA_STREAM_WAIT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (MODE == K_LOCK32_ATTN_MLD || MODE >= K_LOCK32_ATTN_MDMA) stage();
        mseg(); __builtin_amdgcn_s_barrier();
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) t += C[i][j];
#pragma unroll
  for (int i = 0; i < 8; ++i) t += c[i][0] + c[i][1] + c[i][2] + c[i][3] + x[i];
  t += (float)(ld[0][0] + ld[1][1] + ld[2][2] + ld[3][3]) + (float)lds_sink[(threadIdx.x * 7) & 16383];
  if (t == 12345.678f) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> void run_mode(unsigned long long* dout, int cus, int iters) {
  static char* src = nullptr;
  if (!src) { CK(hipMalloc(&src, (size_t)cus * 64 * 16384)); CK(hipMemset(src, 1, (size_t)cus * 64 * 16384)); }
  for (int wps = (MODE >= K_LOCK32_V && MODE != K_FREE32_ATTN) ? 2 : 1; wps <= 2; ++wps) {
    CK(hipMemset(dout, 0, (1 + cus * 8) * 8));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(256 * wps), 0, 0, dout, iters, src);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(1 + cus * 8);
    CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> v;
    for (int b = 0; b < cus; ++b) for (int w = 0; w < 4 * wps; ++w) v.push_back((double)h[1 + b * 8 + w] / iters);
    std::sort(v.begin(), v.end());
    printf("%-34s %d wave(s)/SIMD: %7.1f cycles per wave and body (p10 %.1f p90 %.1f) -> %7.1f per SIMD and body\n", NAMES[MODE], wps, v[v.size() / 2],
           v[v.size() / 10], v[v.size() * 9 / 10], v[v.size() / 2] / wps);
  }
}
template <int M> void run_all(unsigned long long* dout, int cus, int iters) {
  if constexpr (M < K_COUNT) { run_mode<M>(dout, cus, iters); run_all<M + 1>(dout, cus, iters); }
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned long long* dout; CK(hipMalloc(&dout, (1 + cus * 8) * 8));
  printf("device %s, %d CUs; MFMA floor of a body: 384 cycles\n", prop.gcnArchName, cus);
  run_all<0>(dout, cus, 2000);
  return 0;
}
