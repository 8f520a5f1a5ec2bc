// Shared device/host helpers for libpaths_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error convention (include/paths_hip.h): 0 = ok, negative = error, message in thread-local buffer
#define PATHS_OK 0
#define PATHS_EINVAL (-1)
#define PATHS_ELAUNCH (-2)
#define PATHS_EUNSUPPORTED (-3)

int paths_set_error(int code, const char* fmt, ...);

#define PATHS_REQUIRE(cond, ...)                                     \
  do {                                                               \
    if (!(cond)) return paths_set_error(PATHS_EINVAL, __VA_ARGS__);  \
  } while (0)

// A launch that carries the host thread's pending STOP EVENT, if one was set (paths_set_stop_event): the event is attached to the
// kernel's own completion signal (hipExtLaunchKernelGGL) instead of being recorded as a packet of its own behind it - an event record
// costs the launching queue ~8 us before its next kernel starts (profiles/r04_experiments.md, selection-queue timeline), and the
// recursion forks twice per level from its critical queue (tokens ready -> aggregator; top-K ready -> child expansion).
hipEvent_t paths_take_stop_event(void);      // runtime.hip: returns and clears this thread's pending event (nullptr: none)
#define PATHS_LAUNCH_STOP(kern, grid, block, lds, stream, ...)                                                  \
  do {                                                                                                          \
    hipEvent_t ev_ = paths_take_stop_event();                                                                   \
    if (ev_ != nullptr) hipExtLaunchKernelGGL(kern, grid, block, lds, stream, nullptr, ev_, 0, __VA_ARGS__);   \
    else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                       \
  } while (0)

#define PATHS_LAUNCH_CHECK(name)                                                              \
  do {                                                                                        \
    hipError_t e_ = hipGetLastError();                                                        \
    if (e_ != hipSuccess) return paths_set_error(PATHS_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

// Opt a kernel in to `bytes` of dynamic LDS (> the 64 KiB default), once per DEVICE (the attribute is per device: a process that
// drives several GPUs must set it on each), with the return code checked: a failure here would otherwise only surface as a generic
// launch error.  Use inside a function that returns the int error code.
#define PATHS_LDS_OPT_IN(kernel_ptr, bytes, name)                                                                              \
  do {                                                                                                                         \
    static unsigned char done_[64] = {0};                                                                                      \
    int dev_ = 0;                                                                                                              \
    if (hipGetDevice(&dev_) != hipSuccess) return paths_set_error(PATHS_ELAUNCH, "%s: hipGetDevice failed", name);            \
    if (dev_ < 0 || dev_ >= 64 || !done_[dev_]) {                                                                              \
      const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_ptr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
      if (e_ != hipSuccess) return paths_set_error(PATHS_ELAUNCH, "%s: hipFuncSetAttribute(%d bytes of LDS): %s", name, (int)(bytes), hipGetErrorString(e_)); \
      if (dev_ >= 0 && dev_ < 64) done_[dev_] = 1;                                                                             \
    }                                                                                                                          \
  } while (0)

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) - indices stay constants without relying on the unroller
// Residuals of a packed fp16 pair: ra = a - (float)h.lo, rb = b - (float)h.hi, one v_fma_mix_f32 each (the mixed-precision FMA reads
// the half straight out of the packed register; written as a - (float)h hipcc emits v_cvt_f32_f16 + v_sub_f32, and turns an
// fmaf(h, -1, a) back into that).  Exact either way: the same fp32 subtraction.
__device__ __forceinline__ void f16_pair_residuals(uint32_t h, float a, float b, float& ra, float& rb) {
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(h), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(h), "v"(b));
}

// The same residuals ROUNDED to fp16 and packed (the lo plane of the pair): v_fma_mixlo_f16 / v_fma_mixhi_f16 compute the fp32
// FMA and write its fp16 rounding (nearest even, as v_cvt_pk_f16_f32) into one half of the destination - two instructions where
// residuals + pack took three.
__device__ __forceinline__ uint32_t f16_pair_residuals_pk(uint32_t h, float a, float b) {
  uint32_t r;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(b));
  return r;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ---- device math (accurate forms: the selection chain must stay within ~1e-7 of the fp32 CPU path)
// 1/d for d in [1, 1e30]: hardware reciprocal (1 ulp) + one Newton step = correctly rounded in all but a few cases; 3
// instructions against the ~12 of an IEEE division (these run 256 times per lane in a 256x256 GEMM epilogue)
__device__ __forceinline__ float rcp_nr(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}
// e^x, <= 1 ulp: exp2 of x*log2(e) with the product carried in two floats (hi + lo), the integer part applied by ldexp.
// libm's expf is this plus range checks (2 compares + 2 selects per call) that the callers' clamps already make redundant:
// v_exp_f32 / v_ldexp_f32 saturate to +inf and 0 by themselves.
__device__ __forceinline__ float exp_acc(float x) {
  const float l2e_hi = 1.44269502162933349609375f, l2e_lo = 1.925963033500011306e-8f;   // log2(e) = hi + lo
  const float ph = x * l2e_hi;
  const float pl = fmaf(x, l2e_lo, fmaf(x, l2e_hi, -ph));
  const float e = rintf(ph);
  const float r = __builtin_amdgcn_exp2f((ph - e) + pl);
  return __builtin_amdgcn_ldexpf(r, (int)e);
}
// the clamp keeps d finite (exp overflow would turn the Newton step into inf*0); sigmoid(-69) = 1e-30 either way
__device__ __forceinline__ float sigmoid_acc(float x) { return rcp_nr(1.0f + fminf(exp_acc(-x), 1e30f)); }
// branch-free tanh: 1 - 2/(1+e^{2x}); absolute error ~1e-7 (what the h = o*tanh(.) and c-update chains need),
// saturates correctly at +-1 (e^{2x} -> inf / 0).  libm's tanhf branches per lane and serialises epilogues.
__device__ __forceinline__ float tanh_acc(float x) { return 1.0f - 2.0f * rcp_nr(1.0f + fminf(exp_acc(2.0f * x), 1e30f)); }

// ---- MFMA wrappers.  f32-input MFMA = exact k-ordered fp32 FMA chain (guide §3 "FP32-input MFMA").
// 32x32x2: A lane l -> A[l&31][l>>5], B lane l -> B[l>>5][l&31]; C: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// 16x16x4: A lane l -> A[l&15][l>>4], B lane l -> B[l>>4][l&15]; C: col=l&15, row=4*(l>>4)+r
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// 16-byte load through an explicit GLOBAL address-space pointer.  A pointer obtained by selecting between two kernel
// arguments is "generic" to hipcc, which then emits flat_load (counted by BOTH vmcnt and lgkmcnt -> forces full drains).
__device__ __forceinline__ f32x4 ldg_f32x4(const float* p) {
  typedef const f32x4 __attribute__((address_space(1))) * gptr;
  return *reinterpret_cast<gptr>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ int c32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Is [m0, m0+rows) entirely padding?  Rows are laid out [slide][rows_per_slide]; slide b has
// num_ims[b] valid rows at its start.  num_ims == nullptr disables skipping.
// extra: valid rows per slide beyond num_ims[b] (1 for token-ordered rows: the special token's slot comes first).
__device__ __forceinline__ bool block_all_padding(const int64_t* num_ims, int rows_per_slide, int m0, int rows, int M, int extra = 0) {
  if (num_ims == nullptr) return false;
  int last = min(m0 + rows, M) - 1;
  int b0 = m0 / rows_per_slide, b1 = last / rows_per_slide;
  for (int b = b0; b <= b1; ++b) {
    int lo = max(m0, b * rows_per_slide);
    if (lo - b * rows_per_slide < (int)num_ims[b] + extra) return false;
  }
  return true;
}
