// Training-path attention of the LAST decoder layer, which is read at token 0 only (reference model/aggregator.py:70-75: agg =
// out[:, 0]): one query per (slide, head) against all keys, forward (+ log-sum-exp, + dropout on the probabilities as
// nn.MultiheadAttention applies it) and backward.  O(T d) work: the keys are split over S workgroups per (slide, head) so that the
// launch fills the chip instead of running B*H workgroups over 2,049 keys each (the round-2 kernels: 80-90 us per launch, 15
// launches per training step); every key is independent except for the softmax statistics and dq, which go through per-split
// partials and a fixed-order combine (deterministic: same seed -> bit-identical gradients).
//
// Conventions of attn_f32.hip / attn_bwd.hip: q, k, v head-major [B, H, T, 32], q PRE-SCALED by log2(e)/sqrt(32), lse in the log2
// domain, dropout mask element ((slide*H + head)*T + 0)*T + key of the site (csrc/dropout.h).
#include "common.h"
#include "dropout.h"

DropSite paths_make_drop_site(uint64_t key, float p);      // dropout.hip

namespace {

constexpr int HD = 32;
constexpr int S_MAX = 16;
constexpr int PSTRIDE = 36;       // floats per partial record: m, l, -, -, o[32]
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ void split_range(int len, int S, int part, int& k0, int& k1) {
  const int chunk = ((len + S - 1) / S + 63) & ~63;
  k0 = min(len, part * chunk);
  k1 = min(len, k0 + chunk);
}

// forward partials: workgroup (split, head, slide), 256 threads; thread t owns keys k0 + t, k0 + t + 256, ...
__global__ void __launch_bounds__(256)
token0_fwd_partial_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                          const int64_t* __restrict__ num_ims, float* __restrict__ partials, int T, int H, int S, DropSite drop) {
  __shared__ float sP[1024];                 // p * mask of the chunk's keys (chunk <= 1024: T <= 16384 at S = 16)
  __shared__ float red[8][HD + 1];
  __shared__ float sred[4];
  const int part = blockIdx.x, head = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = min((int)num_ims[b] + 1, T);
  const DropWin dwin = drop_window(drop, drop_attn_row((uint64_t)b * H + head, T, 0));       // (this pair's T x T' mask elements: csrc/dropout.h)
  int k0, k1;
  split_range(len, S, part, k0, k1);
  const int64_t base = ((int64_t)b * H + head) * T * HD;
  float* dst = partials + (((int64_t)b * H + head) * S + part) * PSTRIDE;
  if (k0 >= k1) {                            // empty split (short slide): neutral record
    if (tid < PSTRIDE) dst[tid] = tid == 0 ? -INFINITY : 0.f;
    return;
  }
  f32x4 qv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) qv[i] = *reinterpret_cast<const f32x4*>(q + base + 4 * i);
  float sc[4];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = k0 + tid + 256 * j;
    sc[j] = -INFINITY;
    if (key < k1) {
      const float* kp = k + base + (int64_t)key * HD;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 kk = *reinterpret_cast<const f32x4*>(kp + 4 * i);
        s += (kk[0] * qv[i][0] + kk[1] * qv[i][1]) + (kk[2] * qv[i][2] + kk[3] * qv[i][3]);
      }
      sc[j] = s;
      m = fmaxf(m, s);
    }
  }
  m = wave_max(m);
  if (lane == 0) sred[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
  __syncthreads();
  float l = 0.f;
  const uint64_t row = drop_attn_row((uint64_t)b * H + head, T, 0);          // query 0
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = k0 + tid + 256 * j;
    if (key < k1) {
      const float p = __builtin_amdgcn_exp2f(sc[j] - m);
      l += p;                                                      // the normaliser is the un-dropped softmax's
      sP[tid + 256 * j] = drop.thr != 0u ? p * drop_mult_w(drop, dwin, row + (uint64_t)key) : p;
    }
  }
  l = wave_sum(l);
  if (lane == 0) sred[wave] = l;
  __syncthreads();
  l = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  // o[c] = sum_key p[key] v[key][c]: thread (c = tid & 31, g = tid >> 5) walks keys g, g + 8, ... (coalesced 128-byte rows)
  const int c = tid & 31, g = tid >> 5;
  float o = 0.f;
  for (int kk = g; kk < k1 - k0; kk += 8) o = fmaf(sP[kk], v[base + (int64_t)(k0 + kk) * HD + c], o);
  red[g][c] = o;
  __syncthreads();
  if (tid < HD) {
    float t = 0.f;
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) t += red[gg][tid];
    dst[4 + tid] = t;
  }
  if (tid == 0) { dst[0] = m; dst[1] = l; }
}

// combine: one wave per (slide, head): lanes 0..31 own output channel c
__global__ void __launch_bounds__(64)
token0_fwd_combine_kernel(const float* __restrict__ partials, float* __restrict__ a0 /*[B, H*32]*/, float* __restrict__ lse0 /*[B,H]*/,
                          int H, int S) {
  const int head = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const float* sp = partials + ((int64_t)b * H + head) * S * PSTRIDE;
  float M = -INFINITY;
  for (int pt = 0; pt < S; ++pt) M = fmaxf(M, sp[pt * PSTRIDE]);
  float num = 0.f, den = 0.f;
  for (int pt = 0; pt < S; ++pt) {
    const float mp = sp[pt * PSTRIDE];
    const float w = mp == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mp - M);
    den += w * sp[pt * PSTRIDE + 1];
    if (lane < HD) num += w * sp[pt * PSTRIDE + 4 + lane];
  }
  if (lane < HD) a0[((int64_t)b * H + head) * HD + lane] = num / den;
  if (lane == 0) lse0[(int64_t)b * H + head] = M + __builtin_amdgcn_logf(den);
}

// backward: workgroup (split, head, slide); thread per key: dk, dv rows written in place, dq partial per split
__global__ void __launch_bounds__(256)
token0_bwd_partial_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                          const float* __restrict__ a0, const float* __restrict__ da0, const float* __restrict__ lse0,
                          const int64_t* __restrict__ num_ims, float* __restrict__ dqkv, float* __restrict__ dq_part, int T, int H, int S,
                          DropSite drop) {
  __shared__ float red[4][HD];
  const int part = blockIdx.x, head = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = min((int)num_ims[b] + 1, T);
  const DropWin dwin = drop_window(drop, drop_attn_row((uint64_t)b * H + head, T, 0));       // (this pair's T x T' mask elements: csrc/dropout.h)
  int k0, k1;
  split_range(len, S, part, k0, k1);
  const int64_t base = ((int64_t)b * H + head) * T * HD;
  f32x4 qv[8], gv[8];
  float dsum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    qv[i] = *reinterpret_cast<const f32x4*>(q + base + 4 * i);
    gv[i] = *reinterpret_cast<const f32x4*>(da0 + ((int64_t)b * H + head) * HD + 4 * i);
    const f32x4 av = *reinterpret_cast<const f32x4*>(a0 + ((int64_t)b * H + head) * HD + 4 * i);
    dsum += (gv[i][0] * av[0] + gv[i][1] * av[1]) + (gv[i][2] * av[2] + gv[i][3] * av[3]);
  }
  const float L = lse0[(int64_t)b * H + head];
  const uint64_t row = drop_attn_row((uint64_t)b * H + head, T, 0);
  f32x4 dq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int key = k0 + tid; key < k1; key += 256) {
    const float* kp = k + base + (int64_t)key * HD;
    const float* vp = v + base + (int64_t)key * HD;
    f32x4 kk[8], vv[8];
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      kk[i] = *reinterpret_cast<const f32x4*>(kp + 4 * i);
      vv[i] = *reinterpret_cast<const f32x4*>(vp + 4 * i);
      s += (kk[i][0] * qv[i][0] + kk[i][1] * qv[i][1]) + (kk[i][2] * qv[i][2] + kk[i][3] * qv[i][3]);
      dp += (vv[i][0] * gv[i][0] + vv[i][1] * gv[i][1]) + (vv[i][2] * gv[i][2] + vv[i][3] * gv[i][3]);
    }
    const float p = __builtin_amdgcn_exp2f(s - L);
    const float mk = drop.thr != 0u ? drop_mult_w(drop, dwin, row + (uint64_t)key) : 1.f;
    const float pd = p * mk;
    const float ds = LN2 * p * (dp * mk - dsum);
    float* dkp = dqkv + ((int64_t)b * T + key) * (3 * H * HD) + H * HD + head * HD;
    float* dvp = dkp + H * HD;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      *reinterpret_cast<f32x4*>(dkp + 4 * i) = qv[i] * ds;
      *reinterpret_cast<f32x4*>(dvp + 4 * i) = gv[i] * pd;
      dq[i] += kk[i] * ds;
    }
  }
  // dq partial of this split: wave sums, then the four waves in a fixed order
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = wave_sum(dq[i][e]);
      if (lane == 0) red[wave][4 * i + e] = t;
    }
  __syncthreads();
  if (tid < HD) dq_part[(((int64_t)b * H + head) * S + part) * HD + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

__global__ void __launch_bounds__(64)
token0_bwd_combine_kernel(const float* __restrict__ dq_part, float* __restrict__ dqkv, int T, int H, int S) {
  const int head = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  if (lane >= HD) return;
  const float* sp = dq_part + ((int64_t)b * H + head) * S * HD;
  float t = 0.f;
  for (int pt = 0; pt < S; ++pt) t += sp[pt * HD + lane];
  dqkv[((int64_t)b * T) * (3 * H * HD) + head * HD + lane] = t;      // row 0 of slide b, q block
}

// ---- the same single-query attention for any head_dim (multiple of 4, <= 64) on the token-major in_proj output qkv [B*T, 3d]
// (inference of the shape-generic path: no dropout, no lse): partials (m, l, o[HD]) per key split, then the combine
template <int HD>
__global__ void __launch_bounds__(256)
token0_any_partial_kernel(const float* __restrict__ qkv, int64_t ld, int d, float qscale, const int64_t* __restrict__ num_ims,
                          float* __restrict__ partials, int T, int H, int S) {
  constexpr int PS = HD + 4;
  __shared__ float sP[1024];
  __shared__ float red[8][HD + 1];
  __shared__ float sred[4];
  const int part = blockIdx.x, head = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = min((int)num_ims[b] + 1, T);
  int k0, k1;
  split_range(len, S, part, k0, k1);
  const float* qp = qkv + (int64_t)b * T * ld + head * HD;          // token 0 of slide b
  const float* kb = qp + d;
  const float* vb = qp + 2 * d;
  float* dst = partials + (((int64_t)b * H + head) * S + part) * PS;
  if (k0 >= k1) {
    if (tid < PS) dst[tid] = tid == 0 ? -INFINITY : 0.f;
    return;
  }
  f32x4 qv[HD / 4];
#pragma unroll
  for (int i = 0; i < HD / 4; ++i) qv[i] = *reinterpret_cast<const f32x4*>(qp + 4 * i) * qscale;
  float sc[4];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = k0 + tid + 256 * j;
    sc[j] = -INFINITY;
    if (key < k1) {
      const float* kp = kb + (int64_t)key * ld;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < HD / 4; ++i) {
        const f32x4 kk = *reinterpret_cast<const f32x4*>(kp + 4 * i);
        s += (kk[0] * qv[i][0] + kk[1] * qv[i][1]) + (kk[2] * qv[i][2] + kk[3] * qv[i][3]);
      }
      sc[j] = s;
      m = fmaxf(m, s);
    }
  }
  m = wave_max(m);
  if (lane == 0) sred[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
  __syncthreads();
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = k0 + tid + 256 * j;
    if (key < k1) {
      const float p = __builtin_amdgcn_exp2f(sc[j] - m);
      l += p;
      sP[tid + 256 * j] = p;
    }
  }
  l = wave_sum(l);
  if (lane == 0) sred[wave] = l;
  __syncthreads();
  l = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  // o[c] = sum_key p[key] v[key][c]: thread (c = tid % 64, g = tid / 64) walks keys g, g + 4, ... (c < HD active)
  const int c = tid & 63, g = tid >> 6;
  float o = 0.f;
  if (c < HD)
    for (int kk = g; kk < k1 - k0; kk += 4) o = fmaf(sP[kk], vb[(int64_t)(k0 + kk) * ld + c], o);
  if (c < HD) red[g][c] = o;
  __syncthreads();
  if (tid < HD) dst[4 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  if (tid == 0) { dst[0] = m; dst[1] = l; }
}

template <int HD>
__global__ void __launch_bounds__(64)
token0_any_combine_kernel(const float* __restrict__ partials, float* __restrict__ a0 /*[B, H*HD]*/, int H, int S) {
  constexpr int PS = HD + 4;
  const int head = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const float* sp = partials + ((int64_t)b * H + head) * S * PS;
  float M = -INFINITY;
  for (int pt = 0; pt < S; ++pt) M = fmaxf(M, sp[pt * PS]);
  float num = 0.f, den = 0.f;
  for (int pt = 0; pt < S; ++pt) {
    const float mp = sp[pt * PS];
    const float w = mp == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mp - M);
    den += w * sp[pt * PS + 1];
    if (lane < HD) num += w * sp[pt * PS + 4 + lane];
  }
  if (lane < HD) a0[((int64_t)b * H + head) * HD + lane] = num / den;
}

template <int HD>
int launch_token0_any(const float* qkv, int64_t ld, int d, float qscale, const int64_t* num_ims, float* a0, float* ws, int B, int T, int H, int S,
                      hipStream_t stream) {
  hipLaunchKernelGGL(token0_any_partial_kernel<HD>, dim3(S, H, B), dim3(256), 0, stream, qkv, ld, d, qscale, num_ims, ws, T, H, S);
  PATHS_LAUNCH_CHECK("attention_token0_any(partials)");
  hipLaunchKernelGGL(token0_any_combine_kernel<HD>, dim3(H, B), dim3(64), 0, stream, ws, a0, H, S);
  PATHS_LAUNCH_CHECK("attention_token0_any(combine)");
  return PATHS_OK;
}

int pick_splits(int B, int T, int H) {
  int S = 1;
  while (S < S_MAX && B * H * S < 512 && (T + 2 * S - 1) / (2 * S) >= 64) S *= 2;
  while ((((T + S - 1) / S + 63) & ~63) > 1024) S *= 2;          // the probability chunk lives in 4 KiB of LDS
  return S;
}

}  // namespace

extern "C" {

// floats of scratch for the two entry points below (split partials)
int64_t paths_attention_token0_workspace(int B, int T, int H) { return (int64_t)B * H * 64 * 68; }     // (68 >= PSTRIDE, head_dim 64 + 4)

// a0 [B, H*hd] = attention output of token 0 per head for any head_dim in {16, 32, 48, 64} on the token-major in_proj output qkv
// [B*T, 3d] (row stride ld, q unscaled: qscale = log2(e) / sqrt(hd)); inference form (no dropout); ws as above
int paths_attention_token0_any(const float* qkv, int64_t ld, const int64_t* num_ims, float* a0, float* ws, int B, int T, int H, int head_dim,
                               float qscale, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && T > 0 && H > 0 && qkv && num_ims && a0 && ws && ld % 4 == 0 && (uintptr_t)qkv % 16 == 0, "attention_token0_any: bad arguments");
  const int S = pick_splits(B, T, H);
  PATHS_REQUIRE(S <= 64 && (((T + S - 1) / S + 63) & ~63) <= 1024, "attention_token0_any: T = %d is too long", T);
  const int d = H * head_dim;
  switch (head_dim) {
    case 16: return launch_token0_any<16>(qkv, ld, d, qscale, num_ims, a0, ws, B, T, H, S, stream);
    case 32: return launch_token0_any<32>(qkv, ld, d, qscale, num_ims, a0, ws, B, T, H, S, stream);
    case 48: return launch_token0_any<48>(qkv, ld, d, qscale, num_ims, a0, ws, B, T, H, S, stream);
    case 64: return launch_token0_any<64>(qkv, ld, d, qscale, num_ims, a0, ws, B, T, H, S, stream);
    default: return paths_set_error(PATHS_EUNSUPPORTED, "attention_token0_any: head_dim %d (supported: 16, 32, 48, 64)", head_dim);
  }
}

// a0 [B, H*32] = token-0 attention output per head (dropout p on the probabilities, site drop_key), lse0 [B, H] (log2 domain,
// un-dropped softmax); q, k, v head-major [B, H, T, 32] with q pre-scaled (as paths_token_layer_f32 writes them)
int paths_attention_token0_fwd(const float* q, const float* k, const float* v, const int64_t* num_ims, float* a0, float* lse0, float* ws,
                               int B, int T, int H, int head_dim, uint64_t drop_key, float drop_p, hipStream_t stream) {
  PATHS_REQUIRE(head_dim == HD && B > 0 && T > 0 && H > 0 && q && k && v && num_ims && a0 && lse0 && ws, "attention_token0_fwd: bad arguments (head_dim must be 32)");
  PATHS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention_token0_fwd: p must be in [0, 1)");
  const int S = pick_splits(B, T, H);
  PATHS_REQUIRE(S <= 64 && (((T + S - 1) / S + 63) & ~63) <= 1024, "attention_token0_fwd: T = %d is too long", T);
  const DropSite site = paths_make_drop_site(drop_key, drop_p);
  hipLaunchKernelGGL(token0_fwd_partial_kernel, dim3(S, H, B), dim3(256), 0, stream, q, k, v, num_ims, ws, T, H, S, site);
  PATHS_LAUNCH_CHECK("attention_token0_fwd(partials)");
  hipLaunchKernelGGL(token0_fwd_combine_kernel, dim3(H, B), dim3(64), 0, stream, ws, a0, lse0, H, S);
  PATHS_LAUNCH_CHECK("attention_token0_fwd(combine)");
  return PATHS_OK;
}

// dqkv [B, T, 3*H*32] (ZERO on entry: only the dk / dv rows of valid keys and dq of row 0 are written) from a0, da0 [B, H*32], lse0
int paths_attention_token0_bwd(const float* q, const float* k, const float* v, const float* a0, const float* da0, const float* lse0,
                                const int64_t* num_ims, float* dqkv, float* ws, int B, int T, int H, int head_dim, uint64_t drop_key,
                                float drop_p, hipStream_t stream) {
  PATHS_REQUIRE(head_dim == HD && B > 0 && T > 0 && H > 0 && q && k && v && a0 && da0 && lse0 && num_ims && dqkv && ws,
                "attention_token0_bwd: bad arguments (head_dim must be 32)");
  PATHS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention_token0_bwd: p must be in [0, 1)");
  const int S = pick_splits(B, T, H);
  const DropSite site = paths_make_drop_site(drop_key, drop_p);
  hipLaunchKernelGGL(token0_bwd_partial_kernel, dim3(S, H, B), dim3(256), 0, stream, q, k, v, a0, da0, lse0, num_ims, dqkv, ws, T, H, S, site);
  PATHS_LAUNCH_CHECK("attention_token0_bwd(partials)");
  hipLaunchKernelGGL(token0_bwd_combine_kernel, dim3(H, B), dim3(64), 0, stream, ws, dqkv, T, H, S);
  PATHS_LAUNCH_CHECK("attention_token0_bwd(combine)");
  return PATHS_OK;
}

}  // extern "C"
