// Fused masked self-attention (flash style, fp32-exact MFMA) for the PATHS aggregator.
//
// Replaces nn.MultiheadAttention's self-attention inside nn.TransformerDecoderLayer as called at
// reference model/aggregator.py:70-72 with tgt_key_padding_mask = arange(T) >= num_ims+1
// (reference utils.py:97-103): softmax(q k^T / sqrt(hd)) v per head, never materialising T x T.
//
// Layout: q,k,v are head-major [B][h][T][32] fp32 (written by the token-layer kernel); q is pre-scaled
// by log2(e)/sqrt(hd) so the softmax runs on exp2.  Output o is token-major [B][T][h*32].
//
// Mapping (MI355X): one wave = 16 queries of one (slide, head); a 4-wave workgroup shares 64-key K/V
// tiles through LDS (double-buffered).  The kernel computes the TRANSPOSED products
//     S^T[key][q] = K Q^T         (A = K tile from LDS, B = Q^T held in registers)
//     O^T[dv][q] += V^T P^T       (A = V tile from LDS, B = P^T = the S^T accumulator registers themselves)
// with v_mfma_f32_16x16x4_f32, so (a) each lane owns ONE query: running max / sum / rescale are per-lane
// scalars and the key reduction is 15 in-lane ops + 2 cross-lane swaps per 64 keys, and (b) P never
// leaves the register file: the C layout of S^T (lane group g holds keys 4g..4g+3) is consumed directly
// as the B operand of the PV product by reading V rows in that same key order.
#include "common.h"

namespace {

constexpr int HD = 32;       // head_dim (trans_dim 128 / 4 heads)
constexpr int KT = 64;       // keys per LDS tile
constexpr int LDKs = 40;     // K tile row stride (floats): conflict-free b128 reads for the 16x16x4 A operand
constexpr int LDVs = 36;     // V tile row stride: conflict-free b32 reads

__global__ void __launch_bounds__(256)
attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                float* __restrict__ o, float* __restrict__ lse, const int64_t* __restrict__ num_ims, int T, int H) {
  __shared__ __attribute__((aligned(16))) float sK[2][KT * LDKs];
  __shared__ __attribute__((aligned(16))) float sV[2][KT * LDVs];

  const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 64;
  const int len = (int)num_ims[b] + 1;          // valid keys = special token + patches
  if (q0 >= len) return;                        // every query of this block is padding
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 15, g4 = lane >> 4;
  const int64_t base = ((int64_t)b * H + head) * T * HD;

  // Q^T operand: lane (q, g) holds Q[q][16u + 4g + e], u = 0..1, e = 0..3
  const int qrow = min(q0 + wave * 16 + ql, T - 1);
  f32x4 qreg[2];
  qreg[0] = *reinterpret_cast<const f32x4*>(q + base + (int64_t)qrow * HD + 4 * g4);
  qreg[1] = *reinterpret_cast<const f32x4*>(q + base + (int64_t)qrow * HD + 16 + 4 * g4);

  f32x4 oacc[2];
  oacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  oacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int nkt = (len + KT - 1) / KT;
  // staging: 64 keys x 8 float4 per matrix = 512 float4 -> 2 per thread per matrix
  f32x4 rk[2], rv[2];
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = tid + p * 256, row = idx >> 3, c4 = idx & 7;
      const int key = kt * KT + row;
      if (key < len) {
        rk[p] = *reinterpret_cast<const f32x4*>(k + base + (int64_t)key * HD + 4 * c4);
        rv[p] = *reinterpret_cast<const f32x4*>(v + base + (int64_t)key * HD + 4 * c4);
      } else {                                  // masked keys: K irrelevant (score forced to -inf), V must be 0
        rk[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        rv[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = tid + p * 256, row = idx >> 3, c4 = idx & 7;
      *reinterpret_cast<f32x4*>(&sK[buf][row * LDKs + 4 * c4]) = rk[p];
      *reinterpret_cast<f32x4*>(&sV[buf][row * LDVs + 4 * c4]) = rv[p];
    }
  };

  gload(0);
  swrite(0);
  __syncthreads();
  int buf = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) gload(kt + 1);

    // ---- S^T = K Q^T for 4 sub-tiles of 16 keys
    f32x4 s[4], ka[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* kp = &sK[buf][(16 * t + ql) * LDKs + 4 * g4];
      ka[t][0] = *reinterpret_cast<const f32x4*>(kp);
      ka[t][1] = *reinterpret_cast<const f32x4*>(kp + 16);
    }
    // 4 independent accumulators interleaved: back-to-back MFMAs never wait on their own result
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) s[t] = mfma16(ka[t][u][e], qreg[u][e], s[t]);
    // ---- mask (last tile only) + online softmax (lane: one query; keys 16t + 4g + r)
    if (kt == nkt - 1) {
      const int kbase = kt * KT + 4 * g4;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kbase + 16 * t + r >= len) s[t][r] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);       // finite: key 0 (special token) is always valid
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // raw v_exp_f32: arguments are <= 0, no denormal fix-up needed
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] = __builtin_amdgcn_exp2f(s[t][r] - m_new);
        psum += s[t][r];
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    oacc[0] *= alpha;
    oacc[1] *= alpha;
    // ---- O^T += V^T P^T
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* vp = &sV[buf][(16 * t + 4 * g4) * LDVs + ql];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        oacc[0] = mfma16(vp[r * LDVs], s[t][r], oacc[0]);
        oacc[1] = mfma16(vp[r * LDVs + 16], s[t][r], oacc[1]);
      }
    }
    if (kt + 1 < nkt) swrite(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // l: sum the 4 lane groups' partial sums
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_run;
  const int qi = q0 + wave * 16 + ql;
  if (qi < T) {
    float* op = o + ((int64_t)b * T + qi) * (H * HD) + head * HD + 4 * g4;
    *reinterpret_cast<f32x4*>(op) = oacc[0] * inv;
    *reinterpret_cast<f32x4*>(op + 16) = oacc[1] * inv;
    // training: log2-domain log-sum-exp of the (pre-scaled) scores, P = exp2(s - lse) in the backward kernels
    if (lse && g4 == 0) lse[((int64_t)b * H + head) * T + qi] = m_run + log2f(l_run);
  }
}

}  // namespace

extern "C" int paths_attention_f32(const float* q, const float* k, const float* v, float* o,
                                   float* lse /*[B,H,T] or null*/, const int64_t* num_ims, int B, int T, int H, int head_dim,
                                   int max_queries, hipStream_t stream) {
  PATHS_REQUIRE(head_dim == HD, "attention: head_dim must be %d (got %d)", HD, head_dim);
  PATHS_REQUIRE(B > 0 && T > 0 && H > 0 && num_ims != nullptr, "attention: bad shape B=%d T=%d H=%d", B, T, H);
  PATHS_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16 == 0, "attention: buffers must be 16-byte aligned");
  // max_queries > 0: only queries [0, max_queries) are needed (last decoder layer: token 0 only, aggregator.py:75)
  const int nq = max_queries > 0 && max_queries < T ? max_queries : T;
  dim3 grid((nq + 63) / 64, H, B);
  hipLaunchKernelGGL(attn_f32_kernel, grid, dim3(256), 0, stream, q, k, v, o, lse, num_ims, T, H);
  PATHS_LAUNCH_CHECK("attention");
  return PATHS_OK;
}
