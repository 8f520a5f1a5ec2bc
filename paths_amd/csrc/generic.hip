// Shape-generic kernels of the aggregator / importance path: any trans_dim (multiple of 32), trans_heads with head_dim in
// {16, 32, 48, 64} and importance_mlp_hidden_dim (reference config.py:30-36: the dataclass defaults are trans_dim 192 / 4 heads =
// head_dim 48; the shipped configs use 128 / 4 / 128, which run on the specialised kernels of tlayer_ws.hip / attn_x6.hip /
// token0_ws.hip / gemm_epi.h instead).  Together with the generic GEMM entry points (paths_gemm_nt_f32: exact fp32 MFMA) they
// evaluate the same reference code: model/paths.py:95-98,119-139, model/aggregator.py:37-76, utils.py:16-23,47-67,97-115.
// Exact fp32 arithmetic throughout (f32-input MFMA = a k-ordered fp32 FMA chain); written for correctness and reasonable speed,
// not tuned like the 128-wide path.
#include "common.h"
#include "dropout.h"

DropSite paths_make_drop_site(uint64_t key, float p);      // dropout.hip

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- fused masked self-attention, fp32 MFMA (the scheme of attn_f32.hip), templated on the head dim; q, k, v are read in place
// from the token-major in_proj output [B*T, 3d] (q | k | v blocks of d columns, head h at columns h*HD), q scaled here.
// TRAIN (paths_attention_any_train): also writes the log2-domain log-sum-exp of the un-dropped softmax and multiplies the probabilities
// that enter the PV product by the regenerated dropout mask / (1 - p) (element ((slide*H + head)*T + query)*T + key, csrc/dropout.h).
constexpr int KT = 64;
template <int HD, bool TRAIN>
__global__ void __launch_bounds__(256)
attn_any_kernel(const float* __restrict__ qkv, int64_t ld, int d, float qscale, float* __restrict__ o,
                const int64_t* __restrict__ num_ims, int T, int H, float* __restrict__ lse, DropSite drop) {
  constexpr int NU = HD / 16, LDKs = HD + 8, LDVs = HD + 4, C4 = HD / 4;
  __shared__ __attribute__((aligned(16))) float sK[2][KT * LDKs];
  __shared__ __attribute__((aligned(16))) float sV[2][KT * LDVs];
  const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 64;
  const int len = min((int)num_ims[b] + 1, T);
  const DropWin dwin = drop_window(drop, drop_attn_row((uint64_t)b * H + head, T, 0));       // (this pair's T x T' mask elements: csrc/dropout.h)
  if (q0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ql = lane & 15, g4 = lane >> 4;
  const float* qb = qkv + (int64_t)b * T * ld + head * HD;
  const float* kb = qb + d;
  const float* vb = qb + 2 * d;
  const int qrow = min(q0 + wave * 16 + ql, T - 1);
  f32x4 qreg[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) qreg[u] = *reinterpret_cast<const f32x4*>(qb + (int64_t)qrow * ld + 16 * u + 4 * g4) * qscale;
  f32x4 oacc[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) oacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int nkt = (len + KT - 1) / KT;
  constexpr int NP = (KT * C4 + 255) / 256;            // float4 pieces per thread and matrix
  f32x4 rk[NP], rv[NP];
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int idx = tid + p * 256, row = idx / C4, c4 = idx % C4;
      const int key = kt * KT + row;
      if (idx < KT * C4 && key < len) {
        rk[p] = *reinterpret_cast<const f32x4*>(kb + (int64_t)key * ld + 4 * c4);
        rv[p] = *reinterpret_cast<const f32x4*>(vb + (int64_t)key * ld + 4 * c4);
      } else {                                         // masked keys: K irrelevant (score forced to -inf), V must be 0
        rk[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        rv[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int idx = tid + p * 256, row = idx / C4, c4 = idx % C4;
      if (idx < KT * C4) {
        *reinterpret_cast<f32x4*>(&sK[buf][row * LDKs + 4 * c4]) = rk[p];
        *reinterpret_cast<f32x4*>(&sV[buf][row * LDVs + 4 * c4]) = rv[p];
      }
    }
  };
  gload(0);
  swrite(0);
  __syncthreads();
  int buf = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) gload(kt + 1);
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      f32x4 ka[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) ka[t] = *reinterpret_cast<const f32x4*>(&sK[buf][(16 * t + ql) * LDKs + 16 * u + 4 * g4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) s[t] = mfma16(ka[t][e], qreg[u][e], s[t]);
    }
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
    const float m_new = fmaxf(m_run, mx);              // finite: key 0 (special token) is always valid
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
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
    if constexpr (TRAIN) {
      if (drop.thr != 0u) {
        const uint64_t rowi = drop_attn_row((uint64_t)b * H + head, T, min(q0 + wave * 16 + ql, T - 1)) + (uint64_t)(kt * KT + 4 * g4);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {              // keys 4 g4 + r, + 1: one hash per pair (dropout.h)
            float m0, m1;
            drop_mult2_w(drop, dwin, rowi + (uint64_t)(16 * t + r), m0, m1);
            s[t][r] *= m0; s[t][r + 1] *= m1;
          }
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) oacc[u] *= alpha;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* vp = &sV[buf][(16 * t + 4 * g4) * LDVs + ql];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < NU; ++u) oacc[u] = mfma16(vp[r * LDVs + 16 * u], s[t][r], oacc[u]);
    }
    if (kt + 1 < nkt) swrite(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_run;
  const int qi = q0 + wave * 16 + ql;
  if (qi < T) {
    float* op = o + ((int64_t)b * T + qi) * d + head * HD + 4 * g4;
#pragma unroll
    for (int u = 0; u < NU; ++u) *reinterpret_cast<f32x4*>(op + 16 * u) = oacc[u] * inv;
    if constexpr (TRAIN) {
      if (lse != nullptr && g4 == 0) lse[((int64_t)b * H + head) * T + qi] = m_run + __builtin_amdgcn_logf(l_run);
    }
  }
}

// ---- y[row] = LayerNorm(x[row] (+ add)) * gamma + beta over d <= 2048 features, one wave per row (two-pass statistics in registers)
__global__ void __launch_bounds__(256)
layernorm_rows_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ add, const float* __restrict__ g,
                      const float* __restrict__ bta, float* __restrict__ y, int64_t ldy, int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = 4 * lane + 256 * i;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < d) {
      v[i] = *reinterpret_cast<const f32x4*>(x + row * ldx + c);
      if (add) v[i] += *reinterpret_cast<const f32x4*>(add + c);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (4 * lane + 256 * i < d) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float c = v[i][e] - mean; q += c * c; }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = 4 * lane + 256 * i;
    if (c < d) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c), bb = *reinterpret_cast<const f32x4*>(bta + c);
      f32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
      *reinterpret_cast<f32x4*>(y + row * ldy + c) = out;
    }
  }
}

// ---- y[row] = LayerNorm2(LayerNorm1(x[row]) + add) for d <= 2048: the post-attention norm1 -> (+ cross-attention bias) -> norm2 pair of a
// decoder layer over an empty memory (reference model/aggregator.py:25-33) in one pass, the intermediate row stays in registers
__global__ void __launch_bounds__(256)
layernorm2_rows_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ g1, const float* __restrict__ b1,
                       const float* __restrict__ add, const float* __restrict__ g2, const float* __restrict__ b2, float* __restrict__ y,
                       int64_t ldy, int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[8];
  const float inv_d = 1.0f / (float)d;
  auto stats = [&](float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (4 * lane + 256 * i < d) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (4 * lane + 256 * i < d) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float c = v[i][e] - mean; q += c * c; }
      }
    rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = 4 * lane + 256 * i;
    v[i] = c < d ? *reinterpret_cast<const f32x4*>(x + row * ldx + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float mean, rstd;
  stats(mean, rstd);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = 4 * lane + 256 * i;
    if (c < d) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g1 + c), bb = *reinterpret_cast<const f32x4*>(b1 + c);
      const f32x4 aa = add ? *reinterpret_cast<const f32x4*>(add + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = ((v[i][e] - mean) * rstd * gg[e] + bb[e]) + aa[e];
    }
  }
  stats(mean, rstd);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = 4 * lane + 256 * i;
    if (c < d) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g2 + c), bb = *reinterpret_cast<const f32x4*>(b2 + c);
      f32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
      *reinterpret_cast<f32x4*>(y + row * ldy + c) = out;
    }
  }
}

// ---- importance[m] = valid ? sigmoid(hid[m] . w2 + b2) : 0   (hid = relu(Y W1^T + b1) from the GEMM), one wave per row
__global__ void __launch_bounds__(256)
importance_rows_kernel(const float* __restrict__ hid, int64_t ldh, const float* __restrict__ w2, const float* __restrict__ b2,
                       const int64_t* __restrict__ num_ims, int rows_per_slide, int64_t M, int Hi, float* __restrict__ importance, int relu) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float acc = 0.f;
  // relu: hid holds the PRE-activation Y W1^T + b1 (the importance and projection products as ONE GEMM: the relu cannot sit in its epilogue)
  for (int c = lane; c < Hi; c += 64) { const float h = hid[row * ldh + c]; acc = fmaf(relu ? fmaxf(h, 0.f) : h, w2[c], acc); }
  acc = wave_sum(acc);
  const int b = (int)(row / rows_per_slide), idx = (int)(row - (int64_t)b * rows_per_slide);
  if (lane == 0) importance[row] = idx < (int)num_ims[b] ? sigmoid_acc(acc + *b2) : 0.f;
}

// ---- tokens[b, 0] = special ; tokens[b, 1 + n] = alpha * P[b n] + bp + PE(position)   (reference model/aggregator.py:37-65,
// utils.py:16-23,47-67): one wave per token row.  PE: channel c of the 2-D code uses x for c < d/2 and y otherwise, sin for even
// c' = c mod d/2 and cos for odd c', angle = position * div[c' >> 1]; the 1-D code uses the patch index and div[c >> 1].
__global__ void __launch_bounds__(256)
tokens_assemble_kernel(const float* __restrict__ P, int64_t ldp, const float* __restrict__ importance, int imp_mul,
                       const float* __restrict__ bp, const float* __restrict__ special, const float* __restrict__ div_term,
                       const int64_t* __restrict__ locs, int rows_per_slide, int patch_size, int pe_mode, int d, int B,
                       float* __restrict__ tokens) {
  const int lane = threadIdx.x & 63;
  const int64_t trow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int T = rows_per_slide + 1;
  if (trow >= (int64_t)B * T) return;
  const int b = (int)(trow / T), t = (int)(trow - (int64_t)b * T);
  float* out = tokens + trow * d;
  if (t == 0) {
    for (int c = lane; c < d; c += 64) out[c] = special[c];
    return;
  }
  const int64_t m = (int64_t)b * rows_per_slide + (t - 1);
  const float a = imp_mul ? importance[m] : 1.f;
  const int64_t px = locs[2 * m] / patch_size, py = locs[2 * m + 1] / patch_size;
  const int half = d / 2;
  for (int c = lane; c < d; c += 64) {
    float pe;
    if (pe_mode == 2) {
      const int cc = c < half ? c : c - half;
      const float ang = (float)(c < half ? px : py) * div_term[cc >> 1];
      pe = (cc & 1) ? cosf(ang) : sinf(ang);
    } else {
      const float ang = (float)(t - 1) * div_term[c >> 1];
      pe = (c & 1) ? cosf(ang) : sinf(ang);
    }
    // a == 0: a padded row (importance is written as exactly 0 there) - its P may be undefined (a skipped all-padding tile): select, do not multiply
    out[c] = ((imp_mul && a == 0.f) ? 0.f : a * P[m * ldp + c]) + bp[c] + pe;
  }
}

// ---- importance_rows + tokens_assemble in ONE pass over the rows of the [W1 ; Wp] product (hid [M, >= Hi + d]: hidden pre-activations |
// projection): one wave per patch row computes the importance logit, writes importance[m] and the token row alpha * P + bp + PE with
// the sin / cos values read from paths_pe_table's table (the values tokens_assemble computes: same expression, same libm calls);
// the wave of a slide's first row also writes the special token.  Padded rows: importance 0, token bp + PE (P is not read).
__global__ void __launch_bounds__(256)
importance_tokens_rows_kernel(const float* __restrict__ hid, int64_t ldh, const float* __restrict__ w2, const float* __restrict__ b2,
                              const int64_t* __restrict__ num_ims, int rows_per_slide, int64_t M, int Hi, float* __restrict__ importance,
                              int relu, int imp_mul, const float* __restrict__ bp, const float* __restrict__ special,
                              const float* __restrict__ pe_table, int pe_rows, const int64_t* __restrict__ locs, int patch_size, int pe_mode,
                              int d, float* __restrict__ tokens) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int b = (int)(row / rows_per_slide), idx = (int)(row - (int64_t)b * rows_per_slide);
  const bool valid = idx < (int)num_ims[b];
  const float* h = hid + row * ldh;
  float acc = 0.f;
  if (valid)
    for (int c = lane; c < Hi; c += 64) { const float v = h[c]; acc = fmaf(relu ? fmaxf(v, 0.f) : v, w2[c], acc); }
  acc = wave_sum(acc);
  const float imp = valid ? sigmoid_acc(acc + *b2) : 0.f;
  if (lane == 0) importance[row] = imp;
  const float a = imp_mul ? imp : 1.f;
  float* out = tokens + ((int64_t)b * (rows_per_slide + 1) + idx + 1) * d;
  const int half = d / 2;
  const int px = (int)min(max(locs[2 * row] / patch_size, (int64_t)0), (int64_t)(pe_rows - 1));
  const int py = (int)min(max(locs[2 * row + 1] / patch_size, (int64_t)0), (int64_t)(pe_rows - 1));
  const int pi = min(idx, pe_rows - 1);
  for (int c = lane; c < d; c += 64) {
    float pe;
    if (pe_mode == 2) {
      const int cc = c < half ? c : c - half;
      pe = pe_table[(int64_t)(c < half ? px : py) * half + cc];
    } else {
      pe = pe_table[(int64_t)pi * d + c];
    }
    out[c] = (valid ? a * h[Hi + c] : 0.f) + bp[c] + pe;
  }
  if (idx == 0) {
    float* sp = tokens + (int64_t)b * (rows_per_slide + 1) * d;
    for (int c = lane; c < d; c += 64) sp[c] = special[c];
  }
}

// ---- final head for any d <= 2048: decoder.norm(token 0) + slide-context residual -> ctx_out ; classifier over [concat ctx | f]
__global__ void __launch_bounds__(64)
final_head_any_kernel(const float* __restrict__ x, int64_t slide_stride, const float* __restrict__ lng, const float* __restrict__ lnb,
                      const float* __restrict__ ctx_prev, int64_t ctx_stride, const float* __restrict__ ctx_all, int ctx_depth,
                      const float* __restrict__ wcls, const float* __restrict__ bcls, int num_logits, int cls_in,
                      float* __restrict__ ctx_out, float* __restrict__ logits, int d, float eps) {
  __shared__ float f[2048];
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* row = x + (int64_t)b * slide_stride;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += row[c];
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int c = lane; c < d; c += 64) { const float e = row[c] - mean; q += e * e; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
  for (int c = lane; c < d; c += 64) {
    float v = (row[c] - mean) * rstd * lng[c] + lnb[c];
    if (ctx_prev) v += ctx_prev[(int64_t)b * ctx_stride + c];
    f[c] = v;
    ctx_out[(int64_t)b * d + c] = v;
  }
  __syncthreads();
  for (int j = 0; j < num_logits; ++j) {
    const float* w = wcls + (int64_t)j * cls_in;
    float acc = 0.f;
    if (ctx_all) {
      for (int i = lane; i < ctx_depth * d; i += 64) acc += w[i] * ctx_all[(int64_t)b * ctx_depth * d + i];
      w += ctx_depth * d;
    }
    for (int c = lane; c < d; c += 64) acc += w[c] * f[c];
    acc = wave_sum(acc);
    if (lane == 0) logits[(int64_t)b * num_logits + j] = acc + bcls[j];
  }
}

}  // namespace

extern "C" {

// softmax(q k^T / sqrt(hd)) v per head with keys >= num_ims[b] + 1 masked (reference model/aggregator.py:70-72 + utils.py:97-103) for
// head_dim in {16, 32, 48, 64}; qkv [B*T, 3 d] token-major (the in_proj output: q | k | v), qscale = log2(e) / sqrt(head_dim),
// o [B, T, d].  max_queries > 0: only queries [0, max_queries) are computed (last layer: token 0).
int paths_attention_any(const float* qkv, int64_t ld, float* o, const int64_t* num_ims, int B, int T, int H, int head_dim, float qscale,
                        int max_queries, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && T > 0 && H > 0 && qkv && o && num_ims, "attention_any: bad arguments");
  PATHS_REQUIRE(ld % 4 == 0 && ((uintptr_t)qkv | (uintptr_t)o) % 16 == 0, "attention_any: buffers must be 16-byte aligned, ld a multiple of 4");
  const int d = H * head_dim;
  const int nq = max_queries > 0 && max_queries < T ? max_queries : T;
  dim3 grid((nq + 63) / 64, H, B);
  const DropSite none{0u, 0u, 0u, 1.f};
  switch (head_dim) {
    case 16: hipLaunchKernelGGL((attn_any_kernel<16, false>), grid, dim3(256), 0, stream, qkv, ld, d, qscale, o, num_ims, T, H, nullptr, none); break;
    case 32: hipLaunchKernelGGL((attn_any_kernel<32, false>), grid, dim3(256), 0, stream, qkv, ld, d, qscale, o, num_ims, T, H, nullptr, none); break;
    case 48: hipLaunchKernelGGL((attn_any_kernel<48, false>), grid, dim3(256), 0, stream, qkv, ld, d, qscale, o, num_ims, T, H, nullptr, none); break;
    case 64: hipLaunchKernelGGL((attn_any_kernel<64, false>), grid, dim3(256), 0, stream, qkv, ld, d, qscale, o, num_ims, T, H, nullptr, none); break;
    default: return paths_set_error(PATHS_EUNSUPPORTED, "attention_any: head_dim %d (supported: 16, 32, 48, 64)", head_dim);
  }
  PATHS_LAUNCH_CHECK("attention_any");
  return PATHS_OK;
}

// paths_attention_any for the training forward: also lse [B, H, T] (log2 domain, un-dropped softmax; may be null) and dropout p on the
// probabilities (site key drop_key; p = 0: none).  Rows >= max_queries of o / lse are not written.
int paths_attention_any_train(const float* qkv, int64_t ld, float* o, float* lse, const int64_t* num_ims, int B, int T, int H, int head_dim,
                              float qscale, int max_queries, uint64_t drop_key, float drop_p, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && T > 0 && H > 0 && qkv && o && num_ims, "attention_any_train: bad arguments");
  PATHS_REQUIRE(ld % 4 == 0 && ((uintptr_t)qkv | (uintptr_t)o) % 16 == 0, "attention_any_train: buffers must be 16-byte aligned, ld a multiple of 4");
  PATHS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention_any_train: p must be in [0, 1)");
  const int d = H * head_dim;
  const int nq = max_queries > 0 && max_queries < T ? max_queries : T;
  dim3 grid((nq + 63) / 64, H, B);
  const DropSite site = paths_make_drop_site(drop_key, drop_p);
  switch (head_dim) {
    case 16: hipLaunchKernelGGL((attn_any_kernel<16, true>), grid, dim3(256), 0, stream, qkv, ld, d, qscale, o, num_ims, T, H, lse, site); break;
    case 32: hipLaunchKernelGGL((attn_any_kernel<32, true>), grid, dim3(256), 0, stream, qkv, ld, d, qscale, o, num_ims, T, H, lse, site); break;
    case 48: hipLaunchKernelGGL((attn_any_kernel<48, true>), grid, dim3(256), 0, stream, qkv, ld, d, qscale, o, num_ims, T, H, lse, site); break;
    case 64: hipLaunchKernelGGL((attn_any_kernel<64, true>), grid, dim3(256), 0, stream, qkv, ld, d, qscale, o, num_ims, T, H, lse, site); break;
    default: return paths_set_error(PATHS_EUNSUPPORTED, "attention_any_train: head_dim %d (supported: 16, 32, 48, 64)", head_dim);
  }
  PATHS_LAUNCH_CHECK("attention_any_train");
  return PATHS_OK;
}

// y[rows, d] (ldy) = LayerNorm(x (ldx) (+ add [d])) * gamma + beta, d <= 2048 and a multiple of 4 (torch native_layer_norm, biased variance)
int paths_layernorm_rows(const float* x, int64_t ldx, const float* add, const float* gamma, const float* beta, float* y, int64_t ldy,
                         int64_t rows, int d, float eps, hipStream_t stream) {
  PATHS_REQUIRE(rows > 0 && d > 0 && d <= 2048 && d % 4 == 0 && x && gamma && beta && y, "layernorm_rows: bad arguments (d = %d)", d);
  PATHS_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)x | (uintptr_t)y | (uintptr_t)add | (uintptr_t)gamma | (uintptr_t)beta) % 16 == 0, "layernorm_rows: alignment");
  hipLaunchKernelGGL(layernorm_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, ldx, add, gamma, beta, y, ldy, rows, d, eps);
  PATHS_LAUNCH_CHECK("layernorm_rows");
  return PATHS_OK;
}

// y = LayerNorm(LayerNorm(x) * g1 + b1 + add) * g2 + b2 per row (norm1 -> + cross-attention bias -> norm2 of a decoder layer), d <= 2048
int paths_layernorm2_rows(const float* x, int64_t ldx, const float* g1, const float* b1, const float* add, const float* g2, const float* b2,
                          float* y, int64_t ldy, int64_t rows, int d, float eps, hipStream_t stream) {
  PATHS_REQUIRE(rows > 0 && d > 0 && d <= 2048 && d % 4 == 0 && x && g1 && b1 && g2 && b2 && y, "layernorm2_rows: bad arguments (d = %d)", d);
  PATHS_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)x | (uintptr_t)y | (uintptr_t)add | (uintptr_t)g1 | (uintptr_t)b1 | (uintptr_t)g2 | (uintptr_t)b2) % 16 == 0,
                "layernorm2_rows: alignment");
  hipLaunchKernelGGL(layernorm2_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, ldx, g1, b1, add, g2, b2, y, ldy, rows, d, eps);
  PATHS_LAUNCH_CHECK("layernorm2_rows");
  return PATHS_OK;
}

// importance[M] from the hidden layer of the importance MLP (reference model/paths.py:95 + utils.py:106-115), any hidden width
int paths_importance_rows(const float* hid, int64_t ldh, const float* w2, const float* b2, const int64_t* num_ims, int rows_per_slide,
                          int64_t M, int Hi, float* importance, int relu, hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && Hi > 0 && rows_per_slide > 0 && hid && w2 && b2 && num_ims && importance, "importance_rows: bad arguments");
  hipLaunchKernelGGL(importance_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, hid, ldh, w2, b2, num_ims, rows_per_slide, M, Hi, importance, relu ? 1 : 0);
  PATHS_LAUNCH_CHECK("importance_rows");
  return PATHS_OK;
}

// paths_importance_rows + paths_tokens_assemble in one launch (reference model/paths.py:95-98,119-124, model/aggregator.py:37-65):
// hid [M, ldh] = [hidden pre-activations (Hi) | projection (d)] per patch row; pe_table: paths_pe_table(div, pe_mode, d, pe_rows)
// covering every position of the batch (pe_mode 2: locs / patch_size < pe_rows; 1: rows_per_slide <= pe_rows).
int paths_importance_tokens_rows(const float* hid, int64_t ldh, const float* w2, const float* b2, const int64_t* num_ims, int rows_per_slide,
                                 int64_t M, int Hi, float* importance, int relu, int imp_mul, const float* bp, const float* special,
                                 const float* pe_table, int pe_rows, const int64_t* locs, int patch_size, int pe_mode, int d, float* tokens,
                                 hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && Hi > 0 && rows_per_slide > 0 && M % rows_per_slide == 0 && hid && w2 && b2 && num_ims && importance, "importance_tokens_rows: bad arguments");
  PATHS_REQUIRE(d > 0 && d % 2 == 0 && ldh >= Hi + d && patch_size > 0 && (pe_mode == 1 || pe_mode == 2) && bp && special && pe_table && pe_rows > 0 && locs && tokens,
                "importance_tokens_rows: bad token arguments");
  PATHS_REQUIRE(pe_mode == 2 || pe_rows >= rows_per_slide, "importance_tokens_rows: the 1-D table must cover %d rows (has %d)", rows_per_slide, pe_rows);
  hipLaunchKernelGGL(importance_tokens_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, hid, ldh, w2, b2, num_ims, rows_per_slide, M, Hi,
                     importance, relu ? 1 : 0, imp_mul ? 1 : 0, bp, special, pe_table, pe_rows, locs, patch_size, pe_mode, d, tokens);
  PATHS_LAUNCH_CHECK("importance_tokens_rows");
  return PATHS_OK;
}

// tokens [B, N+1, d] from P = Y Wp^T [B*N, d] (reference model/paths.py:96-98,119-124, model/aggregator.py:37-65)
int paths_tokens_assemble(const float* P, int64_t ldp, const float* importance, int imp_mul, const float* bp, const float* special,
                          const float* div_term, const int64_t* locs, int rows_per_slide, int patch_size, int pe_mode, int d, int B,
                          float* tokens, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && rows_per_slide > 0 && d > 0 && patch_size > 0 && (pe_mode == 1 || pe_mode == 2) && P && importance && bp && special && div_term && locs && tokens,
                "tokens_assemble: bad arguments");
  const int64_t rows = (int64_t)B * (rows_per_slide + 1);
  hipLaunchKernelGGL(tokens_assemble_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, P, ldp, importance, imp_mul, bp, special, div_term,
                     locs, rows_per_slide, patch_size, pe_mode, d, B, tokens);
  PATHS_LAUNCH_CHECK("tokens_assemble");
  return PATHS_OK;
}

// paths_final_head for any trans_dim <= 2048 (reference model/aggregator.py:75, model/paths.py:130-139)
int paths_final_head_any(const float* x, int64_t slide_stride, const float* lng, const float* lnb, const float* ctx_prev, int64_t ctx_stride,
                         const float* ctx_all, int ctx_depth, const float* wcls, const float* bcls, int num_logits, int cls_in,
                         float* ctx_out, float* logits, int B, int d, float eps, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && d > 0 && d <= 2048 && num_logits > 0 && cls_in == (ctx_all ? (ctx_depth + 1) * d : d), "final_head_any: bad shape");
  hipLaunchKernelGGL(final_head_any_kernel, dim3(B), dim3(64), 0, stream, x, slide_stride, lng, lnb, ctx_prev, ctx_stride, ctx_all, ctx_depth,
                     wcls, bcls, num_logits, cls_in, ctx_out, logits, d, eps);
  PATHS_LAUNCH_CHECK("final_head_any");
  return PATHS_OK;
}

}  // extern "C"
