// Token-row chain of one post-LN nn.TransformerDecoderLayer with an EMPTY memory sequence, plus the
// QKV projection that feeds the next attention call (fp32-exact MFMA).
//
// Replaces, for the nn.Transformer built at reference model/aggregator.py:25-33 and called at :70-72:
//   x  = norm1(x + out_proj(attn))                          (self-attention block tail)
//   x  = norm2(x + multihead_attn.out_proj.bias)            (cross-attention over 0 keys == its bias)
//   x  = norm3(x + linear2(relu(linear1(x))))               (feed-forward block)
//   q,k,v = in_proj(x) of the NEXT layer (or of layer 0 when do_post == 0), q pre-scaled for exp2 softmax
//
// Mapping (MI355X): every product is computed TRANSPOSED, Y^T[out][token] = W[out][:] . X^T[:][token], with
// v_mfma_f32_16x16x4_f32: the weight is the A operand (streamed through LDS in 36 KB chunks, double
// buffered, shared by the 4 waves of a workgroup), the activation is the B operand and lives in
// registers for the whole chain.  One wave owns 16 tokens (lane&15 = token); the C layout of a product
// (lane group g = lane>>4 holds features 16t+4g+r) is exactly the B-operand layout of the next product
// when the weight's k index is read in that same order -> no LDS round trip, no shuffles between GEMMs,
// and LayerNorm's feature reduction is 31 in-lane adds + 2 cross-lane swaps.  16-token waves (rather
// than 32) keep >= 1000 waves in flight at 8 slides x 2049 tokens, enough for all 1024 SIMDs.
#include "common.h"

namespace {

constexpr int DM = 128;          // trans_dim
constexpr int DFF = 512;         // dim_feedforward = 4 * trans_dim
constexpr int LDA_ = DM + 8;     // LDS stride of a [64 rows][128 k] chunk   (stride/4 % 16 == 2: conflict-free)
constexpr int LDB_ = 64 + 8;     // LDS stride of a [128 rows][64 k] chunk
constexpr int CHUNK_FLOATS = 64 * LDA_ > 128 * LDB_ ? 64 * LDA_ : 128 * LDB_;   // 9216 floats = 36 KB

struct TLayerParams {
  const float* x_in; const float* attn; float* x_out;
  const float *wo, *bo, *ln1g, *ln1b, *cab, *ln2g, *ln2b, *w1, *b1, *w2, *b2, *ln3g, *ln3b;
  const float *wqkv, *bqkv;
  float *q, *k, *v;
  const int64_t* num_ims;
  int T, H; int do_post, do_qkv, skip_padding; float qscale, eps;
};

typedef f32x4 act_t[8];   // 128 features of 16 tokens: tile t, reg r, lane group g -> feature 16t + 4g + r

__device__ __forceinline__ void layernorm_t(act_t& x, const float* gamma, const float* beta, int g4, float eps) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) s += (x[t][0] + x[t][1]) + (x[t][2] + x[t][3]);
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  const float mean = s * (1.0f / DM);
  float v = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float c = x[t][r] - mean; v += c * c; }
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  const float rstd = 1.0f / sqrtf(v * (1.0f / DM) + eps);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 16 * t + 4 * g4);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 16 * t + 4 * g4);
#pragma unroll
    for (int r = 0; r < 4; ++r) x[t][r] = (x[t][r] - mean) * rstd * gm[r] + bt[r];
  }
}

// acc[ot] += W_chunk[16 ot + (lane&15)][16 kt + 4 g + r] * x[kt][r]   (NOT out tiles, NKT k tiles)
template <int NOT, int NKT>
__device__ __forceinline__ void mm_chunk(const float* sW, int ldw, f32x4 (&acc)[NOT], const f32x4* x, int ql, int g4) {
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    f32x4 a[NOT];
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) a[ot] = *reinterpret_cast<const f32x4*>(sW + (16 * ot + ql) * ldw + 16 * kt + 4 * g4);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ot = 0; ot < NOT; ++ot) acc[ot] = mfma16(a[ot][r], x[kt][r], acc[ot]);
  }
}

__global__ void __launch_bounds__(256, 2)
tlayer_f32_kernel(TLayerParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][CHUNK_FLOATS] + in-loop biases [512 + 384]
  float* s_b1 = smem + 2 * CHUNK_FLOATS;
  float* s_bqkv = s_b1 + DFF;
  const int b = blockIdx.y, t0 = blockIdx.x * 64;
  if (p.skip_padding && t0 >= (int)p.num_ims[b] + 1) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ql = lane & 15, g4 = lane >> 4;
  const int tok = t0 + wave * 16 + ql;
  const int tokc = min(tok, p.T - 1);
  const int64_t rowoff = ((int64_t)b * p.T + tokc) * DM;

  // ---- weight chunk stream: ids 0-1 Wo | 2+2h W1 rows 64h | 3+2h W2[:, 64h:64h+64] | 18-23 Wqkv rows
  const int c_first = p.do_post ? 0 : 18, c_last = p.do_qkv ? 24 : 18;
  f32x4 rs[8];
  auto stage_load = [&](int c) {
    if (c >= 2 && c < 18 && (c & 1)) {            // [128 rows][64 k] slice of W2 (row stride 512)
      const float* src = p.w2 + 64 * ((c - 3) >> 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int idx = tid + i * 256; rs[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)(idx >> 4) * DFF + 4 * (idx & 15)); }
    } else {                                       // [64 rows][128 k]
      const float* src = c < 2 ? p.wo + (int64_t)c * 64 * DM : c < 18 ? p.w1 + (int64_t)((c - 2) >> 1) * 64 * DM : p.wqkv + (int64_t)(c - 18) * 64 * DM;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int idx = tid + i * 256; rs[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)(idx >> 5) * DM + 4 * (idx & 31)); }
    }
  };
  auto stage_store = [&](int c, int buf) {
    float* dst = smem + buf * CHUNK_FLOATS;
    if (c >= 2 && c < 18 && (c & 1)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int idx = tid + i * 256; *reinterpret_cast<f32x4*>(dst + (idx >> 4) * LDB_ + 4 * (idx & 15)) = rs[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int idx = tid + i * 256; *reinterpret_cast<f32x4*>(dst + (idx >> 5) * LDA_ + 4 * (idx & 31)) = rs[i]; }
    }
  };
  // In-loop biases live in LDS: an LDS read is counted by lgkmcnt, so consuming it never forces the vmcnt wait that
  // would drain the weight-chunk prefetch (vector-memory loads retire in order; hipcc sinks small global loads next
  // to their use, i.e. behind the prefetch).
  act_t x;      // activations first (oldest loads), then biases, then the first weight chunk
#pragma unroll
  for (int t = 0; t < 8; ++t) x[t] = *reinterpret_cast<const f32x4*>(p.x_in + rowoff + 16 * t + 4 * g4);
  if (p.do_post) for (int i = threadIdx.x; i < DFF; i += 256) s_b1[i] = p.b1[i];
  if (p.do_qkv) for (int i = threadIdx.x; i < 3 * DM; i += 256) s_bqkv[i] = p.bqkv[i];
  int buf = 0, c = c_first;
  stage_load(c);
  stage_store(c, 0);
  __syncthreads();
  // begin(): start fetching the chunk after the current one; end(): publish it and flip buffers
  auto begin = [&]() { if (c + 1 < c_last) stage_load(c + 1); };
  auto end = [&]() { if (c + 1 < c_last) stage_store(c + 1, buf ^ 1); __syncthreads(); buf ^= 1; ++c; };

  if (p.do_post) {
    // ---- out_proj(attn) + residual -> norm1 -> + cross-attn bias -> norm2
    act_t at, y;
#pragma unroll
    for (int t = 0; t < 8; ++t) at[t] = *reinterpret_cast<const f32x4*>(p.attn + rowoff + 16 * t + 4 * g4);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      begin();
      f32x4 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      mm_chunk<4, 8>(smem + buf * CHUNK_FLOATS, LDA_, acc, at, ql, g4);
#pragma unroll
      for (int i = 0; i < 4; ++i) y[4 * half + i] = acc[i];
      end();
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 bo = *reinterpret_cast<const f32x4*>(p.bo + 16 * t + 4 * g4);
      x[t] = x[t] + (y[t] + bo);
    }
    layernorm_t(x, p.ln1g, p.ln1b, g4, p.eps);
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = x[t] + *reinterpret_cast<const f32x4*>(p.cab + 16 * t + 4 * g4);
    layernorm_t(x, p.ln2g, p.ln2b, g4, p.eps);

    // ---- feed-forward: 8 hidden chunks of 64; y accumulates linear2
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int hc = 0; hc < 8; ++hc) {
      f32x4 hid[4], b1v[4];
      begin();
#pragma unroll
      for (int i = 0; i < 4; ++i) b1v[i] = *reinterpret_cast<const f32x4*>(s_b1 + 64 * hc + 16 * i + 4 * g4);
#pragma unroll
      for (int i = 0; i < 4; ++i) hid[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      mm_chunk<4, 8>(smem + buf * CHUNK_FLOATS, LDA_, hid, x, ql, g4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[i][r] = fmaxf(hid[i][r] + b1v[i][r], 0.f);
      end();
      begin();
      mm_chunk<8, 4>(smem + buf * CHUNK_FLOATS, LDB_, y, hid, ql, g4);
      end();
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.b2 + 16 * t + 4 * g4);
      x[t] = x[t] + (y[t] + b2);
    }
    layernorm_t(x, p.ln3g, p.ln3b, g4, p.eps);
    if (tok < p.T) {
#pragma unroll
      for (int t = 0; t < 8; ++t) *reinterpret_cast<f32x4*>(p.x_out + rowoff + 16 * t + 4 * g4) = x[t];
    }
  }

  if (p.do_qkv) {
    // ---- in_proj: 6 chunks of 64 output rows: q(0,1) k(2,3) v(4,5); head = feature >> 5
#pragma unroll 1
    for (int qc = 0; qc < 6; ++qc) {
      f32x4 bb[4];
      begin();
#pragma unroll
      for (int i = 0; i < 4; ++i) bb[i] = *reinterpret_cast<const f32x4*>(s_bqkv + 128 * (qc >> 1) + 64 * (qc & 1) + 16 * i + 4 * g4);
      f32x4 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      mm_chunk<4, 8>(smem + buf * CHUNK_FLOATS, LDA_, acc, x, ql, g4);
      float* dst = qc < 2 ? p.q : qc < 4 ? p.k : p.v;
      const float sc = qc < 2 ? p.qscale : 1.0f;
      if (tok < p.T) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = 64 * (qc & 1) + 16 * i + 4 * g4;          // feature within q / k / v
          const int head = f >> 5, dd = f & 31;
          *reinterpret_cast<f32x4*>(dst + (((int64_t)b * p.H + head) * p.T + tok) * 32 + dd) = (acc[i] + bb[i]) * sc;
        }
      }
      end();
    }
  }
}

// ---- final head: LN_final(token 0) + slide-context residual + classifier (reference aggregator.py:75,
//      paths.py:130-139).  One wave per slide; d = 128 -> 2 features per lane.
__global__ void __launch_bounds__(64)
final_head_kernel(const float* __restrict__ x, int64_t slide_stride, const float* __restrict__ lng, const float* __restrict__ lnb,
                  const float* __restrict__ ctx_prev, int64_t ctx_stride,      // residual source (last ctx_slide row) or null
                  const float* __restrict__ ctx_all, int ctx_depth,            // concat mode: [B, depth, d] or null
                  const float* __restrict__ wcls, const float* __restrict__ bcls, int num_logits, int cls_in,
                  float* __restrict__ ctx_out, float* __restrict__ logits, float eps) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* row = x + (int64_t)b * slide_stride;
  float v0 = row[lane], v1 = row[lane + 64];
  float s = v0 + v1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
  const float mean = s * (1.0f / DM);
  const float c0 = v0 - mean, c1 = v1 - mean;
  float var = c0 * c0 + c1 * c1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) var += __shfl_xor(var, o);
  const float rstd = 1.0f / sqrtf(var * (1.0f / DM) + eps);
  float f0 = c0 * rstd * lng[lane] + lnb[lane];
  float f1 = c1 * rstd * lng[lane + 64] + lnb[lane + 64];
  if (ctx_prev) {
    f0 += ctx_prev[(int64_t)b * ctx_stride + lane];
    f1 += ctx_prev[(int64_t)b * ctx_stride + lane + 64];
  }
  ctx_out[(int64_t)b * DM + lane] = f0;
  ctx_out[(int64_t)b * DM + lane + 64] = f1;
  for (int j = 0; j < num_logits; ++j) {
    const float* w = wcls + (int64_t)j * cls_in;
    float acc = 0.f;
    if (ctx_all) {   // concat mode: classifier input = [flatten(ctx_slide) | agg]
      for (int i = lane; i < ctx_depth * DM; i += 64) acc += w[i] * ctx_all[(int64_t)b * ctx_depth * DM + i];
      w += ctx_depth * DM;
    }
    acc += w[lane] * f0 + w[lane + 64] * f1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) logits[(int64_t)b * num_logits + j] = acc + bcls[j];
  }
}


// ------------------------------------------------------------------------------------------------
// Last decoder layer, token 0 only (the aggregator reads out[:, 0], reference model/aggregator.py:75): that
// layer needs K/V of every token but the query / out_proj / LayerNorms / FFN of ONE row per slide.  One
// workgroup per slide: wave h = attention head h (lane-strided keys, per-lane online softmax, wave merge), then
// the row chain as wave-cooperative GEMVs (weights read straight from L2, 512 B coalesced rows), decoder.norm,
// slide-context residual / concat and the classifier (reference model/paths.py:130-139).
// ------------------------------------------------------------------------------------------------
constexpr int T0_SPLITS = 16;     // key partitions per (slide, head) in the single-query attention
constexpr int T0_PSTRIDE = 36;    // floats per partial: m, l, o[32], pad

struct Token0Params {
  const float* x_in; const float* partials; const int64_t* num_ims;
  const float *wo, *bo, *ln1g, *ln1b, *cab, *ln2g, *ln2b, *w1, *b1, *w2, *b2, *ln3g, *ln3b, *lnfg, *lnfb;
  const float* ctx_prev; int64_t ctx_stride; const float* ctx_all; int ctx_depth;
  const float *wcls, *bcls; int num_logits, cls_in;
  float *ctx_out, *logits; int T, H; float eps, eps_f;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

// y[f] = dot(W[f][0:K], xin[0:K]) for f in [f0, f0+nf), nf a multiple of R: R rows in flight per wave so the
// row loads and the cross-lane reductions of different rows overlap (this kernel is latency-, not bandwidth-bound)
template <int K, int R>
__device__ __forceinline__ void wave_gemv(const float* __restrict__ W, int ldw, const float* xin, float* yout, int f0, int nf, int lane) {
  constexpr int PER = K / 64;
  float xv[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) xv[i] = xin[lane * PER + i];
  for (int f = f0; f < f0 + nf; f += R) {
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float* w = W + (int64_t)(f + r) * ldw + lane * PER;
      acc[r] = 0.f;
#pragma unroll
      for (int i = 0; i < PER; ++i) acc[r] += w[i] * xv[i];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] += __shfl_xor(acc[r], o);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) yout[f + r] = acc[r];
    }
  }
}

__device__ __forceinline__ void block_layernorm128(float* v, const float* g, const float* bta, float eps, int tid) {
  // wave 0 normalises v[0:128] in place (2 values per lane)
  if (tid < 64) {
    const float a = v[tid], c = v[tid + 64];
    const float mean = wave_sum(a + c) * (1.0f / DM);
    const float da = a - mean, dc = c - mean;
    const float rstd = 1.0f / sqrtf(wave_sum(da * da + dc * dc) * (1.0f / DM) + eps);
    v[tid] = da * rstd * g[tid] + bta[tid];
    v[tid + 64] = dc * rstd * g[tid + 64] + bta[tid + 64];
  }
  __syncthreads();
}

// Single-query (token 0) attention partials: one wave per (key partition, head, slide); lane-strided keys, per-lane
// online softmax, wave merge.  Writes (m, l, o[32]) per partition; token0_tail_kernel combines them.
__global__ void __launch_bounds__(64)
attn_token0_partial_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                           const int64_t* __restrict__ num_ims, float* __restrict__ partials, int T, int H) {
  const int part = blockIdx.x, head = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  const int len = (int)num_ims[b] + 1;
  const int chunk = ((len + T0_SPLITS - 1) / T0_SPLITS + 63) & ~63;
  const int k0 = part * chunk, k1 = min(len, k0 + chunk);
  const int64_t base = ((int64_t)b * H + head) * T * 32;
  float qv[32];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(q + base + 4 * i);        // query row 0 (already scaled)
    qv[4 * i] = t[0]; qv[4 * i + 1] = t[1]; qv[4 * i + 2] = t[2]; qv[4 * i + 3] = t[3];
  }
  float m = -INFINITY, l = 0.f, o[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) o[i] = 0.f;
#pragma unroll 2
  for (int key = k0 + lane; key < k1; key += 64) {
    const float* kp = k + base + (int64_t)key * 32;
    const float* vp = v + base + (int64_t)key * 32;
    f32x4 kk[8], vv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { kk[i] = *reinterpret_cast<const f32x4*>(kp + 4 * i); vv[i] = *reinterpret_cast<const f32x4*>(vp + 4 * i); }
    float sc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      sc += kk[i][0] * qv[4 * i] + kk[i][1] * qv[4 * i + 1] + kk[i][2] * qv[4 * i + 2] + kk[i][3] * qv[4 * i + 3];
    const float mn = fmaxf(m, sc);
    const float al = exp2f(m - mn), pr = exp2f(sc - mn);
    l = l * al + pr;
    m = mn;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o[4 * i] = o[4 * i] * al + pr * vv[i][0];
      o[4 * i + 1] = o[4 * i + 1] * al + pr * vv[i][1];
      o[4 * i + 2] = o[4 * i + 2] * al + pr * vv[i][2];
      o[4 * i + 3] = o[4 * i + 3] * al + pr * vv[i][3];
    }
  }
  float mg = m;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) mg = fmaxf(mg, __shfl_xor(mg, off));
  const float scale = (m == -INFINITY) ? 0.f : exp2f(m - mg);          // lanes without keys contribute nothing
  l *= scale;
#pragma unroll
  for (int i = 0; i < 32; ++i) o[i] *= scale;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    l += __shfl_xor(l, off);
#pragma unroll
    for (int i = 0; i < 32; ++i) o[i] += __shfl_xor(o[i], off);
  }
  float* dst = partials + (((int64_t)b * H + head) * T0_SPLITS + part) * T0_PSTRIDE;
  if (lane == 0) { dst[0] = mg; dst[1] = l; }
#pragma unroll
  for (int i = 0; i < 32; ++i) if (lane == 0) dst[2 + i] = o[i];
}

constexpr int T0_WAVES = 16;

__global__ void __launch_bounds__(64 * T0_WAVES)
token0_tail_kernel(Token0Params p) {
  __shared__ float s_x[DM], s_a[DM], s_y[DM], s_h[DFF];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < DM) {                                  // merge the key partitions of head tid>>5 (flash-decoding style)
    const int head = tid >> 5, i = tid & 31;
    const float* sp = p.partials + ((int64_t)b * p.H + head) * T0_SPLITS * T0_PSTRIDE;
    float M = -INFINITY;
#pragma unroll
    for (int pt = 0; pt < T0_SPLITS; ++pt) M = fmaxf(M, sp[pt * T0_PSTRIDE]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int pt = 0; pt < T0_SPLITS; ++pt) {
      const float mp = sp[pt * T0_PSTRIDE];
      const float w = (mp == -INFINITY) ? 0.f : exp2f(mp - M);
      num += sp[pt * T0_PSTRIDE + 2 + i] * w; den += sp[pt * T0_PSTRIDE + 1] * w;
    }
    s_a[tid] = num / den;
    s_x[tid] = p.x_in[(int64_t)b * p.T * DM + tid];
  }
  __syncthreads();
  // ---- x = norm1(x + out_proj(a)) ; x = norm2(x + cab)
  wave_gemv<DM, 8>(p.wo, DM, s_a, s_y, wave * 8, 8, lane);
  __syncthreads();
  if (tid < DM) s_x[tid] = s_x[tid] + (s_y[tid] + p.bo[tid]);
  __syncthreads();
  block_layernorm128(s_x, p.ln1g, p.ln1b, p.eps, tid);
  if (tid < DM) s_x[tid] += p.cab[tid];
  __syncthreads();
  block_layernorm128(s_x, p.ln2g, p.ln2b, p.eps, tid);
  // ---- x = norm3(x + linear2(relu(linear1(x))))
  wave_gemv<DM, 16>(p.w1, DM, s_x, s_h, wave * 32, 32, lane);
  __syncthreads();
  if (tid < DFF) s_h[tid] = fmaxf(s_h[tid] + p.b1[tid], 0.f);
  __syncthreads();
  wave_gemv<DFF, 4>(p.w2, DFF, s_h, s_y, wave * 8, 8, lane);
  __syncthreads();
  if (tid < DM) s_x[tid] = s_x[tid] + (s_y[tid] + p.b2[tid]);
  __syncthreads();
  block_layernorm128(s_x, p.ln3g, p.ln3b, p.eps, tid);
  // ---- decoder.norm, slide-context residual, classifier
  block_layernorm128(s_x, p.lnfg, p.lnfb, p.eps_f, tid);
  if (tid < DM) {
    float f = s_x[tid];
    if (p.ctx_prev) f += p.ctx_prev[(int64_t)b * p.ctx_stride + tid];
    s_x[tid] = f;
    p.ctx_out[(int64_t)b * DM + tid] = f;
  }
  __syncthreads();
  for (int j = wave; j < p.num_logits; j += T0_WAVES) {
    const float* w = p.wcls + (int64_t)j * p.cls_in;
    float acc = 0.f;
    if (p.ctx_all) {
      for (int i = lane; i < p.ctx_depth * DM; i += 64) acc += w[i] * p.ctx_all[(int64_t)b * p.ctx_depth * DM + i];
      w += p.ctx_depth * DM;
    }
    acc += w[lane] * s_x[lane] + w[lane + 64] * s_x[lane + 64];
    acc = wave_sum(acc);
    if (lane == 0) p.logits[(int64_t)b * p.num_logits + j] = acc + p.bcls[j];
  }
}

// ---- stand-alone LayerNorm over rows of width 128 (one wave per row; wavefront reduction).
__global__ void __launch_bounds__(256)
layernorm128_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ bta,
                    float* __restrict__ y, int64_t rows, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float2 v = *reinterpret_cast<const float2*>(x + row * DM + 2 * lane);
  float s = v.x + v.y;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
  const float mean = s * (1.0f / DM);
  const float c0 = v.x - mean, c1 = v.y - mean;
  float var = c0 * c0 + c1 * c1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) var += __shfl_xor(var, o);
  const float rstd = 1.0f / sqrtf(var * (1.0f / DM) + eps);
  const float2 gg = *reinterpret_cast<const float2*>(g + 2 * lane), bb = *reinterpret_cast<const float2*>(bta + 2 * lane);
  float2 out = {c0 * rstd * gg.x + bb.x, c1 * rstd * gg.y + bb.y};
  *reinterpret_cast<float2*>(y + row * DM + 2 * lane) = out;
}

}  // namespace

extern "C" {

int paths_token_layer_f32(const float* x_in, const float* attn, float* x_out,
                          const float* wo, const float* bo, const float* ln1g, const float* ln1b, const float* cab,
                          const float* ln2g, const float* ln2b, const float* w1, const float* b1, const float* w2,
                          const float* b2, const float* ln3g, const float* ln3b, const float* wqkv, const float* bqkv,
                          float* q, float* k, float* v, const int64_t* num_ims, int B, int T, int d, int H,
                          int do_post, int do_qkv, int skip_padding, float qscale, float eps, int max_tokens,
                          hipStream_t stream) {
  PATHS_REQUIRE(d == DM && H == 4, "token_layer: this build supports trans_dim=128, 4 heads (got %d, %d)", d, H);
  PATHS_REQUIRE(B > 0 && T > 0 && (do_post || do_qkv), "token_layer: nothing to do");
  PATHS_REQUIRE(!skip_padding || num_ims, "token_layer: skip_padding needs num_ims");
  PATHS_REQUIRE(x_in && (!do_post || (attn && x_out && wo && w1 && w2)) && (!do_qkv || (wqkv && q && k && v)), "token_layer: null operand");
  TLayerParams p{x_in, attn, x_out, wo, bo, ln1g, ln1b, cab, ln2g, ln2b, w1, b1, w2, b2, ln3g, ln3b, wqkv, bqkv,
                 q, k, v, num_ims, T, H, do_post, do_qkv, skip_padding, qscale, eps};
  constexpr size_t lds_min = (2ull * CHUNK_FLOATS + DFF + 3 * DM) * sizeof(float);      // 77,312 B: two workgroups per CU
  constexpr size_t lds_solo = 84 * 1024;                                                 // > 80 KiB: one workgroup per CU
  PATHS_LDS_OPT_IN(tlayer_f32_kernel, lds_solo, "token_layer");
  // max_tokens > 0: only token rows [0, max_tokens) are needed (last layer: only token 0 is read downstream)
  const int nt = max_tokens > 0 && max_tokens < T ? max_tokens : T;
  const int nblk = ((nt + 63) / 64) * B;
  // Placement (speed only): with about one workgroup per CU in the grid, two co-resident workgroups would share a CU's
  // matrix pipes while other CUs idle and the kernel lasts as long as the doubled-up CUs (measured 0.51 waves/SIMD
  // average).  Asking for > half the LDS makes the dispatcher spread workgroups one per CU.
  const size_t lds = nblk <= 2 * 256 ? lds_solo : lds_min;
  hipLaunchKernelGGL(tlayer_f32_kernel, dim3((nt + 63) / 64, B), dim3(256), lds, stream, p);
  PATHS_LAUNCH_CHECK("token_layer");
  return PATHS_OK;
}

int paths_token0_tail(const float* x_in, const float* q, const float* k, const float* v, const int64_t* num_ims,
                      const float* wo, const float* bo, const float* ln1g, const float* ln1b, const float* cab,
                      const float* ln2g, const float* ln2b, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* ln3g, const float* ln3b, const float* lnfg, const float* lnfb,
                      const float* ctx_prev, int64_t ctx_stride, const float* ctx_all, int ctx_depth,
                      const float* wcls, const float* bcls, int num_logits, int cls_in,
                      float* ctx_out, float* logits, float* ws_partials /*[B*H*16*36]*/, int B, int T, int d, int H,
                      float eps, float eps_final, hipStream_t stream) {
  PATHS_REQUIRE(d == DM && H == 4, "token0_tail: this build supports trans_dim=128, 4 heads (got %d, %d)", d, H);
  PATHS_REQUIRE(B > 0 && T > 0 && num_ims && x_in && q && k && v && wo && w1 && w2 && wcls && ctx_out && logits && ws_partials, "token0_tail: null operand");
  PATHS_REQUIRE(num_logits > 0 && cls_in == (ctx_all ? (ctx_depth + 1) * DM : DM), "token0_tail: bad classifier shape");
  hipLaunchKernelGGL(attn_token0_partial_kernel, dim3(T0_SPLITS, H, B), dim3(64), 0, stream, q, k, v, num_ims, ws_partials, T, H);
  PATHS_LAUNCH_CHECK("token0_tail(attention partials)");
  Token0Params p{x_in, ws_partials, num_ims, wo, bo, ln1g, ln1b, cab, ln2g, ln2b, w1, b1, w2, b2, ln3g, ln3b, lnfg, lnfb,
                 ctx_prev, ctx_stride, ctx_all, ctx_depth, wcls, bcls, num_logits, cls_in, ctx_out, logits, T, H, eps, eps_final};
  hipLaunchKernelGGL(token0_tail_kernel, dim3(B), dim3(64 * T0_WAVES), 0, stream, p);
  PATHS_LAUNCH_CHECK("token0_tail");
  return PATHS_OK;
}

int paths_final_head(const float* x, int64_t slide_stride, const float* lng, const float* lnb,
                     const float* ctx_prev, int64_t ctx_stride, const float* ctx_all, int ctx_depth,
                     const float* wcls, const float* bcls, int num_logits, int cls_in,
                     float* ctx_out, float* logits, int B, int d, float eps, hipStream_t stream) {
  PATHS_REQUIRE(d == DM, "final_head: trans_dim must be %d", DM);
  PATHS_REQUIRE(B > 0 && num_logits > 0 && cls_in == (ctx_all ? (ctx_depth + 1) * DM : DM), "final_head: bad classifier shape");
  hipLaunchKernelGGL(final_head_kernel, dim3(B), dim3(64), 0, stream, x, slide_stride, lng, lnb, ctx_prev, ctx_stride,
                     ctx_all, ctx_depth, wcls, bcls, num_logits, cls_in, ctx_out, logits, eps);
  PATHS_LAUNCH_CHECK("final_head");
  return PATHS_OK;
}

int paths_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int d,
                        float eps, hipStream_t stream) {
  PATHS_REQUIRE(d == DM, "layernorm: width must be %d", DM);
  PATHS_REQUIRE(rows > 0, "layernorm: rows must be > 0");
  hipLaunchKernelGGL(layernorm128_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, gamma, beta, y, rows, eps);
  PATHS_LAUNCH_CHECK("layernorm");
  return PATHS_OK;
}

}  // extern "C"
