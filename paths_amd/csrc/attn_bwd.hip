// Attention backward (fp32-exact MFMA), flash style: scores are recomputed from q, k and the forward's log-sum-exp,
// the T x T matrices never exist in memory.  Differentiates the self-attention of nn.TransformerDecoderLayer as called
// at reference model/aggregator.py:70-72 (what autograd does for it in train.py:65).
//
// Conventions shared with attn_f32.hip: q is stored PRE-SCALED (q_s = q * log2(e)/sqrt(hd)), lse is in the log2 domain,
// so P = exp2(q_s . k - lse).  With ds = ln2 * P * (dP - D), D[q] = sum_dv dO O:
//      dq_s = ds K        dk = ds^T q_s        dv = P^T dO
// (the caller multiplies dq_s by the forward's q scale when it back-propagates through in_proj).
// Gradients are written token-major into dqkv [B, T, 3*H*32] = [dq | dk | dv], the layout the in_proj backward GEMMs read.
//
//   attn_bwd_prep_kernel   D[b,h,q]
//   attn_bwd_kv_kernel     one wave = 16 keys, loops over all queries:   dV^T += dO^T P,  dK^T += Q_s^T ds   (P, ds in the
//                          accumulator layout [q rows][key lane] are directly the B operands)
//   attn_bwd_q_kernel      one wave = 16 queries, loops over all keys:   dQ_s^T += K^T ds^T                 (transposed form,
//                          as the forward: per-lane query state)
//   (the last layer's single-query form lives in attn_token0.hip)
#include <cstdlib>

#include "common.h"
#include "dropout.h"

DropSite paths_make_drop_site(uint64_t key, float p);      // dropout.hip
int paths_attention_bwd_x6_launch(const float* q, const float* k, const float* v, const float* d_o, const float* lse, const float* dsum,
                                  const int64_t* num_ims, float* dqkv, void* images, int B, int T, int H, DropSite site, int kv_too,
                                  int planes, hipStream_t stream);       // attn_bwd_x6.hip

namespace {

constexpr int HD = 32;
constexpr int LD40 = 40;     // row stride for 16-byte row reads of the 16x16x4 A operand (conflict-free)
constexpr int LD36 = 36;     // row stride for 4-byte "column" reads
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

// D[b,h,q] = sum_dv dO[b,q,h*32+dv] * O[b,q,h*32+dv]; one wave per token row, 16 lanes per head
__global__ void __launch_bounds__(256)
attn_bwd_prep_kernel(const float* __restrict__ o, const float* __restrict__ d_o, float* __restrict__ dsum, int64_t rows, int T, int H) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float2 a = *reinterpret_cast<const float2*>(o + row * 128 + 2 * lane);
  const float2 g = *reinterpret_cast<const float2*>(d_o + row * 128 + 2 * lane);
  float s = a.x * g.x + a.y * g.y;
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
  if ((lane & 15) == 0) {
    const int64_t b = row / T, q = row % T;
    dsum[(b * H + (lane >> 4)) * T + q] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// dK, dV: workgroup = 64 keys (4 waves x 16) of one (slide, head); loop over query tiles of 16.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
attn_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                   const float* __restrict__ d_o /*[B,T,128]*/, const float* __restrict__ lse, const float* __restrict__ dsum,
                   const int64_t* __restrict__ num_ims, float* __restrict__ dqkv, int T, int H, DropSite drop) {
  __shared__ __attribute__((aligned(16))) float sQ40[16 * LD40], sQ36[16 * LD36], sG40[16 * LD40], sG36[16 * LD36];
  __shared__ float sLse[16], sD[16];
  const int b = blockIdx.z, head = blockIdx.y, k0 = blockIdx.x * 64;
  const int len = (int)num_ims[b] + 1;
  const DropWin dwin = drop_window(drop, drop_attn_row((uint64_t)b * H + head, T, 0));       // (this pair's T x T' mask elements: csrc/dropout.h)
  if (k0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kl = lane & 15, g4 = lane >> 4;
  const int64_t base = ((int64_t)b * H + head) * T * HD;
  const int key = k0 + wave * 16 + kl;
  const int keyc = min(key, T - 1);
  // B operands held in registers: K^T and V^T of this lane's key: element d = 16u + 4g + e
  f32x4 kreg[2], vreg[2];
  kreg[0] = *reinterpret_cast<const f32x4*>(k + base + (int64_t)keyc * HD + 4 * g4);
  kreg[1] = *reinterpret_cast<const f32x4*>(k + base + (int64_t)keyc * HD + 16 + 4 * g4);
  vreg[0] = *reinterpret_cast<const f32x4*>(v + base + (int64_t)keyc * HD + 4 * g4);
  vreg[1] = *reinterpret_cast<const f32x4*>(v + base + (int64_t)keyc * HD + 16 + 4 * g4);
  const bool key_ok = key < len;
  f32x4 dk[2], dv[2];
  dk[0] = dk[1] = dv[0] = dv[1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging registers: threads 0-127 carry Q_s (16 rows x 8 float4), 128-255 carry dO; threads 0-15 also lse / D.
  // The NEXT query tile is fetched into registers while the current one is being consumed from LDS (the loop was a
  // global round trip per 16 queries before).
  const int sr = (tid & 127) >> 3, sc4 = tid & 7;
  f32x4 stage;
  float st_lse = 0.f, st_d = 0.f;
  auto fetch = [&](int q0) {
    const int qi = min(q0 + sr, T - 1);
    if (tid < 128) {
      stage = *reinterpret_cast<const f32x4*>(q + base + (int64_t)qi * HD + 4 * sc4);
    } else {
      stage = *reinterpret_cast<const f32x4*>(d_o + ((int64_t)b * T + qi) * (H * HD) + head * HD + 4 * sc4);
      if (q0 + sr >= len) stage = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < 16) {
      const int qj = min(q0 + tid, T - 1);
      st_lse = lse[((int64_t)b * H + head) * T + qj];
      st_d = dsum[((int64_t)b * H + head) * T + qj];
    }
  };
  auto publish = [&]() {
    if (tid < 128) {
      *reinterpret_cast<f32x4*>(&sQ40[sr * LD40 + 4 * sc4]) = stage;
      *reinterpret_cast<f32x4*>(&sQ36[sr * LD36 + 4 * sc4]) = stage;
    } else {
      *reinterpret_cast<f32x4*>(&sG40[sr * LD40 + 4 * sc4]) = stage;
      *reinterpret_cast<f32x4*>(&sG36[sr * LD36 + 4 * sc4]) = stage;
    }
    if (tid < 16) { sLse[tid] = st_lse; sD[tid] = st_d; }
  };
  fetch(0);
  for (int q0 = 0; q0 < len; q0 += 16) {
    __syncthreads();                  // everyone is done reading the previous tile
    publish();
    __syncthreads();
    if (q0 + 16 < len) fetch(q0 + 16);
    // S[q][key] and dP[q][key]: A = Q_s / dO rows (16-byte reads), B = K^T / V^T registers
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const f32x4 aq = *reinterpret_cast<const f32x4*>(&sQ40[kl * LD40 + 16 * u + 4 * g4]);
      const f32x4 ag = *reinterpret_cast<const f32x4*>(&sG40[kl * LD40 + 16 * u + 4 * g4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s = mfma16(aq[e], kreg[u][e], s);
        dp = mfma16(ag[e], vreg[u][e], dp);
      }
    }
    // accumulator: column = this lane's key, rows q = q0 + 4 g + r
    // with attention dropout (drop.thr != 0): O = (P * m) V, so dV takes P * m and dP = (dO V^T) * m; m = mask / (1 - p)
    f32x4 p, ds;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ql = 4 * g4 + r;
      const bool ok = key_ok && (q0 + ql < len);
      const float pr = ok ? __builtin_amdgcn_exp2f(s[r] - sLse[ql]) : 0.f;
      const float m = drop.thr ? drop_mult_w(drop, dwin, drop_attn_row((uint64_t)b * H + head, T, min(q0 + ql, T - 1)) + (uint64_t)keyc) : 1.0f;
      ds[r] = LN2 * pr * (dp[r] * m - sD[ql]);
      p[r] = pr * m;
    }
    // dV^T[dv][key] += dO^T[dv][q] P[q][key] ; dK^T[d][key] += Q_s^T[d][q] ds[q][key]   (A: 4-byte column reads)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* gp = &sG36[(4 * g4 + r) * LD36 + kl];
      const float* qp = &sQ36[(4 * g4 + r) * LD36 + kl];
      dv[0] = mfma16(gp[0], p[r], dv[0]);
      dv[1] = mfma16(gp[16], p[r], dv[1]);
      dk[0] = mfma16(qp[0], ds[r], dk[0]);
      dk[1] = mfma16(qp[16], ds[r], dk[1]);
    }
  }
  if (key < len) {            // accumulator rows = feature 16 t + 4 g + r, column = key
    float* dst = dqkv + ((int64_t)b * T + key) * (3 * H * HD) + head * HD + 4 * g4;
    *reinterpret_cast<f32x4*>(dst + H * HD) = dk[0];
    *reinterpret_cast<f32x4*>(dst + H * HD + 16) = dk[1];
    *reinterpret_cast<f32x4*>(dst + 2 * H * HD) = dv[0];
    *reinterpret_cast<f32x4*>(dst + 2 * H * HD + 16) = dv[1];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// dQ: workgroup = 64 queries (4 waves x 16) of one (slide, head); loop over key tiles of 64 (transposed form).
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
attn_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                  const float* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ dsum,
                  const int64_t* __restrict__ num_ims, float* __restrict__ dqkv, int T, int H, DropSite drop) {
  __shared__ __attribute__((aligned(16))) float sK40[64 * LD40], sK36[64 * LD36], sV40[64 * LD40];
  const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 64;
  const int len = (int)num_ims[b] + 1;
  const DropWin dwin = drop_window(drop, drop_attn_row((uint64_t)b * H + head, T, 0));       // (this pair's T x T' mask elements: csrc/dropout.h)
  if (q0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ql = lane & 15, g4 = lane >> 4;
  const int64_t base = ((int64_t)b * H + head) * T * HD;
  const int qi = q0 + wave * 16 + ql;
  const int qc = min(qi, T - 1);
  f32x4 qreg[2], greg[2];
  qreg[0] = *reinterpret_cast<const f32x4*>(q + base + (int64_t)qc * HD + 4 * g4);
  qreg[1] = *reinterpret_cast<const f32x4*>(q + base + (int64_t)qc * HD + 16 + 4 * g4);
  const float* gp = d_o + ((int64_t)b * T + qc) * (H * HD) + head * HD;
  greg[0] = *reinterpret_cast<const f32x4*>(gp + 4 * g4);
  greg[1] = *reinterpret_cast<const f32x4*>(gp + 16 + 4 * g4);
  const float my_lse = lse[((int64_t)b * H + head) * T + qc];
  const float my_d = dsum[((int64_t)b * H + head) * T + qc];
  f32x4 dq[2];
  dq[0] = dq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nkt = (len + 63) / 64;
  // the NEXT key tile is fetched into registers while the current one is consumed from LDS (the loop was a global round trip per
  // 64 keys: 413 us per launch at T = 2049, 40 % of the f32-MFMA rate)
  f32x4 tk[2], tv[2];
  auto fetch = [&](int kt_) {
#pragma unroll
    for (int pss = 0; pss < 2; ++pss) {
      const int idx = tid + pss * 256, row = idx >> 3, c4 = idx & 7;
      const int key = kt_ * 64 + row, keyc = min(key, T - 1);
      tk[pss] = *reinterpret_cast<const f32x4*>(k + base + (int64_t)keyc * HD + 4 * c4);
      tv[pss] = *reinterpret_cast<const f32x4*>(v + base + (int64_t)keyc * HD + 4 * c4);
      if (key >= len) tk[pss] = tv[pss] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  fetch(0);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
#pragma unroll
    for (int pss = 0; pss < 2; ++pss) {
      const int idx = tid + pss * 256, row = idx >> 3, c4 = idx & 7;
      *reinterpret_cast<f32x4*>(&sK40[row * LD40 + 4 * c4]) = tk[pss];
      *reinterpret_cast<f32x4*>(&sK36[row * LD36 + 4 * c4]) = tk[pss];
      *reinterpret_cast<f32x4*>(&sV40[row * LD40 + 4 * c4]) = tv[pss];
    }
    __syncthreads();
    if (kt + 1 < nkt) fetch(kt + 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 ak = *reinterpret_cast<const f32x4*>(&sK40[(16 * t + ql) * LD40 + 16 * u + 4 * g4]);
        const f32x4 av = *reinterpret_cast<const f32x4*>(&sV40[(16 * t + ql) * LD40 + 16 * u + 4 * g4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = mfma16(ak[e], qreg[u][e], s);          // S^T[key][q]
          dp = mfma16(av[e], greg[u][e], dp);        // dP^T[key][q]
        }
      }
      f32x4 ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 64 + 16 * t + 4 * g4 + r;
        const float p = key < len ? __builtin_amdgcn_exp2f(s[r] - my_lse) : 0.f;
        const float m = drop.thr ? drop_mult_w(drop, dwin, drop_attn_row((uint64_t)b * H + head, T, qc) + (uint64_t)min(key, T - 1)) : 1.0f;
        ds[r] = LN2 * p * (dp[r] * m - my_d);
      }
      const float* kp = &sK36[(16 * t + 4 * g4) * LD36 + ql];     // dQ_s^T[d][q] += K^T[d][key] ds^T[key][q]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dq[0] = mfma16(kp[r * LD36], ds[r], dq[0]);
        dq[1] = mfma16(kp[r * LD36 + 16], ds[r], dq[1]);
      }
    }
  }
  if (qi < len) {
    float* dst = dqkv + ((int64_t)b * T + qi) * (3 * H * HD) + head * HD + 4 * g4;
    *reinterpret_cast<f32x4*>(dst) = dq[0];
    *reinterpret_cast<f32x4*>(dst + 16) = dq[1];
  }
}

}  // namespace

extern "C" {

// dqkv [B,T,384] must be zero-initialised by the caller (rows of padded tokens are never written).
static int attention_bwd_impl(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                              const int64_t* num_ims, float* dqkv, float* ws_dsum, int B, int T, int H, int head_dim, DropSite site,
                              hipStream_t stream, void* images = nullptr, int planes = 3);

int paths_attention_bwd_f32(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                            const int64_t* num_ims, float* dqkv, float* ws_dsum /*[B*H*T]*/, int B, int T, int H, int head_dim,
                            hipStream_t stream) {
  return attention_bwd_impl(q, k, v, o, d_o, lse, num_ims, dqkv, ws_dsum, B, T, H, head_dim, paths_make_drop_site(0, 0.f), stream);
}

// the same with the forward's attention-probability dropout (paths_attention_x6_dropout: same drop_key, same p)
int paths_attention_bwd_f32_dropout(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                                    const int64_t* num_ims, float* dqkv, float* ws_dsum, int B, int T, int H, int head_dim,
                                    uint64_t drop_key, float drop_p, hipStream_t stream) {
  PATHS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention_bwd_dropout: p must be in [0, 1)");
  return attention_bwd_impl(q, k, v, o, d_o, lse, num_ims, dqkv, ws_dsum, B, T, H, head_dim, paths_make_drop_site(drop_key, drop_p), stream);
}

// the whole backward on the split-bf16 kernels (paths_attention_bwd_x6_workspace bytes of images), three exact planes
int paths_attention_bwd_x6_dropout(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                                   const int64_t* num_ims, float* dqkv, float* ws_dsum, void* images, int B, int T, int H, int head_dim,
                                   uint64_t drop_key, float drop_p, hipStream_t stream) {
  PATHS_REQUIRE(drop_p >= 0.f && drop_p < 1.f && images != nullptr && (uintptr_t)images % 16 == 0, "attention_bwd_x6: p in [0, 1), 16-byte aligned images");
  return attention_bwd_impl(q, k, v, o, d_o, lse, num_ims, dqkv, ws_dsum, B, T, H, head_dim, paths_make_drop_site(drop_key, drop_p), stream, images);
}

// the same with the operand split chosen by the caller: planes 3 = hi | mid | lo (exact fp32 products, 6 MFMAs per block), planes 2 =
// hi | mid (16 significant bits at fp32's exponent range, 3 MFMAs per block: the setting of the training step's other gradient
// products, PATHS_TRAIN_PLANES=4; relative error of a product ~2e-5)
int paths_attention_bwd_x6_planes(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                                  const int64_t* num_ims, float* dqkv, float* ws_dsum, void* images, int B, int T, int H, int head_dim,
                                  uint64_t drop_key, float drop_p, int planes, hipStream_t stream) {
  PATHS_REQUIRE(drop_p >= 0.f && drop_p < 1.f && images != nullptr && (uintptr_t)images % 16 == 0, "attention_bwd_x6: p in [0, 1), 16-byte aligned images");
  PATHS_REQUIRE(planes == 2 || planes == 3, "attention_bwd_x6: planes must be 3 (exact) or 2 (hi | mid)");
  return attention_bwd_impl(q, k, v, o, d_o, lse, num_ims, dqkv, ws_dsum, B, T, H, head_dim, paths_make_drop_site(drop_key, drop_p), stream, images, planes);
}

static int attention_bwd_impl(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                              const int64_t* num_ims, float* dqkv, float* ws_dsum, int B, int T, int H, int head_dim, DropSite site,
                              hipStream_t stream, void* images, int planes) {
  PATHS_REQUIRE(head_dim == HD && H == 4, "attention_bwd: head_dim must be 32 and H 4");
  PATHS_REQUIRE(B > 0 && T > 0 && q && k && v && o && d_o && lse && num_ims && dqkv && ws_dsum, "attention_bwd: bad arguments");
  const int64_t rows = (int64_t)B * T;
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, o, d_o, ws_dsum, rows, T, H);
  PATHS_LAUNCH_CHECK("attention_bwd(prep)");
  dim3 grid((T + 63) / 64, H, B);
  static const int x6_kv = getenv("PATHS_ATTN_BWD_KV_X6") == nullptr || atoi(getenv("PATHS_ATTN_BWD_KV_X6")) != 0;   // A/B switch
  if (images != nullptr && x6_kv) return paths_attention_bwd_x6_launch(q, k, v, d_o, lse, ws_dsum, num_ims, dqkv, images, B, T, H, site, 1, planes, stream);
  hipLaunchKernelGGL(attn_bwd_kv_kernel, grid, dim3(256), 0, stream, q, k, v, d_o, lse, ws_dsum, num_ims, dqkv, T, H, site);
  PATHS_LAUNCH_CHECK("attention_bwd(kv)");
  if (images != nullptr) return paths_attention_bwd_x6_launch(q, k, v, d_o, lse, ws_dsum, num_ims, dqkv, images, B, T, H, site, 0, planes, stream);
  hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(256), 0, stream, q, k, v, d_o, lse, ws_dsum, num_ims, dqkv, T, H, site);
  PATHS_LAUNCH_CHECK("attention_bwd(q)");
  return PATHS_OK;
}

}  // extern "C"
