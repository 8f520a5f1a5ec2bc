// Self-attention for WIDE heads (head_dim > 64: e.g. trans_dim 256 / 2 heads, or the stress form trans_dim 1536 / 4 heads = 384 of
// SURVEY 8(d); reference model/aggregator.py:25-33 accepts any trans_dim % trans_heads == 0, config.py:30-36), forward and backward.
//
// The flash-style kernels of generic.hip / generic_bwd.hip keep a query's (or key's) whole head row in registers, which stops at
// head_dim 64.  Wide heads take the plain three-step form instead, per (slide, head), with the score matrix in a scratch buffer:
//     S = q k^T                      paths_gemm_nt_f32 (f32-input MFMA: exact fp32 chains), operands read in place from the token-major
//                                    in_proj output qkv [B*T, 3d]
//     P = softmax(qscale S)          wide_softmax_kernel: masked keys, log2-domain log-sum-exp, optional dropout on the probabilities
//     O = P V                        paths_gemm_nt_f32 against V^T (paths_transpose_f32)
// and for the backward  dP = dO V^T,  dS = qscale ln2 P (dP m - D),  dV = (P m)^T dO,  dK = dS^T q,  dQ = dS K  (GEMMs + one row kernel).
// ~4 launches per (slide, head) forward, ~10 backward: a correctness-first path for geometries outside the tuned ones - the same
// conventions as paths_attention_any_train / paths_attention_bwd_any (q unscaled in qkv, lse in the log2 domain, mask element
// ((slide*H + head)*T + query)*T + key, rows of dqkv the call does not own left untouched).
#include "common.h"
#include "dropout.h"

DropSite paths_make_drop_site(uint64_t key, float p);      // dropout.hip
extern "C" int paths_gemm_nt_f32(const float* a, int64_t lda, const float* w, int64_t ldw, const float* b, float* out, int64_t ldo,
                                 int M, int N, int Npad, int K, int act, const float* residual, int64_t ldr, const float* mask,
                                 int64_t ldm, int accumulate, hipStream_t stream);
extern "C" int paths_gemm_tn_f32(const float* a, int64_t lda, const float* b0, int64_t ldb0, int nb0, const float* b1, int64_t ldb1,
                                 float* out, int64_t ldo, int M, int N1, int N2, int splits, int accumulate, float* workspace,
                                 hipStream_t stream);
extern "C" int paths_transpose_f32(const float* in, int64_t ldi, int R, int C, float* out, int64_t ldo, hipStream_t stream);

namespace {

constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One wave per query row of S [nq, Tp] (in place): P[c] = exp2(qscale S[c] - lse) for c < len, 0 for len <= c < Tp.
// lse_in == nullptr: the softmax statistics are computed here (and written to lse_out if given); otherwise the saved ones are used
// (the backward's recompute).  drop.thr != 0 and `dropped`: the probabilities that enter the PV product, P * mask / (1 - p), are
// written to `dropped` [nq, Tp] (may alias S: the forward only needs those); the backward passes a second buffer and keeps both.
__global__ void __launch_bounds__(256)
wide_softmax_kernel(float* __restrict__ S, int Tp, int nq, const int64_t* __restrict__ num_ims, int b, float qscale, const float* __restrict__ lse_in,
                    float* __restrict__ lse_out, float* __restrict__ dropped, DropSite drop, uint64_t site_base /* ((b*H + h)*T) */, int T) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nq) return;
  const int len = min((int)num_ims[b] + 1, T);
  float* row = S + (int64_t)r * Tp;
  if (r >= len) {                                      // a padded query: no probabilities (its q row may be anything)
    for (int c = lane; c < Tp; c += 64) { row[c] = 0.f; if (dropped != nullptr) dropped[(int64_t)r * Tp + c] = 0.f; }
    if (lse_in == nullptr && lse_out != nullptr && lane == 0) lse_out[r] = 0.f;
    return;
  }
  float L;
  if (lse_in != nullptr) {
    L = lse_in[r];
  } else {
    float m = -INFINITY;
    for (int c = lane; c < len; c += 64) m = fmaxf(m, row[c] * qscale);
    m = wave_max(m);                                   // finite: key 0 (the special token) is always valid
    float l = 0.f;
    for (int c = lane; c < len; c += 64) l += __builtin_amdgcn_exp2f(row[c] * qscale - m);
    l = wave_sum(l);
    L = m + __builtin_amdgcn_logf(l);                  // v_log_f32 = log2
    if (lse_out != nullptr && lane == 0) lse_out[r] = L;
  }
  const uint64_t mrow = (site_base + (uint64_t)r) * drop_attn_stride(T);
  float* drow = dropped != nullptr ? dropped + (int64_t)r * Tp : nullptr;
  for (int c = lane; c < Tp; c += 64) {
    const float p = c < len ? __builtin_amdgcn_exp2f(row[c] * qscale - L) : 0.f;
    if (drow != nullptr) {
      const float m = (drop.thr != 0u && c < len) ? drop_mult(drop, mrow + (uint64_t)c) : 1.f;
      if (drow != row) row[c] = p;
      drow[c] = p * m;
    } else {
      row[c] = p;
    }
  }
}

// One wave per query row: D = sum_c dO[c] O[c] over the head's columns; dS[c] = qscale ln2 P[c] (dP[c] m - D) written over dP, P m over P.
__global__ void __launch_bounds__(256)
wide_ds_kernel(float* __restrict__ P, float* __restrict__ dP, int Tp, int nq, const int64_t* __restrict__ num_ims, int b, float qscale,
               const float* __restrict__ o, const float* __restrict__ d_o, int64_t ldo, int hd, DropSite drop, uint64_t site_base, int T) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nq) return;
  const int len = min((int)num_ims[b] + 1, T);
  if (r >= len) {                                      // a padded query contributes nothing
    for (int c = lane; c < Tp; c += 64) { P[(int64_t)r * Tp + c] = 0.f; dP[(int64_t)r * Tp + c] = 0.f; }
    return;
  }
  float D = 0.f;
  for (int c = lane; c < hd; c += 64) D = fmaf(d_o[(int64_t)r * ldo + c], o[(int64_t)r * ldo + c], D);
  D = wave_sum(D);
  const uint64_t mrow = (site_base + (uint64_t)r) * drop_attn_stride(T);
  float* prow = P + (int64_t)r * Tp;
  float* drow = dP + (int64_t)r * Tp;
  for (int c = lane; c < Tp; c += 64) {
    if (c < len) {
      const float m = drop.thr != 0u ? drop_mult(drop, mrow + (uint64_t)c) : 1.f;
      const float p = prow[c];
      drow[c] = qscale * LN2 * p * (drow[c] * m - D);
      prow[c] = p * m;
    } else {
      drow[c] = 0.f;
      prow[c] = 0.f;
    }
  }
}

// dst[r][c] = src[r][c] for r < rows, c < cols (the [keys, head_dim] gradient blocks written into the token-major dqkv)
__global__ void __launch_bounds__(256)
wide_copy2d_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd, int rows, int cols) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds + c];
}

struct WideScratch { float *S, *dP, *XT, *OUT; int Tp, Np, hdp; };
inline int rnd(int x, int m) { return (x + m - 1) / m * m; }
inline WideScratch carve(float* ws, int T, int hd) {
  WideScratch w;
  w.Tp = rnd(T, 128);                                // row stride of the score buffers = K of the PV-type products = padded N of the score products
  w.Np = w.Tp;
  w.hdp = rnd(hd, 128);
  w.S = ws;
  w.dP = w.S + (int64_t)T * w.Tp;
  w.XT = w.dP + (int64_t)T * w.Tp;
  w.OUT = w.XT + (int64_t)w.hdp * w.Tp;
  return w;
}

}  // namespace

extern "C" {

// floats of scratch for the two calls below (two [T, Tp] score buffers, a transposed [hd_pad, Tp] operand, a [Tp, hd_pad] output)
int64_t paths_attention_wide_workspace(int T, int head_dim) {
  const int64_t Tp = rnd(T, 128), hdp = rnd(head_dim, 128);
  return 2 * (int64_t)T * Tp + 2 * hdp * Tp + 64;
}

// Forward for head_dim > 64 (a multiple of 32).  qkv [B*T, 3d] token-major, q unscaled, with AT LEAST 128 READABLE ROWS behind its
// last one (the score product reads whole 128-row tiles of k); o [B, T, d]; lse [B, H, T] (log2 domain) or null; max_queries > 0:
// only queries [0, max_queries) (the last layer: token 0); dropout p on the probabilities (site drop_key), p = 0: none.
int paths_attention_wide_fwd(const float* qkv, int64_t ld, float* o, float* lse, const int64_t* num_ims, int B, int T, int H, int head_dim,
                             float qscale, int max_queries, uint64_t drop_key, float drop_p, float* workspace, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && T > 0 && H > 0 && qkv && o && num_ims && workspace, "attention_wide_fwd: bad arguments");
  PATHS_REQUIRE(head_dim > 64 && head_dim % 32 == 0 && head_dim <= 1024, "attention_wide_fwd: head_dim %d (this entry: multiples of 32 in (64, 1024])", head_dim);
  PATHS_REQUIRE(ld % 4 == 0 && ((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)workspace) % 16 == 0, "attention_wide_fwd: 16-byte aligned buffers, ld %% 4 == 0");
  PATHS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention_wide_fwd: p must be in [0, 1)");
  const int d = H * head_dim, nq = max_queries > 0 && max_queries < T ? max_queries : T;
  const WideScratch w = carve(workspace, T, head_dim);
  const DropSite site = paths_make_drop_site(drop_key, drop_p);
  if (hipMemsetAsync(w.XT, 0, (size_t)w.hdp * w.Tp * 4, stream) != hipSuccess) return paths_set_error(PATHS_ELAUNCH, "attention_wide_fwd: memset failed");
  for (int b = 0; b < B; ++b) {
    const int nqb = nq;                               // (padded queries / keys are handled on the device: num_ims is never read on the host)
    for (int h = 0; h < H; ++h) {
      const float* qp = qkv + (int64_t)b * T * ld + h * head_dim;
      int rc = paths_gemm_nt_f32(qp, ld, qp + d, ld, nullptr, w.S, w.Tp, nqb, T, w.Np, head_dim, 0, nullptr, 0, nullptr, 0, 0, stream);
      if (rc) return rc;
      float* lrow = lse ? lse + ((int64_t)b * H + h) * T : nullptr;
      hipLaunchKernelGGL(wide_softmax_kernel, dim3((nqb + 3) / 4), dim3(256), 0, stream, w.S, w.Tp, nqb, num_ims, b, qscale, (const float*)nullptr, lrow,
                         site.thr ? w.S : (float*)nullptr, site, ((uint64_t)b * H + h) * (uint64_t)T, T);
      PATHS_LAUNCH_CHECK("attention_wide_fwd(softmax)");
      rc = paths_transpose_f32(qp + 2 * d, ld, T, head_dim, w.XT, w.Tp, stream);
      if (rc) return rc;
      rc = paths_gemm_nt_f32(w.S, w.Tp, w.XT, w.Tp, nullptr, o + (int64_t)b * T * d + h * head_dim, d, nqb, head_dim, w.hdp, w.Tp, 0, nullptr, 0,
                             nullptr, 0, 0, stream);
      if (rc) return rc;
    }
  }
  return PATHS_OK;
}

// Backward of the same: dqkv [B*T, 3d] = [dq | dk | dv] (gradient of the UNSCALED q); dq rows >= max_queries are left as they are (the
// caller zero-fills dqkv, as for paths_attention_bwd_any), every dk / dv row is written (padded keys: zeros).  o = the forward's output, d_o its gradient, lse as saved.
int paths_attention_wide_bwd(const float* qkv, int64_t ld, const float* o, const float* d_o, const float* lse, const int64_t* num_ims,
                             float* dqkv, int B, int T, int H, int head_dim, float qscale, int max_queries, uint64_t drop_key, float drop_p,
                             float* workspace, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && T > 0 && H > 0 && qkv && o && d_o && lse && num_ims && dqkv && workspace, "attention_wide_bwd: bad arguments");
  PATHS_REQUIRE(head_dim > 64 && head_dim % 32 == 0 && head_dim <= 1024, "attention_wide_bwd: head_dim %d (this entry: multiples of 32 in (64, 1024])", head_dim);
  PATHS_REQUIRE(ld % 4 == 0 && ((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dqkv | (uintptr_t)workspace) % 16 == 0, "attention_wide_bwd: 16-byte aligned buffers");
  const int d = H * head_dim, nq = max_queries > 0 && max_queries < T ? max_queries : T;
  const WideScratch w = carve(workspace, T, head_dim);
  const DropSite site = paths_make_drop_site(drop_key, drop_p);
  if (hipMemsetAsync(w.XT, 0, (size_t)w.hdp * w.Tp * 4, stream) != hipSuccess) return paths_set_error(PATHS_ELAUNCH, "attention_wide_bwd: memset failed");
  float* const tn_ws = w.OUT + (int64_t)w.Tp * w.hdp;   // (unused by the single-split weight-gradient form; must be non-null)
  for (int b = 0; b < B; ++b) {
    const int nqb = nq;
    for (int h = 0; h < H; ++h) {
      const float* qp = qkv + (int64_t)b * T * ld + h * head_dim;
      const float* op = o + (int64_t)b * T * d + h * head_dim;
      const float* gp = d_o + (int64_t)b * T * d + h * head_dim;
      float* dq = dqkv + (int64_t)b * T * (3 * d) + h * head_dim;
      const uint64_t sb = ((uint64_t)b * H + h) * (uint64_t)T;
      int rc = paths_gemm_nt_f32(qp, ld, qp + d, ld, nullptr, w.S, w.Tp, nqb, T, w.Np, head_dim, 0, nullptr, 0, nullptr, 0, 0, stream);     // S
      if (rc) return rc;
      hipLaunchKernelGGL(wide_softmax_kernel, dim3((nqb + 3) / 4), dim3(256), 0, stream, w.S, w.Tp, nqb, num_ims, b, qscale, lse + ((int64_t)b * H + h) * T,
                         (float*)nullptr, (float*)nullptr, site, sb, T);                                                                  // P (un-dropped)
      PATHS_LAUNCH_CHECK("attention_wide_bwd(softmax)");
      rc = paths_gemm_nt_f32(gp, d, qp + 2 * d, ld, nullptr, w.dP, w.Tp, nqb, T, w.Np, head_dim, 0, nullptr, 0, nullptr, 0, 0, stream);   // dP = dO V^T
      if (rc) return rc;
      hipLaunchKernelGGL(wide_ds_kernel, dim3((nqb + 3) / 4), dim3(256), 0, stream, w.S, w.dP, w.Tp, nqb, num_ims, b, qscale, op, gp, (int64_t)d, head_dim,
                         site, sb, T);                                                                                                     // dS over dP, P m over P
      PATHS_LAUNCH_CHECK("attention_wide_bwd(ds)");
      // dV = (P m)^T dO and dK = dS^T q: [Tp, head_dim] blocks through OUT, rows [0, T) copied into dqkv (padded keys: exact zeros)
      rc = paths_gemm_tn_f32(w.S, w.Tp, gp, d, 0, nullptr, 0, w.OUT, w.hdp, nqb, w.Tp, head_dim, 1, 0, tn_ws, stream);
      if (rc) return rc;
      hipLaunchKernelGGL(wide_copy2d_kernel, dim3((unsigned)(((int64_t)T * head_dim + 255) / 256)), dim3(256), 0, stream, w.OUT, (int64_t)w.hdp,
                         dq + 2 * d, (int64_t)3 * d, T, head_dim);
      PATHS_LAUNCH_CHECK("attention_wide_bwd(dv)");
      rc = paths_gemm_tn_f32(w.dP, w.Tp, qp, ld, 0, nullptr, 0, w.OUT, w.hdp, nqb, w.Tp, head_dim, 1, 0, tn_ws, stream);
      if (rc) return rc;
      hipLaunchKernelGGL(wide_copy2d_kernel, dim3((unsigned)(((int64_t)T * head_dim + 255) / 256)), dim3(256), 0, stream, w.OUT, (int64_t)w.hdp,
                         dq + d, (int64_t)3 * d, T, head_dim);
      PATHS_LAUNCH_CHECK("attention_wide_bwd(dk)");
      rc = paths_transpose_f32(qp + d, ld, T, head_dim, w.XT, w.Tp, stream);                                                              // K^T
      if (rc) return rc;
      rc = paths_gemm_nt_f32(w.dP, w.Tp, w.XT, w.Tp, nullptr, dq, (int64_t)3 * d, nqb, head_dim, w.hdp, w.Tp, 0, nullptr, 0, nullptr, 0, 0, stream);   // dQ = dS K
      if (rc) return rc;
    }
  }
  return PATHS_OK;
}

}  // extern "C"
