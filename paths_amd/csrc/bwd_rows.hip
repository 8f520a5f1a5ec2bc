// Row-wise / element-wise kernels of the backward pass (HBM-bound streaming work; 16-byte accesses, wavefront
// reductions).  They implement the derivative formulas of
//   LSTMCell.forward           reference model/interface.py:49-56  (c1 = c0 f + r m ; h1 = o tanh(Wc c1 + bc))
//   importance MLP + scaling   reference model/paths.py:95-98, utils.py:106-115
//   nn.LayerNorm               (post-LN decoder layers + decoder.norm of the nn.Transformer at model/aggregator.py:25-33)
// as applied by autograd in the reference train step (train.py:65).
#include "common.h"

namespace {

// ---- LSTM, phase A: from dh1 (gradient of h1 incl. the residual Y = X + h1) to the two pre-activation gradients
//   dpre_o = dh1 * tc * o (1 - o)        (o = sigmoid(.))        -> dG[:, 3Hc:]
//   dpre_h = dh1 * o * (1 - tc^2)        (tc = tanh(Wc c1 + bc))
// padded rows (idx >= num_ims) get zeros so that the weight-gradient GEMMs see no garbage.
__global__ void __launch_bounds__(256)
lstm_bwd_a_kernel(const float* __restrict__ dh1, int64_t ldd, const float* __restrict__ dh1b, int64_t lddb,
                  const float* __restrict__ o, const float* __restrict__ tc, const int64_t* __restrict__ num_ims,
                  int rows_per_slide, int D, float* __restrict__ dpre_o, int64_t ldo, float* __restrict__ dpre_h) {
  const int64_t row = blockIdx.x;
  const int b = (int)(row / rows_per_slide), idx = (int)(row % rows_per_slide);
  const bool valid = idx < (int)num_ims[b];
  for (int i = threadIdx.x; i < D / 4; i += 256) {
    f32x4 po{0.f, 0.f, 0.f, 0.f}, ph{0.f, 0.f, 0.f, 0.f};
    if (valid) {
      f32x4 g = *reinterpret_cast<const f32x4*>(dh1 + row * ldd + 4 * i);
      if (dh1b) g += *reinterpret_cast<const f32x4*>(dh1b + row * lddb + 4 * i);
      const f32x4 ov = *reinterpret_cast<const f32x4*>(o + row * D + 4 * i);
      const f32x4 tv = *reinterpret_cast<const f32x4*>(tc + row * D + 4 * i);
      po = g * tv * ov * (1.0f - ov);
      ph = g * ov * (1.0f - tv * tv);
    }
    *reinterpret_cast<f32x4*>(dpre_o + row * ldo + 4 * i) = po;
    *reinterpret_cast<f32x4*>(dpre_h + row * D + 4 * i) = ph;
  }
}

// ---- LSTM, phase B: dc1 = dc1_h (+ dc1_ext) ; packed gate gradients [df | dr | dm] per 32-unit block (same column
// packing as the gate weights) and dc0 = dc1 * f.
__global__ void __launch_bounds__(256)
lstm_bwd_b_kernel(const float* __restrict__ dc1_h, const float* __restrict__ dc1_ext, int64_t lde,
                  const float* __restrict__ frm /*[M,3Hc] packed*/, const float* __restrict__ c0, int64_t ldc0,
                  const int64_t* __restrict__ num_ims, int rows_per_slide, int Hc,
                  float* __restrict__ dg, int64_t ldg, float* __restrict__ dc0, int64_t lddc0) {
  const int64_t row = blockIdx.x;
  const int b = (int)(row / rows_per_slide), idx = (int)(row % rows_per_slide);
  const bool valid = idx < (int)num_ims[b];
  for (int j = threadIdx.x; j < Hc; j += 256) {
    const int blk = j >> 5, jj = j & 31;
    float df = 0.f, dr = 0.f, dm = 0.f, d0 = 0.f;
    if (valid) {
      float dc = dc1_h[row * Hc + j];
      if (dc1_ext) dc += dc1_ext[row * lde + j];
      const float f = frm[row * 3 * Hc + blk * 96 + jj];
      const float r = frm[row * 3 * Hc + blk * 96 + 32 + jj];
      const float m = frm[row * 3 * Hc + blk * 96 + 64 + jj];
      const float cp = c0 ? c0[row * ldc0 + j] : 0.f;
      df = dc * cp * f * (1.0f - f);
      dr = dc * m * r * (1.0f - r);
      dm = dc * r * (1.0f - m * m);
      d0 = dc * f;
    }
    dg[row * ldg + blk * 96 + jj] = df;
    dg[row * ldg + blk * 96 + 32 + jj] = dr;
    dg[row * ldg + blk * 96 + 64 + jj] = dm;
    if (dc0) dc0[row * lddc0 + j] = d0;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- importance + projection backward, one wave per patch row (width d = Hi = 128 -> 2 values per lane)
//   tokens = alpha * P + bp + PE ; alpha = valid * sigmoid(a) ; a = w2 . hid + b2 ; hid = relu(Y W1^T + b1)
//   dU[row] = [ dhid (128) | dP (128) ] feeds dY = dU W_ip and dW_ip = dU^T Y ;  da[row] for dw2 / db2
__global__ void __launch_bounds__(256)
imp_bwd_kernel(const float* __restrict__ dtok /*[B,T,128]*/, const float* __restrict__ pproj, const float* __restrict__ hid,
               const float* __restrict__ alpha, const float* __restrict__ w2, const int64_t* __restrict__ num_ims,
               int rows_per_slide, int64_t M, int imp_mul, float* __restrict__ du /*[M,256]*/, float* __restrict__ da,
               float* __restrict__ dah /*[M,128] = da * hid*/) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int b = (int)(row / rows_per_slide), idx = (int)(row % rows_per_slide);
  const bool valid = idx < (int)num_ims[b];
  const float* dt = dtok + ((int64_t)b * (rows_per_slide + 1) + idx + 1) * 128;
  float2 g = valid ? *reinterpret_cast<const float2*>(dt + 2 * lane) : float2{0.f, 0.f};
  const float2 pv = *reinterpret_cast<const float2*>(pproj + row * 128 + 2 * lane);
  const float2 hv = *reinterpret_cast<const float2*>(hid + row * 128 + 2 * lane);
  const float a = alpha[row];
  float dalpha = 0.f;
  float2 dp = g;
  if (imp_mul) {
    dalpha = wave_sum(g.x * pv.x + g.y * pv.y);
    dp.x = a * g.x; dp.y = a * g.y;
  }
  const float dz = valid ? dalpha * a * (1.0f - a) : 0.f;      // through sigmoid; padded rows have alpha == 0 by mask
  const float2 wv = *reinterpret_cast<const float2*>(w2 + 2 * lane);
  float2 dh = {hv.x > 0.f ? dz * wv.x : 0.f, hv.y > 0.f ? dz * wv.y : 0.f};
  *reinterpret_cast<float2*>(du + row * 256 + 2 * lane) = dh;
  *reinterpret_cast<float2*>(du + row * 256 + 128 + 2 * lane) = dp;
  *reinterpret_cast<float2*>(dah + row * 128 + 2 * lane) = float2{dz * hv.x, dz * hv.y};
  if (lane == 0) da[row] = dz;
}

// lstm = false variant (reference model/paths.py:95-109): Z = alpha * X (+ hctx), alpha = valid * sigmoid(w2 . relu(X W1^T + b1) + b2).
//   dalpha[row] = dZ[row] . X[row] (D columns), dz = valid * dalpha * alpha (1 - alpha);  dh [M,128] = (hid > 0) dz w2 feeds
//   dW1 = dh^T X and db1; dah = dz * hid feeds dw2; da = dz feeds db2.  One wave per row.
__global__ void __launch_bounds__(256)
imp_rows_bwd_kernel(const float* __restrict__ dz_rows /*[M,D]*/, const float* __restrict__ x /*[M,D]*/, int D, const float* __restrict__ hid,
                    const float* __restrict__ alpha, const float* __restrict__ w2, const int64_t* __restrict__ num_ims, int rows_per_slide,
                    int64_t M, float* __restrict__ dh /*[M,128]*/, float* __restrict__ da, float* __restrict__ dah) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int b = (int)(row / rows_per_slide), idx = (int)(row % rows_per_slide);
  const bool valid = idx < (int)num_ims[b];
  float acc = 0.f;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(dz_rows + row * D);
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x + row * D);
  for (int i = lane; i < D / 4; i += 64) { const f32x4 g = g4[i], v = x4[i]; acc += (g[0] * v[0] + g[1] * v[1]) + (g[2] * v[2] + g[3] * v[3]); }
  const float dalpha = wave_sum(acc);
  const float a = alpha[row];
  const float dz = valid ? dalpha * a * (1.0f - a) : 0.f;
  const float2 hv = *reinterpret_cast<const float2*>(hid + row * 128 + 2 * lane);
  const float2 wv = *reinterpret_cast<const float2*>(w2 + 2 * lane);
  *reinterpret_cast<float2*>(dh + row * 128 + 2 * lane) = float2{hv.x > 0.f ? dz * wv.x : 0.f, hv.y > 0.f ? dz * wv.y : 0.f};
  *reinterpret_cast<float2*>(dah + row * 128 + 2 * lane) = float2{dz * hv.x, dz * hv.y};
  if (lane == 0) da[row] = dz;
}

// ---- LayerNorm forward with saved statistics (recompute pass of the backward) and backward; width 128, one wave/row
__global__ void __launch_bounds__(256)
ln_fwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ add /*[128] or null: x + add first*/,
                    const float* __restrict__ g, const float* __restrict__ bta, float* __restrict__ y,
                    float* __restrict__ xhat, float* __restrict__ rstd_out, int64_t rows, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float2 v = *reinterpret_cast<const float2*>(x + row * 128 + 2 * lane);
  if (add) { v.x += add[2 * lane]; v.y += add[2 * lane + 1]; }
  const float mean = wave_sum(v.x + v.y) * (1.0f / 128);
  const float c0 = v.x - mean, c1 = v.y - mean;
  const float rstd = 1.0f / sqrtf(wave_sum(c0 * c0 + c1 * c1) * (1.0f / 128) + eps);
  const float2 xh = {c0 * rstd, c1 * rstd};
  *reinterpret_cast<float2*>(xhat + row * 128 + 2 * lane) = xh;
  if (y) *reinterpret_cast<float2*>(y + row * 128 + 2 * lane) = float2{xh.x * g[2 * lane] + bta[2 * lane], xh.y * g[2 * lane + 1] + bta[2 * lane + 1]};
  if (lane == 0) rstd_out[row] = rstd;
}

//   dx = rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)),  dyg = dy * gamma ;  also writes dy*xhat (for dgamma colsum)
__global__ void __launch_bounds__(256)
ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
              const float* __restrict__ g, float* __restrict__ dx, float* __restrict__ dyxhat, int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float2 d = *reinterpret_cast<const float2*>(dy + row * 128 + 2 * lane);
  const float2 xh = *reinterpret_cast<const float2*>(xhat + row * 128 + 2 * lane);
  const float2 dg = {d.x * g[2 * lane], d.y * g[2 * lane + 1]};
  const float m1 = wave_sum(dg.x + dg.y) * (1.0f / 128);
  const float m2 = wave_sum(dg.x * xh.x + dg.y * xh.y) * (1.0f / 128);
  const float rs = rstd[row];
  *reinterpret_cast<float2*>(dx + row * 128 + 2 * lane) = float2{rs * (dg.x - m1 - xh.x * m2), rs * (dg.y - m1 - xh.y * m2)};
  *reinterpret_cast<float2*>(dyxhat + row * 128 + 2 * lane) = float2{d.x * xh.x, d.y * xh.y};
}

// ln_bwd_kernel + the column sums the affine gradients need, in one pass: a workgroup owns rows_per_block rows (wave w takes
// rows w, w+4, ...), keeps sum(dy * xhat) (-> dgamma), sum(dy) (-> dbeta) and sum(dx) (-> the bias in front of this LayerNorm) of its rows in
// registers and writes one 384-float slab per workgroup (fixed order: deterministic); paths_reduce_slabs_f32 adds the slabs.  Replaces the dy*xhat tensor (written,
// then read back by a column-sum launch) and two column-sum launch pairs per LayerNorm.
__global__ void __launch_bounds__(256)
ln_bwd_sums_kernel(const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
                   const float* __restrict__ g, float* __restrict__ dx, float* __restrict__ slabs, int64_t rows, int rows_per_block) {
  __shared__ float2 pg[4][64], pb[4][64], px[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block, row1 = min(rows, row0 + rows_per_block);
  const float g0 = g[2 * lane], g1 = g[2 * lane + 1];
  float2 sg{0.f, 0.f}, sb{0.f, 0.f}, sx{0.f, 0.f};
  for (int64_t row = row0 + wave; row < row1; row += 4) {
    const float2 d = *reinterpret_cast<const float2*>(dy + row * 128 + 2 * lane);
    const float2 xh = *reinterpret_cast<const float2*>(xhat + row * 128 + 2 * lane);
    const float2 dg = {d.x * g0, d.y * g1};
    const float m1 = wave_sum(dg.x + dg.y) * (1.0f / 128);
    const float m2 = wave_sum(dg.x * xh.x + dg.y * xh.y) * (1.0f / 128);
    const float rs = rstd[row];
    const float2 o2{rs * (dg.x - m1 - xh.x * m2), rs * (dg.y - m1 - xh.y * m2)};
    *reinterpret_cast<float2*>(dx + row * 128 + 2 * lane) = o2;
    sg.x += d.x * xh.x; sg.y += d.y * xh.y;
    sb.x += d.x; sb.y += d.y;
    sx.x += o2.x; sx.y += o2.y;
  }
  pg[wave][lane] = sg; pb[wave][lane] = sb; px[wave][lane] = sx;
  __syncthreads();
  if (wave == 0) {
    float* o = slabs + (int64_t)blockIdx.x * 384;
    const float2 a = pg[0][lane], b = pg[1][lane], c = pg[2][lane], d = pg[3][lane];
    *reinterpret_cast<float2*>(o + 2 * lane) = float2{(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y)};
    const float2 e = pb[0][lane], f = pb[1][lane], h = pb[2][lane], k = pb[3][lane];
    *reinterpret_cast<float2*>(o + 128 + 2 * lane) = float2{(e.x + f.x) + (h.x + k.x), (e.y + f.y) + (h.y + k.y)};
    const float2 q0 = px[0][lane], q1 = px[1][lane], q2 = px[2][lane], q3 = px[3][lane];
    *reinterpret_cast<float2*>(o + 256 + 2 * lane) = float2{(q0.x + q1.x) + (q2.x + q3.x), (q0.y + q1.y) + (q2.y + q3.y)};
  }
}

}  // namespace

extern "C" {

int paths_lstm_bwd_a(const float* dh1, int64_t ldd, const float* dh1b, int64_t lddb, const float* o, const float* tc,
                     const int64_t* num_ims, int rows_per_slide, int64_t M, int D, float* dpre_o, int64_t ldo,
                     float* dpre_h, hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && D % 4 == 0 && ldd % 4 == 0 && ldo % 4 == 0 && num_ims && dh1 && o && tc && dpre_o && dpre_h, "lstm_bwd_a: bad arguments");
  hipLaunchKernelGGL(lstm_bwd_a_kernel, dim3((unsigned)M), dim3(256), 0, stream, dh1, ldd, dh1b, lddb, o, tc, num_ims, rows_per_slide, D, dpre_o, ldo, dpre_h);
  PATHS_LAUNCH_CHECK("lstm_bwd_a");
  return PATHS_OK;
}

int paths_lstm_bwd_b(const float* dc1_h, const float* dc1_ext, int64_t lde, const float* frm, const float* c0, int64_t ldc0,
                     const int64_t* num_ims, int rows_per_slide, int64_t M, int Hc, float* dg, int64_t ldg, float* dc0,
                     int64_t lddc0, hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && Hc % 32 == 0 && num_ims && dc1_h && frm && dg, "lstm_bwd_b: bad arguments");
  hipLaunchKernelGGL(lstm_bwd_b_kernel, dim3((unsigned)M), dim3(256), 0, stream, dc1_h, dc1_ext, lde, frm, c0, ldc0, num_ims, rows_per_slide, Hc, dg, ldg, dc0, lddc0);
  PATHS_LAUNCH_CHECK("lstm_bwd_b");
  return PATHS_OK;
}

int paths_importance_bwd(const float* dtok, const float* pproj, const float* hid, const float* alpha, const float* w2,
                         const int64_t* num_ims, int rows_per_slide, int64_t M, int imp_mul, float* du, float* da, float* dah,
                         hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && dtok && pproj && hid && alpha && w2 && num_ims && du && da && dah, "importance_bwd: bad arguments");
  hipLaunchKernelGGL(imp_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, dtok, pproj, hid, alpha, w2, num_ims, rows_per_slide, M, imp_mul, du, da, dah);
  PATHS_LAUNCH_CHECK("importance_bwd");
  return PATHS_OK;
}

int paths_importance_rows_bwd(const float* dz_rows, const float* x, int D, const float* hid, const float* alpha, const float* w2,
                              const int64_t* num_ims, int rows_per_slide, int64_t M, float* dh, float* da, float* dah, hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && D % 4 == 0 && dz_rows && x && hid && alpha && w2 && num_ims && dh && da && dah, "importance_rows_bwd: bad arguments");
  hipLaunchKernelGGL(imp_rows_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, dz_rows, x, D, hid, alpha, w2, num_ims, rows_per_slide, M, dh, da, dah);
  PATHS_LAUNCH_CHECK("importance_rows_bwd");
  return PATHS_OK;
}

int paths_layernorm_fwd_stats(const float* x, const float* add, const float* gamma, const float* beta, float* y, float* xhat,
                              float* rstd, int64_t rows, int d, float eps, hipStream_t stream) {
  PATHS_REQUIRE(rows > 0 && d == 128 && x && xhat && rstd, "layernorm_fwd_stats: bad arguments");
  hipLaunchKernelGGL(ln_fwd_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, add, gamma, beta, y, xhat, rstd, rows, eps);
  PATHS_LAUNCH_CHECK("layernorm_fwd_stats");
  return PATHS_OK;
}

int paths_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx, float* dyxhat,
                        int64_t rows, int d, hipStream_t stream) {
  PATHS_REQUIRE(rows > 0 && d == 128 && dy && xhat && rstd && gamma && dx && dyxhat, "layernorm_bwd: bad arguments");
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, dy, xhat, rstd, gamma, dx, dyxhat, rows);
  PATHS_LAUNCH_CHECK("layernorm_bwd");
  return PATHS_OK;
}

// LayerNorm backward with the affine-gradient partial sums: slabs [ceil(rows / rows_per_block)][384] = sum(dy * xhat) | sum(dy) | sum(dx)
int paths_layernorm_bwd_sums(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx, float* slabs,
                             int64_t rows, int d, int rows_per_block, hipStream_t stream) {
  PATHS_REQUIRE(rows > 0 && d == 128 && rows_per_block >= 4 && dy && xhat && rstd && gamma && dx && slabs, "layernorm_bwd_sums: bad arguments");
  hipLaunchKernelGGL(ln_bwd_sums_kernel, dim3((unsigned)((rows + rows_per_block - 1) / rows_per_block)), dim3(256), 0, stream,
                     dy, xhat, rstd, gamma, dx, slabs, rows, rows_per_block);
  PATHS_LAUNCH_CHECK("layernorm_bwd_sums");
  return PATHS_OK;
}

}  // extern "C"
