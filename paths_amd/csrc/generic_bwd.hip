// Shape-generic kernels of the TRAINING path: any trans_dim (multiple of 32, <= 2048), head_dim in {16, 32, 48, 64} (wider heads: attn_wide.hip), any
// importance_mlp_hidden_dim (multiple of 4).  The shipped geometry (128 / 4 heads / 128) trains on the specialised kernels
// (bwd_rows.hip, attn_bwd.hip, attn_bwd_x6.hip); these evaluate the same derivatives for every other configuration of the
// reference's config surface (config.py:30-36: the dataclass default is trans_dim 192 = head_dim 48).  What they differentiate:
//   nn.LayerNorm                        (post-LN decoder layers + decoder.norm of the nn.Transformer, model/aggregator.py:25-33)
//   importance MLP + scaling + proj_in  (model/paths.py:95-98,119-124)
//   masked multi-head self-attention    (model/aggregator.py:70-72, nn.MultiheadAttention incl. its dropout on the probabilities)
// as autograd applies them in the reference train step (train.py:65).  Exact fp32 arithmetic, written for correctness and
// reasonable speed (VALU dot products over LDS-staged tiles), not tuned like the 128-wide path.
#include "common.h"
#include "dropout.h"

DropSite paths_make_drop_site(uint64_t key, float p);      // dropout.hip

namespace {

constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm, any width d <= 2048 (d % 4 == 0): one wave per row, lane l owns columns 4 l + 256 i, i < NI
// ---------------------------------------------------------------------------------------------------------------
constexpr int NI = 8;
__global__ void __launch_bounds__(256)
ln_fwd_stats_any_kernel(const float* __restrict__ x, const float* __restrict__ add, const float* __restrict__ g,
                        const float* __restrict__ bta, float* __restrict__ y, float* __restrict__ xhat,
                        float* __restrict__ rstd_out, int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[NI];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = 4 * lane + 256 * i;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < d) {
      v[i] = *reinterpret_cast<const f32x4*>(x + row * d + c);
      if (add) v[i] += *reinterpret_cast<const f32x4*>(add + c);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (4 * lane + 256 * i < d) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float c = v[i][e] - mean; q += c * c; }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = 4 * lane + 256 * i;
    if (c < d) {
      f32x4 xh;
#pragma unroll
      for (int e = 0; e < 4; ++e) xh[e] = (v[i][e] - mean) * rstd;
      *reinterpret_cast<f32x4*>(xhat + row * d + c) = xh;
      if (y) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c), bb = *reinterpret_cast<const f32x4*>(bta + c);
        *reinterpret_cast<f32x4*>(y + row * d + c) = xh * gg + bb;
      }
    }
  }
  if (lane == 0) rstd_out[row] = rstd;
}

//   dx = rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)),  dyg = dy * gamma.  SUMS: a workgroup owns rows_per_block rows and
//   writes one slab [sum dy*xhat | sum dy | sum dx] of 3 d floats (fixed order: deterministic); otherwise dy*xhat is written out.
template <bool SUMS>
__global__ void __launch_bounds__(256)
ln_bwd_any_kernel(const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
                  const float* __restrict__ g, float* __restrict__ dx, float* __restrict__ aux /*dyxhat | slabs*/, int64_t rows,
                  int d, int rows_per_block) {
  __shared__ f32x4 part[3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = SUMS ? (int64_t)blockIdx.x * rows_per_block : (int64_t)blockIdx.x * 4;
  const int64_t row1 = min(rows, row0 + (SUMS ? rows_per_block : 4));
  f32x4 gg[NI], sg[NI], sb[NI], sx[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = 4 * lane + 256 * i;
    gg[i] = c < d ? *reinterpret_cast<const f32x4*>(g + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    sg[i] = sb[i] = sx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float inv_d = 1.0f / (float)d;
  for (int64_t row = row0 + wave; row < row1; row += 4) {
    f32x4 dv[NI], xh[NI], dg[NI];
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = 4 * lane + 256 * i;
      dv[i] = xh[i] = dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < d) {
        dv[i] = *reinterpret_cast<const f32x4*>(dy + row * d + c);
        xh[i] = *reinterpret_cast<const f32x4*>(xhat + row * d + c);
        dg[i] = dv[i] * gg[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a1 += dg[i][e]; a2 += dg[i][e] * xh[i][e]; }
      }
    }
    const float m1 = wave_sum(a1) * inv_d, m2 = wave_sum(a2) * inv_d;
    const float rs = rstd[row];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = 4 * lane + 256 * i;
      if (c < d) {
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = rs * (dg[i][e] - m1 - xh[i][e] * m2);
        *reinterpret_cast<f32x4*>(dx + row * d + c) = o4;
        if constexpr (SUMS) { sg[i] += dv[i] * xh[i]; sb[i] += dv[i]; sx[i] += o4; }
        else *reinterpret_cast<f32x4*>(aux + row * d + c) = dv[i] * xh[i];
      }
    }
  }
  if constexpr (SUMS) {
    float* o = aux + (int64_t)blockIdx.x * 3 * d;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = 4 * lane + 256 * i;      // (uniform trip count: every wave passes the barriers)
      __syncthreads();
      part[0][wave][lane] = sg[i]; part[1][wave][lane] = sb[i]; part[2][wave][lane] = sx[i];
      __syncthreads();
      if (wave < 3 && c < d) {
        const f32x4 t = (part[wave][0][lane] + part[wave][1][lane]) + (part[wave][2][lane] + part[wave][3][lane]);
        *reinterpret_cast<f32x4*>(o + wave * d + c) = t;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// importance + projection backward for any (Hi, d), one wave per patch row (cf. bwd_rows.hip: imp_bwd_kernel)
//   tokens = alpha * P + bp + PE ; alpha = valid * sigmoid(a) ; a = w2 . hid + b2 ; hid = relu(Y W1^T + b1)
//   dU[row] = [ dhid (Hi) | dP (d) | 0 pad ] (row stride ldu) ; da[row] ; dah[row] = da * hid
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
imp_bwd_any_kernel(const float* __restrict__ dtok /*[B,T,d]*/, const float* __restrict__ pproj /*[M,d]*/, const float* __restrict__ hid /*[M,Hi]*/,
                   const float* __restrict__ alpha, const float* __restrict__ w2, const int64_t* __restrict__ num_ims,
                   int rows_per_slide, int64_t M, int imp_mul, int Hi, int d, int64_t ldu, float* __restrict__ du,
                   float* __restrict__ da, float* __restrict__ dah /*[M,Hi]*/) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int b = (int)(row / rows_per_slide), idx = (int)(row % rows_per_slide);
  const bool valid = idx < (int)num_ims[b];
  const float* dt = dtok + ((int64_t)b * (rows_per_slide + 1) + idx + 1) * d;
  const float a = alpha[row];
  float dalpha = 0.f;
  if (imp_mul) {
    float acc = 0.f;
    if (valid)
      for (int c = lane; c < d; c += 64) acc = fmaf(dt[c], pproj[row * d + c], acc);
    dalpha = wave_sum(acc);
  }
  for (int c = lane; c < d; c += 64) {
    const float gv = valid ? dt[c] : 0.f;
    du[row * ldu + Hi + c] = imp_mul ? a * gv : gv;
  }
  const float dz = valid ? dalpha * a * (1.0f - a) : 0.f;      // through the sigmoid; padded rows have alpha == 0 by the mask
  for (int c = lane; c < Hi; c += 64) {
    const float hv = hid[row * Hi + c];
    du[row * ldu + c] = hv > 0.f ? dz * w2[c] : 0.f;
    dah[row * Hi + c] = dz * hv;
  }
  for (int c = Hi + d + lane; c < ldu; c += 64) du[row * ldu + c] = 0.f;
  if (lane == 0) da[row] = dz;
}

// lstm = false variant (reference model/paths.py:95-109; cf. bwd_rows.hip: imp_rows_bwd_kernel) for any hidden width Hi:
//   dalpha[row] = dZ[row] . X[row] (D columns), dz = valid * dalpha * alpha (1 - alpha);  dh [M,Hi] = (hid > 0) dz w2, dah = dz * hid, da = dz
__global__ void __launch_bounds__(256)
imp_rows_bwd_any_kernel(const float* __restrict__ dz_rows /*[M,D]*/, const float* __restrict__ x /*[M,D]*/, int D, const float* __restrict__ hid,
                        const float* __restrict__ alpha, const float* __restrict__ w2, const int64_t* __restrict__ num_ims, int rows_per_slide,
                        int64_t M, int Hi, float* __restrict__ dh, float* __restrict__ da, float* __restrict__ dah) {
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
  for (int c = lane; c < Hi; c += 64) {
    const float hv = hid[row * Hi + c];
    dh[row * Hi + c] = hv > 0.f ? dz * w2[c] : 0.f;
    dah[row * Hi + c] = dz * hv;
  }
  if (lane == 0) da[row] = dz;
}

// ---------------------------------------------------------------------------------------------------------------
// attention backward, any head_dim: q, k, v read in place from the token-major in_proj output qkv [B*T, 3 d] (q UNscaled),
// S = (q qscale) . k in the log2 domain, P = exp2(S - lse), dropout mask m regenerated from (site, element index):
//    dP = (dO . v) m,  ds = ln2 P (dP - D),  D[q] = sum_dv dO O
//    dq = qscale ds K       dk = ds^T (q qscale)       dv = (P m)^T dO
// written token-major into dqkv [B*T, 3 d] = [dq | dk | dv] (rows the kernels do not own must be zero on entry).
//   attn_bwd_any_prep_kernel  D[b,h,q]
//   attn_bwd_any_kv_kernel    workgroup = 64 keys of one (slide, head); lane = key, the 4 waves split the queries of every
//                             staged tile (their q / dO rows are wave-uniform LDS broadcasts), partial dk / dv joined through LDS
//   attn_bwd_any_q_kernel     the same with the roles swapped: lane = query, waves split the keys of every staged tile
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
attn_bwd_any_prep_kernel(const float* __restrict__ o, const float* __restrict__ d_o, float* __restrict__ dsum, int64_t rows, int T,
                         int H, int hd) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (row, head)
  if (i >= rows * H) return;
  const int64_t row = i / H;
  const int h = (int)(i % H);
  const float* a = o + row * (int64_t)(H * hd) + h * hd;
  const float* g = d_o + row * (int64_t)(H * hd) + h * hd;
  float s = 0.f;
  for (int c = 0; c < hd; c += 4) {
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + c), gv = *reinterpret_cast<const f32x4*>(g + c);
    s += (av[0] * gv[0] + av[1] * gv[1]) + (av[2] * gv[2] + av[3] * gv[3]);
  }
  const int64_t b = row / T, q = row % T;
  dsum[(b * H + h) * T + q] = s;
}

constexpr int BT = 32;       // rows of the other side staged per tile

template <int HD>
__global__ void __launch_bounds__(256)
attn_bwd_any_kv_kernel(const float* __restrict__ qkv, int64_t ld, int d, float qscale, const float* __restrict__ d_o,
                       const float* __restrict__ lse, const float* __restrict__ dsum, const int64_t* __restrict__ num_ims,
                       float* __restrict__ dqkv, int T, int H, int max_q, DropSite drop) {
  __shared__ __attribute__((aligned(16))) float sQ[BT * HD], sG[BT * HD];
  __shared__ float sL[BT], sD[BT];
  __shared__ float red[64 * (HD + 1)];
  const int b = blockIdx.z, head = blockIdx.y, k0 = blockIdx.x * 64;
  const int len = min((int)num_ims[b] + 1, T);
  const DropWin dwin = drop_window(drop, drop_attn_row((uint64_t)b * H + head, T, 0));       // (this pair's T x T' mask elements: csrc/dropout.h)
  if (k0 >= len) return;
  const int nq = max_q > 0 ? min(len, max_q) : len;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int key = k0 + lane;
  const bool key_ok = key < len;
  const float* base = qkv + (int64_t)b * T * ld + head * HD;
  float kr[HD], vr[HD], dk[HD], dv[HD];
  {
    const float* kp = base + d + (int64_t)min(key, T - 1) * ld;
    const float* vp = base + 2 * d + (int64_t)min(key, T - 1) * ld;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + c), vv = *reinterpret_cast<const f32x4*>(vp + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) { kr[c + e] = kv[e]; vr[c + e] = vv[e]; dk[c + e] = 0.f; dv[c + e] = 0.f; }
    }
  }
  const uint64_t site_row = ((uint64_t)b * H + head) * T;
  for (int q0 = 0; q0 < nq; q0 += BT) {
    __syncthreads();
    for (int i = tid; i < BT * HD / 4; i += 256) {
      const int r = i / (HD / 4), c4 = i % (HD / 4);
      const int qi = min(q0 + r, T - 1);
      const f32x4 qv = *reinterpret_cast<const f32x4*>(base + (int64_t)qi * ld + 4 * c4) * qscale;
      const f32x4 gv = *reinterpret_cast<const f32x4*>(d_o + ((int64_t)b * T + qi) * d + head * HD + 4 * c4);
      *reinterpret_cast<f32x4*>(&sQ[r * HD + 4 * c4]) = qv;
      *reinterpret_cast<f32x4*>(&sG[r * HD + 4 * c4]) = gv;
    }
    if (tid < BT) {
      const int qi = min(q0 + tid, T - 1);
      sL[tid] = lse[((int64_t)b * H + head) * T + qi];
      sD[tid] = dsum[((int64_t)b * H + head) * T + qi];
    }
    __syncthreads();
    const int nr = min(BT, nq - q0);
    for (int r = wave; r < nr; r += 4) {
      const float* qrow = &sQ[r * HD];
      const float* grow = &sG[r * HD];
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) { s = fmaf(qrow[c], kr[c], s); dp = fmaf(grow[c], vr[c], dp); }
      const float p = key_ok ? __builtin_amdgcn_exp2f(s - sL[r]) : 0.f;
      const float m = drop.thr != 0u ? drop_mult_w(drop, dwin, (site_row + (uint64_t)(q0 + r)) * drop_attn_stride(T) + (uint64_t)key) : 1.f;
      const float pd = p * m;
      const float ds = LN2 * p * (dp * m - sD[r]);
#pragma unroll
      for (int c = 0; c < HD; ++c) { dv[c] = fmaf(pd, grow[c], dv[c]); dk[c] = fmaf(ds, qrow[c], dk[c]); }
    }
  }
  // join the four waves' partial sums (fixed order), one matrix at a time through red[key][HD + 1]
  float* out_k = dqkv + ((int64_t)b * T + min(key, T - 1)) * (3 * d) + d + head * HD;
  float* out_v = out_k + d;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    float* acc = which == 0 ? dk : dv;
    for (int w = 1; w < 4; ++w) {
      __syncthreads();
      if (wave == w) {
#pragma unroll
        for (int c = 0; c < HD; ++c) red[lane * (HD + 1) + c] = acc[c];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int c = 0; c < HD; ++c) acc[c] += red[lane * (HD + 1) + c];
      }
    }
    if (wave == 0 && key_ok) {
      float* op = which == 0 ? out_k : out_v;
#pragma unroll
      for (int c = 0; c < HD; c += 4) *reinterpret_cast<f32x4*>(op + c) = f32x4{acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
    }
  }
}

template <int HD>
__global__ void __launch_bounds__(256)
attn_bwd_any_q_kernel(const float* __restrict__ qkv, int64_t ld, int d, float qscale, const float* __restrict__ d_o,
                      const float* __restrict__ lse, const float* __restrict__ dsum, const int64_t* __restrict__ num_ims,
                      float* __restrict__ dqkv, int T, int H, int max_q, DropSite drop) {
  __shared__ __attribute__((aligned(16))) float sK[BT * HD], sV[BT * HD];
  __shared__ float red[64 * (HD + 1)];
  const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 64;
  const int len = min((int)num_ims[b] + 1, T);
  const DropWin dwin = drop_window(drop, drop_attn_row((uint64_t)b * H + head, T, 0));       // (this pair's T x T' mask elements: csrc/dropout.h)
  const int nq = max_q > 0 ? min(len, max_q) : len;
  if (q0 >= nq) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int query = q0 + lane;
  const bool q_ok = query < nq;
  const int qi = min(query, T - 1);
  const float* base = qkv + (int64_t)b * T * ld + head * HD;
  float qr[HD], gr[HD], dq[HD];
  {
    const float* qp = base + (int64_t)qi * ld;
    const float* gp = d_o + ((int64_t)b * T + qi) * d + head * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + c) * qscale, gv = *reinterpret_cast<const f32x4*>(gp + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) { qr[c + e] = qv[e]; gr[c + e] = gv[e]; dq[c + e] = 0.f; }
    }
  }
  const float L = lse[((int64_t)b * H + head) * T + qi], Dv = dsum[((int64_t)b * H + head) * T + qi];
  const uint64_t site_row = drop_attn_row((uint64_t)b * H + head, T, qi);
  for (int k0 = 0; k0 < len; k0 += BT) {
    __syncthreads();
    for (int i = tid; i < BT * HD / 4; i += 256) {
      const int r = i / (HD / 4), c4 = i % (HD / 4);
      const int ki = min(k0 + r, T - 1);
      *reinterpret_cast<f32x4*>(&sK[r * HD + 4 * c4]) = *reinterpret_cast<const f32x4*>(base + d + (int64_t)ki * ld + 4 * c4);
      *reinterpret_cast<f32x4*>(&sV[r * HD + 4 * c4]) = *reinterpret_cast<const f32x4*>(base + 2 * d + (int64_t)ki * ld + 4 * c4);
    }
    __syncthreads();
    const int nr = min(BT, len - k0);
    for (int r = wave; r < nr; r += 4) {
      const float* krow = &sK[r * HD];
      const float* vrow = &sV[r * HD];
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) { s = fmaf(qr[c], krow[c], s); dp = fmaf(gr[c], vrow[c], dp); }
      const float p = __builtin_amdgcn_exp2f(s - L);
      const float m = drop.thr != 0u ? drop_mult_w(drop, dwin, site_row + (uint64_t)(k0 + r)) : 1.f;
      const float ds = LN2 * p * (dp * m - Dv);
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] = fmaf(ds, krow[c], dq[c]);
    }
  }
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < HD; ++c) red[lane * (HD + 1) + c] = dq[c];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] += red[lane * (HD + 1) + c];
    }
  }
  if (wave == 0 && q_ok) {
    float* op = dqkv + ((int64_t)b * T + query) * (3 * d) + head * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) *reinterpret_cast<f32x4*>(op + c) = f32x4{dq[c], dq[c + 1], dq[c + 2], dq[c + 3]} * qscale;
  }
}

template <int HD>
int launch_attn_bwd_any(const float* qkv, int64_t ld, int d, float qscale, const float* d_o, const float* lse, const float* dsum,
                        const int64_t* num_ims, float* dqkv, int B, int T, int H, int max_q, DropSite site, hipStream_t stream) {
  const int nq = max_q > 0 && max_q < T ? max_q : T;
  hipLaunchKernelGGL(attn_bwd_any_kv_kernel<HD>, dim3((T + 63) / 64, H, B), dim3(256), 0, stream, qkv, ld, d, qscale, d_o, lse, dsum, num_ims,
                     dqkv, T, H, max_q, site);
  PATHS_LAUNCH_CHECK("attention_bwd_any(kv)");
  hipLaunchKernelGGL(attn_bwd_any_q_kernel<HD>, dim3((nq + 63) / 64, H, B), dim3(256), 0, stream, qkv, ld, d, qscale, d_o, lse, dsum, num_ims,
                     dqkv, T, H, max_q, site);
  PATHS_LAUNCH_CHECK("attention_bwd_any(q)");
  return PATHS_OK;
}

}  // namespace

extern "C" {

// LayerNorm forward keeping xhat / rstd for the backward (y may be null), any width; x (+ add) -> y
int paths_layernorm_fwd_stats_any(const float* x, const float* add, const float* gamma, const float* beta, float* y, float* xhat,
                                  float* rstd, int64_t rows, int d, float eps, hipStream_t stream) {
  PATHS_REQUIRE(rows > 0 && d > 0 && d <= 2048 && d % 4 == 0 && x && xhat && rstd && (y == nullptr || (gamma && beta)), "layernorm_fwd_stats_any: bad arguments (d = %d)", d);
  hipLaunchKernelGGL(ln_fwd_stats_any_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, add, gamma, beta, y, xhat, rstd, rows, d, eps);
  PATHS_LAUNCH_CHECK("layernorm_fwd_stats_any");
  return PATHS_OK;
}

int paths_layernorm_bwd_any(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx, float* dyxhat,
                            int64_t rows, int d, hipStream_t stream) {
  PATHS_REQUIRE(rows > 0 && d > 0 && d <= 2048 && d % 4 == 0 && dy && xhat && rstd && gamma && dx && dyxhat, "layernorm_bwd_any: bad arguments (d = %d)", d);
  hipLaunchKernelGGL(ln_bwd_any_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, dy, xhat, rstd, gamma, dx, dyxhat, rows, d, 4);
  PATHS_LAUNCH_CHECK("layernorm_bwd_any");
  return PATHS_OK;
}

// dx + one slab [sum dy*xhat (d) | sum dy (d) | sum dx (d)] per block of rows_per_block rows (paths_reduce_slabs_f32 adds them)
int paths_layernorm_bwd_sums_any(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx, float* slabs,
                                 int64_t rows, int d, int rows_per_block, hipStream_t stream) {
  PATHS_REQUIRE(rows > 0 && d > 0 && d <= 2048 && d % 4 == 0 && rows_per_block >= 4 && dy && xhat && rstd && gamma && dx && slabs,
                "layernorm_bwd_sums_any: bad arguments (d = %d)", d);
  const unsigned nblk = (unsigned)((rows + rows_per_block - 1) / rows_per_block);
  hipLaunchKernelGGL(ln_bwd_any_kernel<true>, dim3(nblk), dim3(256), 0, stream, dy, xhat, rstd, gamma, dx, slabs, rows, d, rows_per_block);
  PATHS_LAUNCH_CHECK("layernorm_bwd_sums_any");
  return PATHS_OK;
}

// importance MLP / scaling / proj_in backward for any widths: du [M, ldu] = [dhid (Hi) | dP (d) | zeros], da [M], dah [M, Hi]
int paths_importance_bwd_any(const float* dtok, const float* pproj, const float* hid, const float* alpha, const float* w2,
                             const int64_t* num_ims, int rows_per_slide, int64_t M, int imp_mul, int Hi, int d, int64_t ldu, float* du,
                             float* da, float* dah, hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && Hi > 0 && d > 0 && ldu >= Hi + d && dtok && pproj && hid && alpha && w2 && num_ims && du && da && dah,
                "importance_bwd_any: bad arguments");
  hipLaunchKernelGGL(imp_bwd_any_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, dtok, pproj, hid, alpha, w2, num_ims, rows_per_slide, M,
                     imp_mul, Hi, d, ldu, du, da, dah);
  PATHS_LAUNCH_CHECK("importance_bwd_any");
  return PATHS_OK;
}

// lstm = false: importance MLP backward through Z = alpha X for any hidden width (dh [M, Hi], da [M], dah [M, Hi])
int paths_importance_rows_bwd_any(const float* dz_rows, const float* x, int D, const float* hid, const float* alpha, const float* w2,
                                  const int64_t* num_ims, int rows_per_slide, int64_t M, int Hi, float* dh, float* da, float* dah,
                                  hipStream_t stream) {
  PATHS_REQUIRE(M > 0 && D % 4 == 0 && Hi > 0 && dz_rows && x && hid && alpha && w2 && num_ims && dh && da && dah, "importance_rows_bwd_any: bad arguments");
  hipLaunchKernelGGL(imp_rows_bwd_any_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, dz_rows, x, D, hid, alpha, w2, num_ims,
                     rows_per_slide, M, Hi, dh, da, dah);
  PATHS_LAUNCH_CHECK("importance_rows_bwd_any");
  return PATHS_OK;
}

// Attention backward for head_dim in {16, 32, 48, 64}: qkv [B*T, 3d] token-major (q unscaled), o / d_o [B*T, d], lse [B,H,T] (log2
// domain), dqkv [B*T, 3d] (ZERO on entry: rows >= num_ims + 1 and, with max_queries > 0, the dq of the other queries are not written),
// ws_dsum [B*H*T] scratch.  max_queries > 0: only queries [0, max_queries) carry an output gradient (last layer: token 0).
int paths_attention_bwd_any(const float* qkv, int64_t ld, const float* o, const float* d_o, const float* lse, const int64_t* num_ims,
                            float* dqkv, float* ws_dsum, int B, int T, int H, int head_dim, float qscale, int max_queries,
                            uint64_t drop_key, float drop_p, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && T > 0 && H > 0 && qkv && o && d_o && lse && num_ims && dqkv && ws_dsum, "attention_bwd_any: bad arguments");
  PATHS_REQUIRE(ld % 4 == 0 && ((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dqkv) % 16 == 0, "attention_bwd_any: alignment");
  PATHS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention_bwd_any: p must be in [0, 1)");
  const int d = H * head_dim;
  const int64_t rows = (int64_t)B * T;
  hipLaunchKernelGGL(attn_bwd_any_prep_kernel, dim3((unsigned)((rows * H + 255) / 256)), dim3(256), 0, stream, o, d_o, ws_dsum, rows, T, H, head_dim);
  PATHS_LAUNCH_CHECK("attention_bwd_any(prep)");
  const DropSite site = paths_make_drop_site(drop_key, drop_p);
  switch (head_dim) {
    case 16: return launch_attn_bwd_any<16>(qkv, ld, d, qscale, d_o, lse, ws_dsum, num_ims, dqkv, B, T, H, max_queries, site, stream);
    case 32: return launch_attn_bwd_any<32>(qkv, ld, d, qscale, d_o, lse, ws_dsum, num_ims, dqkv, B, T, H, max_queries, site, stream);
    case 48: return launch_attn_bwd_any<48>(qkv, ld, d, qscale, d_o, lse, ws_dsum, num_ims, dqkv, B, T, H, max_queries, site, stream);
    case 64: return launch_attn_bwd_any<64>(qkv, ld, d, qscale, d_o, lse, ws_dsum, num_ims, dqkv, B, T, H, max_queries, site, stream);
    default: return paths_set_error(PATHS_EUNSUPPORTED, "attention_bwd_any: head_dim %d (supported: 16, 32, 48, 64)", head_dim);
  }
}

}  // extern "C"
