// Attention backward, dQ part, on the bf16 matrix cores with fp32 accuracy (three exact bf16 planes per operand, the six largest
// partial products, smallest first: the arithmetic of gemm_x6.hip / attn_x6.hip).  Same conventions as attn_bwd.hip (reference
// model/aggregator.py:70-72 differentiated; q pre-scaled, lse in the log2 domain, P = exp2(q_s . k - lse),
// ds = ln2 P (dP m - D), dq_s = ds K), same outputs (rows [dq | . | .] of dqkv), same dropout-mask regeneration.
//
//   attn_bwd_x6_prep_kernel   q, k, v, dO -> MFMA fragment images, three planes each (1 KiB per 16 x 32 fragment and plane, lane l
//                             owns bytes [16 l, 16 l + 16)):
//                               Qr / Kr / Vr / Gr : rows = tokens, k = the 32 head dims        [Tp/16][3][64][8]
//                               Kt                : rows = 16 head dims, k = 32 keys, permuted  [Tp/32][2][3][64][8]
//                             k-slot (g, j) <-> key 4 g + (j & 3) + 16 (j >> 2): the order in which the dS^T accumulators of two
//                             16-key tiles sit in a lane, so dS never leaves the registers (the V^T trick of attn_x6.hip).
//   attn_bwd_q_x6_kernel      one wave = 32 queries (two 16-query tiles), a 4-wave workgroup shares 64-key fragment sets through
//                             LDS (double-buffered).  Per 32-key group:
//                               S^T[key][q]  = Kr Qr^T     2 key tiles x 2 query tiles x 6 MFMAs
//                               dP^T[key][q] = Vr Gr^T     2 x 2 x 6
//                               dS^T split in registers    3 planes
//                               dQ^T[d][q]  += Kt dS^T     2 d tiles x 2 query tiles x 6
//                             72 v_mfma_f32_16x16x32_bf16 (16 cycles) per 32 x 32 block against 96 v_mfma_f32_16x16x4_f32 (32 cycles).
#include "common.h"
#include "dropout.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 32;
constexpr int FRAG = 1024;
constexpr int KSTEP = 64;
constexpr int QT = 2;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// 8 fp32 -> PL planes of 8 bf16: PL = 3 hi, mid, lo (exact: 3 x 8 = the 24 bits of fp32); PL = 2 hi, mid only (16 significant bits at
// fp32's exponent range - the setting of the training step's gradient GEMMs, PATHS_TRAIN_PLANES=4: no scales, nothing overflows)
template <int PL>
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 (&pl)[PL]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const uint32_t h = pk_bf16(a, b);
    const float ra = a - bf_lo(h), rb = b - bf_hi(h);
    const uint32_t m = pk_bf16(ra, rb);
    pl[0][i] = h; pl[1][i] = m;
    if constexpr (PL == 3) {
      const float sa = ra - bf_lo(m), sb = rb - bf_hi(m);
      pl[2][i] = pk_bf16(sa, sb);
    }
  }
}
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_split(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4 c) {   // hi, mid, lo: six largest of nine
  c = mfma_bf16(a[2], b[0], c);
  c = mfma_bf16(a[0], b[2], c);
  c = mfma_bf16(a[1], b[1], c);
  c = mfma_bf16(a[1], b[0], c);
  c = mfma_bf16(a[0], b[1], c);
  c = mfma_bf16(a[0], b[0], c);
  return c;
}
__device__ __forceinline__ f32x4 mfma_split(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x4 c) {   // hi, mid: all but mid*mid
  c = mfma_bf16(a[1], b[0], c);
  c = mfma_bf16(a[0], b[1], c);
  c = mfma_bf16(a[0], b[0], c);
  return c;
}

template <int PL>
__global__ void __launch_bounds__(256)
attn_bwd_x6_prep_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                        const float* __restrict__ d_o /*[B,T,H*32]*/, char* __restrict__ qr, char* __restrict__ kr, char* __restrict__ vr,
                        char* __restrict__ gr, char* __restrict__ kt, char* __restrict__ qt, char* __restrict__ gt,
                        const int64_t* __restrict__ num_ims, int T, int Tp, int H) {
  __shared__ float sk[3][KSTEP][HD + 1];               // K, Q, dO rows of this 64-token block
  const int b = blockIdx.z, head = blockIdx.y, t0 = blockIdx.x * KSTEP;
  const int len = min((int)num_ims[b] + 1, T);
  const int tid = threadIdx.x;
  const int64_t base = ((int64_t)b * H + head) * T * HD;
  const int64_t ibase = ((int64_t)b * H + head) * (int64_t)Tp * HD * 2 * PL;          // bytes of one (slide, head) image
  {
    const int tl = tid >> 2, g = tid & 3, tok = t0 + tl;
    float xq[8], xk[8], xv[8], xg[8];
    const bool valid = tok < len;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xk[i] = valid ? k[base + (int64_t)tok * HD + 8 * g + i] : 0.f;
      xv[i] = valid ? v[base + (int64_t)tok * HD + 8 * g + i] : 0.f;
      xq[i] = valid ? q[base + (int64_t)tok * HD + 8 * g + i] : 0.f;
      xg[i] = valid ? d_o[((int64_t)b * T + tok) * (H * HD) + head * HD + 8 * g + i] : 0.f;
    }
    const int64_t off = ibase + ((int64_t)(tok >> 4) * PL) * FRAG + ((tok & 15) + 16 * g) * 16;
    u32x4 pl[PL];
    split8<PL>(xk, pl);
#pragma unroll
    for (int p = 0; p < PL; ++p) *reinterpret_cast<u32x4*>(kr + off + p * FRAG) = pl[p];
    split8<PL>(xv, pl);
#pragma unroll
    for (int p = 0; p < PL; ++p) *reinterpret_cast<u32x4*>(vr + off + p * FRAG) = pl[p];
    split8<PL>(xq, pl);
#pragma unroll
    for (int p = 0; p < PL; ++p) *reinterpret_cast<u32x4*>(qr + off + p * FRAG) = pl[p];
    split8<PL>(xg, pl);
#pragma unroll
    for (int p = 0; p < PL; ++p) *reinterpret_cast<u32x4*>(gr + off + p * FRAG) = pl[p];
  }
  // K^T, Q^T, dO^T through LDS (coalesced rows in, transposed + token-permuted fragments out)
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int idx = tid + 256 * p, tl = idx >> 5, dcol = idx & 31, tok = t0 + tl;
    const bool valid = tok < len;
    sk[0][tl][dcol] = valid ? k[base + (int64_t)tok * HD + dcol] : 0.f;
    sk[1][tl][dcol] = valid ? q[base + (int64_t)tok * HD + dcol] : 0.f;
    sk[2][tl][dcol] = valid ? d_o[((int64_t)b * T + tok) * (H * HD) + head * HD + dcol] : 0.f;
  }
  __syncthreads();
  {
    const int kg = tid >> 7, dt = (tid >> 6) & 1, l = tid & 63, dd = l & 15, g = l >> 4;
    const int64_t off = ibase + ((int64_t)(((t0 >> 5) + kg) * 2 + dt) * PL) * FRAG + l * 16;
    char* const dst[3] = {kt, qt, gt};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float xt[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xt[j] = sk[a][32 * kg + 4 * g + (j & 3) + 16 * (j >> 2)][16 * dt + dd];
      u32x4 pl[PL];
      split8<PL>(xt, pl);
#pragma unroll
      for (int p = 0; p < PL; ++p) *reinterpret_cast<u32x4*>(dst[a] + off + p * FRAG) = pl[p];
    }
  }
}

// LDS per 64-key step: Kr 4 tiles x 3 planes | Vr 4 x 3 | Kt 2 groups x 2 d tiles x 3 = 36 KiB, double-buffered
template <int PL> constexpr int part_bytes() { return 4 * PL * FRAG; }      // bytes of one of the three parts of a step
template <int PL> constexpr int step_bytes() { return 3 * part_bytes<PL>(); }

template <int PL>
__global__ void __launch_bounds__(256, 2)
attn_bwd_q_x6_kernel(const char* __restrict__ qr, const char* __restrict__ kr, const char* __restrict__ vr, const char* __restrict__ gr,
                     const char* __restrict__ kt, const float* __restrict__ lse, const float* __restrict__ dsum,
                     const int64_t* __restrict__ num_ims, float* __restrict__ dqkv, int T, int Tp, int H, int npairs, int nqb, DropSite drop) {
  constexpr int PART = part_bytes<PL>(), STEP = step_bytes<PL>();
  extern __shared__ __attribute__((aligned(16))) char smem[];           // [2][STEP]
  // XCD-aware placement as in attn_x6.hip: pair p only ever runs on the XCD group p % 8
  const int lin = blockIdx.x, xg = lin & 7, jx = lin >> 3;
  const int cnt = (npairs - xg + 7) >> 3;
  if (cnt <= 0) return;
  const int pair = xg + 8 * (jx % cnt), qb = jx / cnt;
  if (qb >= nqb) return;
  const int b = pair / H, head = pair - b * H, q0 = qb * 64 * QT;
  const int len = min((int)num_ims[b] + 1, T);
  const DropWin dwin = drop_window(drop, drop_attn_row((uint64_t)b * H + head, T, 0));       // (this pair's T x T' mask elements: csrc/dropout.h)
  if (q0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 15, g4 = lane >> 4;
  const int64_t ibase = ((int64_t)b * H + head) * (int64_t)Tp * HD * 2 * PL;
  const int qw = q0 + wave * 16 * QT;

  u32x4 qf[QT][PL], gf[QT][PL];
  float my_lse[QT], my_d[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int64_t off = ibase + ((int64_t)(min(qw + 16 * qt, Tp - 16) >> 4) * PL) * FRAG + lane * 16;
#pragma unroll
    for (int p = 0; p < PL; ++p) {
      qf[qt][p] = *reinterpret_cast<const u32x4*>(qr + off + p * FRAG);
      gf[qt][p] = *reinterpret_cast<const u32x4*>(gr + off + p * FRAG);
    }
    const int qc = min(qw + 16 * qt + ql, T - 1);
    my_lse[qt] = lse[((int64_t)b * H + head) * T + qc];
    my_d[qt] = dsum[((int64_t)b * H + head) * T + qc];
  }
  f32x4 dq[2][QT];                                      // [d tile][query tile]: rows = dims 4 g4 .. +3, col = query ql
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < QT; ++j) dq[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (len + KSTEP - 1) / KSTEP;
  // staging: a step is three contiguous 12-KiB pieces (Kr, Vr, Kt of 64 keys); 9 x 16 bytes per thread
  u32x4 st[3 * PL];
  auto gload = [&](int kt_) {
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      st[i] = *reinterpret_cast<const u32x4*>(kr + ibase + (int64_t)kt_ * PART + (tid + 256 * i) * 16);
      st[PL + i] = *reinterpret_cast<const u32x4*>(vr + ibase + (int64_t)kt_ * PART + (tid + 256 * i) * 16);
      st[2 * PL + i] = *reinterpret_cast<const u32x4*>(kt + ibase + (int64_t)kt_ * PART + (tid + 256 * i) * 16);
    }
  };
  auto swrite = [&](int kt_) {
    char* d = smem + (kt_ & 1) * STEP;
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      *reinterpret_cast<u32x4*>(d + (tid + 256 * i) * 16) = st[i];
      *reinterpret_cast<u32x4*>(d + PART + (tid + 256 * i) * 16) = st[PL + i];
      *reinterpret_cast<u32x4*>(d + 2 * PART + (tid + 256 * i) * 16) = st[2 * PL + i];
    }
  };
  gload(0);
  swrite(0);
  __syncthreads();
  for (int kt_ = 0; kt_ < nkt; ++kt_) {
    if (kt_ + 1 < nkt) gload(kt_ + 1);
    const char* sK = smem + (kt_ & 1) * STEP + lane * 16;
    const char* sV = sK + PART;
    const char* sT = sK + 2 * PART;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      f32x4 s[QT][2], dp[QT][2];                        // [query tile][key tile of the group]: rows = keys 4 g4 .. +3, col = query ql
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        u32x4 kf[PL], vf[PL];
#pragma unroll
        for (int p = 0; p < PL; ++p) {
          kf[p] = *reinterpret_cast<const u32x4*>(sK + ((2 * kg + t) * PL + p) * FRAG);
          vf[p] = *reinterpret_cast<const u32x4*>(sV + ((2 * kg + t) * PL + p) * FRAG);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          s[qt][t] = mfma_split(kf, qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
          dp[qt][t] = mfma_split(vf, gf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
        }
      }
      u32x4 dsf[QT][PL];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float dsv[8];
        const uint64_t drow = drop_attn_row((uint64_t)b * H + head, T, min(qw + 16 * qt + ql, T - 1));
        float mk[8];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {                // keys 4 g4 + (j & 3), + 1 of a query: one hash per pair (dropout.h)
          const int key = kt_ * KSTEP + 32 * kg + 16 * (j >> 2) + 4 * g4 + (j & 3);
          mk[j] = mk[j + 1] = 1.0f;
          if (drop.thr) drop_mult2_w(drop, dwin, drow + (uint64_t)min(key, (int)drop_attn_stride(T) - 2), mk[j], mk[j + 1]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {                   // k-slot (g4, j) = key 4 g4 + (j & 3) + 16 (j >> 2) of the group
          const int key = kt_ * KSTEP + 32 * kg + 16 * (j >> 2) + 4 * g4 + (j & 3);
          const float p = key < len ? __builtin_amdgcn_exp2f(s[qt][j >> 2][j & 3] - my_lse[qt]) : 0.f;
          dsv[j] = LN2 * p * (dp[qt][j >> 2][j & 3] * mk[j] - my_d[qt]);
        }
        split8<PL>(dsv, dsf[qt]);
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        u32x4 tf[PL];
#pragma unroll
        for (int p = 0; p < PL; ++p) tf[p] = *reinterpret_cast<const u32x4*>(sT + ((kg * 2 + dt) * PL + p) * FRAG);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) dq[dt][qt] = mfma_split(tf, dsf[qt], dq[dt][qt]);
      }
    }
    if (kt_ + 1 < nkt) swrite(kt_ + 1);                 // the other buffer: read one step ago, everyone passed the barrier since
    __syncthreads();
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qi = qw + 16 * qt + ql;
    if (qi < len) {
      float* dst = dqkv + ((int64_t)b * T + qi) * (3 * H * HD) + head * HD + 4 * g4;
      *reinterpret_cast<f32x4*>(dst) = dq[0][qt];
      *reinterpret_cast<f32x4*>(dst + 16) = dq[1][qt];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// dK, dV: one wave = 32 keys (two 16-key tiles, their Kr / Vr fragments in registers), a 4-wave workgroup = 128 keys of one
// (slide, head), looping over 64-query steps staged through LDS (Qr, Gr rows; Qt, Gt transposed; lse, D).  Per 32-query group:
//   S[q][key]  = Qr Kr^T      2 query tiles x 2 key tiles x 6 MFMAs        (accumulator rows = queries, column = key)
//   dP[q][key] = Gr Vr^T      2 x 2 x 6
//   P m and dS split in registers: the B operands of the next two products, k = the group's 32 queries in accumulator order
//   dV^T[dv][key] += Gt (P m)    2 dv tiles x 2 key tiles x 6
//   dK^T[d][key]  += Qt dS       2 d tiles x 2 key tiles x 6
// ---------------------------------------------------------------------------------------------------------------
template <int PL> constexpr int kv_step_bytes() { return 4 * part_bytes<PL>(); }   // Qr | Gr | Qt | Gt of 64 queries: 48 KiB at three planes (one buffer; the next step waits in registers)

template <int PL>
__global__ void __launch_bounds__(256, 2)
attn_bwd_kv_x6_kernel(const char* __restrict__ qr, const char* __restrict__ kr, const char* __restrict__ vr, const char* __restrict__ gr,
                      const char* __restrict__ qt, const char* __restrict__ gt, const float* __restrict__ lse, const float* __restrict__ dsum,
                      const int64_t* __restrict__ num_ims, float* __restrict__ dqkv, int T, int Tp, int H, int npairs, int nkb, DropSite drop) {
  constexpr int PART = part_bytes<PL>(), KV_STEP = kv_step_bytes<PL>();
  extern __shared__ __attribute__((aligned(16))) char smem[];           // [KV_STEP] + lse[64] + D[64]
  const int lin = blockIdx.x, xg = lin & 7, jx = lin >> 3;
  const int cnt = (npairs - xg + 7) >> 3;
  if (cnt <= 0) return;
  const int pair = xg + 8 * (jx % cnt), kb = jx / cnt;
  if (kb >= nkb) return;
  const int b = pair / H, head = pair - b * H, k0 = kb * 128;
  const int len = min((int)num_ims[b] + 1, T);
  const DropWin dwin = drop_window(drop, drop_attn_row((uint64_t)b * H + head, T, 0));       // (this pair's T x T' mask elements: csrc/dropout.h)
  if (k0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kl = lane & 15, g4 = lane >> 4;
  const int64_t ibase = ((int64_t)b * H + head) * (int64_t)Tp * HD * 2 * PL;
  const int kw = k0 + wave * 32;                        // this wave's first key
  float* sL = reinterpret_cast<float*>(smem + KV_STEP);
  float* sD = sL + 64;

  u32x4 kf[2][PL], vf[2][PL];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int64_t off = ibase + ((int64_t)(min(kw + 16 * t, Tp - 16) >> 4) * PL) * FRAG + lane * 16;
#pragma unroll
    for (int p = 0; p < PL; ++p) {
      kf[t][p] = *reinterpret_cast<const u32x4*>(kr + off + p * FRAG);
      vf[t][p] = *reinterpret_cast<const u32x4*>(vr + off + p * FRAG);
    }
  }
  f32x4 dk[2][2], dv[2][2];                             // [key tile][d tile]: rows = dims 4 g4 .. +3, col = key kl
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) { dk[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nqs = (len + KSTEP - 1) / KSTEP;            // 64-query steps
  u32x4 st[4 * PL];
  float st_s = 0.f;
  auto gload = [&](int qs) {
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      st[i] = *reinterpret_cast<const u32x4*>(qr + ibase + (int64_t)qs * PART + (tid + 256 * i) * 16);
      st[PL + i] = *reinterpret_cast<const u32x4*>(gr + ibase + (int64_t)qs * PART + (tid + 256 * i) * 16);
      st[2 * PL + i] = *reinterpret_cast<const u32x4*>(qt + ibase + (int64_t)qs * PART + (tid + 256 * i) * 16);
      st[3 * PL + i] = *reinterpret_cast<const u32x4*>(gt + ibase + (int64_t)qs * PART + (tid + 256 * i) * 16);
    }
    if (tid < 128) {
      const int qi = min(qs * KSTEP + (tid & 63), T - 1);
      st_s = (tid < 64 ? lse : dsum)[((int64_t)b * H + head) * T + qi];
    }
  };
  auto swrite = [&]() {
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      *reinterpret_cast<u32x4*>(smem + (tid + 256 * i) * 16) = st[i];
      *reinterpret_cast<u32x4*>(smem + PART + (tid + 256 * i) * 16) = st[PL + i];
      *reinterpret_cast<u32x4*>(smem + 2 * PART + (tid + 256 * i) * 16) = st[2 * PL + i];
      *reinterpret_cast<u32x4*>(smem + 3 * PART + (tid + 256 * i) * 16) = st[3 * PL + i];
    }
    if (tid < 64) sL[tid] = st_s;
    else if (tid < 128) sD[tid - 64] = st_s;
  };
  gload(0);
  for (int qs = 0; qs < nqs; ++qs) {
    __syncthreads();                                    // everyone is done reading the previous step
    swrite();
    __syncthreads();
    if (qs + 1 < nqs) gload(qs + 1);
    const char* sQ = smem + lane * 16;
    const char* sG = sQ + PART;
    const char* sQt = sQ + 2 * PART;
    const char* sGt = sQ + 3 * PART;
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      f32x4 s[2][2], dp[2][2];                          // [query tile of the group][key tile]: rows = queries 4 g4 .. +3, col = key kl
#pragma unroll
      for (int qt_ = 0; qt_ < 2; ++qt_) {
        u32x4 qf[PL], gf[PL];
#pragma unroll
        for (int p = 0; p < PL; ++p) {
          qf[p] = *reinterpret_cast<const u32x4*>(sQ + ((2 * qg + qt_) * PL + p) * FRAG);
          gf[p] = *reinterpret_cast<const u32x4*>(sG + ((2 * qg + qt_) * PL + p) * FRAG);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          s[qt_][t] = mfma_split(qf, kf[t], f32x4{0.f, 0.f, 0.f, 0.f});
          dp[qt_][t] = mfma_split(gf, vf[t], f32x4{0.f, 0.f, 0.f, 0.f});
        }
      }
      u32x4 pf[2][PL], dsf[2][PL];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int key = kw + 16 * t + kl;
        const bool key_ok = key < len;
        float pv[8], dsv[8], mk[8];
        if (drop.thr) {
          // A lane owns ONE key and eight queries, its neighbour (lane ^ 1) the other key of the pair and the same queries: each
          // of the two hashes four of the eight (query, key pair) elements and they swap through DPP (one hash per pair: dropout.h)
          const int odd = kl & 1;                       // = key & 1 (kw + 16 t is even)
          const uint64_t kev = (uint64_t)(min(key, T - 1) & ~1);
          uint32_t hm[4], ho[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int qi = qs * KSTEP + 32 * qg + 16 * odd + 4 * g4 + jj;
            hm[jj] = drop_hash_w(drop, dwin, drop_attn_row((uint64_t)b * H + head, T, min(qi, T - 1)) + kev);
            ho[jj] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hm[jj], 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const uint32_t hsh = ((j >> 2) == odd) ? hm[j & 3] : ho[j & 3];
            mk[j] = (odd ? hsh >> 16 : hsh & 0xFFFFu) >= drop.thr ? drop.scale : 0.f;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) mk[j] = 1.0f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {                   // k-slot (g4, j) = query 4 g4 + (j & 3) + 16 (j >> 2) of the group
          const int qloc = 32 * qg + 16 * (j >> 2) + 4 * g4 + (j & 3), qi = qs * KSTEP + qloc;
          const bool ok = key_ok && qi < len;
          const float pr = ok ? __builtin_amdgcn_exp2f(s[j >> 2][t][j & 3] - sL[qloc]) : 0.f;
          dsv[j] = LN2 * pr * (dp[j >> 2][t][j & 3] * mk[j] - sD[qloc]);
          pv[j] = pr * mk[j];
        }
        split8<PL>(pv, pf[t]);
        split8<PL>(dsv, dsf[t]);
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        u32x4 qtf[PL], gtf[PL];
#pragma unroll
        for (int p = 0; p < PL; ++p) {
          qtf[p] = *reinterpret_cast<const u32x4*>(sQt + ((qg * 2 + dt) * PL + p) * FRAG);
          gtf[p] = *reinterpret_cast<const u32x4*>(sGt + ((qg * 2 + dt) * PL + p) * FRAG);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          dv[t][dt] = mfma_split(gtf, pf[t], dv[t][dt]);
          dk[t][dt] = mfma_split(qtf, dsf[t], dk[t][dt]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int key = kw + 16 * t + kl;
    if (key < len) {
      float* dst = dqkv + ((int64_t)b * T + key) * (3 * H * HD) + head * HD + 4 * g4;
      *reinterpret_cast<f32x4*>(dst + H * HD) = dk[t][0];
      *reinterpret_cast<f32x4*>(dst + H * HD + 16) = dk[t][1];
      *reinterpret_cast<f32x4*>(dst + 2 * H * HD) = dv[t][0];
      *reinterpret_cast<f32x4*>(dst + 2 * H * HD + 16) = dv[t][1];
    }
  }
}

}  // namespace

// bytes of the fragment images the split-bf16 backward kernels need (seven images of three bf16 planes)
extern "C" int64_t paths_attention_bwd_x6_workspace(int B, int T, int H, int head_dim) {
  const int64_t Tp = ((int64_t)T + KSTEP - 1) / KSTEP * KSTEP;
  return 7 * (int64_t)B * H * Tp * head_dim * 6;
}

// dqkv from the images: dK / dV and dQ (called by attention_bwd_impl in attn_bwd.hip in place of its f32-MFMA kernels)
template <int PL>
static int attention_bwd_x6_launch_pl(const float* q, const float* k, const float* v, const float* d_o, const float* lse, const float* dsum,
                                      const int64_t* num_ims, float* dqkv, void* images, int B, int T, int H, DropSite site, int kv_too,
                                      hipStream_t stream) {
  constexpr int STEP = step_bytes<PL>(), KV_STEP = kv_step_bytes<PL>();
  const int Tp = (T + KSTEP - 1) / KSTEP * KSTEP;
  const int64_t img = (int64_t)B * H * Tp * HD * 2 * PL;
  char* qr = reinterpret_cast<char*>(images);
  char* kr = qr + img; char* vr = kr + img; char* gr = vr + img; char* ktp = gr + img; char* qtp = ktp + img; char* gtp = qtp + img;
  hipLaunchKernelGGL(attn_bwd_x6_prep_kernel<PL>, dim3(Tp / KSTEP, H, B), dim3(256), 0, stream, q, k, v, d_o, qr, kr, vr, gr, ktp, qtp, gtp, num_ims, T, Tp, H);
  PATHS_LAUNCH_CHECK("attention_bwd_x6(prep)");
  PATHS_LDS_OPT_IN(attn_bwd_q_x6_kernel<PL>, 2 * STEP, "attention_bwd_x6(dq)");
  PATHS_LDS_OPT_IN(attn_bwd_kv_x6_kernel<PL>, KV_STEP + 512, "attention_bwd_x6(dk, dv)");
  const int npairs = H * B;
  if (kv_too) {
    const int nkb = (T + 127) / 128;
    hipLaunchKernelGGL(attn_bwd_kv_x6_kernel<PL>, dim3(8 * ((npairs + 7) / 8) * nkb), dim3(256), KV_STEP + 512, stream, qr, kr, vr, gr, qtp, gtp, lse, dsum,
                       num_ims, dqkv, T, Tp, H, npairs, nkb, site);
    PATHS_LAUNCH_CHECK("attention_bwd_x6(kv)");
  }
  const int nqb = (T + 64 * QT - 1) / (64 * QT);
  hipLaunchKernelGGL(attn_bwd_q_x6_kernel<PL>, dim3(8 * ((npairs + 7) / 8) * nqb), dim3(256), 2 * STEP, stream, qr, kr, vr, gr, ktp, lse, dsum,
                     num_ims, dqkv, T, Tp, H, npairs, nqb, site);
  PATHS_LAUNCH_CHECK("attention_bwd_x6(q)");
  return PATHS_OK;
}

// planes: 3 = exact (hi, mid, lo; 6 MFMAs per product block), 2 = hi, mid (16 bits, 3 MFMAs: the training default, PATHS_TRAIN_PLANES=4)
int paths_attention_bwd_x6_launch(const float* q, const float* k, const float* v, const float* d_o, const float* lse, const float* dsum,
                                  const int64_t* num_ims, float* dqkv, void* images, int B, int T, int H, DropSite site, int kv_too,
                                  int planes, hipStream_t stream) {
  if (planes == 2) return attention_bwd_x6_launch_pl<2>(q, k, v, d_o, lse, dsum, num_ims, dqkv, images, B, T, H, site, kv_too, stream);
  return attention_bwd_x6_launch_pl<3>(q, k, v, d_o, lse, dsum, num_ims, dqkv, images, B, T, H, site, kv_too, stream);
}
