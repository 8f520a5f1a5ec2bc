// Masked self-attention for ANY head_dim in {16, 32, 48, 64} on the 16-bit matrix cores with fp32-accurate operands: the shape-generic
// counterpart of attn_x6.hip's default mode (reference model/aggregator.py:70-72 + utils.py:97-103; the reference's dataclass default
// trans_dim 192 / 4 heads is head_dim 48).  Every operand is split into two fp16 planes (hi + lo = 22 significant bits), a product
// block is hi*hi + hi*lo + lo*hi = three v_mfma_f32_16x16x32_f16, accumulation in fp32; the probabilities are split in registers.
// Same structure as attn_fp8.hip / attn_x6.hip: S^T / O^T form (keys on the MFMA rows, queries on the lanes: per-lane softmax state),
// one wave = 32 queries, a 4-wave workgroup shares 64-key K / V^T fragment sets through double-buffered LDS, S(k+1) issued before the
// softmax of S(k), the accumulator layout of P is directly the B operand of the PV product (k-slot order of the V image).  head_dim is
// padded to a multiple of 32 on the score side (16 -> 32, 48 -> 64: zero dims) - correctness first, not tuned like attn_x6.hip.
// q, k, v are read from the token-major in_proj output [B*T, 3d] (q unscaled: log2(e)/sqrt(hd) is applied while the images are written).
#include "common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int FB = 1024;                   // bytes of one 16-row x 32-k fp16 fragment plane (64 lanes x 16 bytes)
constexpr int KSTEP = 64;                  // keys per LDS buffer
constexpr int QT = 2;                      // 16-query tiles per wave

__device__ __forceinline__ uint32_t pk_f16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// 8 fp32 -> two planes of 8 fp16 (hi, lo): 22 significant bits
__device__ __forceinline__ void split8h(const float (&x)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const uint32_t h = pk_f16(a, b);
    float ra, rb;
    f16_pair_residuals(h, a, b, ra, rb);
    hi[i] = h; lo[i] = pk_f16(ra, rb);
  }
}
__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// acc += A * B from (hi, lo) planes, smallest partial products first: all but lo * lo
__device__ __forceinline__ f32x4 mfma_split(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x4 c) {
  c = mfma_f16(a[1], b[0], c);
  c = mfma_f16(a[0], b[1], c);
  return mfma_f16(a[0], b[0], c);
}
__device__ __forceinline__ float rows_max(float x) { x = fmaxf(x, __shfl_xor(x, 16)); return fmaxf(x, __shfl_xor(x, 32)); }
__device__ __forceinline__ float rows_sum(float x) { x += __shfl_xor(x, 16); return x + __shfl_xor(x, 32); }

// Fragment images per (slide, head), Tp = T rounded up to 64, NK = ceil(HD / 32) k-steps on the score side, NDV = HD / 16 output tiles:
//   Q / K : [Tp/16 tiles][NK][2 planes][64 lanes][16 B]    lane (r = l&15, g = l>>4): token 16 tile + r, dims 32 kk + 8g .. + 7 (0 past HD)
//   V     : [Tp/32 groups][NDV][2 planes][64 lanes][16 B]   lane (dv = l&15, g): dim 16 dvt + dv, keys 32 grp + 4g + (j&3) + 16 (j>>2)
template <int HD>
__global__ void __launch_bounds__(256)
attn_h3_prep_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int64_t ld, int64_t hstride,
                    int64_t bstride, float qmul, char* __restrict__ qi, char* __restrict__ ki, char* __restrict__ vi,
                    const int64_t* __restrict__ num_ims, int T, int Tp, int H) {
  constexpr int NK = (HD + 31) / 32, NDV = HD / 16;
  __shared__ float sv[KSTEP][HD + 1];
  const int b = blockIdx.z, head = blockIdx.y, t0 = blockIdx.x * KSTEP;
  const int len = min((int)num_ims[b] + 1, T);
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)b * bstride + (int64_t)head * hstride;
  const int64_t qkbase = ((int64_t)b * H + head) * (int64_t)Tp * (NK * 32) * 4;     // bytes: 2 planes x 2 bytes per element
  const int64_t vbase = ((int64_t)b * H + head) * (int64_t)Tp * HD * 4;
  {
    const int tl = tid >> 2, g = tid & 3, tok = t0 + tl;
    const bool kvalid = tok < len, qvalid = tok < T;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      float xq[8], xk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int dcol = 32 * kk + 8 * g + i;
        const bool in = dcol < HD;
        xk[i] = (kvalid && in) ? k[base + (int64_t)tok * ld + dcol] : 0.f;
        xq[i] = (qvalid && in) ? q[base + (int64_t)tok * ld + dcol] * qmul : 0.f;
      }
      const int64_t off = qkbase + ((int64_t)(tok >> 4) * NK + kk) * (2 * FB) + ((tok & 15) + 16 * g) * 16;
      u32x4 hi, lo;
      split8h(xk, hi, lo);
      *reinterpret_cast<u32x4*>(ki + off) = hi;
      *reinterpret_cast<u32x4*>(ki + off + FB) = lo;
      split8h(xq, hi, lo);
      *reinterpret_cast<u32x4*>(qi + off) = hi;
      *reinterpret_cast<u32x4*>(qi + off + FB) = lo;
    }
  }
  for (int idx = tid; idx < KSTEP * HD; idx += 256) {
    const int tl = idx / HD, dcol = idx % HD, tok = t0 + tl;
    sv[tl][dcol] = tok < len ? v[base + (int64_t)tok * ld + dcol] : 0.f;
  }
  __syncthreads();
  for (int job = tid; job < 2 * NDV * 64; job += 256) {
    const int kg = job / (NDV * 64), dvt = (job / 64) % NDV, l = job & 63, dv = l & 15, g = l >> 4;
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = sv[32 * kg + 4 * g + (j & 3) + 16 * (j >> 2)][16 * dvt + dv];
    const int64_t off = vbase + (int64_t)(((t0 >> 5) + kg) * NDV + dvt) * (2 * FB) + l * 16;
    u32x4 hi, lo;
    split8h(xv, hi, lo);
    *reinterpret_cast<u32x4*>(vi + off) = hi;
    *reinterpret_cast<u32x4*>(vi + off + FB) = lo;
  }
}

template <int HD>
__global__ void __launch_bounds__(256, 2)
attn_h3_any_kernel(const char* __restrict__ qi, const char* __restrict__ ki, const char* __restrict__ vi,
                   float* __restrict__ o, const int64_t* __restrict__ num_ims, int T, int Tp, int H, int npairs, int nqb) {
  constexpr int NK = (HD + 31) / 32, NDV = HD / 16;
  constexpr int KB = 4 * NK * 2 * FB, VB = 2 * NDV * 2 * FB;      // bytes of K / V^T fragments per 64-key step
  constexpr int NCK = KB / 2048, NCV = VB / 2048, NC = NCK > NCV ? NCK : NCV;   // 16-byte pieces per carrying thread
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [2][KB] | [2][VB]
  char* sKb = smem;
  char* sVb = smem + 2 * KB;
  // XCD-aware placement as in attn_x6.hip: pair p only ever runs on the XCD group p % 8
  const int lin = blockIdx.x, xg = lin & 7, jx = lin >> 3;
  const int cnt = (npairs - xg + 7) >> 3;
  if (cnt <= 0) return;
  const int pair = xg + 8 * (jx % cnt), qb = jx / cnt;
  if (qb >= nqb) return;
  const int b = pair / H, head = pair - b * H, q0 = qb * 64 * QT;
  const int len = min((int)num_ims[b] + 1, T);
  if (q0 >= len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 15, g4 = lane >> 4;
  const int64_t qkbase = ((int64_t)b * H + head) * (int64_t)Tp * (NK * 32) * 4;
  const int64_t vbase = ((int64_t)b * H + head) * (int64_t)Tp * HD * 4;
  // A block with at most one wave's worth of queries (the last block of a completely full slide: 2,049 = 16 x 128 + 1 tokens) takes the
  // SMALL path below: its four waves share the one query tile pair and split the KEY steps (block-uniform branch).  As an ordinary block
  // it ran all 33 key steps with four waves of garbage queries - a second round of workgroups as long as the first (trans_dim 192,
  // level 0 of the bench's slides: 131 us against 82 for the ragged levels).
  // (head_dim 64 keeps its registers for the main loop: it is at the 256-register cap of two waves per SIMD already)
  const bool small = HD <= 48 && len - q0 <= 16 * QT;
  const int qw = small ? q0 : q0 + wave * 16 * QT;

  u32x4 qf[QT][NK][2];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int kk = 0; kk < NK; ++kk)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        qf[qt][kk][pl] = *reinterpret_cast<const u32x4*>(qi + qkbase + ((int64_t)(min(qw + 16 * qt, Tp - 16) >> 4) * NK + kk) * (2 * FB) + pl * FB + lane * 16);
  f32x4 oacc[NDV][QT];
#pragma unroll
  for (int i = 0; i < NDV; ++i)
#pragma unroll
    for (int j = 0; j < QT; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int j = 0; j < QT; ++j) { m_run[j] = -INFINITY; l_run[j] = 0.f; }

  const int nkt = (len + KSTEP - 1) / KSTEP;
  auto write_out = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const float l = rows_sum(l_run[qt]);
      const float inv = 1.0f / l;
      const int qidx = qw + 16 * qt + ql;
      if (qidx < T) {
        float* op = o + ((int64_t)b * T + qidx) * (H * HD) + head * HD + 4 * g4;
#pragma unroll
        for (int dvt = 0; dvt < NDV; ++dvt) *reinterpret_cast<f32x4*>(op + 16 * dvt) = oacc[dvt][qt] * inv;
      }
    }
  };
  if constexpr (HD <= 48) if (small) {
    // ---- SMALL path: key steps wave, wave + 4, ... per wave, fragments straight from the images in global memory (operand order),
    // the four (m, l, O) states merged through LDS by wave 0
    for (int kt = wave; kt < nkt; kt += 4) {
      f32x4 s[QT][4];
      const char* gK = ki + qkbase + (int64_t)kt * KB + lane * 16;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          u32x4 kf[2];
          kf[0] = *reinterpret_cast<const u32x4*>(gK + (t * NK + kk) * (2 * FB));
          kf[1] = *reinterpret_cast<const u32x4*>(gK + (t * NK + kk) * (2 * FB) + FB);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) s[qt][t] = mfma_split(kf, qf[qt][kk], s[qt][t]);
        }
      }
      if (kt == nkt - 1) {
        const int kbase = kt * KSTEP + 4 * g4;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kbase + 16 * t + r >= len) s[qt][t][r] = -INFINITY;
      }
      u32x4 pf[QT][2][2];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          mx = fmaxf(fmaxf(mx, s[qt][t][0]), s[qt][t][1]);
          mx = fmaxf(fmaxf(mx, s[qt][t][2]), s[qt][t][3]);
        }
        mx = rows_max(mx);
        const float m_new = fmaxf(m_run[qt], mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
          float pv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            pv[j] = __builtin_amdgcn_exp2f(s[qt][2 * kg + (j >> 2)][j & 3] - m_new);
            psum += pv[j];
          }
          split8h(pv, pf[qt][kg][0], pf[qt][kg][1]);
        }
        l_run[qt] = l_run[qt] * alpha + psum;
        m_run[qt] = m_new;
#pragma unroll
        for (int dvt = 0; dvt < NDV; ++dvt) oacc[dvt][qt] *= alpha;
      }
      const char* gV = vi + vbase + (int64_t)kt * VB + lane * 16;
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int dvt = 0; dvt < NDV; ++dvt) {
          u32x4 vf[2];
          vf[0] = *reinterpret_cast<const u32x4*>(gV + (kg * NDV + dvt) * (2 * FB));
          vf[1] = *reinterpret_cast<const u32x4*>(gV + (kg * NDV + dvt) * (2 * FB) + FB);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) oacc[dvt][qt] = mfma_split(vf, pf[qt][kg], oacc[dvt][qt]);
        }
    }
    // per-lane states -> LDS [4 waves][2 QT + 4 NDV QT][64]  (a wave without a key step: m = -inf, l = 0, O = 0: weight 0 in the merge)
    constexpr int NST = 2 * QT + 4 * NDV * QT;
    static_assert(4 * NST * 64 * 4 <= 2 * (KB + VB), "the merge scratch fits the staging LDS");
    float* const stf = reinterpret_cast<float*>(smem) + wave * (NST * 64);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      stf[qt * 64 + lane] = m_run[qt];
      stf[(QT + qt) * 64 + lane] = l_run[qt];
#pragma unroll
      for (int dvt = 0; dvt < NDV; ++dvt)
#pragma unroll
        for (int r = 0; r < 4; ++r) stf[(2 * QT + (dvt * QT + qt) * 4 + r) * 64 + lane] = oacc[dvt][qt][r];
    }
    __syncthreads();
    if (wave != 0) return;
    const float* const all = reinterpret_cast<const float*>(smem);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float M = -INFINITY;
#pragma nounroll
      for (int w = 0; w < 4; ++w) M = fmaxf(M, all[w * (NST * 64) + qt * 64 + lane]);
      float lsum = 0.f;
#pragma unroll
      for (int dvt = 0; dvt < NDV; ++dvt) oacc[dvt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma nounroll
      for (int w = 0; w < 4; ++w) {
        const float* a = all + w * (NST * 64);
        const float sc = __builtin_amdgcn_exp2f(a[qt * 64 + lane] - M);
        lsum = fmaf(a[(QT + qt) * 64 + lane], sc, lsum);
#pragma unroll
        for (int dvt = 0; dvt < NDV; ++dvt)
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[dvt][qt][r] = fmaf(a[(2 * QT + (dvt * QT + qt) * 4 + r) * 64 + lane], sc, oacc[dvt][qt][r]);
      }
      l_run[qt] = lsum;
      m_run[qt] = M;
    }
    write_out();
    return;
  }
  // staging: threads 0-127 carry the K fragments of a 64-key step (one step ahead), 128-255 its V^T fragments
  const bool carriesK = tid < 128;
  const int chunk = (tid & 127) * 16;
  u32x4 st[NC];
  auto gload = [&](int ktk, int ktv) {        // K of step ktk / V of step ktv (the caller checks the ranges)
    const char* src = carriesK ? ki + qkbase + (int64_t)ktk * KB : vi + vbase + (int64_t)ktv * VB;
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (c < (carriesK ? NCK : NCV)) st[c] = *reinterpret_cast<const u32x4*>(src + chunk + 2048 * c);
  };
  auto swrite = [&](int ktk, int ktv, bool dok, bool dov) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (carriesK) { if (dok && c < NCK) *reinterpret_cast<u32x4*>(sKb + (ktk & 1) * KB + chunk + 2048 * c) = st[c]; }
      else if (dov && c < NCV) *reinterpret_cast<u32x4*>(sVb + (ktv & 1) * VB + chunk + 2048 * c) = st[c];
    }
  };
  auto qk = [&](int kt, f32x4 (&s)[QT][4]) __attribute__((always_inline)) {
    const char* sK = sKb + (kt & 1) * KB + lane * 16;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) s[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        u32x4 kf[2];
        kf[0] = *reinterpret_cast<const u32x4*>(sK + (t * NK + kk) * (2 * FB));
        kf[1] = *reinterpret_cast<const u32x4*>(sK + (t * NK + kk) * (2 * FB) + FB);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[qt][t] = mfma_split(kf, qf[qt][kk], s[qt][t]);
      }
    }
  };
  gload(0, 0);
  swrite(0, 0, true, true);
  if (nkt > 1) { if (carriesK) gload(1, 0); swrite(1, 0, true, false); }
  __syncthreads();
  f32x4 sA[QT][4], sB[QT][4];
  qk(0, sA);
  auto step = [&](int kt, f32x4 (&s)[QT][4], f32x4 (&sn)[QT][4], bool last) __attribute__((always_inline)) {
    const bool morek = kt + 2 < nkt, morev = kt + 1 < nkt;
    if (carriesK ? morek : morev) gload(kt + 2, kt + 1);
    const char* sV = sVb + (kt & 1) * VB + lane * 16;
    if (last) {
      const int kbase = kt * KSTEP + 4 * g4;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kbase + 16 * t + r >= len) s[qt][t][r] = -INFINITY;
    }
    qk(kt + 1, sn);                                     // (past the end: stale K fragments, finite garbage nobody reads)
    u32x4 pf[QT][2][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        mx = fmaxf(fmaxf(mx, s[qt][t][0]), s[qt][t][1]);
        mx = fmaxf(fmaxf(mx, s[qt][t][2]), s[qt][t][3]);
      }
      mx = rows_max(mx);
      const float m_new = fmaxf(m_run[qt], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
      float psum = 0.f;
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                   // k-slot (g4, j) of the PV product = key 4 g4 + (j&3) + 16 (j>>2) of the group
          pv[j] = __builtin_amdgcn_exp2f(s[qt][2 * kg + (j >> 2)][j & 3] - m_new);
          psum += pv[j];
        }
        split8h(pv, pf[qt][kg][0], pf[qt][kg][1]);
      }
      l_run[qt] = l_run[qt] * alpha + psum;
      m_run[qt] = m_new;
#pragma unroll
      for (int dvt = 0; dvt < NDV; ++dvt) oacc[dvt][qt] *= alpha;
    }
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int dvt = 0; dvt < NDV; ++dvt) {
        u32x4 vf[2];
        vf[0] = *reinterpret_cast<const u32x4*>(sV + (kg * NDV + dvt) * (2 * FB));
        vf[1] = *reinterpret_cast<const u32x4*>(sV + (kg * NDV + dvt) * (2 * FB) + FB);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) oacc[dvt][qt] = mfma_split(vf, pf[qt][kg], oacc[dvt][qt]);
      }
    swrite(kt + 2, kt + 1, morek, morev);               // K over K(kt) (read one step ago), V over V(kt-1)
    __syncthreads();
  };
  {
    int kt = 0;
    for (; kt + 2 < nkt; kt += 2) {
      step(kt, sA, sB, false);
      step(kt + 1, sB, sA, false);
    }
    if (kt + 1 < nkt) { step(kt, sA, sB, false); step(kt + 1, sB, sA, true); }
    else step(kt, sA, sB, true);
  }
  write_out();
}

template <int HD>
int launch_h3_any(const float* qkv, int64_t ld, float qscale, float* o, const int64_t* num_ims, int B, int T, int H, void* workspace,
                  hipStream_t stream, bool images_ready = false) {
  constexpr int NK = (HD + 31) / 32, NDV = HD / 16;
  constexpr int KB = 4 * NK * 2 * FB, VB = 2 * NDV * 2 * FB;
  const int Tp = (T + KSTEP - 1) / KSTEP * KSTEP;
  const int d = H * HD;
  const int64_t qk_img = (int64_t)B * H * Tp * (NK * 32) * 4;
  char* qi = reinterpret_cast<char*>(workspace);
  char* ki = qi + qk_img;
  char* vi = ki + qk_img;
  if (!images_ready) {
    hipLaunchKernelGGL(attn_h3_prep_kernel<HD>, dim3(Tp / KSTEP, H, B), dim3(256), 0, stream, qkv, qkv + d, qkv + 2 * d, ld, (int64_t)HD,
                       (int64_t)T * ld, qscale, qi, ki, vi, num_ims, T, Tp, H);
    PATHS_LAUNCH_CHECK("attention_h3_any(prep)");
  }
  PATHS_LDS_OPT_IN(attn_h3_any_kernel<HD>, 2 * (KB + VB), "attention_h3_any");
  const int nqb = (T + 64 * QT - 1) / (64 * QT), npairs = H * B;
  hipLaunchKernelGGL(attn_h3_any_kernel<HD>, dim3(8 * ((npairs + 7) / 8) * nqb), dim3(256), 2 * (KB + VB), stream, qi, ki, vi, o, num_ims, T, Tp, H,
                     npairs, nqb);
  PATHS_LAUNCH_CHECK("attention_h3_any");
  return PATHS_OK;
}

}  // namespace

extern "C" {

// bytes of the workspace paths_attention_h3_any needs (three two-plane fp16 fragment images)
int64_t paths_attention_h3_any_workspace(int B, int T, int H, int head_dim) {
  const int64_t Tp = ((int64_t)T + KSTEP - 1) / KSTEP * KSTEP;
  const int64_t hdp = (head_dim + 31) / 32 * 32;
  return 3 * (int64_t)B * H * Tp * hdp * 4;
}

// o[B, T, H*hd] = softmax(q k^T / sqrt(hd)) v per head, keys >= num_ims[b] + 1 masked, on the token-major in_proj output qkv [B*T, 3d]
// (row stride ld; q unscaled, qscale = log2(e) / sqrt(head_dim)); head_dim in {16, 32, 48, 64}; two fp16 planes per operand.
int paths_attention_h3_any(const float* qkv, int64_t ld, float* o, const int64_t* num_ims, int B, int T, int H, int head_dim, float qscale,
                           void* workspace, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && T > 0 && H > 0 && qkv && o && num_ims && workspace && ld >= 3 * H * head_dim, "attention_h3_any: bad arguments B=%d T=%d H=%d", B, T, H);
  PATHS_REQUIRE(((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)workspace) % 16 == 0, "attention_h3_any: buffers must be 16-byte aligned");
  switch (head_dim) {
    case 16: return launch_h3_any<16>(qkv, ld, qscale, o, num_ims, B, T, H, workspace, stream);
    case 32: return launch_h3_any<32>(qkv, ld, qscale, o, num_ims, B, T, H, workspace, stream);
    case 48: return launch_h3_any<48>(qkv, ld, qscale, o, num_ims, B, T, H, workspace, stream);
    case 64: return launch_h3_any<64>(qkv, ld, qscale, o, num_ims, B, T, H, workspace, stream);
    default: return paths_set_error(PATHS_EUNSUPPORTED, "attention_h3_any: head_dim %d (supported: 16, 32, 48, 64)", head_dim);
  }
}

// The same attention on operand images that are already in the workspace (paths_token_layer_ws wrote them: trans_dim 192 / 4 heads =
// head_dim 48; q scaled there): no prep launch.
int paths_attention_h3_any_img(float* o, const int64_t* num_ims, int B, int T, int H, int head_dim, void* workspace, hipStream_t stream) {
  PATHS_REQUIRE(B > 0 && T > 0 && H > 0 && o && num_ims && workspace, "attention_h3_any_img: bad arguments B=%d T=%d H=%d", B, T, H);
  PATHS_REQUIRE(((uintptr_t)o | (uintptr_t)workspace) % 16 == 0, "attention_h3_any_img: buffers must be 16-byte aligned");
  switch (head_dim) {
    case 32: return launch_h3_any<32>(nullptr, 0, 1.0f, o, num_ims, B, T, H, workspace, stream, true);
    case 48: return launch_h3_any<48>(nullptr, 0, 1.0f, o, num_ims, B, T, H, workspace, stream, true);
    default: return paths_set_error(PATHS_EUNSUPPORTED, "attention_h3_any_img: head_dim %d (supported: 32, 48)", head_dim);
  }
}

}  // extern "C"
