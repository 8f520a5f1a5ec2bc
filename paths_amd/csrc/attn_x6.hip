// Fused masked self-attention on the bf16 matrix cores with fp32 accuracy (the "x6" operand split of gemm_x6.hip).
//
// Same contract as paths_attention_f32 (attn_f32.hip; reference model/aggregator.py:70-72 + utils.py:97-103): q, k, v
// head-major [B][h][T][32] fp32, q pre-scaled by log2(e)/sqrt(hd), o token-major [B][T][h*32], keys >= num_ims[b]+1 masked.
//
// Two launches:
//  1. attn_x6_prep_kernel: every fp32 value of q, k, v becomes three bf16 (hi + mid + lo = the value, exactly), stored as
//     MFMA FRAGMENTS: one 16-row x 32-k operand of v_mfma_f32_16x16x32_bf16 for one plane is 1 KiB, lane l owns bytes
//     [16 l, 16 l + 16) = its 8 k-values.  Q and K fragments: rows = tokens, k = the 32 head dims.  V fragments are
//     TRANSPOSED (rows = 16 head dims, k = 32 keys) and key-PERMUTED, k-slot (g, j) <-> key 4g + (j&3) + 16 (j>>2), which is
//     the order in which the S^T accumulators of two 16-key tiles sit in a lane - so P never leaves the registers.
//     Keys that are masked (or beyond T) are written as zeros.
//  2. attn_x6_kernel: one wave = 32 queries (two 16-query tiles) of one (slide, head); a 4-wave workgroup shares 64-key
//     K / V^T fragment sets through LDS (plain 16-byte copies, double-buffered).  Per 32-key group and wave:
//        S^T[key][q] = K Q^T          2 key tiles x 2 query tiles x 6 MFMAs (A = K fragment, B = Q fragment in registers)
//        online softmax               per lane: one query of each query tile, keys in registers, 2 cross-lane swaps
//        P^T split in registers       3 x bf16x8 per query tile
//        O^T[dv][q] += V^T P^T        2 dv tiles x 2 query tiles x 6 MFMAs
//     48 MFMAs of 16 cycles against 128 of 32 cycles (v_mfma_f32_16x16x4_f32) in the f32 kernel for the same 32 x 32 block.
// The six kept partial products per operand pair are accumulated smallest first, exactly as in gemm_x6.hip.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 32;
constexpr int FRAG = 1024;                 // bytes of one fragment (64 lanes x 8 bf16)
constexpr int KSTEP = 64;                  // keys staged per LDS buffer
// per 64 keys: K = 4 key tiles x 3 planes, V^T = 2 key groups x 2 dv tiles x 3 planes -> 24 fragments = 24 KiB
constexpr int STEP_BYTES = 24 * FRAG;

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// 8 fp32 -> three planes of 8 bf16 (hi, mid, lo), exact
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const uint32_t h = pk_bf16(a, b);
    const float ra = a - bf_lo(h), rb = b - bf_hi(h);
    const uint32_t m = pk_bf16(ra, rb);
    const float sa = ra - bf_lo(m), sb = rb - bf_hi(m);
    hi[i] = h; mid[i] = m; lo[i] = pk_bf16(sa, sb);
  }
}

// Fragment images (per (slide, head), Tp = T rounded up to 64):
//   Q6 / K6 : [Tp/16 tiles][3 planes][64 lanes][8 bf16]        lane (r = l&15, g = l>>4): token 16 tile + r, dims 8g .. 8g+7
//   V6      : [Tp/32 groups][2 dv tiles][3 planes][64 lanes][8] lane (dv = l&15, g):      dim 16 dvt + dv, keys 32 grp + kappa(g, j)
__global__ void __launch_bounds__(256)
attn_x6_prep_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                    char* __restrict__ q6, char* __restrict__ k6, char* __restrict__ v6,
                    const int64_t* __restrict__ num_ims, int T, int Tp, int H, int nq) {
  __shared__ float sv[KSTEP][HD + 1];
  const int b = blockIdx.z, head = blockIdx.y, t0 = blockIdx.x * KSTEP;
  const int len = min((int)num_ims[b] + 1, T);
  const int tid = threadIdx.x;
  const int64_t base = ((int64_t)b * H + head) * T * HD;
  const int64_t ibase = ((int64_t)b * H + head) * (int64_t)Tp * HD * 6;     // bytes of one (slide, head) image
  // ---- Q and K: thread = (token t0 + tid/4, dims 8 (tid%4) ..)
  {
    const int tl = tid >> 2, g = tid & 3, tok = t0 + tl;
    float xq[8], xk[8];
    const bool kvalid = tok < len, qvalid = tok < T && tok < nq;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xk[i] = kvalid ? k[base + (int64_t)tok * HD + 8 * g + i] : 0.f;
      xq[i] = qvalid ? q[base + (int64_t)tok * HD + 8 * g + i] : 0.f;
    }
    u32x4 h, m, l;
    const int64_t off = ibase + ((int64_t)(tok >> 4) * 3) * FRAG + ((tok & 15) + 16 * g) * 16;
    split8(xk, h, m, l);
    *reinterpret_cast<u32x4*>(k6 + off) = h; *reinterpret_cast<u32x4*>(k6 + off + FRAG) = m; *reinterpret_cast<u32x4*>(k6 + off + 2 * FRAG) = l;
    split8(xq, h, m, l);
    *reinterpret_cast<u32x4*>(q6 + off) = h; *reinterpret_cast<u32x4*>(q6 + off + FRAG) = m; *reinterpret_cast<u32x4*>(q6 + off + 2 * FRAG) = l;
  }
  // ---- V^T: through LDS (coalesced rows in, transposed + key-permuted fragments out)
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int idx = tid + 256 * p, tl = idx >> 5, dcol = idx & 31, tok = t0 + tl;
    sv[tl][dcol] = tok < len ? v[base + (int64_t)tok * HD + dcol] : 0.f;
  }
  __syncthreads();
  {
    const int kg = tid >> 7, dvt = (tid >> 6) & 1, l = tid & 63, dv = l & 15, g = l >> 4;
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = sv[32 * kg + 4 * g + (j & 3) + 16 * (j >> 2)][16 * dvt + dv];
    u32x4 h, m, lo;
    split8(xv, h, m, lo);
    const int64_t off = ibase + ((int64_t)(((t0 >> 5) + kg) * 2 + dvt) * 3) * FRAG + l * 16;
    *reinterpret_cast<u32x4*>(v6 + off) = h; *reinterpret_cast<u32x4*>(v6 + off + FRAG) = m; *reinterpret_cast<u32x4*>(v6 + off + 2 * FRAG) = lo;
  }
}

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// acc += A * B with A = (a[0] hi, a[1] mid, a[2] lo), B alike: the six largest partial products, smallest first
__device__ __forceinline__ f32x4 mfma_x6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c) {
  c = mfma_bf16(a[2], b[0], c);
  c = mfma_bf16(a[0], b[2], c);
  c = mfma_bf16(a[1], b[1], c);
  c = mfma_bf16(a[1], b[0], c);
  c = mfma_bf16(a[0], b[1], c);
  c = mfma_bf16(a[0], b[0], c);
  return c;
}

__global__ void __launch_bounds__(256)
attn_x6_kernel(const char* __restrict__ q6, const char* __restrict__ k6, const char* __restrict__ v6,
               float* __restrict__ o, float* __restrict__ lse, const int64_t* __restrict__ num_ims, int T, int Tp, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];          // 2 x STEP_BYTES (+ occupancy padding, see the launcher)
  char (*smem)[STEP_BYTES] = reinterpret_cast<char (*)[STEP_BYTES]>(smem_raw);
  const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 128;
  const int len = min((int)num_ims[b] + 1, T);          // valid keys = special token + patches
  if (q0 >= len) return;                                // every query of this block is padding
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 15, g4 = lane >> 4;
  const int64_t ibase = ((int64_t)b * H + head) * (int64_t)Tp * HD * 6;
  const int qw = q0 + wave * 32;                        // this wave's first query

  // Q fragments (B operand of S^T): two 16-query tiles x 3 planes, kept in registers
  bf16x8 qf[2][3];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int p = 0; p < 3; ++p)
      qf[qt][p] = *reinterpret_cast<const bf16x8*>(q6 + ibase + ((int64_t)(min(qw + 16 * qt, Tp - 16) >> 4) * 3 + p) * FRAG + lane * 16);

  f32x4 oacc[2][2];                                     // [dv tile][query tile]: rows = dims 4 g4 .. +3, col = query ql
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

  // staging: one 64-key step = 12 KiB of K fragments + 12 KiB of V^T fragments, both contiguous in their images
  const int nkt = (len + KSTEP - 1) / KSTEP;
  u32x4 st[6];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      st[i] = *reinterpret_cast<const u32x4*>(k6 + ibase + (int64_t)kt * 12 * FRAG + (tid + 256 * i) * 16);
      st[3 + i] = *reinterpret_cast<const u32x4*>(v6 + ibase + (int64_t)kt * 12 * FRAG + (tid + 256 * i) * 16);
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      *reinterpret_cast<u32x4*>(smem[buf] + (tid + 256 * i) * 16) = st[i];
      *reinterpret_cast<u32x4*>(smem[buf] + 12 * FRAG + (tid + 256 * i) * 16) = st[3 + i];
    }
  };
  gload(0);
  swrite(0);
  __syncthreads();
  int buf = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) gload(kt + 1);
    const char* sK = smem[buf] + lane * 16;
    const char* sV = smem[buf] + 12 * FRAG + lane * 16;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      // ---- S^T = K Q^T: key tiles 2 kg, 2 kg + 1 of this step
      f32x4 s[2][2];                                    // [query tile][key tile]: rows = keys 4 g4 .. +3, col = query ql
      bf16x8 kf[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) kf[t][p] = *reinterpret_cast<const bf16x8*>(sK + ((2 * kg + t) * 3 + p) * FRAG);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int t = 0; t < 2; ++t) s[qt][t] = mfma_x6(kf[t], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
      // ---- mask (last step only) + online softmax (lane: query ql of each tile; keys 16 t + 4 g4 + r)
      if (kt == nkt - 1) {
        const int kbase = kt * KSTEP + 32 * kg + 4 * g4;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kbase + 16 * t + r >= len) s[qt][t][r] = -INFINITY;
      }
      bf16x8 pf[2][3];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        float mx = fmaxf(fmaxf(fmaxf(s[qt][0][0], s[qt][0][1]), fmaxf(s[qt][0][2], s[qt][0][3])),
                         fmaxf(fmaxf(s[qt][1][0], s[qt][1][1]), fmaxf(s[qt][1][2], s[qt][1][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run[qt], mx);       // finite: key 0 (special token) is always valid
        const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
        float pv[8], psum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {                   // k-slot (g4, j) of the PV product = key 4 g4 + (j&3) + 16 (j>>2)
          pv[j] = __builtin_amdgcn_exp2f(s[qt][j >> 2][j & 3] - m_new);
          psum += pv[j];
        }
        l_run[qt] = l_run[qt] * alpha + psum;
        m_run[qt] = m_new;
        oacc[0][qt] *= alpha;
        oacc[1][qt] *= alpha;
        u32x4 h, m, l;
        split8(pv, h, m, l);
        pf[qt][0] = __builtin_bit_cast(bf16x8, h); pf[qt][1] = __builtin_bit_cast(bf16x8, m); pf[qt][2] = __builtin_bit_cast(bf16x8, l);
      }
      // ---- O^T += V^T P^T
#pragma unroll
      for (int dvt = 0; dvt < 2; ++dvt) {
        bf16x8 vf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) vf[p] = *reinterpret_cast<const bf16x8*>(sV + ((kg * 2 + dvt) * 3 + p) * FRAG);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) oacc[dvt][qt] = mfma_x6(vf, pf[qt], oacc[dvt][qt]);
      }
    }
    if (kt + 1 < nkt) swrite(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float l = l_run[qt];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    const int qi = qw + 16 * qt + ql;
    if (qi < T) {
      float* op = o + ((int64_t)b * T + qi) * (H * HD) + head * HD + 4 * g4;
      *reinterpret_cast<f32x4*>(op) = oacc[0][qt] * inv;
      *reinterpret_cast<f32x4*>(op + 16) = oacc[1][qt] * inv;
      if (lse && g4 == 0) lse[((int64_t)b * H + head) * T + qi] = m_run[qt] + log2f(l);
    }
  }
}

}  // namespace

extern "C" {

// bytes of the workspace paths_attention_x6 needs (three fragment images)
int64_t paths_attention_x6_workspace(int B, int T, int H, int head_dim) {
  const int64_t Tp = ((int64_t)T + KSTEP - 1) / KSTEP * KSTEP;
  return 3 * (int64_t)B * H * Tp * head_dim * 6;
}

int paths_attention_x6(const float* q, const float* k, const float* v, float* o, float* lse /*[B,H,T] or null*/,
                       const int64_t* num_ims, int B, int T, int H, int head_dim, int max_queries, void* workspace,
                       hipStream_t stream) {
  PATHS_REQUIRE(head_dim == HD, "attention_x6: head_dim must be %d (got %d)", HD, head_dim);
  PATHS_REQUIRE(B > 0 && T > 0 && H > 0 && num_ims != nullptr && workspace != nullptr, "attention_x6: bad arguments B=%d T=%d H=%d", B, T, H);
  PATHS_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)workspace) % 16 == 0, "attention_x6: buffers must be 16-byte aligned");
  const int Tp = (T + KSTEP - 1) / KSTEP * KSTEP;
  const int64_t img = (int64_t)B * H * Tp * HD * 6;
  char* q6 = reinterpret_cast<char*>(workspace);
  char* k6 = q6 + img;
  char* v6 = k6 + img;
  const int nq = max_queries > 0 && max_queries < T ? max_queries : T;
  hipLaunchKernelGGL(attn_x6_prep_kernel, dim3(Tp / KSTEP, H, B), dim3(256), 0, stream, q, k, v, q6, k6, v6, num_ims, T, Tp, H, nq);
  PATHS_LAUNCH_CHECK("attention_x6(prep)");
  // Workgroups per CU: registers allow 2 (170 VGPRs; capping them at 168 for 3 cost more than it gave), LDS would allow 3.
  // The dispatcher fills a CU to its limit before it moves on, so small grids ask for more LDS than needed to spread out:
  // depth = ceil(grid / 256).
  const int nblk = ((nq + 127) / 128) * H * B;
  const int depth = nblk <= 256 ? 1 : nblk <= 512 ? 2 : 3;
  const int lds = depth == 1 ? 96 * 1024 : depth == 2 ? 64 * 1024 : 2 * STEP_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_x6_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_x6_kernel, dim3((nq + 127) / 128, H, B), dim3(256), lds, stream, q6, k6, v6, o, lse, num_ims, T, Tp, H);
  PATHS_LAUNCH_CHECK("attention_x6");
  return PATHS_OK;
}

}  // extern "C"
