// LAST decoder layer at token 0 only + decoder.norm + slide-context residual / concat + classifier, ONE launch, and without the
// layer's K / V projections (reference model/aggregator.py:70-75 for the final layer, model/paths.py:130-139).
//
// Only out[:, 0] of the final layer is read.  paths_token0_tail (tlayer_f32.hip) already evaluates one query per slide, but it
// needs k = Wk x + bk and v = Wv x + bv of EVERY token (a whole in_proj launch over [B, T, d], 3 MB of fp32 q, k, v per slide).
// With one query per head the projections fold into the query and the output (exact algebra, fp32 rounding differs at 1e-7):
//   score_h,t = q_h . (Wk_h x_t + bk_h) = (Wk_h^T q_h) . x_t + const    ->   softmax over t does not see the constant
//   o_h       = sum_t p_h,t (Wv_h x_t + bv_h) = Wv_h (sum_t p_h,t x_t) + bv_h                    (sum_t p = 1)
// and q_h = Wq_h x_0 + bq_h, so qt_h = Wk_h^T q_h * log2(e)/sqrt(hd) = A_h x_0 + a0_h with A_h = c Wk_h^T Wq_h, a0_h = c Wk_h^T bq_h
// built once per weight version (paths_token0_pack_ws).  What is left per slide is O(T d) work on the layer INPUT rows x_t:
//   phase 0  qt_h = A_h x_0 + a0_h                               (one 128 x 128 GEMV per workgroup)
//   phase 1  s_t = qt_h . x_t, online softmax, z_h = sum_t p_t x_t   over this workgroup's token range (one head per workgroup)
//   publish  (m, l, z[128]) partial -> global, agent-scope release, arrival ticket per slide
//   phase 2  LAST arriver of the slide: merge partials, o = Wv z + bv, out_proj, norm1, + cross-attn bias, norm2, FFN, norm3,
//            decoder.norm, slide-context residual, classifier  -  GEMVs over transposed fp32 weights (lane = output row, 16-byte
//            coalesced loads, no cross-lane reductions), all weight loads of the chain issued before the first dependent stage.
// Exact fp32 FMA chains throughout (no operand split).
#include "common.h"

namespace {

constexpr int DM = 128, DFF = 512, NH = 4, HD = 32;
constexpr int NT = 512;                      // threads per workgroup (8 waves)
constexpr int SLOTS = NT / 32;               // half-waves: one token each per iteration
constexpr int REC = 4 + DM;                  // floats of one partial record: m, l, pad, pad, z[128]
constexpr int MAX_TS = 16;                   // token splits per (slide, head)
constexpr int TS_TOKENS = 128;               // tokens per split (up to MAX_TS splits): two 64-token rounds, all loads issued up front

// offsets (floats) into the packed weight image of paths_token0_pack_ws
constexpr int OFF_A = 0;                                 // [4 heads][32 k4][128 c][4]  A_h^T4
constexpr int OFF_A0 = OFF_A + NH * DM * DM;             // [4][128]
constexpr int OFF_WV = OFF_A0 + NH * DM;                 // [32 k4][128 f][4]
constexpr int OFF_WO = OFF_WV + DM * DM;                 // [32 k4][128 f][4]
constexpr int OFF_W1 = OFF_WO + DM * DM;                 // [32 k4][512 n][4]
constexpr int OFF_W2 = OFF_W1 + DFF * DM;                // [128 k4][128 f][4]
constexpr int IMG_FLOATS = OFF_W2 + DM * DFF;

#ifdef PATHS_T0_STAMPS
#define T0_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (p.stamps && tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define T0_STAMP(i) do { } while (0)
#endif

struct T0Params {
  const float* x1;                 // [B][T][128] input rows of the last layer
  const int64_t* num_ims;
  const float* img;                // packed weights (above)
  const float *bv, *bo, *ln1g, *ln1b, *cab, *ln2g, *ln2b, *b1, *b2, *ln3g, *ln3b, *lnfg, *lnfb;
  const float* ctx_prev; int64_t ctx_stride; const float* ctx_all; int ctx_depth;
  const float* wcls; const float* bcls; int num_logits, cls_in;
  float* ctx_out; float* logits;
  float* partials;                 // [B][4][nts][REC]
  int* counters;                   // [B] arrival tickets, zero on entry, left zero
  int T, nts; float eps, eps_f;
#ifdef PATHS_T0_STAMPS
  unsigned long long* stamps;
#endif
};

// sum over the 16 lanes of a row (DPP: two quad permutes, two mirrors), over 32 (+ v_permlane16_swap) and over 64 lanes
// (+ v_permlane32_swap); every lane ends up with the total.  No LDS round trips (ds_bpermute) in the dependent chains.
__device__ __forceinline__ float row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}
__device__ __forceinline__ float half_sum32(float v) {
  float a = row_sum16(v), b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // (inline asm: see tlayer_ws.hip sum_xor16)
  return a + b;
}
__device__ __forceinline__ float wave_sum64(float v) {
  float a = half_sum32(v), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
// partial records are written by OTHER workgroups of this launch: read them on the vector path with agent-scope (sc1) loads - a
// wave-uniform address would otherwise become an s_load through the scalar cache, which the acquire fence does not invalidate
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]))); }

// wave 0 normalises v[0:128] in place (2 values per lane); every thread must call it
__device__ __forceinline__ void block_layernorm(float* v, const float* g, const float* bta, float eps, int tid) {
  if (tid < 64) {
    const float a = v[tid], c = v[tid + 64];
    const float mean = wave_sum64(a + c) * (1.0f / DM);
    const float da = a - mean, dc = c - mean;
    const float rstd = 1.0f / sqrtf(wave_sum64(da * da + dc * dc) * (1.0f / DM) + eps);
    v[tid] = da * rstd * g[tid] + bta[tid];
    v[tid + 64] = dc * rstd * g[tid + 64] + bta[tid + 64];
  }
  __syncthreads();
}

__global__ void __launch_bounds__(NT)
token0_ws_kernel(T0Params p) {
  __shared__ __attribute__((aligned(16))) float smem[SLOTS * DM + 4 * DFF + 8 * DM + 64];
  float* const sZ = smem;                    // phase 1: [16 slots][128]; phase 2: zc[4][128] | scratch
  float* const sRed = smem + SLOTS * DM;     // [4][512] k-split partial sums
  float* const sV = sRed + 4 * DFF;          // vectors: x0 | qt | o | xa | y | h (512, overlays sRed rows? no: own) ...
  float* const sX0 = sV, *const sQ = sV + DM, *const sO = sV + 2 * DM, *const sXa = sV + 3 * DM, *const sY = sV + 4 * DM;
  float* const sML = sV + 5 * DM;            // [16][2] slot (m, l)
  int* const sFlag = reinterpret_cast<int*>(sV + 5 * DM + 2 * SLOTS);
  const int b = blockIdx.y, head = blockIdx.x & 3, ts = blockIdx.x >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = p.x1 + (int64_t)b * p.T * DM;

  T0_STAMP(0);
  // ---- phase 0: qt = A_head x0 + a0_head
  if (tid < DM) sX0[tid] = xb[tid];
  __syncthreads();
  {
    const int c = tid & 127, kq = tid >> 7;
    const float* A = p.img + OFF_A + (int64_t)head * DM * DM;
    f32x4 w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = ldg_f32x4(A + ((8 * kq + i) * DM + c) * 4);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += dot4(w[i], *reinterpret_cast<const f32x4*>(sX0 + 32 * kq + 4 * i));
    sRed[kq * DM + c] = acc;
  }
  __syncthreads();
  if (tid < DM) sQ[tid] = ((sRed[tid] + sRed[DM + tid]) + (sRed[2 * DM + tid] + sRed[3 * DM + tid])) + p.img[OFF_A0 + head * DM + tid];
  __syncthreads();

  T0_STAMP(1);
  // ---- phase 1: this workgroup's token range, one token per half-wave and iteration, 4 features per lane
  const int len = min((int)p.num_ims[b] + 1, p.T);
  const int chunk = (len + p.nts - 1) / p.nts;
  const int k0 = ts * chunk, k1 = min(len, k0 + chunk);
  const int l5 = lane & 31, slot = wave * 2 + (lane >> 5);
  const f32x4 qv = *reinterpret_cast<const f32x4*>(sQ + 4 * l5);
  float m = -1e30f, l = 0.f;
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  // rounds of 8 tokens per half-wave (128 per workgroup): all 8 row pieces are in flight before the first dot product
  for (int base = k0; base < k1; base += 8 * SLOTS) {
    f32x4 x[8];
    float s[8];
    bool ok[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int tok = base + u * SLOTS + slot;
      ok[u] = tok < k1;
      x[u] = ldg_f32x4(xb + (int64_t)min(tok, k1 - 1) * DM + 4 * l5);
    }
    float mx = m;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s[u] = ok[u] ? half_sum32(dot4(qv, x[u])) : -1e30f;
      mx = fmaxf(mx, s[u]);
    }
    const float alpha = __builtin_amdgcn_exp2f(m - mx);
    float ps = 0.f;
    z = z * alpha;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float pu = ok[u] ? __builtin_amdgcn_exp2f(s[u] - mx) : 0.f;
      ps += pu;
      z = z + x[u] * pu;
    }
    l = l * alpha + ps;
    m = mx;
  }
  T0_STAMP(2);
  *reinterpret_cast<f32x4*>(sZ + slot * DM + 4 * l5) = z;
  if (l5 == 0) { sML[2 * slot] = m; sML[2 * slot + 1] = l; }
  __syncthreads();
  float* rec = p.partials + (((int64_t)b * NH + head) * p.nts + ts) * REC;
  if (tid < DM) {
    float M = -1e30f;
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) M = fmaxf(M, sML[2 * sl]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      const float w = __builtin_amdgcn_exp2f(sML[2 * sl] - M);
      num = fmaf(sZ[sl * DM + tid], w, num);
      den = fmaf(sML[2 * sl + 1], w, den);
    }
    st_agent(rec + 4 + tid, num);
    if (tid == 0) { st_agent(rec, M); st_agent(rec + 1, den); }
  }
  // ---- publish: the record is stored write-through (sc1: no release fence, no L2 write-back), every storing wave drains its
  // stores, then one lane draws the slide's ticket; the last arriver reads the records with sc1 loads (L1 bypassed)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int total = NH * p.nts;
    const int t = __hip_atomic_fetch_add(p.counters + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == total - 1) ? 1 : 0;
    if (last) {
      // (every load of the records below is an sc1 load, which bypasses this CU's L1: the invalidate is issued for good measure
      // and not waited for - guide, Guideline 16 "Valid forms": sc1 stores drained before the ticket, sc1 loads after it)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(p.counters + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // left zero for the next launch
    }
    *sFlag = last;
  }
  __syncthreads();
  T0_STAMP(3);
  if (*sFlag == 0) return;

  // ---- phase 2 (one workgroup per slide): the row chain of token 0.  All weight loads that do not depend on data go first.
  const float* W = p.img;
  const int f = tid & 127, kq = tid >> 7;                  // N = 128 stages: output f, k quarter kq
  // the split records of head kq, feature f go first (vmcnt retires in order: behind the weight stream they would wait for it)
  float pm[MAX_TS], pl[MAX_TS], pz[MAX_TS];
  {
    const float* hp = p.partials + ((int64_t)b * NH + kq) * p.nts * REC;
#pragma unroll
    for (int t = 0; t < MAX_TS; ++t) {
      const int tc = min(t, p.nts - 1);
      pm[t] = ld_agent(hp + tc * REC); pl[t] = ld_agent(hp + tc * REC + 1); pz[t] = ld_agent(hp + tc * REC + 4 + f);
    }
  }
  f32x4 wv[8], wo[8], w1[32];
#pragma unroll
  for (int i = 0; i < 8; ++i) wv[i] = ldg_f32x4(W + OFF_WV + ((8 * kq + i) * DM + f) * 4);
#pragma unroll
  for (int i = 0; i < 8; ++i) wo[i] = ldg_f32x4(W + OFF_WO + ((8 * kq + i) * DM + f) * 4);
#pragma unroll
  for (int i = 0; i < 32; ++i) w1[i] = ldg_f32x4(W + OFF_W1 + (i * DFF + tid) * 4);
  // merge the splits (flash-decoding style): zc[h][c] = sum z / sum l
  {
    float M = -1e30f;
#pragma unroll
    for (int t = 0; t < MAX_TS; ++t) if (t < p.nts) M = fmaxf(M, pm[t]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int t = 0; t < MAX_TS; ++t)
      if (t < p.nts) {
        const float w = __builtin_amdgcn_exp2f(pm[t] - M);
        num = fmaf(pz[t], w, num);
        den = fmaf(pl[t], w, den);
      }
    sZ[kq * DM + f] = num / den;
  }
  __syncthreads();
  T0_STAMP(4);
  // o = Wv z_{head of row} + bv
  {
    const float* zc = sZ + (f >> 5) * DM + 32 * kq;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += dot4(wv[i], *reinterpret_cast<const f32x4*>(zc + 4 * i));
    sRed[kq * DM + f] = acc;
  }
  __syncthreads();
  T0_STAMP(5);
  if (tid < DM) sO[tid] = ((sRed[tid] + sRed[DM + tid]) + (sRed[2 * DM + tid] + sRed[3 * DM + tid])) + p.bv[tid];
  __syncthreads();
  f32x4 w2[32];                                            // linear2: rows f, k quarter kq (128 k): first half issued while out_proj /
#pragma unroll                                             // norms run, second half once linear1's weights are dead (256 registers)
  for (int i = 0; i < 16; ++i) w2[i] = ldg_f32x4(W + OFF_W2 + ((32 * kq + i) * DM + f) * 4);
  // x = norm1(x0 + out_proj(o)) ; x = norm2(x + cab)
  {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += dot4(wo[i], *reinterpret_cast<const f32x4*>(sO + 32 * kq + 4 * i));
    sRed[kq * DM + f] = acc;
  }
  __syncthreads();
  if (tid < DM) sXa[tid] = sX0[tid] + (((sRed[tid] + sRed[DM + tid]) + (sRed[2 * DM + tid] + sRed[3 * DM + tid])) + p.bo[tid]);
  __syncthreads();
  block_layernorm(sXa, p.ln1g, p.ln1b, p.eps, tid);
  if (tid < DM) sXa[tid] += p.cab[tid];
  __syncthreads();
  block_layernorm(sXa, p.ln2g, p.ln2b, p.eps, tid);
  T0_STAMP(6);
  // h = relu(W1 x + b1): one hidden unit per thread
  {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += dot4(w1[i], *reinterpret_cast<const f32x4*>(sXa + 4 * i));
    sRed[tid] = fmaxf(acc + p.b1[tid], 0.f);               // h[512] (sRed row 0..)
  }
#pragma unroll
  for (int i = 16; i < 32; ++i) w2[i] = ldg_f32x4(W + OFF_W2 + ((32 * kq + i) * DM + f) * 4);
  __syncthreads();
  T0_STAMP(7);
  // y = W2 h + b2
  {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += dot4(w2[i], *reinterpret_cast<const f32x4*>(sRed + 128 * kq + 4 * i));
    sZ[kq * DM + f] = acc;                                 // (zc is dead)
  }
  __syncthreads();
  if (tid < DM) sXa[tid] = sXa[tid] + (((sZ[tid] + sZ[DM + tid]) + (sZ[2 * DM + tid] + sZ[3 * DM + tid])) + p.b2[tid]);
  __syncthreads();
  T0_STAMP(8);
  block_layernorm(sXa, p.ln3g, p.ln3b, p.eps, tid);
  // ---- decoder.norm, slide-context residual, classifier
  block_layernorm(sXa, p.lnfg, p.lnfb, p.eps_f, tid);
  if (tid < DM) {
    float v = sXa[tid];
    if (p.ctx_prev) v += p.ctx_prev[(int64_t)b * p.ctx_stride + tid];
    sXa[tid] = v;
    p.ctx_out[(int64_t)b * DM + tid] = v;
  }
  __syncthreads();
  for (int j = wave; j < p.num_logits; j += NT / 64) {
    const float* w = p.wcls + (int64_t)j * p.cls_in;
    float acc = 0.f;
    if (p.ctx_all) {
      for (int i = lane; i < p.ctx_depth * DM; i += 64) acc += w[i] * p.ctx_all[(int64_t)b * p.ctx_depth * DM + i];
      w += p.ctx_depth * DM;
    }
    acc += w[lane] * sXa[lane] + w[lane + 64] * sXa[lane + 64];
    acc = wave_sum64(acc);
    if (lane == 0) p.logits[(int64_t)b * p.num_logits + j] = acc + p.bcls[j];
  }
  T0_STAMP(9);
}

// ---- packing: A_h = c Wk_h^T Wq_h (fp32 FMA chains over the 32 head dims), a0_h = c Wk_h^T bq_h, and the T4 transposes
// out[(k4 * N + n) * 4 + e] = W[n][4 k4 + e]
__global__ void __launch_bounds__(256)
token0_pack_kernel(const float* __restrict__ wqkv, const float* __restrict__ bqkv, const float* __restrict__ wo,
                   const float* __restrict__ w1, const float* __restrict__ w2, float qscale, float* __restrict__ out) {
  const int job = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (job == 0) {                    // A: [h][k4][c][e] <- c * sum_i Wk[32h+i][c] Wq[32h+i][4 k4 + e];  i in [0, 4*32*128*4)
    if (i >= NH * DM * DM) return;
    const int e = i & 3, c = (i >> 2) & 127, k4 = (i >> 9) & 31, h = i >> 14;
    const float* wq = wqkv + (int64_t)(HD * h) * DM + 4 * k4 + e;
    const float* wk = wqkv + (int64_t)(DM + HD * h) * DM + c;
    float acc = 0.f;
    for (int r = 0; r < HD; ++r) acc = fmaf(wk[r * DM], wq[r * DM], acc);
    out[OFF_A + i] = acc * qscale;
  } else if (job == 1) {             // a0: [h][c]
    if (i >= NH * DM) return;
    const int c = i & 127, h = i >> 7;
    const float* wk = wqkv + (int64_t)(DM + HD * h) * DM + c;
    float acc = 0.f;
    for (int r = 0; r < HD; ++r) acc = fmaf(wk[r * DM], bqkv[HD * h + r], acc);
    out[OFF_A0 + i] = acc * qscale;
  } else {                           // T4 transposes: Wv (rows 2 DM.. of wqkv), Wo, W1, W2
    const float* src; int N, K, off;
    if (job == 2) { src = wqkv + 2 * DM * DM; N = DM; K = DM; off = OFF_WV; }
    else if (job == 3) { src = wo; N = DM; K = DM; off = OFF_WO; }
    else if (job == 4) { src = w1; N = DFF; K = DM; off = OFF_W1; }
    else { src = w2; N = DM; K = DFF; off = OFF_W2; }
    if (i >= N * K) return;
    const int e = i & 3, n = (i >> 2) % N, k4 = (i >> 2) / N;
    out[off + i] = src[(int64_t)n * K + 4 * k4 + e];
  }
}

}  // namespace

#ifdef PATHS_T0_STAMPS
static unsigned long long* g_t0_stamps = nullptr;
extern "C" void paths_t0_stamp_buffer(unsigned long long* p) { g_t0_stamps = p; }     // development hook (tools/t0_time.py)
#endif

extern "C" {

int64_t paths_token0_ws_image_bytes(void) { return (int64_t)IMG_FLOATS * 4; }

// floats of the partials scratch of paths_token0_tail_ws
int64_t paths_token0_ws_partials(int B, int T) {
  int nts = (T + TS_TOKENS - 1) / TS_TOKENS;
  nts = nts < 1 ? 1 : nts > MAX_TS ? MAX_TS : nts;
  return (int64_t)B * NH * nts * REC;
}

// Weight image of paths_token0_tail_ws for one (last) decoder layer: wqkv [384,128], bqkv [384], wo [128,128], w1 [512,128],
// w2 [128,512]; qscale = log2(e) / sqrt(head_dim).  Rebuilt whenever the weights change.
int paths_token0_pack_ws(const float* wqkv, const float* bqkv, const float* wo, const float* w1, const float* w2, float qscale, void* out,
                         hipStream_t stream) {
  PATHS_REQUIRE(wqkv && bqkv && wo && w1 && w2 && out && (uintptr_t)out % 16 == 0, "token0_pack_ws: bad arguments");
  hipLaunchKernelGGL(token0_pack_kernel, dim3(256, 6), dim3(256), 0, stream, wqkv, bqkv, wo, w1, w2, qscale, reinterpret_cast<float*>(out));
  PATHS_LAUNCH_CHECK("token0_pack_ws");
  return PATHS_OK;
}

// The last decoder layer at token 0 (reference model/aggregator.py:70-75) + decoder.norm + slide-context residual / concat +
// classifier (model/paths.py:130-139) from the layer's INPUT rows x1 [B,T,128]: no K / V projection, one launch.
// img: paths_token0_pack_ws image; bv = in_proj_bias + 256; partials: paths_token0_ws_partials(B, T) floats of scratch;
// counters: B int32 words that are ZERO on entry (they are left zero: the last arriver of a slide resets its word).
int paths_token0_tail_ws(const float* x1, const int64_t* num_ims, const void* img, const float* bv, const float* bo,
                         const float* ln1g, const float* ln1b, const float* cab, const float* ln2g, const float* ln2b,
                         const float* b1, const float* b2, const float* ln3g, const float* ln3b, const float* lnfg, const float* lnfb,
                         const float* ctx_prev, int64_t ctx_stride, const float* ctx_all, int ctx_depth,
                         const float* wcls, const float* bcls, int num_logits, int cls_in,
                         float* ctx_out, float* logits, float* partials, int* counters, int B, int T, int d, int H,
                         float eps, float eps_final, hipStream_t stream) {
  PATHS_REQUIRE(d == DM && H == NH, "token0_tail_ws: this build supports trans_dim=128, 4 heads (got %d, %d)", d, H);
  PATHS_REQUIRE(B > 0 && T > 0 && x1 && num_ims && img && bv && bo && ln1g && ln1b && cab && ln2g && ln2b && b1 && b2 && ln3g && ln3b && lnfg && lnfb,
                "token0_tail_ws: null operand");
  PATHS_REQUIRE(wcls && bcls && ctx_out && logits && partials && counters, "token0_tail_ws: null output / scratch");
  PATHS_REQUIRE(num_logits > 0 && cls_in == (ctx_all ? (ctx_depth + 1) * DM : DM), "token0_tail_ws: bad classifier shape");
  PATHS_REQUIRE(((uintptr_t)x1 | (uintptr_t)img) % 16 == 0, "token0_tail_ws: buffers must be 16-byte aligned");
  int nts = (T + TS_TOKENS - 1) / TS_TOKENS;
  nts = nts < 1 ? 1 : nts > MAX_TS ? MAX_TS : nts;
  T0Params p{x1, num_ims, reinterpret_cast<const float*>(img), bv, bo, ln1g, ln1b, cab, ln2g, ln2b, b1, b2, ln3g, ln3b, lnfg, lnfb,
             ctx_prev, ctx_stride, ctx_all, ctx_depth, wcls, bcls, num_logits, cls_in, ctx_out, logits, partials, counters, T, nts, eps, eps_final
#ifdef PATHS_T0_STAMPS
             , g_t0_stamps
#endif
  };
  hipLaunchKernelGGL(token0_ws_kernel, dim3(NH * nts, B), dim3(NT), 0, stream, p);
  PATHS_LAUNCH_CHECK("token0_tail_ws");
  return PATHS_OK;
}

}  // extern "C"
