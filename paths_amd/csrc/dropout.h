// Counter-based dropout masks (training only; reference: nn.Transformer(..., dropout=p), model/aggregator.py:25-33 - five sites per
// decoder layer: attention probabilities, dropout1 after out_proj, dropout2 on the (degenerate) cross-attention output, the
// feed-forward's inner dropout and dropout3).
//
// A mask is never stored: every kernel that needs it (forward, the backward's recompute, the gradient masking) regenerates it from
// the same (key, idx).  One 32-bit hash serves a PAIR of elements (round 4: the hash was ~half of the vector work of the attention
// backward kernels): element idx of site (key_lo, key_hi) is kept iff
//     half(idx & 1) of hash(idx & ~1) >= thr16,     thr16 = round(p * 65536)      (the low half for even idx, the high half for odd)
//     hash(i) = fmix32( ((uint32) i * 0x9E3779B1 + key_lo)  ^  fmix32( (uint32)(i >> 32) * 0x85EBCA77 + key_hi ) )
// (murmur3's 32-bit finaliser: a bijection with full avalanche; its two 16-bit halves are independent draws).  The drop probability
// actually applied is p16 = thr16 / 65536 (0.05 -> 3277 / 65536 = 0.0500031) and the kept elements are scaled by 1 / (1 - p16), so the
// mask stays unbiased.  The inner term depends on the HIGH word of the index only: the attention kernels, which hash T x T' elements
// per (slide, head), compute it once per workgroup for the two high words their indices can take (DropWin) and pay one finaliser per
// PAIR of probabilities.  Attention rows are T' = T rounded up to even elements apart (drop_attn_stride), so that a key pair
// (2j, 2j + 1) of any query shares a hash - the lanes of the attention kernels own runs of four consecutive keys.
// The host derives one key per (step seed, level, layer, site), so sites, layers, levels and steps draw independent masks.
#pragma once
#include <stdint.h>

struct DropSite {
  uint32_t key_lo, key_hi;
  uint32_t thr;        // keep iff the element's 16-bit half >= thr;  thr = round(p * 65536), 0 = dropout off
  float scale;         // 1 / (1 - thr / 65536)
};

__host__ __device__ __forceinline__ uint32_t drop_fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__host__ __device__ __forceinline__ uint32_t drop_hterm(uint32_t hi, uint32_t key_hi) { return drop_fmix32(hi * 0x85EBCA77u + key_hi); }
__host__ __device__ __forceinline__ uint32_t drop_hash(uint64_t idx, uint32_t key_lo, uint32_t key_hi) {
  return drop_fmix32(((uint32_t)idx * 0x9E3779B1u + key_lo) ^ drop_hterm((uint32_t)(idx >> 32), key_hi));
}
__host__ __device__ __forceinline__ uint32_t drop_half(uint32_t h, uint64_t idx) { return (idx & 1u) ? h >> 16 : h & 0xFFFFu; }
// multiplier of element idx: 0 (dropped) or 1 / (1 - p16)
__host__ __device__ __forceinline__ float drop_mult(const DropSite& s, uint64_t idx) {
  return drop_half(drop_hash(idx & ~1ull, s.key_lo, s.key_hi), idx) >= s.thr ? s.scale : 0.f;
}
// the multipliers of elements idx (even) and idx + 1 from their one hash
__host__ __device__ __forceinline__ void drop_mult2(const DropSite& s, uint64_t idx_even, float& m0, float& m1) {
  const uint32_t h = drop_hash(idx_even, s.key_lo, s.key_hi);
  m0 = (h & 0xFFFFu) >= s.thr ? s.scale : 0.f;
  m1 = (h >> 16) >= s.thr ? s.scale : 0.f;
}

// Attention probabilities of (slide, head) pair `pair`: element (query q, key k) has index ((pair * T + q) * stride + k), stride = T
// rounded up to even.
__host__ __device__ __forceinline__ uint64_t drop_attn_stride(int T) { return (uint64_t)((T + 1) & ~1); }
__host__ __device__ __forceinline__ uint64_t drop_attn_row(uint64_t pair, int T, int q) { return (pair * (uint64_t)T + (uint64_t)q) * drop_attn_stride(T); }

// The same multiplier for indices inside a window [idx_min, idx_min + 2^32): the hashed high-word terms of the two high words the
// window covers are computed once (a (slide, head) pair's T x T' attention probabilities: T < 65536); an index outside the window
// still gets the right value (the rare branch recomputes its term).
struct DropWin { uint32_t hi, h0, h1; };
__host__ __device__ __forceinline__ DropWin drop_window(const DropSite& s, uint64_t idx_min) {
  const uint32_t hi = (uint32_t)(idx_min >> 32);
  return DropWin{hi, drop_hterm(hi, s.key_hi), drop_hterm(hi + 1u, s.key_hi)};
}
__host__ __device__ __forceinline__ uint32_t drop_hash_w(const DropSite& s, const DropWin& w, uint64_t idx_even) {
  const uint32_t d = (uint32_t)(idx_even >> 32) - w.hi;
  uint32_t ht = d == 0u ? w.h0 : w.h1;
  if (__builtin_expect(d > 1u, 0)) ht = drop_hterm((uint32_t)(idx_even >> 32), s.key_hi);
  return drop_fmix32(((uint32_t)idx_even * 0x9E3779B1u + s.key_lo) ^ ht);
}
__host__ __device__ __forceinline__ float drop_mult_w(const DropSite& s, const DropWin& w, uint64_t idx) {
  return drop_half(drop_hash_w(s, w, idx & ~1ull), idx) >= s.thr ? s.scale : 0.f;
}
__host__ __device__ __forceinline__ void drop_mult2_w(const DropSite& s, const DropWin& w, uint64_t idx_even, float& m0, float& m1) {
  const uint32_t h = drop_hash_w(s, w, idx_even);
  m0 = (h & 0xFFFFu) >= s.thr ? s.scale : 0.f;
  m1 = (h >> 16) >= s.thr ? s.scale : 0.f;
}
