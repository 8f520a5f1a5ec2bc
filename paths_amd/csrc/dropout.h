// Counter-based dropout masks (training only; reference: nn.Transformer(..., dropout=p), model/aggregator.py:25-33 - five sites per
// decoder layer: attention probabilities, dropout1 after out_proj, dropout2 on the (degenerate) cross-attention output, the
// feed-forward's inner dropout and dropout3).
//
// A mask is never stored: element `idx` of site (key_lo, key_hi) is kept iff hash(idx, key) >= p * 2^32, and every kernel that
// needs the mask (forward, the backward's recompute, the gradient masking) regenerates it from the same (key, idx).  The hash:
//     hash(idx) = fmix32( ((uint32) idx * 0x9E3779B1 + key_lo)  ^  fmix32( (uint32)(idx >> 32) * 0x85EBCA77 + key_hi ) )
// (murmur3's 32-bit finaliser: a bijection with full avalanche, so for a fixed high word consecutive indices walk a permutation of
// the 32-bit values).  The inner term depends on the HIGH word of the index only: the attention kernels, which hash T^2 elements per
// (slide, head), compute it once per workgroup for the two high words their indices can take (DropWin: T^2 < 2^32) and pay one
// finaliser per element instead of three (round 2: 28 of ~40 VALU operations per probability were the hash).  The host derives
// one key per (step seed, level, layer, site), so sites, layers, levels and steps draw independent masks.
#pragma once
#include <stdint.h>

struct DropSite {
  uint32_t key_lo, key_hi;
  uint32_t thr;        // keep iff hash >= thr;  thr = round(p * 2^32), 0 = dropout off
  float scale;         // 1 / (1 - p)
};

__host__ __device__ __forceinline__ uint32_t drop_fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__host__ __device__ __forceinline__ uint32_t drop_hterm(uint32_t hi, uint32_t key_hi) { return drop_fmix32(hi * 0x85EBCA77u + key_hi); }
__host__ __device__ __forceinline__ uint32_t drop_hash(uint64_t idx, uint32_t key_lo, uint32_t key_hi) {
  return drop_fmix32(((uint32_t)idx * 0x9E3779B1u + key_lo) ^ drop_hterm((uint32_t)(idx >> 32), key_hi));
}
// multiplier of element idx: 0 (dropped) or 1 / (1 - p)
__host__ __device__ __forceinline__ float drop_mult(const DropSite& s, uint64_t idx) {
  return drop_hash(idx, s.key_lo, s.key_hi) >= s.thr ? s.scale : 0.f;
}

// The same multiplier for indices inside a window [idx_min, idx_min + 2^32): the hashed high-word terms of the two high words the
// window covers are computed once (a (slide, head) pair's T^2 attention probabilities: T < 65536); an index outside the window
// still gets the right value (the rare branch recomputes its term).
struct DropWin { uint32_t hi, h0, h1; };
__host__ __device__ __forceinline__ DropWin drop_window(const DropSite& s, uint64_t idx_min) {
  const uint32_t hi = (uint32_t)(idx_min >> 32);
  return DropWin{hi, drop_hterm(hi, s.key_hi), drop_hterm(hi + 1u, s.key_hi)};
}
__host__ __device__ __forceinline__ float drop_mult_w(const DropSite& s, const DropWin& w, uint64_t idx) {
  const uint32_t d = (uint32_t)(idx >> 32) - w.hi;
  uint32_t ht = d == 0u ? w.h0 : w.h1;
  if (__builtin_expect(d > 1u, 0)) ht = drop_hterm((uint32_t)(idx >> 32), s.key_hi);
  return drop_fmix32(((uint32_t)idx * 0x9E3779B1u + s.key_lo) ^ ht) >= s.thr ? s.scale : 0.f;
}
