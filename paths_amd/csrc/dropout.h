// Counter-based dropout masks (training only; reference: nn.Transformer(..., dropout=p), model/aggregator.py:25-33 - five sites per
// decoder layer: attention probabilities, dropout1 after out_proj, dropout2 on the (degenerate) cross-attention output, the
// feed-forward's inner dropout and dropout3).
//
// A mask is never stored: element `idx` of site (key_lo, key_hi) is kept iff hash(idx, key) >= p * 2^32, and every kernel that
// needs the mask (forward, the backward's recompute, the gradient masking) regenerates it from the same (key, idx).  The hash is
// two rounds of the murmur3 finaliser over a 64-bit element index and a 64-bit site key; the host derives one key per
// (step seed, level, layer, site), so sites, layers, levels and steps draw independent masks.
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
__host__ __device__ __forceinline__ uint32_t drop_hash(uint64_t idx, uint32_t key_lo, uint32_t key_hi) {
  const uint32_t a = drop_fmix32((uint32_t)idx * 0x9E3779B1u + key_lo);
  return drop_fmix32(a ^ drop_fmix32((uint32_t)(idx >> 32) * 0x85EBCA77u + key_hi));
}
// multiplier of element idx: 0 (dropped) or 1 / (1 - p)
__host__ __device__ __forceinline__ float drop_mult(const DropSite& s, uint64_t idx) {
  return drop_hash(idx, s.key_lo, s.key_hi) >= s.thr ? s.scale : 0.f;
}
