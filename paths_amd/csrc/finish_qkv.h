// TOKEN ORDER: patch i of slide b is token i, the special token sits at index num_ims[b] (behind the valid patches; the reference
// prepends it, model/aggregator.py:62-64 - masked self-attention is invariant under that permutation and only the special token's
// output row is read).  A 64-row block of the GEMM result (rows b * N + 64 j ..) is then exactly token tile j of slide b; the extra
// tile j = N / 64 exists for the one slot N the special token takes when a slide is full.
//
// Hand-over between the split-K importance / projection GEMM (gemm_x6.hip) and its fused finish (tlayer_ws.hip:
// tlayer_ws_kernel<128, false, true, false, FIN = true>): the finish workgroup of a 64-token tile sums the two k-half slabs, runs the
// importance MLP's second layer + sigmoid + mask, builds the tokens (alpha * proj + bias + positional encoding, special token in slot 0)
// AND multiplies them by the first decoder layer's in_proj, writing the attention kernel's q | k | v operand images - the launch that
// used to re-read the tokens for that product (paths_token_layer_ws with do_qkv only) is gone.
// Reference: model/paths.py:95-98,119-124; model/aggregator.py:37-65; nn.TransformerDecoderLayer self_attn in_proj (aggregator.py:70-72).
#pragma once
#include "common.h"

struct FinQkvParams {
  // ---- the raw GEMM result: ws [nz][M_pad / 32][8][1024] floats in the accumulator layout (EpiRaw), rows = patches (M = B * N)
  const float* ws; int64_t zstride; int nz;
  // ---- importance MLP tail + token assembly (EpiImpProj's operands)
  const float* b1; const float* w2; const float* b2;
  const float* bp; const float* special;
  const float* pe_table; int pe_rows;          // paths_pe_table output (required: an inline sinf / cosf path cost 6.8k instructions per lane)
  const int64_t* locs;                         // [B * N, 2] pixel coordinates (2-D mode)
  const int64_t* num_ims;                      // [B]
  int N, T, Tp, B;                             // patches per slide (capacity), tokens = N + 1, T rounded up to 64
  int patch_size, pe_mode, imp_mul, skip_padding;
  float acc_scale;                             // 1 / (w_scale * a_scale) of the GEMM
  int alpha_from_importance;                   // 1: alpha is READ from `importance` (an importance-only finish already ran), 0: computed and written
  float* importance;                           // [B * N]
  float* tokens;                               // [B, T, 128]
  // ---- first decoder layer's in_proj (paths_tlayer_pack_ws part 1 image, scale s_wqkv) -> operand images of paths_attention_h3_img
  const void* w_qkv; const float* bqkv; float inv_wqkv; float qscale;
  void* qkv_img;
  int xcd_order;                               // finish workgroups in the XCD order of the GEMM that wrote the slabs (tlayer_ws.hip: fin_tile_of_workgroup)
};

// top-K operands of the fused importance + top-K finish (phase 8; paths_topk_rows' outputs)
struct FinTopkParams {
  int keep;                        // < 0: keep all, original order
  int* keep_idx; int64_t ldk;      // [B, ldk] kept indices by (score descending, index ascending)
  int* keep_count;                 // [B]
  const float* row_base; int64_t row_ld;   // optional: kept_rows[b, i] = address of row_base[b, keep_idx[b, i], :] (rows of N per slide)
  int64_t* kept_rows; const float* zero_row;
  int* counters;                   // [2 B] int32: arrival / departure counts per slide, zero on entry, left zero
  int* status;                     // optional: bit 4 set if the bounded arrival wait gave up
};

// defined in tlayer_ws.hip; launches on `stream` (stop-event capable: PATHS_LAUNCH_STOP)
int paths_launch_finish_qkv(const FinQkvParams& p, hipStream_t stream);
int paths_launch_finish_importance(const FinQkvParams& p, hipStream_t stream);
int paths_launch_finish_importance_topk(const FinQkvParams& p, const FinTopkParams& k, hipStream_t stream);
