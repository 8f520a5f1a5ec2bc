/* libpaths_hip.so — C ABI of the MI355X (gfx950) PATHS hot path.
 *
 * The reference (zzbuzzard/PATHS) is pure Python/PyTorch and has NO FFI of its own; this header is the
 * thin boundary the build defines underneath the reference's Python surface.  Each entry
 * point names the reference code it replaces (file:line relative to the reference repo root).
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is a DEVICE pointer unless stated; no torch types;
 *   - returns 0 on success, a negative code on error (-1 invalid argument, -2 launch failure,
 *     -3 unsupported configuration); paths_last_error() returns a thread-local message;
 *   - never allocates, frees or synchronises; launches on the caller's stream (hipStream_t passed as
 *     void*); no global mutable state, callable from any host thread;
 *   - all floating-point buffers are fp32, row-major, 16-byte aligned, leading dimensions in elements
 *     and multiples of 4; integer fields follow the reference's dtypes (int64 locs / num_ims /
 *     parent_inds);
 *   - arithmetic: fp32 inputs, outputs and accumulation everywhere.  The *_x6 / *_h3 entry points multiply SPLIT operands on
 *     the 16-bit matrix cores: planes = 2 -> x = hi + lo as two fp16 planes (22 significant bits; 3 x
 *     v_mfma_f32_32x32x16_f16 per product block; operands pre-scaled by powers of two, w_scale / a_scale arguments, and
 *     |activation| * a_scale must stay below 65504: the Python host checks max|x| of every resident grid / input batch
 *     (paths_tissue_mask_absmax) and runs out-of-range data on planes = 3); planes = 3 -> x = hi + mid + lo EXACTLY as three
 *     bf16 planes (6 x v_mfma_f32_32x32x16_bf16, no range limits).  The entry points without suffix use the f32-input
 *     matrix cores (v_mfma_f32_32x32x2_f32 / 16x16x4_f32: a k-ordered fp32 FMA chain).  Measured error of all three vs
 *     fp64 is that of an fp32 FMA chain.
 */
#ifndef PATHS_HIP_H
#define PATHS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* paths_stream_t; /* hipStream_t */

const char* paths_last_error(void);
const char* paths_build_info(void);
/* 2 since round 5.  Differences a consumer built against ABI 1 must know: (a) dropout masks (paths_dropout_mask and every *_dropout /
 * drop_key entry) are one 16-bit hash half per element - thr16 = round(p * 65536) clamped to [1, 65535] for p > 0, kept elements scaled by
 * 1 / (1 - thr16 / 65536), attention mask rows T rounded up to even elements apart - so the same (key, p) yields other masks than ABI 1;
 * p >= 1 is rejected by every entry point (p in (1 - 2^-16, 1) is applied as 65535 / 65536); (b) events of paths_event_create order
 * streams of ONE device (created with hipEventDisableSystemFence): never wait for them on the host. */
int paths_abi_version(void);

/* Stream plumbing of the launch tape (paths_amd/utils.py:TapedRecursion replays a recorded recursion as a flat list of C calls;
 * no reference equivalent).  paths_event_create / paths_event_destroy: a timing-less event handle (host object) for paths_stream_wait, which makes `dst`
 * wait for everything enqueued on `src` so far; paths_memset_zero: hipMemsetAsync(0).  None of them synchronises the host. */
void* paths_event_create(void);
/* a HIP stream restricted to the compute units set in cu_mask (words x 32 bits): measurement aid (CU-partitioned streams) */
void* paths_stream_create_masked(const uint32_t* cu_mask, int words);
int paths_event_destroy(void* event);
int paths_stream_wait(paths_stream_t dst, paths_stream_t src, void* event);
/* Stop events: paths_set_stop_event(ev) makes the next stop-capable launch of this host thread (the importance / projection finish
 * kernel of paths_importance_proj_x6, the kernel of paths_topk / paths_topk_rows) carry ev as its completion event - no event-record
 * packet of its own in the launching queue; paths_flush_stop_event(src) records it on src the ordinary way if no launch took it;
 * paths_stream_wait_event(dst, ev): dst waits for ev.  paths_stop_event_pending(): 1 while the armed event has not been taken;
 * paths_clear_stop_event(): disarm without recording (error paths); paths_record_event(ev, s): plain hipEventRecord - the caller
 * re-records ev behind launches that followed the stop-capable kernel, so the join covers them too (paths_amd/_lib.py:fork_behind). */
int paths_set_stop_event(void* event);
int paths_flush_stop_event(paths_stream_t src);
int paths_stop_event_pending(void);
int paths_clear_stop_event(void);
int paths_record_event(void* event, paths_stream_t stream);
int paths_stream_wait_event(paths_stream_t dst, void* event);
int paths_memset_zero(void* p, size_t bytes, paths_stream_t stream);

/* LSTMCell.forward over depth + residual (reference model/interface.py:31-58, model/paths.py:78-91).
 *   x [M,D] (ldx), h0/c0 = previous state views (both NULL at depth 0), M = B * rows_per_slide.
 *   w_gates [3Hc+D, 2D] PACKED: rows [0,3Hc) in groups of 96 = forget|remember|map rows of one block of
 *   32 memory units, rows [3Hc, 3Hc+D) = out_select_gate; b_gates packed alike; w_mem [D,Hc], b_mem [D].
 *   state_out [M, D+Hc] <- (h1 | c1)  (= "ctx_patch"), y [M,D] <- x + h1, ws_o [M,D] scratch.
 *   num_ims != NULL: tiles that contain only padding rows (row index within slide >= num_ims[b]) are skipped.
 *   phases: bit0 memory-cell GEMM, bit1 output-gate GEMM, bit2 mem_to_out GEMM; pass 7 (all, in this order).
 *   save_frm [M,3Hc] / save_tc [M,D] (both optional): gate activations f|r|m (packed order) and tanh(Wc c1 + bc),
 *   kept for the backward pass; ws_o then holds the output gate o.
 *   hp [rows, 3Hc+D] + hp_row [M] (optional, instead of h0): siblings share their parent's h, so h_parent Wh^T is computed
 *   ONCE per kept parent (paths_gather_kept_rows + paths_gemm_nt_f32 on w_gates[:, D:2D]) and added in the epilogue
 *   through hp_row (-1 = no parent: zero state); the gate GEMM then only runs over the x panel (K = D). */
int paths_lstm_cell(const float* x, int64_t ldx, const float* h0, int64_t ldh0, const float* c0, int64_t ldc0,
                    const float* w_gates, const float* b_gates, const float* w_mem, const float* b_mem,
                    float* state_out, int64_t ldso, float* y, int64_t ldy, float* ws_o, float* save_frm, float* save_tc,
                    const float* hp, const int* hp_row,
                    int M, int D, int Hc, const int64_t* num_ims, int rows_per_slide, int phases, paths_stream_t stream);

/* importance MLP + sigmoid + padding mask, importance scaling, proj_in, positional encoding, special token
 * (reference model/paths.py:95-98,119-124; utils.py:16-23,47-67,106-115; model/aggregator.py:37-65).
 *   w_ip [256, D] = importance_mlp.0.weight (W1) and proj_in.weight (Wp) interleaved in blocks of 64 rows:
 *   [W1[0:64] ; Wp[0:64] ; W1[64:128] ; Wp[64:128]]; b1 / w2 / bp in their natural order; tokens [B, N+1, d]: row 0 = special token.
 *   pe_mode 2 = "2d" (div_term has d/4 entries, locs [M,2] int64 pixel coords), 1 = "1d" (d/2 entries).
 *   pe_table (optional) = paths_pe_table output with pe_rows rows: sin/cos values are then read from it (bit-identical to
 *   evaluating them).  The caller guarantees 0 <= locs / patch_size < pe_rows (paths_amd passes the level's grid size);
 *   positions are clamped into the table for memory safety only. */
int paths_pe_table(const float* div_term, int pe_mode, int d, int rows, float* out, paths_stream_t stream);
int paths_importance_proj(const float* y, int64_t ldy, const float* w_ip, const float* b1, const float* w2, const float* b2 /* device scalar */,
                          const float* bp, const float* special, const float* div_term, const float* pe_table, int pe_rows,
                          const int64_t* locs,
                          const int64_t* num_ims, int rows_per_slide, int patch_size, int pe_mode, int imp_mul,
                          float* importance, float* tokens, float* save_hid, float* save_pproj, int M, int D, int Hi, int d,
                          int skip_padding, paths_stream_t stream);

/* ---- split-operand variants of the three GEMM entry points above: same arithmetic contract (fp32 in, fp32 out, fp32
 * accumulation, error of the order of an fp32 FMA chain's), computed on the 16-bit matrix cores (csrc/gemm_x6.hip).
 *   planes = 3 ("x6"): every fp32 operand is the exact sum of three bf16 (hi+mid+lo); six of the nine partial products are
 *                      kept (dropped: <= 2^-25 |a w|); 6 bf16 MFMAs per product block.  No range restrictions; scales must be 1.
 *   planes = 2 ("h3"): operands are split into two fp16 (hi+lo, 22 significant bits); hi*hi + hi*lo + lo*hi are kept (dropped:
 *                      <= 2^-22 |a w|); 3 fp16 MFMAs per product block.  fp16's exponent range is narrow, so operands are
 *                      pre-scaled by powers of two: weights by w_scale when packed (max|w| w_scale < 65504), activations by
 *                      a_scale inside the kernel (|activation| a_scale < 65504; below 0.125 / a_scale the lo plane is
 *                      subnormal: absolute error 2^-25 / a_scale per element).  Accumulators are un-scaled in the epilogues.
 *   planes = 4        (paths_gemm_nt_x6 / paths_x6_pack_weights only): TWO bf16 planes (hi + mid of the exact split: 16
 *                      significant bits per operand, fp32's exponent range: no scales, nothing overflows); hi*hi + hi*mid + mid*hi,
 *                      3 bf16 MFMAs per block, relative error of a product ~2e-5.  The gradient GEMMs of the training step
 *                      (round 3; paths_gemm_tn_x6 has the same form as its planes = 2).
 * Activations stay fp32 in HBM; WEIGHTS are passed as the image produced once by paths_x6_pack_weights:
 *   [Npad/32][K/16][plane][k-half][32 rows][8 x 16 bit]   (paths_x6_packed_bytes(Npad, K, planes) = 2 planes Npad K bytes).
 * They replace the same reference code as their f32 twins (model/interface.py:31-58, model/paths.py:78-98,119-124). */
int64_t paths_x6_packed_bytes(int Npad, int K, int planes);
int paths_x6_pack_weights(const float* w, int64_t ldw, void* out, int N, int Npad, int K, int planes, float w_scale,
                          paths_stream_t stream);
/* The image of W^T made from W: w is [Kvalid, N] fp32 (row stride ldw); the packed weight is [N, K] (rows n >= N and columns k >= Kvalid zero);
 * planes 3 or 4 (bf16 planes, scale 1).  One launch instead of paths_transpose_f32 + paths_x6_pack_weights (the backward's dX products). */
int paths_x6_pack_weights_t(const float* w, int64_t ldw, void* out, int N, int Npad, int K, int Kvalid, int planes, paths_stream_t stream);
/* w_gates_x6 = pack of the PACKED gate matrix [3Hc+D, 2D] of paths_lstm_cell (scale wg_scale); w_mem_x6 = pack of [D, Hc]
 * (scale wm_scale); D % 256 == 0; y may be NULL (Y = X + h1 not materialised).  x_rows (optional, planes = 2, needs hp / no
 * h0, y = NULL): [M] addresses of the feature rows - x is then read in place (paths_gather_rows row_ptrs), x / ldx unused.
 * ws_o: scratch of ceil(M / 256) * 256 x D floats (with y = NULL and no training saves the output gate crosses from phase 2 to
 * phase 4 as raw pre-activations in whole 32x32 accumulator tiles; otherwise as the [M, D] gate values) */
int paths_lstm_cell_x6(const float* x, int64_t ldx, const int64_t* x_rows, const float* h0, int64_t ldh0, const float* c0, int64_t ldc0,
                       const void* w_gates_x6, const float* b_gates, const void* w_mem_x6, const float* b_mem,
                       float* state_out, int64_t ldso, float* y, int64_t ldy, float* ws_o, float* save_frm, float* save_tc,
                       const float* hp, const int* hp_row,
                       int M, int D, int Hc, const int64_t* num_ims, int rows_per_slide, int phases,
                       int planes, float wg_scale, float wm_scale, float a_scale, paths_stream_t stream);
/* w_ip_x6 = pack of [256, D] (rows interleaved as for paths_importance_proj); y_add (optional): GEMM input = y + y_add summed
 * in fp32 while staging, so that the caller can pass (x, h1) and skip materialising Y = X + h1; y_rows (optional, planes = 2):
 * row addresses of y instead of (y, ldy) */
int paths_importance_proj_x6(const float* y, int64_t ldy, const int64_t* y_rows, const float* y_add, int64_t ldya, const void* w_ip_x6, const float* b1, const float* w2, const float* b2 /* device scalar */,
                             const float* bp, const float* special, const float* div_term, const float* pe_table, int pe_rows,
                             const int64_t* locs,
                             const int64_t* num_ims, int rows_per_slide, int patch_size, int pe_mode, int imp_mul,
                             float* importance, float* tokens, float* save_hid, float* save_pproj, int M, int D, int Hi, int d,
                             int skip_padding, int planes, float w_scale, float a_scale, float* splitk_ws, paths_stream_t stream);
/* splitk_ws (optional, paths_importance_proj_x6_workspace(M) bytes; used with planes = 2, y_add given, no training saves): the
 * [M/128 x 1]-block GEMM covers half the chip, so its k loop runs as two halves on twice the blocks (raw accumulators to the
 * workspace) and a second launch sums them and applies the epilogue on 64-row blocks.  null = one launch. */
int64_t paths_importance_proj_x6_workspace(int M);
/* Round 5: paths_importance_proj_x6 (split-K form) AND the first decoder layer's in_proj (paths_token_layer_ws with do_qkv only) as one
 * GEMM + one fused finish, for trans_dim 128 / 4 heads / importance hidden 128, two fp16 planes, N % 64 == 0 (reference
 * model/paths.py:95-98,119-124; model/aggregator.py:37-65 and the self_attn in_proj of decoder layer 0, model/aggregator.py:70-72).
 * TOKEN ORDER of this form: patch i of slide b is token i and the special token sits at index num_ims[b], right behind the valid
 * patches (the reference prepends it; masked self-attention is invariant under that permutation and only the special token's output
 * row is read) - a 64-row block of the GEMM result is then one 64-token tile of the attention's operand images.  Consumers:
 * paths_attention_h3_img, paths_token_layer_ws (position-agnostic) and paths_token0_tail_ws with special_last = 1.
 * phases: bit 1 = the split-K GEMM of (y | y_rows) + y_add into splitk_ws (paths_importance_proj_x6_workspace(B * N) bytes); bit 2 =
 * importance-only finish (alpha -> importance [B, N]); bit 8 (instead of 2) = that finish AND the top-K of every slide in ONE launch
 * (keep_idx / keep_count / kept_rows as paths_topk_rows writes them; a workgroup publishes its 64 alphas, waits for the slide's other
 * workgroups - all part of the launch - and ranks its elements); bit 4 = tokens [B, N + 1, 128] + importance (or, alpha_from_importance != 0,
 * importance READ back) + q | k | v operand images of paths_attention_h3_img (qkv_images: paths_attention_x6_workspace(B, N + 1, 4, 32,
 * 2) bytes; w_qkv: paths_tlayer_pack_ws part 1 image with scale s_wqkv, qscale = log2(e) / sqrt(32)).  Bits 2 and 4 are stop-event
 * capable launches and may be issued by separate calls on different streams behind bit 1.  pe_table (paths_pe_table) is required;
 * positions must be < pe_rows (they are clamped for memory safety). */
int paths_importance_qkv_x6(const float* y, int64_t ldy, const int64_t* y_rows, const float* y_add, int64_t ldya, const void* w_ip_x6,
                            const float* b1, const float* w2, const float* b2 /* device scalar */, const float* bp, const float* special,
                            const float* pe_table, int pe_rows, const int64_t* locs, const int64_t* num_ims, int B, int N,
                            int patch_size, int pe_mode, int imp_mul, float* importance, float* tokens, int D, int skip_padding,
                            float w_scale, float a_scale, float* splitk_ws, const void* w_qkv, const float* bqkv, float s_wqkv,
                            float qscale, void* qkv_images, int phases, int alpha_from_importance,
                            /* phase 8 (importance finish + top-K in one launch, instead of phase 2): the outputs of paths_topk_rows */
                            int keep, int* keep_idx, int64_t ldk, int* keep_count, const float* row_base, int64_t row_ld, int64_t* kept_rows,
                            const float* zero_row, int* counters /* 2 B int32, zero on entry, left zero */, int* status, paths_stream_t stream);
/* out (+)= maskop(act(a W[:, k0:k0+K]^T + b)) + residual, W = pack of an [Npad, Kpacked] weight; Npad % 128 == 0
 * (256-column tiles when Npad % 256 == 0, else 128-column tiles) */
int paths_gemm_nt_x6(const float* a, int64_t lda, const void* w_x6, int Kpacked, int k0, const float* b, float* out, int64_t ldo,
                     int M, int N, int Npad, int K, int act, const float* residual, int64_t ldr, const float* mask,
                     int64_t ldm, int accumulate, int planes, float w_scale, float a_scale, paths_stream_t stream);
/* out = act((A + A_add) W^T + b), planes = 2: the GEMM input is summed in fp32 while staged (Y = X + h1, reference model/paths.py:89-91,
 * never materialised); A as a matrix or as row addresses (exactly one of a / a_rows); num_ims (optional) skips whole tiles of padding.
 * Npad a multiple of 256 (128 x 256 tiles) or of 192 (128 x 192 tiles: a 320-column product pads to 384 instead of 512). */
int paths_gemm_add_nt_x6(const float* a, int64_t lda, const int64_t* a_rows, const float* a_add, int64_t ld_add, const void* w_x6, int Kpacked,
                         const float* b, float* out, int64_t ldo, int M, int N, int Npad, int K, int act, const int64_t* num_ims,
                         int rows_per_slide, float w_scale, float a_scale, paths_stream_t stream);
/* The same product without bias / activation, A given as row ADDRESSES (planes = 2 only): row m = the K floats at a_rows[m]. */
int paths_gemm_rows_nt_x6(const int64_t* a_rows, const void* w_x6, int Kpacked, int k0, float* out, int64_t ldo,
                          int M, int Npad, int K, int planes, float w_scale, float a_scale, paths_stream_t stream);

/* ---- backward-pass building blocks (reference: autograd of train.py:65 loss.backward()) --------------------------
 * paths_gemm_nt_f32: out (+)= maskop(act(a W^T + b)) + residual; with W = a transposed weight copy this is dX = dY W.
 * paths_gemm_tn_f32: out[N1,N2] (+)= a[M,N1]^T [b0|b1][M,N2]  (weight gradients; split-M slabs summed in a fixed order).
 * paths_colsum_f32 : out[N] (+)= sum over rows (bias gradients).  paths_transpose_f32: out = in^T. */
int paths_gemm_nt_f32(const float* a, int64_t lda, const float* w, int64_t ldw, const float* b, float* out, int64_t ldo,
                      int M, int N, int Npad, int K, int act, const float* residual, int64_t ldr, const float* mask,
                      int64_t ldm, int accumulate, paths_stream_t stream);
int64_t paths_gemm_tn_workspace(int N1, int N2, int splits);
int paths_gemm_tn_f32(const float* a, int64_t lda, const float* b0, int64_t ldb0, int nb0, const float* b1, int64_t ldb1,
                      float* out, int64_t ldo, int M, int N1, int N2, int splits, int accumulate, float* workspace,
                      paths_stream_t stream);
/* The same product on the bf16 matrix cores, operands split in registers: planes = 3: three exact bf16 planes (6 MFMAs per block, as
 * accurate as the f32 MFMA); planes = 2: hi | mid only (3 MFMAs, 16 significant bits per operand at fp32's exponent range, relative
 * error of a product ~2e-5: the training step's default since round 3).  `splits` is an upper bound (>= 32 rows per split are kept);
 * operands must be smaller than 4 GiB each (32-bit buffer offsets). */
int paths_gemm_tn_x6(const float* a, int64_t lda, const float* b0, int64_t ldb0, int nb0, const float* b1, int64_t ldb1,
                     float* out, int64_t ldo, int M, int N1, int N2, int splits, int accumulate, float* workspace, int planes,
                     paths_stream_t stream);
int paths_colsum_f32(const float* a, int64_t lda, int M, int N, float* out, int splits, int accumulate, float* workspace,
                     paths_stream_t stream);
int paths_transpose_f32(const float* in, int64_t ldi, int R, int C, float* out, int64_t ldo, paths_stream_t stream);

/* Multi-tensor AdamW, one launch per step (replaces torch.optim.AdamW's ~12 foreach launches, reference train.py:49-50 /
 * train.py:66 `opt.step()`; same operation order and fp32 rounding as torch/optim/adam.py:_multi_tensor_adam, so a training
 * trajectory is torch's own bit for bit).  All arrays live in DEVICE memory: table [n][4] addresses (param, grad, exp_avg,
 * exp_avg_sq; fp32 contiguous tensors), numel [n], blocks [nblocks][2] = (tensor index, first element) with
 * paths_adamw_chunk() elements per block, step_size [n] = -lr / (1 - beta1^t), bc2_sqrt [n] = sqrt(1 - beta2^t).
 * wd_scale = 1 - lr * weight_decay (has_wd = 0: no decay), lerp_w = 1 - beta1 (< 0.5), value = 1 - beta2;
 * flavor: bit 0 / 1 / 2 = lerp / addcmul / addcdiv evaluated as ONE fused multiply-add (7 = what the installed torch does). */
int paths_adamw_chunk(void);
int paths_adamw_multi(const int64_t* table, const int64_t* numel, const int32_t* blocks, int nblocks, const float* step_size,
                      const float* bc2_sqrt, float wd_scale, int has_wd, float lerp_w, float beta2, float value, float eps, int flavor,
                      paths_stream_t stream);

/* LSTMCell backward, element-wise parts (reference model/interface.py:52-56 differentiated):
 *   a: dpre_o = dh1 tc o(1-o) -> dG[:, 3Hc:], dpre_h = dh1 o (1-tc^2);  b: packed df|dr|dm -> dG[:, :3Hc], dc0 = dc1 f. */
int paths_lstm_bwd_a(const float* dh1, int64_t ldd, const float* dh1b, int64_t lddb, const float* o, const float* tc,
                     const int64_t* num_ims, int rows_per_slide, int64_t M, int D, float* dpre_o, int64_t ldo,
                     float* dpre_h, paths_stream_t stream);
int paths_lstm_bwd_b(const float* dc1_h, const float* dc1_ext, int64_t lde, const float* frm, const float* c0, int64_t ldc0,
                     const int64_t* num_ims, int rows_per_slide, int64_t M, int Hc, float* dg, int64_t ldg, float* dc0,
                     int64_t lddc0, paths_stream_t stream);
/* importance MLP / scaling backward per patch row (reference model/paths.py:95-98 differentiated). */
int paths_importance_bwd(const float* dtok, const float* pproj, const float* hid, const float* alpha, const float* w2,
                         const int64_t* num_ims, int rows_per_slide, int64_t M, int imp_mul, float* du, float* da, float* dah,
                         paths_stream_t stream);
/* LayerNorm (width 128) forward with saved xhat / rstd, and backward (dx, dy*xhat for the gamma gradient). */
int paths_layernorm_fwd_stats(const float* x, const float* add, const float* gamma, const float* beta, float* y, float* xhat,
                              float* rstd, int64_t rows, int d, float eps, paths_stream_t stream);
int paths_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx, float* dyxhat,
                        int64_t rows, int d, paths_stream_t stream);
/* paths_layernorm_bwd without the dy*xhat tensor: every workgroup of rows_per_block rows also writes one 384-float slab
 * sum(dy * xhat) | sum(dy) | sum(dx) of its rows; paths_reduce_slabs_f32 over the ceil(rows / rows_per_block) slabs gives
 * dgamma | dbeta | column sums of dx (the gradient of a bias added in front of the LayerNorm). */
int paths_layernorm_bwd_sums(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx, float* slabs,
                             int64_t rows, int d, int rows_per_block, paths_stream_t stream);
int paths_reduce_slabs_f32(const float* slabs, int splits, int n, float* out, int accumulate, paths_stream_t stream);
/* Deferred slab reductions (csrc/reduce_multi.hip): between paths_defer_reductions(1) and paths_flush_reductions() the final
 * "out (+)= sum of slabs" pass of paths_gemm_tn_f32 / paths_gemm_tn_x6 / paths_colsum_f32 / paths_reduce_slabs_f32 is registered instead
 * of launched; the flush runs all registered reductions in one launch per 32 entries, each in the summation order of the launch it
 * replaces (bit-identical outputs).  The caller keeps the slab workspaces alive until the flush and flushes before anything reads an
 * output; an entry whose output overlaps a pending output or pending slabs flushes first.  Per host thread.  paths_defer_reductions
 * returns the previous setting and does not flush. */
int paths_defer_reductions(int on);
int paths_flush_reductions(int* n_entries, paths_stream_t stream);

/* Self-attention backward, flash style (recompute from q, k, lse; reference: autograd through the attention of
 * model/aggregator.py:70-72).  q is the stored pre-scaled query; gradients land token-major in dqkv [B,T,384] =
 * [dq_scaled | dk | dv] (zero-initialised by the caller); ws_dsum: B*H*T floats. */
int paths_attention_bwd_f32(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                            const int64_t* num_ims, float* dqkv, float* ws_dsum, int B, int T, int H, int head_dim,
                            paths_stream_t stream);
/* The last decoder layer is read at token 0 only (model/aggregator.py:75): single-query attention of the TRAINING path, keys split
 * over workgroups (csrc/attn_token0.hip).  a0 / da0 [B, H*32] = attention output of token 0 and its gradient, lse0 [B, H] (log2
 * domain, un-dropped softmax), dropout p on the probabilities (mask element ((b*H + h)*T + 0)*T + key of site drop_key; p = 0: none);
 * ws: paths_attention_token0_workspace(B, T, H) floats.  The backward writes dk / dv of every valid key and dq of row 0 into dqkv
 * [B, T, 3*H*32] (zero on entry). */
int64_t paths_attention_token0_workspace(int B, int T, int H);
int paths_attention_token0_any(const float* qkv, int64_t ld, const int64_t* num_ims, float* a0, float* ws, int B, int T, int H, int head_dim,
                               float qscale, paths_stream_t stream);      /* inference form for any head_dim on the token-major in_proj output */
int paths_attention_token0_fwd(const float* q, const float* k, const float* v, const int64_t* num_ims, float* a0, float* lse0, float* ws,
                               int B, int T, int H, int head_dim, uint64_t drop_key, float drop_p, paths_stream_t stream);
int paths_attention_token0_bwd(const float* q, const float* k, const float* v, const float* a0, const float* da0, const float* lse0,
                               const int64_t* num_ims, float* dqkv, float* ws, int B, int T, int H, int head_dim, uint64_t drop_key,
                               float drop_p, paths_stream_t stream);

/* Generic out = act(a W^T + b) on the fp32 matrix cores (W rows zero-padded to Npad, a multiple of 128). */
int paths_linear_f32(const float* a, int64_t lda, const float* w, const float* b, float* out, int64_t ldo,
                     int M, int N, int Npad, int K, int act, paths_stream_t stream);

/* Masked multi-head self-attention, flash style (nn.MultiheadAttention inside nn.TransformerDecoderLayer as
 * called at reference model/aggregator.py:70-72, key mask utils.py:97-103).
 *   q,k,v [B,H,T,32] head-major, q pre-scaled by log2(e)/sqrt(32); o [B,T,H*32]; valid keys = num_ims[b]+1.
 *   max_queries > 0 restricts the computed query rows to [0, max_queries) (the last decoder layer is read at
 *   token 0 only, reference model/aggregator.py:75); 0 = all T rows. */
int paths_attention_f32(const float* q, const float* k, const float* v, float* o, float* lse, const int64_t* num_ims,
                        int B, int T, int H, int head_dim, int max_queries, paths_stream_t stream);

/* paths_attention_f32 on the 16-bit matrix cores with fp32 accuracy (csrc/attn_x6.hip): q, k, v are first re-written as
 * split operands in MFMA-fragment order (planes = 3: exact bf16 hi|mid|lo, 6 MFMAs per product block; planes = 2: fp16 hi|lo,
 * 3 MFMAs, |q|, |k|, |v| < 65504; workspace of paths_attention_x6_workspace(B, T, H, head_dim, planes) bytes, caller-owned),
 * then S^T = K Q^T and O^T += V^T P^T run with fp32 accumulation; P is split in registers.
 * Same arguments and results (to fp32 rounding) as paths_attention_f32. */
int64_t paths_attention_x6_workspace(int B, int T, int H, int head_dim, int planes);
int paths_attention_x6(const float* q, const float* k, const float* v, float* o, float* lse, const int64_t* num_ims,
                       int B, int T, int H, int head_dim, int max_queries, void* workspace, int planes, int images_ready,
                       paths_stream_t stream);
/* images_ready != 0 (planes = 2): the workspace already holds the operand images (paths_token_layer_h3 wrote them through its
 * qkv_images argument); q, k, v are not read and the re-write launch is skipped. */

/* Token-row chain of one post-LN decoder layer with empty memory + the next in_proj (same call site):
 *   do_post: x_out = norm3(x' + ffn(x')), x' = norm2(norm1(x_in + out_proj(attn)) + cross_attn_bias)
 *   do_qkv : q,k,v = in_proj(x)  (x = x_out if do_post else x_in) written head-major, q scaled by qscale.
 *   max_tokens > 0 restricts the processed token rows to [0, max_tokens) (last layer); 0 = all T rows. */
int paths_token_layer_f32(const float* x_in, const float* attn, float* x_out,
                          const float* wo, const float* bo, const float* ln1g, const float* ln1b, const float* cab,
                          const float* ln2g, const float* ln2b, const float* w1, const float* b1, const float* w2,
                          const float* b2, const float* ln3g, const float* ln3b, const float* wqkv, const float* bqkv,
                          float* q, float* k, float* v, const int64_t* num_ims, int B, int T, int d, int H,
                          int do_post, int do_qkv, int skip_padding, float qscale, float eps, int max_tokens,
                          paths_stream_t stream);

/* paths_token_layer_f32 on the fp16 matrix cores with fp32 accuracy (csrc/tlayer_h3.hip; two-plane operand split as in the
 * planes = 2 GEMMs, 3 x v_mfma_f32_16x16x32_f16 per product block, activations split in registers).  Weights are passed as
 * images built once per weight version by paths_tlayer_pack_h3: part 0 = (out_proj, linear1, linear2) of THIS layer with
 * power-of-two scales (s_wo, s_w1, s_w2), part 1 = in_proj of the NEXT layer with s_wqkv (max|w| * scale < 65504).
 * Same arguments otherwise, same results to fp32 rounding. */
int64_t paths_tlayer_h3_image_bytes(int part);
int paths_tlayer_pack_h3(int part, const float* wa, const float* wb, const float* wc, float s_a, float s_b, float s_c, void* out,
                         paths_stream_t stream);
int paths_token_layer_h3(const float* x_in, const float* attn, float* x_out, const void* w_post, const void* w_qkv,
                         const float* bo, const float* ln1g, const float* ln1b, const float* cab, const float* ln2g, const float* ln2b,
                         const float* b1, const float* b2, const float* ln3g, const float* ln3b, const float* bqkv,
                         float s_wo, float s_w1, float s_w2, float s_wqkv,
                         float* q, float* k, float* v, const int64_t* num_ims, int B, int T, int d, int H,
                         int do_post, int do_qkv, int skip_padding, float qscale, float eps, int max_tokens, void* qkv_images,
                         paths_stream_t stream);
/* qkv_images (optional, max_tokens = 0): a paths_attention_x6_workspace(B, T, H, 32, 2) buffer; the in_proj outputs are written
 * as the two-plane operand images of paths_attention_x6 (masked keys zeroed) instead of fp32 q, k, v (which may then be null). */

/* paths_token_layer_h3 in WEIGHT-STATIONARY form (csrc/tlayer_ws.hip; reference model/aggregator.py:25-33, 70-72 = torch's post-LN
 * TransformerDecoderLayer after the self-attention, and the next layer's in_proj).  A workgroup = 4 waves = 64 tokens; wave w owns a
 * quarter of every product's output features for all 64 tokens, its weight fragments stream L2 -> registers (never through LDS),
 * activations cross LDS as the fp16 hi | lo B-operand fragments of the next product, LayerNorm statistics as (mean, M2) pairs.
 * w_post / w_qkv: paths_tlayer_pack_ws images (part 0 = out_proj, linear1, linear2 of THIS layer; part 1 = in_proj of the NEXT layer)
 * with their power-of-two scales.  attn: fp32 [B,T,d], or attn_img: the fragment image written by paths_attention_h3_img.
 * qkv_images: a paths_attention_x6_workspace(B, T, 4, 32, 2) buffer that receives the attention operand images (masked keys zeroed).
 * zero_words (optional, do_post): n_zero <= 256 int32 words set to 0 by the launch (arrival counters of paths_token0_tail_ws). */
int64_t paths_tlayer_ws_image_bytes(int part, int d);
int paths_tlayer_pack_ws(int part, const float* wa, const float* wb, const float* wc, float s_a, float s_b, float s_c, void* out, int d,
                         paths_stream_t stream);
int paths_token_layer_ws(const float* x_in, const float* attn, const void* attn_img, float* x_out, const void* w_post, const void* w_qkv,
                         const float* bo, const float* ln1g, const float* ln1b, const float* cab, const float* ln2g, const float* ln2b,
                         const float* b1, const float* b2, const float* ln3g, const float* ln3b, const float* bqkv,
                         float s_wo, float s_w1, float s_w2, float s_wqkv, void* qkv_images, const int64_t* num_ims,
                         int B, int T, int d, int H, int do_post, int do_qkv, int skip_padding, float qscale, float eps,
                         int* zero_words, int n_zero, paths_stream_t stream);
/* The same kernel instantiated at trans_dim 192 (the reference's dataclass default, config.py:30; any head count) with the in_proj result
 * as fp32 token-major rows qkv_rows [B*T][ld_qkv] = [q | k | v] (q UNscaled: the operand of the shape-generic attention kernels)
 * instead of the head_dim-32 fragment images; the attention output enters as fp32 rows.  do_post and / or do_qkv as above. */
int paths_token_layer_ws_rows(const float* x_in, const float* attn, float* x_out, const void* w_post, const void* w_qkv,
                              const float* bo, const float* ln1g, const float* ln1b, const float* cab, const float* ln2g, const float* ln2b,
                              const float* b1, const float* b2, const float* ln3g, const float* ln3b, const float* bqkv,
                              float s_wo, float s_w1, float s_w2, float s_wqkv, float* qkv_rows, int64_t ld_qkv, const int64_t* num_ims,
                              int B, int T, int d, int do_post, int do_qkv, int skip_padding, float eps, paths_stream_t stream);
/* paths_attention_x6 (planes = 2, operand images already in `workspace`) with the output written as the out_proj operand image of
 * paths_token_layer_ws: B * ceil(T/64) * 64 * H * 32 * 4 bytes, [slide][64-token group][head][16-token tile][plane][64 lanes][16 B]. */
int paths_attention_h3_img(void* o_img, const int64_t* num_ims, int B, int T, int H, int head_dim, void* workspace, paths_stream_t stream);

/* paths_token0_tail WITHOUT the last layer's K / V projections, in ONE launch (csrc/token0_ws.hip; reference model/aggregator.py:70-75
 * for the final layer, model/paths.py:130-139).  With one query per head the projections fold into the query and the output:
 * score_h,t = (Wk_h^T q_h) . x_t + const, o_h = Wv_h (sum_t p_h,t x_t) + bv_h, so the launch only reads the layer's INPUT rows
 * x1 [B,T,128].  img: image built once per weight version by paths_token0_pack_ws (A_h = c Wk_h^T Wq_h, a0_h = c Wk_h^T bq_h and
 * 16-byte-transposed Wv, Wo, W1, W2; qscale c = log2(e)/sqrt(head_dim)); bv = in_proj_bias + 2 d.  Workgroups = (token split, head)
 * pairs per slide; each publishes a (max, sum, z[128]) partial, the LAST arriver of a slide (agent-scope release / acquire around an
 * arrival ticket) runs the row chain - or, when all workgroups of the launch fit the chip at once (<= 192), the DISTRIBUTED form:
 * every workgroup pushes its partial through its head's Wv / Wo slices before the ticket and carries a slice of the feed-forward
 * after a flag hop, so that no CU pulls more than ~100 KB of weights.  partials: paths_token0_ws_partials(B, T) floats of scratch;
 * counters: 3 B int32 words, zero on entry, left zero; status (optional): bit 4 set if a bounded hand-off wait gave up.
 * Exact fp32 FMA chains.
 * Widths: trans_dim d = 128 (both forms) and d = 192, the reference's dataclass default (config.py:30; distributed form only, 4 heads
 * of 48: 768 threads per workgroup, x1 [B,T,192]) - the _d entry points take d, the ones without it are the d = 128 forms kept for
 * existing callers.  paths_token0_ws_supported(B, T, d, H): 1 if paths_token0_tail_ws can run the shape (at 192: while the
 * distributed launch fits the chip, B <= 24 on an MI355X, T > 128), else 0 - the caller then takes the generic launches. */
int64_t paths_token0_ws_image_bytes(void);
int64_t paths_token0_ws_image_bytes_d(int d);
int64_t paths_token0_ws_partials(int B, int T);
int64_t paths_token0_ws_partials_d(int B, int T, int d);
int paths_token0_ws_supported(int B, int T, int d, int H);
int paths_token0_pack_ws(const float* wqkv, const float* bqkv, const float* wo, const float* bo, const float* w1, const float* w2, float qscale,
                         void* out, paths_stream_t stream);
int paths_token0_pack_ws_d(const float* wqkv, const float* bqkv, const float* wo, const float* bo, const float* w1, const float* w2, float qscale,
                           int d, void* out, paths_stream_t stream);
int paths_token0_tail_ws(const float* x1, const int64_t* num_ims, const void* img, const float* bv, const float* bo,
                         const float* ln1g, const float* ln1b, const float* cab, const float* ln2g, const float* ln2b,
                         const float* b1, const float* b2, const float* ln3g, const float* ln3b, const float* lnfg, const float* lnfb,
                         const float* ctx_prev, int64_t ctx_stride, const float* ctx_all, int ctx_depth,
                         const float* wcls, const float* bcls, int num_logits, int cls_in,
                         float* ctx_out, float* logits, float* partials, int* counters, int* status, int B, int T, int d, int H,
                         float eps, float eps_final, int special_last /* 1: the special token is row num_ims[b] (paths_importance_qkv_x6's
                         token order) instead of row 0 */, paths_stream_t stream);

/* ---- shape-generic kernels (csrc/generic.hip): any trans_dim (multiple of 32, <= 1024), head_dim in {16, 32, 48, 64}, any
 * importance_mlp_hidden_dim - e.g. the reference's dataclass defaults trans_dim 192 / 4 heads (config.py:30-36).  With
 * paths_gemm_nt_f32 for the products they evaluate model/paths.py:95-98,119-139 and model/aggregator.py:37-76 in exact fp32; the
 * shipped 128 / 4 / 128 geometry runs on the specialised kernels above instead.
 *   paths_attention_any   softmax(q k^T / sqrt(hd)) v, keys >= num_ims[b] + 1 masked; qkv [B*T, 3 d] token-major (in_proj output),
 *                         qscale = log2(e) / sqrt(head_dim), o [B, T, d]; max_queries > 0: only queries [0, max_queries)
 *   paths_layernorm_rows  y = LayerNorm(x (+ add [d])) * gamma + beta per row (row strides ldx / ldy)
 *   paths_importance_rows importance[m] = valid ? sigmoid(hid[m] . w2 + b2) : 0 from hid = relu(Y W1^T + b1)  (utils.py:106-115)
 *   paths_tokens_assemble tokens[b, 0] = special, tokens[b, 1 + n] = alpha P[b n] + bp + PE  (aggregator.py:37-65, utils.py:16-23,47-67)
 *   paths_final_head_any  paths_final_head for any trans_dim */
int paths_attention_any(const float* qkv, int64_t ld, float* o, const int64_t* num_ims, int B, int T, int H, int head_dim, float qscale,
                        int max_queries, paths_stream_t stream);
/* The same attention on the 16-bit matrix cores with fp32-accurate operands (csrc/attn_h3_any.hip: two fp16 planes per operand, three
 * MFMAs per product block - the arithmetic of the tuned head_dim-32 kernel - for head_dim 16 / 32 / 48 / 64); all queries;
 * workspace: paths_attention_h3_any_workspace(B, T, H, head_dim) bytes. */
int64_t paths_attention_h3_any_workspace(int B, int T, int H, int head_dim);
int paths_attention_h3_any(const float* qkv, int64_t ld, float* o, const int64_t* num_ims, int B, int T, int H, int head_dim, float qscale,
                           void* workspace, paths_stream_t stream);
/* ... on operand images already in the workspace (written by paths_token_layer_ws with d = 192, do_qkv only: head_dim 48, q scaled
 * there; or head_dim 32): no prep launch. */
int paths_attention_h3_any_img(float* o, const int64_t* num_ims, int B, int T, int H, int head_dim, void* workspace, paths_stream_t stream);
int paths_layernorm_rows(const float* x, int64_t ldx, const float* add, const float* gamma, const float* beta, float* y, int64_t ldy,
                         int64_t rows, int d, float eps, paths_stream_t stream);
int paths_layernorm2_rows(const float* x, int64_t ldx, const float* g1, const float* b1, const float* add, const float* g2, const float* b2,
                          float* y, int64_t ldy, int64_t rows, int d, float eps, paths_stream_t stream);   /* LN2(LN1(x) + add) in one pass */
int paths_importance_rows(const float* hid, int64_t ldh, const float* w2, const float* b2, const int64_t* num_ims, int rows_per_slide,
                          int64_t M, int Hi, float* importance, int relu /* 1: hid holds the pre-activation, relu applied here */, paths_stream_t stream);
int paths_tokens_assemble(const float* P, int64_t ldp, const float* importance, int imp_mul, const float* bp, const float* special,
                          const float* div_term, const int64_t* locs, int rows_per_slide, int patch_size, int pe_mode, int d, int B,
                          float* tokens, paths_stream_t stream);
/* The importance-rows and tokens-assemble launches (above) as ONE launch over the rows of the [W1 ; Wp] product (csrc/generic.hip; reference
 * model/paths.py:95-98,119-124, model/aggregator.py:37-65, utils.py:16-23,47-67): hid [M, ldh] holds the importance MLP's hidden
 * pre-activations (Hi columns) and the projection (d columns) of every patch row; writes importance [M] (0 for padded rows) and
 * tokens [B, rows_per_slide + 1, d] (special token first; padded rows bp + PE).  The positional encoding is read from pe_table
 * (the table built by paths_pe_table for this pe_mode, d and pe_rows): same values as the sin / cos calls of the tokens-assemble launch. */
int paths_importance_tokens_rows(const float* hid, int64_t ldh, const float* w2, const float* b2, const int64_t* num_ims, int rows_per_slide,
                                 int64_t M, int Hi, float* importance, int relu, int imp_mul, const float* bp, const float* special,
                                 const float* pe_table, int pe_rows, const int64_t* locs, int patch_size, int pe_mode, int d, float* tokens,
                                 paths_stream_t stream);
int paths_final_head_any(const float* x, int64_t slide_stride, const float* lng, const float* lnb, const float* ctx_prev, int64_t ctx_stride,
                         const float* ctx_all, int ctx_depth, const float* wcls, const float* bcls, int num_logits, int cls_in,
                         float* ctx_out, float* logits, int B, int d, float eps, paths_stream_t stream);

/* Shape-generic TRAINING kernels (csrc/generic_bwd.hip + the TRAIN form of the generic attention): what autograd applies in the
 * reference train step (train.py:65) to nn.LayerNorm, the importance MLP / scaling / proj_in (model/paths.py:95-98,119-124) and the
 * masked multi-head self-attention incl. its dropout (model/aggregator.py:25-33,70-72), for any trans_dim % 32 == 0 (<= 1024), head_dim
 * 16 / 32 / 48 / 64 and any importance hidden width; the shipped 128 / 4 / 128 geometry trains on the specialised entry points below.
 *   paths_attention_any_train      paths_attention_any + lse [B,H,T] (log2 domain, un-dropped softmax; may be null) + dropout p on the
 *                                  probabilities (mask element ((b*H + h)*T + q)*T + k of site drop_key; p = 0: none)
 *   paths_attention_bwd_any        dqkv [B*T, 3d] = [dq | dk | dv] (token-major, ZERO on entry) from qkv (q unscaled), o, d_o, lse;
 *                                  ws_dsum: B*H*T floats of scratch; max_queries > 0: only those queries carry an output gradient
 *   paths_layernorm_fwd_stats_any  y (may be null), xhat, rstd of LayerNorm(x (+ add [d])) over contiguous [rows, d]
 *   paths_layernorm_bwd_any        dx and dy * xhat
 *   paths_layernorm_bwd_sums_any   dx and one slab [sum dy*xhat | sum dy | sum dx] (3 d floats) per rows_per_block rows
 *   paths_importance_bwd_any       du [M, ldu] = [dhid (Hi) | dP (d) | zeros], da [M], dah [M, Hi] = da * hid
 *   paths_importance_rows_bwd_any  lstm = false (model/paths.py:95-109): the importance MLP's backward through Z = alpha X, any Hi */
int paths_attention_any_train(const float* qkv, int64_t ld, float* o, float* lse, const int64_t* num_ims, int B, int T, int H, int head_dim,
                              float qscale, int max_queries, uint64_t drop_key, float drop_p, paths_stream_t stream);
int paths_attention_bwd_any(const float* qkv, int64_t ld, const float* o, const float* d_o, const float* lse, const int64_t* num_ims,
                            float* dqkv, float* ws_dsum, int B, int T, int H, int head_dim, float qscale, int max_queries,
                            uint64_t drop_key, float drop_p, paths_stream_t stream);
int paths_layernorm_fwd_stats_any(const float* x, const float* add, const float* gamma, const float* beta, float* y, float* xhat,
                                  float* rstd, int64_t rows, int d, float eps, paths_stream_t stream);
int paths_layernorm_bwd_any(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx, float* dyxhat,
                            int64_t rows, int d, paths_stream_t stream);
int paths_layernorm_bwd_sums_any(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx, float* slabs,
                                 int64_t rows, int d, int rows_per_block, paths_stream_t stream);
int paths_importance_rows_bwd_any(const float* dz_rows, const float* x, int D, const float* hid, const float* alpha, const float* w2,
                                  const int64_t* num_ims, int rows_per_slide, int64_t M, int Hi, float* dh, float* da, float* dah,
                                  paths_stream_t stream);
int paths_importance_bwd_any(const float* dtok, const float* pproj, const float* hid, const float* alpha, const float* w2,
                             const int64_t* num_ims, int rows_per_slide, int64_t M, int imp_mul, int Hi, int d, int64_t ldu, float* du,
                             float* da, float* dah, paths_stream_t stream);

/* LAST decoder layer evaluated at token 0 only + decoder.norm + slide-context residual / concat + classifier, one
 * launch (reference model/aggregator.py:70-75 for the final layer, model/paths.py:130-139).  Legal because only
 * out[:, 0] of the final layer is read: it needs K/V of every token (q,k,v as written by paths_token_layer_f32 for
 * that layer) but queries / out_proj / norms / FFN of one row per slide.  x_in = the layer's input activations;
 * ws_partials = scratch of B*H*16*36 floats (split-key attention partials). */
int paths_token0_tail(const float* x_in, const float* q, const float* k, const float* v, const int64_t* num_ims,
                      const float* wo, const float* bo, const float* ln1g, const float* ln1b, const float* cab,
                      const float* ln2g, const float* ln2b, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* ln3g, const float* ln3b, const float* lnfg, const float* lnfb,
                      const float* ctx_prev, int64_t ctx_stride, const float* ctx_all, int ctx_depth,
                      const float* wcls, const float* bcls, int num_logits, int cls_in,
                      float* ctx_out, float* logits, float* ws_partials, int B, int T, int d, int H, float eps,
                      float eps_final, paths_stream_t stream);

/* decoder.norm on token 0, slide-context residual / concat, classifier
 * (reference model/aggregator.py:75, model/paths.py:130-139). */
int paths_final_head(const float* x, int64_t slide_stride, const float* lng, const float* lnb,
                     const float* ctx_prev, int64_t ctx_stride, const float* ctx_all, int ctx_depth,
                     const float* wcls, const float* bcls, int num_logits, int cls_in,
                     float* ctx_out, float* logits, int B, int d, float eps, paths_stream_t stream);

/* Stand-alone wavefront-reduction LayerNorm over rows of width 128 (aten::native_layer_norm). */
int paths_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int d,
                        float eps, paths_stream_t stream);

/* torch.topk(importance[:n], min(n, keep)).indices per slide (reference data_utils/slide.py:294-301).
 * Order: score descending, ties by index ascending.  keep = -1 keeps every patch in original order. */
int paths_topk(const float* scores, int64_t ld, const int64_t* num_ims, int B, int n_max, int keep,
               int* keep_idx, int64_t ldk, int* keep_count, paths_stream_t stream);
/* paths_topk + kept_rows[b, i] = address of row_base[b, keep_idx[b, i], 0] of a row-major [B, slide_rows, row_ld] float table
 * (entries beyond keep_count[b] = zero_row): lets paths_gemm_rows_nt_x6 read the kept parents' rows in place. */
int paths_topk_rows(const float* scores, int64_t ld, const int64_t* num_ims, int B, int n_max, int keep,
                    int* keep_idx, int64_t ldk, int* keep_count, const float* row_base, int64_t row_ld, int64_t slide_rows,
                    int64_t* kept_rows, const float* zero_row, paths_stream_t stream);

/* 4-child expansion, bounds + background filter, stable compaction (reference data_utils/slide.py:303-331).
 *   mask_ptrs[b] -> uint8 [X*Y] tissue mask of the NEXT level (1 = row sum != 0).  status bit0: a slide
 *   produced zero children (reference fallback slide.py:336-352 needed), bit1: capacity n_next exceeded.
 *   child_pos (optional, [B, 4*ldk]): output row of every candidate child (-1 if dropped), for paths_gather_rows_bwd.
 *   hp_row (optional, [B, n_next]): row b*ldk + i of the kept-parent table for every child (-1 on padding), see paths_lstm_cell. */
int paths_expand_children(const int* keep_idx, int64_t ldk, const int* keep_count, const int64_t* locs, int64_t n_cur,
                          int patch_size, const int* next_x, const int* next_y, const int64_t* mask_ptrs, int B,
                          int64_t n_next, int64_t* num_out, int64_t* locs_out, int64_t* parent_out, int* src_row,
                          int* src_cell, int* status, int* child_pos, int* hp_row, paths_stream_t stream);

/* Rare fallback of reference data_utils/slide.py:336-352 for slides with num_out[b] == 0 after paths_expand_children:
 * continue with every tissue cell of the next grid (every cell if it has no tissue), zero patch context (src_row = -1),
 * parent_inds = cell index.  Other slides are untouched.  status bit1 set if n_next is too small. */
int paths_fallback_all_cells(const int* next_x, const int* next_y, const int64_t* mask_ptrs, int patch_size, int B,
                             int64_t n_next, int64_t* num_out, int64_t* locs_out, int64_t* parent_out, int* src_row,
                             int* src_cell, int* status, int* hp_row, paths_stream_t stream);

/* Gather child features from the next-level grids and parent LSTM state (reference slide.py:318,327-331;
 * zero padding of data_utils/dataset.py:216-227 when zero_pad != 0).  state_cur points at the first state column to copy
 * (row stride ld_state_cur), Dp = number of columns copied into state_out [B, n_next, Dp].
 * fts_out may be NULL when row_ptrs [B, n_next] is given: the features are then not copied at all, row_ptrs receives the
 * ADDRESS of every child's feature row inside its resident grid (padding rows: zero_row, D zeros) and the split-operand
 * GEMMs read the rows in place (x_rows / y_rows of paths_lstm_cell_x6 / paths_importance_proj_x6). */
int paths_gather_rows(const int64_t* grid_ptrs, const int* src_cell, int D, const float* state_cur, int64_t n_cur,
                      int64_t ld_state_cur, const int* src_row, int Dp, const int64_t* num_out, int B, int64_t n_next,
                      float* fts_out, float* state_out, int zero_pad, int64_t* row_ptrs, const float* zero_row,
                      paths_stream_t stream);

/* Compact table of the kept parents' rows: out[b*ldk + i] = src[b, keep_idx[b,i], 0:D] (zeros beyond keep_count). */
int paths_gather_kept_rows(const float* src, int64_t n_cur, int64_t ld_src, const int* keep_idx, int64_t ldk, const int* keep_count,
                           int D, int B, float* out, paths_stream_t stream);

/* Backward of the parent-state gather: d_cur[b, keep_idx[i]] = sum over the surviving children of parent i of d_next
 * (d_cur zero-initialised by the caller). */
int paths_gather_rows_bwd(const int* keep_idx, int64_t ldk, const int* keep_count, const int* child_pos, const float* d_next,
                          int64_t n_next, int Dp, float* d_cur, int64_t n_cur, int B, paths_stream_t stream);

/* The once-per-parent form of the TRAINING step (siblings share their parent's h, reference data_utils/slide.py:303-331 +
 * model/interface.py:49-56): paths_sibling_sum adds the rows of the surviving children of every kept parent -
 * dst[b, row(i), 0:width] = sum_children src[b, child, 0:width], row(i) = keep_idx[b, i] or, with keep_idx NULL, the compact slot i
 * (the gradient of the once-per-parent partial pre-activations: dHP = sum of the children's dG) - and paths_scatter_kept_rows puts
 * a compact per-kept-parent table back into the level's rows: dst[b, keep_idx[b, i], 0:width] = src[b, i, 0:width]. */
int paths_sibling_sum(const int* keep_idx, int64_t ldk, const int* keep_count, const int* child_pos, const float* src, int64_t n_src,
                      int64_t ld_src, int width, float* dst, int64_t n_dst, int64_t ld_dst, int B, paths_stream_t stream);
int paths_scatter_kept_rows(const float* src, int64_t ldk, int64_t ld_src, const int* keep_idx, const int* keep_count, float* dst,
                            int64_t n_dst, int64_t ld_dst, int width, int B, paths_stream_t stream);

/* Level-0 batch: every grid cell in row-major order (reference data_utils/slide.py:257-269,362-381). */
int paths_level0_batch(const int64_t* grid_ptrs, const int* gx, const int* gy, int B, int D, int patch_size, int64_t n0,
                       float* fts, int64_t* locs, int64_t* parent, int64_t* num_ims, int zero_pad, int64_t* row_ptrs,
                       const float* zero_row, paths_stream_t stream);

/* z = alpha * x (+ h on valid rows): importance scaling and the non-LSTM hierarchical-context add
 * (reference model/paths.py:96-109). */
int paths_scale_add_rows(const float* x, const float* alpha, const float* h, const int64_t* num_ims, int rows_per_slide,
                         int D, int64_t M, int use_alpha, float* z, paths_stream_t stream);

/* Tissue mask of a preprocessed grid [cells, D]: 1 iff fp32 row sum != 0 (reference slide.py:324). */
int paths_tissue_mask(const float* grid, int64_t cells, int D, uint8_t* mask, paths_stream_t stream);

/* The same pass (mask may be NULL) + *absmax_bits = max(*absmax_bits, fp32 bit pattern of max|x| over the grid): non-negative
 * floats order like their bit patterns, a NaN reads back above +inf.  Guards the fp16-split range contract of the default mode
 * (replaces nothing in the reference; its fp32 CPU path has no such limit). */
int paths_tissue_mask_absmax(const float* grid, int64_t cells, int D, uint8_t* mask, uint32_t* absmax_bits, paths_stream_t stream);

/* Counter-based synthetic grid (paths_amd/synthetic.py). */
int paths_synth_grid(float* grid, int X, int Y, int D, uint32_t slide_level_key, int level, uint64_t bg_threshold,
                     paths_stream_t stream);

/* lstm = false training (reference model/paths.py:95-109, Z = alpha X + hctx): gradient of the importance MLP through the row
 * scaling: dalpha = dZ . X per row, dz = valid dalpha alpha (1 - alpha); dh [M,128] = (hid > 0) dz w2, dah = dz hid, da = dz. */
int paths_importance_rows_bwd(const float* dz_rows, const float* x, int D, const float* hid, const float* alpha, const float* w2,
                              const int64_t* num_ims, int rows_per_slide, int64_t M, float* dh, float* da, float* dah,
                              paths_stream_t stream);

/* ---- dropout (training; reference nn.Transformer(..., dropout=p), model/aggregator.py:25-33: attention probabilities,
 * dropout1, dropout2, the feed-forward's inner dropout, dropout3).  Masks are never stored: element idx of site `key` is kept iff
 * the 16-bit half (low for even idx, high for odd) of hash(idx & ~1, key) is >= round(p * 65536) (csrc/dropout.h: one 32-bit hash
 * per pair of elements) and is regenerated by every kernel that needs it; kept values are scaled by 1 / (1 - p16), p16 =
 * round(p * 65536) / 65536 the rate actually applied.  Attention probabilities of (slide, head) pair s: element (query q, key k)
 * has idx = (s * T + q) * T' + k with T' = T rounded up to even.  The host derives one 64-bit key per (step seed, level, layer, site). */

/* out[r, c] = (resid ? resid[r, c] : 0) + (vec ? vec[c] : x[r, c]) * mask(r * N + c) / (1 - p16); exactly one of x / vec is given;
 * x, resid, out may alias.  Covers dropout1 / dropout3 (+ residual), the inner feed-forward dropout (in place), dropout2 on the
 * broadcast cross-attention bias (vec) and the masking of gradients in the backward pass. */
int paths_dropout_rows(const float* x, int64_t ldx, const float* vec, const float* resid, int64_t ldr, float* out, int64_t ldo,
                       int64_t M, int N, uint64_t key, float p, paths_stream_t stream);
/* mask[i] = 1 kept / 0 dropped, i in [0, n): tests only */
int paths_dropout_mask(float* mask, int64_t n, uint64_t key, float p, paths_stream_t stream);
/* Opt-in low-precision variant for the stress geometry (BASELINE.json configs[4]): the same attention with OCP e4m3 operands, one
 * v_mfma_f32_16x16x32_fp8_fp8 per product block (csrc/attn_fp8.hip).  4 significant bits per operand: NOT within the 1e-4 logit bar;
 * never used unless PATHS_ATTN_FP8=1.  workspace: paths_attention_fp8_workspace(B, T, H, head_dim) bytes. */
int64_t paths_attention_fp8_workspace(int B, int T, int H, int head_dim);
int paths_attention_fp8(const float* q, const float* k, const float* v, float* o, const int64_t* num_ims, int B, int T, int H,
                        int head_dim, void* workspace, paths_stream_t stream);
/* The same on the token-major in_proj output qkv [B*T, 3d] (row stride ld; q unscaled, qscale = log2(e)/sqrt(head_dim) applied while the
 * operand images are written); head_dim 32 or 64. */
int paths_attention_fp8_qkv(const float* qkv, int64_t ld, float* o, const int64_t* num_ims, int B, int T, int H, int head_dim, float qscale,
                            void* workspace, paths_stream_t stream);

/* e4m3 GEMM of the stress variant (csrc/gemm_fp8.hip: v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales, per-tensor scales) for the
 * aggregator's products over all tokens (reference model/aggregator.py:25-33: in_proj, out_proj, linear1, linear2).  Opt-in, NOT a parity
 * path (4 significant bits per operand).
 *   paths_fp8_scale        *scale = 448 / max|x| over an fp32 [M, K] matrix, on the device (scratch: one zeroed uint32, left zero);
 *                          with num_ims / rows_per_slide only the valid token rows of a token-major activation count
 *   paths_fp8_pack_weight  w8 [ceil(N/256)*256, K] = e4m3(W * *scale) (zero rows behind N), *scale = 448 / max|W|; K % 64 == 0
 *   paths_fp8_quantize     x8 [ceil(M/256)*256, K] = e4m3(x * *scale) of an fp32 [M, K] matrix (zero rows behind M)
 *   paths_gemm_nt_fp8      out[M,N] = act(A W^T + bias) (+ residual) from the two e4m3 images and their scales; K % 128 == 0 */
int paths_fp8_scale(const float* x, int64_t ld, int64_t M, int K, float* scale, unsigned int* scratch, const int64_t* num_ims,
                    int rows_per_slide, paths_stream_t stream);
int paths_fp8_pack_weight(const float* w, int64_t ldw, int N, int K, uint8_t* w8, float* scale, unsigned int* scratch, paths_stream_t stream);
int paths_fp8_quantize(const float* x, int64_t ld, int M, int K, const float* scale, uint8_t* x8, paths_stream_t stream);
int paths_gemm_nt_fp8(const uint8_t* a8, const uint8_t* w8, const float* a_scale, const float* w_scale, const float* bias,
                      float* out, int64_t ldo, int M, int N, int K, int act, const float* residual, int64_t ldr, paths_stream_t stream);
/* The same product handed on in e4m3 (the A image of the next GEMM): out8 [ceil(M/256)*256, N] = e4m3(act(A W^T + bias) * *out_scale),
 * rows >= M zero on entry; *out_scale is a calibrated per-tensor scale, out_absmax (optional) receives max|result| as float bits. */
int paths_gemm_nt_fp8_out8(const uint8_t* a8, const uint8_t* w8, const float* a_scale, const float* w_scale, const float* bias,
                           uint8_t* out8, const float* out_scale, unsigned int* out_absmax, int M, int N, int K, int act,
                           paths_stream_t stream);

/* paths_attention_x6 with dropout on the softmax probabilities: O = (softmax(S) * mask / (1 - p)) V, lse un-dropped; mask element
 * ((b * H + h) * T + q) * T + k.  q, k, v fp32 (no pre-built images). */
int paths_attention_x6_dropout(const float* q, const float* k, const float* v, float* o, float* lse, const int64_t* num_ims, int B,
                               int T, int H, int head_dim, int max_queries, void* workspace, int planes, uint64_t drop_key,
                               float drop_p, paths_stream_t stream);
/* backward twins (same key, same p as the forward) */
int paths_attention_bwd_f32_dropout(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                                    const float* lse, const int64_t* num_ims, float* dqkv, float* ws_dsum, int B, int T, int H,
                                    int head_dim, uint64_t drop_key, float drop_p, paths_stream_t stream);
/* The same with the dQ part on the split-bf16 matrix-core kernel (csrc/attn_bwd_x6.hip: three exact bf16 planes per operand, as
 * accurate as the f32 MFMA); images: paths_attention_bwd_x6_workspace(B, T, H, head_dim) bytes of scratch. */
int64_t paths_attention_bwd_x6_workspace(int B, int T, int H, int head_dim);
int paths_attention_bwd_x6_dropout(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                                   const int64_t* num_ims, float* dqkv, float* ws_dsum, void* images, int B, int T, int H, int head_dim,
                                   uint64_t drop_key, float drop_p, paths_stream_t stream);
/* The same with the operand split chosen by the caller: planes 3 = three exact bf16 planes (6 MFMAs per product block), planes 2 =
 * hi | mid only (16 significant bits at fp32's exponent range, 3 MFMAs: what the training step's gradient GEMMs use by default). */
int paths_attention_bwd_x6_planes(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                                  const int64_t* num_ims, float* dqkv, float* ws_dsum, void* images, int B, int T, int H, int head_dim,
                                  uint64_t drop_key, float drop_p, int planes, paths_stream_t stream);

/* ---- wide heads (head_dim > 64, a multiple of 32: e.g. trans_dim 256 / 2 heads, or 1536 / 4 heads = 384, the stress form of
 * BASELINE configs[4]; reference model/aggregator.py:25-33 accepts any trans_dim % trans_heads == 0).  Per (slide, head): S = q k^T
 * (f32-input MFMA GEMM), masked softmax, O = P V, and the corresponding five products backward, with the score matrix in scratch
 * (paths_attention_wide_workspace floats).  Same conventions as paths_attention_any_train / paths_attention_bwd_any; qkv must have at
 * least 128 readable rows behind its last one. */
int64_t paths_attention_wide_workspace(int T, int head_dim);
int paths_attention_wide_fwd(const float* qkv, int64_t ld, float* o, float* lse, const int64_t* num_ims, int B, int T, int H, int head_dim,
                             float qscale, int max_queries, uint64_t drop_key, float drop_p, float* workspace, paths_stream_t stream);
int paths_attention_wide_bwd(const float* qkv, int64_t ld, const float* o, const float* d_o, const float* lse, const int64_t* num_ims,
                             float* dqkv, int B, int T, int H, int head_dim, float qscale, int max_queries, uint64_t drop_key, float drop_p,
                             float* workspace, paths_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PATHS_HIP_H */
