"""Backward pass launch sequences (hand-written HIP kernels; Python only allocates and launches).

Implements what autograd does in the reference train step (reference train.py:65 ``loss.backward()``) for the
fused forward of paths_amd/ops.py.  Every product runs on the fp32 matrix cores:

    dX  = dY W        paths_gemm_nt_f32 with a transposed weight copy (paths_transpose_f32)
    dW  = dY^T X      paths_gemm_tn_f32 (split-M slabs, fixed summation order -> deterministic, rank-count independent)
    db  = colsum(dY)  paths_colsum_f32

plus the row-wise derivative kernels of csrc/bwd_rows.hip and the attention backward kernels of csrc/attn_bwd.hip.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import _lib

P = _lib.ptr


def _f32(dev):
    return dict(device=dev, dtype=torch.float32)


def _splits(M: int, n_tiles: int) -> int:
    """Split the row reduction so that about 2 x 256 workgroups are in flight, each with >= 256 rows."""
    return max(1, min(64, (512 + n_tiles - 1) // n_tiles, M // 256 if M >= 256 else 1))


def transpose(w: torch.Tensor, rows: int, cols: int, ld: Optional[int] = None, offset: int = 0) -> torch.Tensor:
    """out[cols, rows] = w[rows, offset:offset+cols]^T  (w row-major with leading dimension ld)."""
    out = torch.empty((cols, rows), **_f32(w.device))
    _lib.call("paths_transpose_f32", w.data_ptr() + 4 * offset, ld if ld is not None else w.stride(0), rows, cols,
              P(out), rows, _lib.stream())
    return out


def gemm_nt(a, lda, wt, out, ldo, M, N, K, bias=None, act=0, residual=None, ldr=0, mask=None, ldm=0, accumulate=False,
            ldw=None):
    """out[M,N] (+)= maskop(act(a[M,K] wt[N,K]^T + bias)) + residual; a/out/residual/mask may be raw pointers."""
    assert N % 128 == 0 and K % 32 == 0
    ap = a if isinstance(a, int) else a.data_ptr()
    op = out if isinstance(out, int) else out.data_ptr()
    rp = None if residual is None else (residual if isinstance(residual, int) else residual.data_ptr())
    mp = None if mask is None else (mask if isinstance(mask, int) else mask.data_ptr())
    _lib.call("paths_gemm_nt_f32", ap, lda, P(wt), ldw if ldw is not None else K, P(bias), op, ldo, M, N, N, K, act, rp, ldr,
              mp, ldm, 1 if accumulate else 0, _lib.stream())


def gemm_tn(a, lda, b0, ldb0, out, M, N1, N2, b1=None, ldb1=0, nb0=0, ldo=None, accumulate=False):
    """out[N1,N2] (+)= a[M,N1]^T [b0 | b1][M,N2]."""
    dev = out.device
    splits = _splits(M, (N1 // 128) * (N2 // 128))
    ws = torch.empty((splits * N1 * N2,), **_f32(dev))
    ap = a if isinstance(a, int) else a.data_ptr()
    b0p = b0 if isinstance(b0, int) else b0.data_ptr()
    b1p = None if b1 is None else (b1 if isinstance(b1, int) else b1.data_ptr())
    _lib.call("paths_gemm_tn_f32", ap, lda, b0p, ldb0, nb0, b1p, ldb1, P(out), ldo if ldo is not None else N2, M, N1, N2,
              splits, 1 if accumulate else 0, P(ws), _lib.stream())


def colsum(a, lda, M, N, out=None, accumulate=False):
    dev = a.device if not isinstance(a, int) else out.device
    if out is None:
        out = torch.empty((N,), **_f32(dev))
    splits = max(1, min(256, M // 64))
    ws = torch.empty((splits * N,), **_f32(dev))
    ap = a if isinstance(a, int) else a.data_ptr()
    _lib.call("paths_colsum_f32", ap, lda, M, N, P(out), splits, 1 if accumulate else 0, P(ws), _lib.stream())
    return out


# ---------------------------------------------------------------------------------------------------------------
# selection chain: LSTM cell + importance MLP / projection
# ---------------------------------------------------------------------------------------------------------------
def selection_forward_train(mc, lstm_pack, lvl_pack, fts, locs, num_ims, state_prev) -> Dict[str, torch.Tensor]:
    """Forward of the LSTM + importance/projection part with everything the backward needs kept in HBM.
    Padded rows are computed too (finite values everywhere: 0 * garbage would poison the reductions)."""
    B, N, D = fts.shape
    d = mc.trans_dim
    Hc = lstm_pack["Hc"]
    Dp = D + Hc
    M, T = B * N, N + 1
    f32 = _f32(fts.device)
    st = _lib.stream()
    sv = {"fts": fts, "state_prev": state_prev, "num_ims": num_ims, "locs": locs}
    sv["state_out"] = torch.empty((B, N, Dp), **f32)
    sv["y"] = torch.empty((B, N, D), **f32)
    sv["o"] = torch.empty((B, N, D), **f32)
    sv["frm"] = torch.empty((B, N, 3 * Hc), **f32)
    sv["tc"] = torch.empty((B, N, D), **f32)
    if state_prev is not None:
        assert state_prev.shape == (B, N, Dp) and state_prev.stride(2) == 1 and state_prev.stride(0) == N * state_prev.stride(1)
        ld, h0, c0 = state_prev.stride(1), state_prev.data_ptr(), state_prev.data_ptr() + 4 * D
    else:
        ld, h0, c0 = 0, None, None
    _lib.call("paths_lstm_cell", P(fts), D, h0, ld, c0, ld, P(lstm_pack["w_gates"]), P(lstm_pack["b_gates"]),
              P(lstm_pack["w_mem"]), P(lstm_pack["b_mem"]), P(sv["state_out"]), Dp, P(sv["y"]), D, P(sv["o"]), P(sv["frm"]),
              P(sv["tc"]), M, D, Hc, None, N, 7, st)
    sv["importance"] = torch.empty((B, N), **f32)
    sv["tokens"] = torch.empty((B, T, d), **f32)
    sv["hid"] = torch.empty((B, N, 128), **f32)
    sv["pproj"] = torch.empty((B, N, 128), **f32)
    pe_mode = 2 if mc.pos_encoding_mode == "2d" else 1
    _lib.call("paths_importance_proj", P(sv["y"]), D, P(lvl_pack["w_ip"]), P(lvl_pack["b1"]), P(lvl_pack["w2"]), lvl_pack["b2"],
              P(lvl_pack["bp"]), P(lvl_pack["special"]), P(lvl_pack["div_2d" if pe_mode == 2 else "div_1d"]), P(locs),
              P(num_ims), N, mc.patch_size, pe_mode, 1 if mc.importance_mode == "mul" else 0, P(sv["importance"]),
              P(sv["tokens"]), P(sv["hid"]), P(sv["pproj"]), M, D, mc.importance_mlp_hidden_dim, d, 0, st)
    return sv


def selection_backward(mc, lstm_pack, lvl_pack, sv, d_tokens: torch.Tensor, d_state_out: Optional[torch.Tensor]):
    """d_tokens [B,T,128]: gradient of the token sequence (row 0 = special token);
    d_state_out [B,N,D+Hc] or None: gradient flowing into (h1|c1) from the next level's gather.
    Returns (grads dict in PACKED layouts, d_state_prev [B,N,D+Hc] or None)."""
    fts, state_prev, num_ims = sv["fts"], sv["state_prev"], sv["num_ims"]
    B, N, D = fts.shape
    Hc = lstm_pack["Hc"]
    Dp, M, T = D + Hc, B * N, N + 1
    G = 3 * Hc + D
    dev = fts.device
    f32 = _f32(dev)
    st = _lib.stream()
    grads: Dict[str, torch.Tensor] = {}

    # ---- importance MLP + scaling + proj_in
    du = torch.empty((M, 256), **f32)
    da = torch.empty((M,), **f32)
    dah = torch.empty((M, 128), **f32)
    _lib.call("paths_importance_bwd", P(d_tokens), P(sv["pproj"]), P(sv["hid"]), P(sv["importance"]), P(lvl_pack["w2"]),
              P(num_ims), N, M, 1 if mc.importance_mode == "mul" else 0, P(du), P(da), P(dah), st)
    grads["w2"] = colsum(dah, 128, M, 128)
    grads["b2"] = colsum(da, 1, M, 1)
    grads["b1"] = colsum(du, 256, M, 128)
    grads["special"] = colsum(d_tokens, T * 128, B, 128)
    # proj_in.bias: sum of token gradients over the valid patch rows = colsum of dP / alpha is not usable (alpha may
    # be 0), so sum d_tokens rows 1..N directly; padded token rows carry exact zeros (masked keys, unused queries)
    dbp = torch.zeros((128,), **f32)
    for b in range(B):
        colsum(d_tokens.data_ptr() + 4 * (b * T + 1) * 128, 128, N, 128, out=dbp, accumulate=True)
    grads["bp"] = dbp
    grads["w_ip"] = torch.empty((256, D), **f32)
    gemm_tn(du, 256, sv["y"], D, grads["w_ip"], M, 256, D)
    dy = torch.empty((M, D), **f32)
    w_ip_t = transpose(lvl_pack["w_ip"], 256, D)                       # [D, 256]
    gemm_nt(du, 256, w_ip_t, dy, D, M, D, 256)

    # ---- LSTM cell.  Y = X + h1  =>  dh1 = dY (+ gradient arriving at the h half of state_out)
    dG = torch.empty((M, G), **f32)
    dpre_h = torch.empty((M, D), **f32)
    ext_h = d_state_out.data_ptr() if d_state_out is not None else None
    _lib.call("paths_lstm_bwd_a", P(dy), D, ext_h, Dp, P(sv["o"]), P(sv["tc"]), P(num_ims), N, M, D,
              dG.data_ptr() + 4 * 3 * Hc, G, P(dpre_h), st)
    grads["b_mem"] = colsum(dpre_h, D, M, D)
    grads["w_mem"] = torch.empty((D, Hc), **f32)
    c1_ptr = sv["state_out"].data_ptr() + 4 * D
    gemm_tn(dpre_h, D, c1_ptr, Dp, grads["w_mem"], M, D, Hc)
    dc1_h = torch.empty((M, Hc), **f32)
    w_mem_t = transpose(lstm_pack["w_mem"], D, Hc)                      # [Hc, D]
    gemm_nt(dpre_h, D, w_mem_t, dc1_h, Hc, M, Hc, D)
    d_state_prev = torch.empty((B, N, Dp), **f32) if state_prev is not None else None
    ext_c = d_state_out.data_ptr() + 4 * D if d_state_out is not None else None
    c0_ptr = state_prev.data_ptr() + 4 * D if state_prev is not None else None
    _lib.call("paths_lstm_bwd_b", P(dc1_h), ext_c, Dp, P(sv["frm"]), c0_ptr, state_prev.stride(1) if state_prev is not None else 0,
              P(num_ims), N, M, Hc, P(dG), G, d_state_prev.data_ptr() + 4 * D if d_state_prev is not None else None, Dp, st)
    grads["b_gates"] = colsum(dG, G, M, G)
    grads["w_gates"] = torch.zeros((G, 2 * D), **f32)
    if state_prev is not None:
        gemm_tn(dG, G, fts, D, grads["w_gates"], M, G, 2 * D, b1=state_prev.data_ptr(), ldb1=state_prev.stride(1), nb0=D)
        wh_t = transpose(lstm_pack["w_gates"], G, D, ld=2 * D, offset=D)   # [D, G] = (W_gates[:, D:2D])^T
        gemm_nt(dG, G, wh_t, d_state_prev.data_ptr(), Dp, M, D, G)
    else:
        gemm_tn(dG, G, fts, D, grads["w_gates"], M, G, D, ldo=2 * D)        # only the x panel is live at depth 0
    return grads, d_state_prev


def unpack_lstm_grads(lstm, g: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Packed gate layout -> the reference's parameter names (inverse of ops.pack_lstm)."""
    Hc = lstm.cdim
    wc = g["w_gates"][: 3 * Hc].view(Hc // 32, 3, 32, -1)
    bc = g["b_gates"][: 3 * Hc].view(Hc // 32, 3, 32)
    return {
        "forget_gate.0.weight": wc[:, 0].reshape(Hc, -1), "forget_gate.0.bias": bc[:, 0].reshape(Hc),
        "remember_gate.0.weight": wc[:, 1].reshape(Hc, -1), "remember_gate.0.bias": bc[:, 1].reshape(Hc),
        "remember_map.0.weight": wc[:, 2].reshape(Hc, -1), "remember_map.0.bias": bc[:, 2].reshape(Hc),
        "out_select_gate.0.weight": g["w_gates"][3 * Hc:], "out_select_gate.0.bias": g["b_gates"][3 * Hc:],
        "mem_to_out.0.weight": g["w_mem"], "mem_to_out.0.bias": g["b_mem"],
    }
