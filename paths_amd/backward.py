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
import os
from typing import Dict, Optional

import torch

from . import _lib, ops

P = _lib.ptr
TN_SPLIT2 = int(os.environ.get("PATHS_TN_SPLIT2", "2"))
# weight-gradient GEMMs: "x6" = split-bf16 MFMA kernel (csrc/gemm_tn_x6.hip), "f32" = the f32-MFMA kernel (csrc/gemm_bwd.hip)
TN_MODE = os.environ.get("PATHS_TN_MODE", "x6")
# dX / recompute GEMMs go to the split-operand kernel when the output width is a multiple of this (128: the d = 128 products too)
NT_X6_MIN_N = int(os.environ.get("PATHS_NT_X6_MIN_N", "128"))
PACK_T = os.environ.get("PATHS_PACK_T", "1") != "0"        # dX products image W^T straight from W (one launch instead of transpose + pack)
# Row counts from which the gradient GEMMs run on the split-bf16 kernels (below: the f32-input MFMA kernels, which are as fast there).
# Module constants so that tests can force the split kernels onto the small shapes of the reference's trajectory fixtures.
NT_X6_MIN_M = int(os.environ.get("PATHS_NT_X6_MIN_M", "1024"))
TN_X6_MIN_M = int(os.environ.get("PATHS_TN_X6_MIN_M", "512"))
# attention backward: "x6q" = the split-bf16 kernels (csrc/attn_bwd_x6.hip: dQ, and dK / dV unless PATHS_ATTN_BWD_KV_X6=0), "f32" = all of it on the f32 MFMA (csrc/attn_bwd.hip)
ATTN_BWD_MODE = os.environ.get("PATHS_ATTN_BWD_MODE", "x6q")


TRAIN_SPLITK_IMPORTANCE = os.environ.get("PATHS_TRAIN_SPLITK_IMPORTANCE", "1") != "0"     # training: importance / proj GEMM as two k halves + finish


def _f32(dev):
    return dict(device=dev, dtype=torch.float32)


def _splits(M: int, n_tiles: int) -> int:
    """Split the row reduction so that about 2 x 256 workgroups are in flight, each with >= 256 rows."""
    if n_tiles >= 192:
        # about one workgroup per CU: TN_SPLIT2 = two half-height workgroups per CU instead (they hide each other's barrier and
        # load stalls), at the price of one slab reduction
        return TN_SPLIT2 if (TN_SPLIT2 > 1 and n_tiles <= 256 and M >= 4096) else 1
    return max(1, min(64, (512 + n_tiles - 1) // n_tiles, M // 256 if M >= 256 else 1))


def transpose(w: torch.Tensor, rows: int, cols: int, ld: Optional[int] = None, offset: int = 0, pad_to: Optional[int] = None) -> torch.Tensor:
    """out[cols, rows] = w[rows, offset:offset+cols]^T  (w row-major with leading dimension ld); pad_to > rows: out is [cols, pad_to]
    with zeros behind column ``rows`` (a K dimension padded for the GEMM kernels)."""
    if pad_to is not None and pad_to > rows:
        out = torch.zeros((cols, pad_to), **_f32(w.device))
    else:
        out = torch.empty((cols, rows), **_f32(w.device))
    _lib.call("paths_transpose_f32", w.data_ptr() + 4 * offset, ld if ld is not None else w.stride(0), rows, cols,
              P(out), out.shape[1], _lib.stream())
    return out


def zeros_group(dev, *shapes):
    """Several zero-filled fp32 tensors from ONE fill launch (each starts on a 256-byte boundary of one buffer)."""
    sizes = [math.prod(int(v) for v in sh) for sh in shapes]
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += (n + 63) // 64 * 64
    buf = torch.zeros((tot,), **_f32(dev))
    return [buf[o:o + n].view(*sh) for o, n, sh in zip(offs, sizes, shapes)]


class Transposed:
    """``transpose(w, rows, cols, ld, offset, pad_to)`` not yet materialised: the weight of a dX product (dX = dY W takes W^T).  The
    split-bf16 GEMM images it straight from ``w`` (paths_x6_pack_weights_t: one launch instead of a transpose + a pack); every other
    consumer calls :meth:`tensor`."""
    __slots__ = ("w", "rows", "cols", "ld", "offset", "pad_to")

    def __init__(self, w, rows, cols, ld=None, offset=0, pad_to=None):
        self.w, self.rows, self.cols, self.ld, self.offset, self.pad_to = w, rows, cols, ld, offset, pad_to

    def tensor(self) -> torch.Tensor:
        return transpose(self.w, self.rows, self.cols, self.ld, self.offset, self.pad_to)


def gemm_nt(a, lda, wt, out, ldo, M, N, K, bias=None, act=0, residual=None, ldr=0, mask=None, ldm=0, accumulate=False,
            ldw=None):
    """out[M,N] (+)= maskop(act(a[M,K] wt[N,K]^T + bias)) + residual; a/out/residual/mask may be raw pointers."""
    assert K % 32 == 0
    if isinstance(wt, Transposed):
        kfull = wt.pad_to if (wt.pad_to is not None and wt.pad_to > wt.rows) else wt.rows
        if (ops.GEMM_MODE != "f32" and N % NT_X6_MIN_N == 0 and K >= 128 and K % 128 == 0 and M >= NT_X6_MIN_M and ldw is None and N == wt.cols and K == kfull
                and ops.TRAIN_PLANES in (3, 4) and PACK_T):
            pl = ops.TRAIN_PLANES
            wx = torch.empty((N * K * 2 * (2 if pl == 4 else pl),), device=wt.w.device, dtype=torch.uint8)
            _lib.call("paths_x6_pack_weights_t", wt.w.data_ptr() + 4 * wt.offset, wt.ld if wt.ld is not None else wt.w.stride(0), P(wx), N, N, K, wt.rows,
                      pl, _lib.stream())
            ap = a if isinstance(a, int) else a.data_ptr()
            op = out if isinstance(out, int) else out.data_ptr()
            rp = None if residual is None else (residual if isinstance(residual, int) else residual.data_ptr())
            mp = None if mask is None else (mask if isinstance(mask, int) else mask.data_ptr())
            _lib.call("paths_gemm_nt_x6", ap, lda, P(wx), K, 0, P(bias), op, ldo, M, N, N, K, act, rp, ldr, mp, ldm,
                      1 if accumulate else 0, pl, 1.0, 1.0, _lib.stream())
            return
        wt = wt.tensor()
    ap = a if isinstance(a, int) else a.data_ptr()
    op = out if isinstance(out, int) else out.data_ptr()
    rp = None if residual is None else (residual if isinstance(residual, int) else residual.data_ptr())
    mp = None if mask is None else (mask if isinstance(mask, int) else mask.data_ptr())
    ldw = ldw if ldw is not None else K
    if ops.GEMM_MODE != "f32" and N % NT_X6_MIN_N == 0 and K >= 128 and K % 128 == 0 and M >= NT_X6_MIN_M and isinstance(wt, torch.Tensor) and wt.dim() == 2 \
            and wt.shape[0] >= N and wt.stride(0) == ldw and wt.stride(1) == 1:
        # split-bf16 GEMM: the (transposed) weight is re-imaged per call - 6 N K bytes, microseconds next to an M >= 1024 product
        wx, wx_s = ops.x6_pack(wt[:N, :K], planes=ops.TRAIN_PLANES)
        _lib.call("paths_gemm_nt_x6", ap, lda, P(wx), K, 0, P(bias), op, ldo, M, N, N, K, act, rp, ldr, mp, ldm,
                  1 if accumulate else 0, ops.TRAIN_PLANES, wx_s, 1.0, _lib.stream())
        return
    n_pad = N
    if N % 128:                        # the f32 GEMM reads whole 128-row weight tiles: zero rows behind the last output feature
        assert isinstance(wt, torch.Tensor) and wt.stride(1) == 1 and wt.stride(0) == ldw
        n_pad = (N + 127) // 128 * 128
        if wt.shape[0] < n_pad:
            wt = ops._pad_rows(wt[:N])
    _lib.call("paths_gemm_nt_f32", ap, lda, P(wt), ldw, P(bias), op, ldo, M, N, n_pad, K, act, rp, ldr,
              mp, ldm, 1 if accumulate else 0, _lib.stream())


_TN_WGS = int(os.environ.get("PATHS_TN_WGS", "256"))      # workgroups of 256 x 256 tiles a weight-gradient launch aims at: ONE round of the 256 CUs
                                                           # (two rounds = twice the split-M slabs to write and to sum: 13.83 / 14.68 ms per step against 13.65 / 14.45)


def _splits_x6(M: int, N1: int, N2: int, nb0: int) -> int:
    """Row splits of the split-bf16 weight-gradient kernel: 256 x 256 tiles run one workgroup per CU, 128 x 128 tiles two; aim at
    one full round of the 256 CUs (_TN_WGS) with at least 128 rows (8 stages) per split."""
    big = N1 % 256 == 0 and N2 % 256 == 0 and nb0 % 256 == 0
    tiles = (N1 // 256) * (N2 // 256) if big else (N1 // 128) * (N2 // 128)
    return max(1, min(64, (_TN_WGS if big else 2 * _TN_WGS) // tiles, M // 128))


# ---- the gradient side stream.  In a level's backward the dX chain is the critical path (each product feeds the next); the weight
# / bias gradients (dW = dY^T X, column sums) only CONSUME what it produces.  They run on a second HIP stream: the many short
# launches of the chain (LayerNorm, dropout masks, column sums: 3-10 us each, a few workgroups) then share the chip with the large
# weight-gradient GEMMs instead of queueing behind them.  Same kernels, same arguments, same order within each stream: results are
# bit-identical to the single-stream schedule.  MEASURED (round 4, K = 2048, 8 slides) and therefore OFF by default
# (PATHS_BWD_SIDE=1 enables it): the device side of a step gets shorter (the host no longer waits for it: drain 2.7 -> 0.06 ms), but
# the ~35 fork points per step (event record + stream wait + record_stream + torch's stream context) cost 3.2 ms of HOST time
# (backward enqueue 8.2 -> 11.4 ms) and the step becomes host-bound: 15.9 -> 17.0-17.9 ms.  It pays once the backward is replayed
# from a launch tape (no Python per launch).
BWD_SIDE = os.environ.get("PATHS_BWD_SIDE", "0") != "0"
_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


class side_stream:
    """``with side_stream(dev, t1, t2, ...):`` - the launches inside go to the device's gradient side stream, ordered after
    everything enqueued on the current stream so far; the tensors named are read there (their blocks are not re-used before that
    work is done: ``record_stream``).  :func:`side_join` orders the current stream behind the side stream again - the backward
    functions call it before they return, i.e. before their raw-pointer operands (saved activations) can be freed."""

    def __init__(self, dev, *reads):
        self.on = BWD_SIDE and dev.type == "cuda" and _lib.TAPE is None
        self.dev, self.reads = dev, reads

    def __enter__(self):
        if not self.on:
            return self
        idx = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        side = _SIDE_STREAMS.get(idx)
        if side is None:
            side = _SIDE_STREAMS[idx] = torch.cuda.Stream(device=idx)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        for t in self.reads:
            if isinstance(t, torch.Tensor):
                t.record_stream(side)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False


def side_join(dev):
    if BWD_SIDE and dev.type == "cuda":
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        side = _SIDE_STREAMS.get(idx)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)


# Deferred slab reductions (csrc/reduce_multi.hip): inside ``with deferred_reductions():`` the "out (+)= sum of slabs" pass that ends
# every weight / bias / LayerNorm-affine gradient is registered instead of launched and all of them run in one launch per 32 at the
# exit (~170 launches of 4-10 us per training step -> ~10).  Outputs are bit-identical.  Rules inside the block: nothing may read or
# modify a gradient produced by gemm_tn / colsum / _ln_bwd_sums with a torch op (use after_reductions(fn) for a touch-up such as a
# scale, or flush_reductions() first); the slab workspaces are kept alive here until the flush.
DEFER_REDUCTIONS = os.environ.get("PATHS_DEFER_REDUCTIONS", "1") != "0"
_DEFER = {"depth": 0, "keep": [], "post": []}


def flush_reductions():
    if _DEFER["depth"] > 0:
        _lib.call("paths_flush_reductions", None, _lib.stream())
        _DEFER["keep"].clear()
        post, _DEFER["post"] = _DEFER["post"], []
        for fn in post:
            fn()


def after_reductions(fn):
    """Run ``fn()`` (a torch op on gradients still waiting for their reduction) once they have been reduced: now when nothing is deferred."""
    if _DEFER["depth"] > 0:
        _DEFER["post"].append(fn)
    else:
        fn()


def _keep_slabs(ws):
    if _DEFER["depth"] > 0:
        _DEFER["keep"].append(ws)


class deferred_reductions:
    def __enter__(self):
        self.on = DEFER_REDUCTIONS and not BWD_SIDE and _lib.TAPE is None
        if self.on:
            if _DEFER["depth"] == 0:
                _lib.load().paths_defer_reductions(1)
            _DEFER["depth"] += 1
        return self

    def __exit__(self, *exc):
        if self.on:
            try:
                if _DEFER["depth"] == 1:
                    flush_reductions()                  # (also on an exception: the registered entries point into workspaces freed below)
            finally:
                _DEFER["depth"] -= 1
                if _DEFER["depth"] == 0:
                    _lib.load().paths_defer_reductions(0)
                    _DEFER["keep"].clear()
                    _DEFER["post"].clear()
        return False


def gemm_tn(a, lda, b0, ldb0, out, M, N1, N2, b1=None, ldb1=0, nb0=0, ldo=None, accumulate=False):
    """out[N1,N2] (+)= a[M,N1]^T [b0 | b1][M,N2]."""
    dev = out.device
    if TN_MODE == "x6" and M >= TN_X6_MIN_M and N1 % 128 == 0 and N2 % 128 == 0 and M * 4 * max(lda, ldb0, ldb1) < (1 << 31):
        splits = _splits_x6(M, N1, N2, nb0 if b1 is not None else 0)
        ws = torch.empty((splits * N1 * N2,), **_f32(dev))
        ap = a if isinstance(a, int) else a.data_ptr()
        b0p = b0 if isinstance(b0, int) else b0.data_ptr()
        b1p = None if b1 is None else (b1 if isinstance(b1, int) else b1.data_ptr())
        _lib.call("paths_gemm_tn_x6", ap, lda, b0p, ldb0, nb0, b1p, ldb1, P(out), ldo if ldo is not None else N2, M, N1, N2,
                  splits, 1 if accumulate else 0, P(ws), 2 if ops.TRAIN_PLANES == 4 else 3, _lib.stream())
        _keep_slabs(ws)
        return
    splits = _splits(M, ((N1 + 127) // 128) * ((N2 + 127) // 128))
    ws = torch.empty((splits * N1 * N2,), **_f32(dev))
    ap = a if isinstance(a, int) else a.data_ptr()
    b0p = b0 if isinstance(b0, int) else b0.data_ptr()
    b1p = None if b1 is None else (b1 if isinstance(b1, int) else b1.data_ptr())
    _lib.call("paths_gemm_tn_f32", ap, lda, b0p, ldb0, nb0, b1p, ldb1, P(out), ldo if ldo is not None else N2, M, N1, N2,
              splits, 1 if accumulate else 0, P(ws), _lib.stream())
    _keep_slabs(ws)


def colsum(a, lda, M, N, out=None, accumulate=False):
    dev = a.device if not isinstance(a, int) else out.device
    if out is None:
        out = torch.empty((N,), **_f32(dev))
    splits = max(1, min(256, M // 64))
    ws = torch.empty((splits * N,), **_f32(dev))
    ap = a if isinstance(a, int) else a.data_ptr()
    _lib.call("paths_colsum_f32", ap, lda, M, N, P(out), splits, 1 if accumulate else 0, P(ws), _lib.stream())
    _keep_slabs(ws)
    return out


# ---------------------------------------------------------------------------------------------------------------
# selection chain: LSTM cell + importance MLP / projection
# ---------------------------------------------------------------------------------------------------------------
def selection_forward_train(mc, lstm_pack, lvl_pack, fts, locs, num_ims, state_prev, parent=None) -> Dict[str, torch.Tensor]:
    """Forward of the LSTM + importance/projection part with everything the backward needs kept in HBM.
    Padded rows are computed too (finite values everywhere: 0 * garbage would poison the reductions).
    ``parent`` (the device recursion's once-per-parent form, ``state_prev`` None then) = {"c0": [B,N,Hc] the children's inherited memory
    cell, "h_kept": [B*cap, D] the kept parents' h rows, "hp_row": [B,N] int32 child -> kept slot (-1: no parent)}: siblings share
    their parent's h, so the h half of the gate pre-activations is ONE product over the kept parents (a quarter of the rows) that
    seeds the children's accumulators (reference model/interface.py:49-56 with cat(x, h) split in two panels)."""
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
    G = 3 * Hc + D
    x6 = ops.use_x6(D, Hc)        # forward GEMMs on the split-bf16 path (weight images are re-packed when the optimizer steps)
    hp, hp_row = None, None
    if parent is not None:
        assert state_prev is None
        c0t, hk = parent["c0"], parent["h_kept"]
        assert c0t.shape == (B, N, Hc) and c0t.is_contiguous() and hk.shape[1] == D and hk.is_contiguous()
        ld, h0, c0 = Hc, None, c0t.data_ptr()
        M4 = hk.shape[0]
        sv["parent"] = {"c0": c0t, "h_kept": hk, "hp_row": parent["hp_row"], "child_pos": parent["child_pos"],
                        "keep_count": parent["keep_count"], "cap": parent["cap"]}
        hpt = torch.empty((M4, G), **f32)              # HP = h_kept W_gates[:, D:2D]^T (no bias), packed gate-column order
        if x6 and G % 256 == 0:
            TP = ops.TRAIN_FWD_PLANES
            wg, wg_s = ops._x6_of(lstm_pack, "w_gates", TP, lagged=True)
            _lib.call("paths_gemm_nt_x6", P(hk), D, P(wg), 2 * D, D, None, P(hpt), G, M4, G, G, D, 0, None, 0, None, 0, 0, TP, wg_s,
                      ops.A_SCALE if TP == 2 else 1.0, st)
        else:
            _lib.call("paths_gemm_nt_f32", P(hk), D, lstm_pack["w_gates"].data_ptr() + 4 * D, 2 * D, None, P(hpt), G, M4, G, G, D, 0,
                      None, 0, None, 0, 0, st)
        hp, hp_row = P(hpt), P(parent["hp_row"])
    elif state_prev is not None:
        assert state_prev.shape == (B, N, Dp) and state_prev.stride(2) == 1 and state_prev.stride(0) == N * state_prev.stride(1)
        ld, h0, c0 = state_prev.stride(1), state_prev.data_ptr(), state_prev.data_ptr() + 4 * D
    else:
        ld, h0, c0 = 0, None, None
    if x6:
        TP = ops.TRAIN_FWD_PLANES
        asc = ops.A_SCALE if TP == 2 else 1.0
        (wg, wg_s), (wm, wm_s) = ops._x6_of(lstm_pack, "w_gates", TP, lagged=True), ops._x6_of(lstm_pack, "w_mem", TP, lagged=True)
        _lib.call("paths_lstm_cell_x6", P(fts), D, None, h0, ld, c0, ld, P(wg), P(lstm_pack["b_gates"]), P(wm), P(lstm_pack["b_mem"]),
                  P(sv["state_out"]), Dp, P(sv["y"]), D, P(sv["o"]), P(sv["frm"]), P(sv["tc"]), hp, hp_row, M, D, Hc, None, N, 7,
                  TP, wg_s, wm_s, asc, st)
    else:
        _lib.call("paths_lstm_cell", P(fts), D, h0, ld, c0, ld, P(lstm_pack["w_gates"]), P(lstm_pack["b_gates"]),
                  P(lstm_pack["w_mem"]), P(lstm_pack["b_mem"]), P(sv["state_out"]), Dp,
                  P(sv["y"]), D, P(sv["o"]), P(sv["frm"]), P(sv["tc"]), hp, hp_row, M, D, Hc, None, N, 7, st)
    sv["importance"] = torch.empty((B, N), **f32)
    sv["tokens"] = torch.empty((B, T, d), **f32)
    Hi = mc.importance_mlp_hidden_dim
    sv["hid"] = torch.empty((B, N, Hi), **f32)
    sv["pproj"] = torch.empty((B, N, d), **f32)
    pe_mode = 2 if mc.pos_encoding_mode == "2d" else 1
    if not ops.fast_path(mc):
        # any (trans_dim, hidden) widths: two generic GEMMs + the row kernels of csrc/generic.hip, keeping hid = relu(Y W1^T + b1)
        # and P = Y Wp^T for the backward (reference model/paths.py:95-98,119-124; model/aggregator.py:37-65)
        gp = ops.generic_pack(lvl_pack, mc)
        ops.gemm_f32(sv["y"], D, gp["w1"], lvl_pack["b1"], sv["hid"], Hi, M, Hi, D, act=1)
        _lib.call("paths_importance_rows", P(sv["hid"]), Hi, P(lvl_pack["w2"]), P(lvl_pack["b2"]), P(num_ims), N, M, Hi, P(sv["importance"]), 0, st)
        ops.gemm_f32(sv["y"], D, gp["wp"], None, sv["pproj"], d, M, d, D)
        _lib.call("paths_tokens_assemble", P(sv["pproj"]), d, P(sv["importance"]), 1 if mc.importance_mode == "mul" else 0, P(lvl_pack["bp"]),
                  P(lvl_pack["special"]), P(lvl_pack["div_2d" if pe_mode == 2 else "div_1d"]), P(locs), N, mc.patch_size, pe_mode, d, B,
                  P(sv["tokens"]), st)
        return sv
    tail = (P(lvl_pack["b1"]), P(lvl_pack["w2"]), P(lvl_pack["b2"]),
            P(lvl_pack["bp"]), P(lvl_pack["special"]), P(lvl_pack["div_2d" if pe_mode == 2 else "div_1d"]), None, 0, P(locs),
            P(num_ims), N, mc.patch_size, pe_mode, 1 if mc.importance_mode == "mul" else 0, P(sv["importance"]),
            P(sv["tokens"]), P(sv["hid"]), P(sv["pproj"]), M, D, mc.importance_mlp_hidden_dim, d, 0, st)
    if x6:
        TP = ops.TRAIN_FWD_PLANES
        wip, wip_s = ops._x6_of(lvl_pack, "w_ip_fwd", TP, lagged=True)
        # M / 128 blocks fill half the chip: two k halves on twice the blocks + the epilogue launch, as in inference (round 5)
        splitk_ws = None
        if TRAIN_SPLITK_IMPORTANCE and TP == 2 and (M + 127) // 128 <= 160:
            splitk_ws = torch.empty((int(_lib.load().paths_importance_proj_x6_workspace(M)),), device=fts.device, dtype=torch.uint8)
        _lib.call("paths_importance_proj_x6", P(sv["y"]), D, None, None, 0, P(wip), *tail[:-1], TP, wip_s, ops.A_SCALE if TP == 2 else 1.0,
                  P(splitk_ws), tail[-1])
    else:
        _lib.call("paths_importance_proj", P(sv["y"]), D, P(lvl_pack["w_ip_fwd"]), *tail)
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
    d, Hi = mc.trans_dim, mc.importance_mlp_hidden_dim
    U = (Hi + d + 31) // 32 * 32                     # dU = [dhid (Hi) | dP (d) | zero pad]: the K dimension of dY = dU W_ip
    du = torch.empty((M, U), **f32)
    da = torch.empty((M,), **f32)
    dah = torch.empty((M, Hi), **f32)
    if ops.fast_path(mc):
        _lib.call("paths_importance_bwd", P(d_tokens), P(sv["pproj"]), P(sv["hid"]), P(sv["importance"]), P(lvl_pack["w2"]),
                  P(num_ims), N, M, 1 if mc.importance_mode == "mul" else 0, P(du), P(da), P(dah), st)
    else:
        _lib.call("paths_importance_bwd_any", P(d_tokens), P(sv["pproj"]), P(sv["hid"]), P(sv["importance"]), P(lvl_pack["w2"]),
                  P(num_ims), N, M, 1 if mc.importance_mode == "mul" else 0, Hi, d, U, P(du), P(da), P(dah), st)
    with side_stream(dev, dah, da, du, d_tokens):
        grads["w2"] = colsum(dah, Hi, M, Hi)
        grads["b2"] = colsum(da, 1, M, 1)
        grads["b1"] = colsum(du, U, M, Hi)
        grads["special"] = colsum(d_tokens, T * d, B, d)
        # proj_in.bias: sum of token gradients over the valid patch rows = colsum of dP / alpha is not usable (alpha may
        # be 0), so sum d_tokens rows 1..N directly; padded token rows carry exact zeros (masked keys, unused queries)
        # = (sum over all B*T token rows) - (sum over the B special-token rows): two launches instead of 2 B
        grads["bp"] = bp_all = colsum(d_tokens, d, B * T, d)
        after_reductions(lambda a=bp_all, b=grads["special"]: a.sub_(b))
        grads["w_ip"] = torch.empty((Hi + d, D), **f32)
        gemm_tn(du, U, sv["y"], D, grads["w_ip"], M, Hi + d, D)
    dy = torch.empty((M, D), **f32)
    w_ip_t = Transposed(lvl_pack["w_ip"], Hi + d, D, pad_to=U)          # [D, U]
    gemm_nt(du, U, w_ip_t, dy, D, M, D, U)

    # ---- LSTM cell.  Y = X + h1  =>  dh1 = dY (+ gradient arriving at the h half of state_out)
    dG = torch.empty((M, G), **f32)
    dpre_h = torch.empty((M, D), **f32)
    ext_h = d_state_out.data_ptr() if d_state_out is not None else None
    _lib.call("paths_lstm_bwd_a", P(dy), D, ext_h, Dp, P(sv["o"]), P(sv["tc"]), P(num_ims), N, M, D,
              dG.data_ptr() + 4 * 3 * Hc, G, P(dpre_h), st)
    c1_ptr = sv["state_out"].data_ptr() + 4 * D
    with side_stream(dev, dpre_h):
        grads["b_mem"] = colsum(dpre_h, D, M, D)
        grads["w_mem"] = torch.empty((D, Hc), **f32)
        gemm_tn(dpre_h, D, c1_ptr, Dp, grads["w_mem"], M, D, Hc)
    dc1_h = torch.empty((M, Hc), **f32)
    w_mem_t = Transposed(lstm_pack["w_mem"], D, Hc)                      # [Hc, D]
    gemm_nt(dpre_h, D, w_mem_t, dc1_h, Hc, M, Hc, D)
    par = sv.get("parent")
    d_state_prev = torch.empty((B, N, Dp), **f32) if state_prev is not None else None
    ext_c = d_state_out.data_ptr() + 4 * D if d_state_out is not None else None
    if par is not None:                 # once-per-parent form: the children inherited c0 [B,N,Hc]; its gradient goes back as such
        d_c0 = torch.empty((B, N, Hc), **f32)
        _lib.call("paths_lstm_bwd_b", P(dc1_h), ext_c, Dp, P(sv["frm"]), P(par["c0"]), Hc, P(num_ims), N, M, Hc, P(dG), G, P(d_c0), Hc, st)
    else:
        c0_ptr = state_prev.data_ptr() + 4 * D if state_prev is not None else None
        _lib.call("paths_lstm_bwd_b", P(dc1_h), ext_c, Dp, P(sv["frm"]), c0_ptr, state_prev.stride(1) if state_prev is not None else 0,
                  P(num_ims), N, M, Hc, P(dG), G, d_state_prev.data_ptr() + 4 * D if d_state_prev is not None else None, Dp, st)
    if par is not None:
        # dHP[b, i] = sum of dG over the surviving children of kept parent i (the pre-activations got HP[parent] added): then the h
        # half of the weight gradient and the parents' h gradient are products over the kept parents - a quarter of the rows
        cap, hk = par["cap"], par["h_kept"]
        M4 = hk.shape[0]
        dhp = torch.zeros((M4, G), **f32)
        _lib.call("paths_sibling_sum", None, cap, P(par["keep_count"]), P(par["child_pos"]), P(dG), N, G, G, P(dhp), cap, G, B, st)
        grads["b_gates"] = colsum(dG, G, M, G)
        grads["w_gates"] = torch.empty((G, 2 * D), **f32)
        gemm_tn(dG, G, fts, D, grads["w_gates"], M, G, D, ldo=2 * D)                       # x panel: over the children
        gemm_tn(dhp, G, hk, D, grads["w_gates"][:, D:], M4, G, D, ldo=2 * D)               # h panel: over the kept parents
        wh_t = Transposed(lstm_pack["w_gates"], G, D, ld=2 * D, offset=D)                  # [D, G] = (W_gates[:, D:2D])^T
        d_hk = torch.empty((M4, D), **f32)
        gemm_nt(dhp, G, wh_t, d_hk, D, M4, D, G)
        side_join(dev)
        return grads, (d_c0, d_hk)
    with side_stream(dev, dG, fts):
        grads["b_gates"] = colsum(dG, G, M, G)
        grads["w_gates"] = torch.empty((G, 2 * D), **f32)
        if state_prev is None:
            grads["w_gates"][:, D:].zero_()                                      # the h panel is dead at depth 0
            gemm_tn(dG, G, fts, D, grads["w_gates"], M, G, D, ldo=2 * D)        # only the x panel is live at depth 0
        else:
            gemm_tn(dG, G, fts, D, grads["w_gates"], M, G, 2 * D, b1=state_prev.data_ptr(), ldb1=state_prev.stride(1), nb0=D)
    if state_prev is not None:
        wh_t = Transposed(lstm_pack["w_gates"], G, D, ld=2 * D, offset=D)   # [D, G] = (W_gates[:, D:2D])^T
        gemm_nt(dG, G, wh_t, d_state_prev.data_ptr(), Dp, M, D, G)
    side_join(dev)                     # (before the saved activations behind the raw pointers above can be freed)
    return grads, d_state_prev


# ---------------------------------------------------------------------------------------------------------------
# selection chain of the lstm = false variant (reference model/paths.py:95-109: RNN hierarchical context instead of the LSTM)
#   alpha = valid * sigmoid(importance_mlp(X)) ; Z = alpha * X (+ hctx_mlp(previous Z) on valid rows) ; patch ctx = Z ;
#   tokens = proj_in(Z) + PE.   Re-uses the generic f32-MFMA kernels; not a tuned path.
# ---------------------------------------------------------------------------------------------------------------
def selection_forward_train_nolstm(mc, lvl_pack, fts, locs, num_ims, state_prev) -> Dict[str, torch.Tensor]:
    B, N, D = fts.shape
    d, Hi = mc.trans_dim, mc.importance_mlp_hidden_dim
    M, T = B * N, N + 1
    f32 = _f32(fts.device)
    st = _lib.stream()
    sv = {"fts": fts, "state_prev": state_prev, "num_ims": num_ims, "locs": locs}
    sv["importance"] = torch.empty((B, N), **f32)
    sv["tokens"] = torch.empty((B, T, d), **f32)
    sv["hid"] = torch.empty((B, N, Hi), **f32)
    scratch_p = torch.empty((B, N, d), **f32)
    pe_mode = 2 if mc.pos_encoding_mode == "2d" else 1

    def imp_proj(src, imp_out, hid_out, pproj_out):
        if not ops.fast_path(mc):      # any widths: generic GEMMs + row kernels (tokens = P + bp + PE: src is already scaled)
            gp = ops.generic_pack(lvl_pack, mc)
            ops.gemm_f32(src, D, gp["w1"], lvl_pack["b1"], hid_out, Hi, M, Hi, D, act=1)
            _lib.call("paths_importance_rows", P(hid_out), Hi, P(lvl_pack["w2"]), P(lvl_pack["b2"]), P(num_ims), N, M, Hi, P(imp_out), 0, st)
            ops.gemm_f32(src, D, gp["wp"], None, pproj_out, d, M, d, D)
            _lib.call("paths_tokens_assemble", P(pproj_out), d, P(imp_out), 0, P(lvl_pack["bp"]), P(lvl_pack["special"]),
                      P(lvl_pack["div_2d" if pe_mode == 2 else "div_1d"]), P(locs), N, mc.patch_size, pe_mode, d, B, P(sv["tokens"]), st)
            return
        _lib.call("paths_importance_proj", P(src), D, P(lvl_pack["w_ip_fwd"]), P(lvl_pack["b1"]), P(lvl_pack["w2"]), P(lvl_pack["b2"]),
                  P(lvl_pack["bp"]), P(lvl_pack["special"]), P(lvl_pack["div_2d" if pe_mode == 2 else "div_1d"]), None, 0, P(locs),
                  P(num_ims), N, mc.patch_size, pe_mode, 0, P(imp_out), P(sv["tokens"]), P(hid_out), P(pproj_out), M, D, Hi, d, 0, st)

    imp_proj(fts, sv["importance"], sv["hid"], scratch_p)                    # pass 1: alpha and the importance MLP's hidden layer
    hctx = None
    if state_prev is not None and mc.hierarchical_ctx:
        assert state_prev.shape == (B, N, D) and state_prev.is_contiguous()
        Hh = lvl_pack["wh1"].shape[0]
        assert Hh % 128 == 0, "hierarchical_ctx_mlp_hidden_dim must be a multiple of 128 for lstm=false"
        sv["hid_h"] = torch.empty((B, N, Hh), **f32)
        hctx = torch.empty((B, N, D), **f32)
        _lib.call("paths_linear_f32", P(state_prev), D, P(lvl_pack["wh1"]), P(lvl_pack["bh1"]), P(sv["hid_h"]), Hh, M, Hh, Hh, D, 1, st)
        _lib.call("paths_linear_f32", P(sv["hid_h"]), Hh, P(lvl_pack["wh2"]), P(lvl_pack["bh2"]), P(hctx), D, M, D, D, Hh, 0, st)
    sv["has_hctx"] = hctx is not None
    sv["state_out"] = torch.empty((B, N, D), **f32)
    _lib.call("paths_scale_add_rows", P(fts), P(sv["importance"]), P(hctx), P(num_ims), N, D, M,
              1 if mc.importance_mode == "mul" else 0, P(sv["state_out"]), st)
    scratch_i, scratch_h = torch.empty((B, N), **f32), torch.empty((B, N, Hi), **f32)
    imp_proj(sv["state_out"], scratch_i, scratch_h, scratch_p)               # pass 2: tokens = proj_in(Z) + PE
    return sv


def selection_backward_nolstm(mc, lvl_pack, sv, d_tokens: torch.Tensor, d_state_out: Optional[torch.Tensor]):
    """Returns (grads {"w1","b1","w2","b2","wp","bp","special","wh1","bh1","wh2","bh2"} (None where unused), d_state_prev [B,N,D] or None)."""
    fts, state_prev, num_ims = sv["fts"], sv["state_prev"], sv["num_ims"]
    B, N, D = fts.shape
    M, T = B * N, N + 1
    d, Hi = mc.trans_dim, mc.importance_mlp_hidden_dim
    dev = fts.device
    f32 = _f32(dev)
    st = _lib.stream()
    g: Dict[str, Optional[torch.Tensor]] = {k: None for k in ("w1", "b1", "w2", "b2", "wh1", "bh1", "wh2", "bh2")}
    Z = sv["state_out"]
    # tokens[:, 1:] = Z Wp^T + bp + PE on valid rows (padded token rows carry exact zero gradients: masked keys, unused queries)
    dpp = d_tokens[:, 1:, :].contiguous().view(M, d)
    g["special"] = colsum(d_tokens, T * d, B, d)
    g["bp"] = colsum(dpp, d, M, d)
    wp = lvl_pack["w_ip"][Hi:]                                           # [d, D] proj_in.weight
    g["wp"] = torch.empty((d, D), **f32)
    gemm_tn(dpp, d, Z, D, g["wp"], M, d, D)
    dZ = torch.empty((M, D), **f32)
    gemm_nt(dpp, d, Transposed(wp, d, D), dZ, D, M, D, d, residual=d_state_out, ldr=D)
    d_state_prev = None
    if sv["has_hctx"]:
        # hctx = Wh2 relu(Wh1 s + bh1) + bh2 added on valid rows; dZ is exactly zero on padded rows (no token, never kept)
        Hh = lvl_pack["wh1"].shape[0]
        hid_h = sv["hid_h"]
        g["bh2"] = colsum(dZ, D, M, D)
        g["wh2"] = torch.empty((D, Hh), **f32)
        gemm_tn(dZ, D, hid_h, Hh, g["wh2"], M, D, Hh)
        dhh = torch.empty((M, Hh), **f32)
        gemm_nt(dZ, D, Transposed(lvl_pack["wh2"], D, Hh), dhh, Hh, M, Hh, D, mask=hid_h, ldm=Hh)
        g["bh1"] = colsum(dhh, Hh, M, Hh)
        g["wh1"] = torch.empty((Hh, D), **f32)
        gemm_tn(dhh, Hh, state_prev, D, g["wh1"], M, Hh, D)
        d_state_prev = torch.empty((B, N, D), **f32)
        gemm_nt(dhh, Hh, Transposed(lvl_pack["wh1"], Hh, D), d_state_prev, D, M, D, Hh)
    if mc.importance_mode == "mul":
        dh = torch.empty((M, Hi), **f32)
        da = torch.empty((M,), **f32)
        dah = torch.empty((M, Hi), **f32)
        if ops.fast_path(mc):
            _lib.call("paths_importance_rows_bwd", P(dZ), P(fts), D, P(sv["hid"]), P(sv["importance"]), P(lvl_pack["w2"]), P(num_ims), N, M,
                      P(dh), P(da), P(dah), st)
        else:
            _lib.call("paths_importance_rows_bwd_any", P(dZ), P(fts), D, P(sv["hid"]), P(sv["importance"]), P(lvl_pack["w2"]), P(num_ims), N, M,
                      Hi, P(dh), P(da), P(dah), st)
        g["w2"] = colsum(dah, Hi, M, Hi)
        g["b2"] = colsum(da, 1, M, 1)
        g["b1"] = colsum(dh, Hi, M, Hi)
        g["w1"] = torch.empty((Hi, D), **f32)
        gemm_tn(dh, Hi, fts, D, g["w1"], M, Hi, D)
    return g, d_state_prev


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


# ---------------------------------------------------------------------------------------------------------------
# transformer aggregator (post-LN decoder stack over an empty memory; last layer evaluated at token 0 only)
# ---------------------------------------------------------------------------------------------------------------
LOG2E = 1.4426950408889634


class Drop:
    """Dropout of one level's transformer in one training step (reference nn.Transformer(..., dropout=p): five sites per decoder
    layer).  ``seed`` is drawn once per level forward from the device's default generator (paths_amd/autograd.py:next_dropout_seed:
    ``torch.manual_seed`` controls it, the CPU generator that orders the epochs is untouched);
    every (layer, site) gets its own 64-bit key; masks are regenerated from (key, element index) wherever they are needed
    (csrc/dropout.h) - forward, the backward's recompute and the gradient masking all see the same mask."""
    ATTN, SA_OUT, CA_OUT, FF_INNER, FF_OUT = range(5)

    def __init__(self, p: float, seed: int, depth: int = 0):
        self.p, self.seed, self.depth = float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(depth)

    def key(self, layer: int, site: int) -> int:
        z = (self.seed + 0x9E3779B97F4A7C15 * (1 + site + 8 * layer + 1024 * self.depth)) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF          # splitmix64 finaliser
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)


def dropout_rows(x, ldx, M, N, key, p, out=None, ldo=None, resid=None, ldr=0, vec=None):
    """out = (resid) + (vec broadcast | x) * mask / (1 - p)   (paths_dropout_rows); x / resid / out may be raw pointers."""
    ptr = lambda t: None if t is None else (t if isinstance(t, int) else t.data_ptr())
    dev = next(t for t in (out, x, resid, vec) if t is not None and not isinstance(t, int)).device
    if out is None:
        out = torch.empty((M, N), **_f32(dev))
        ldo = N
    _lib.call("paths_dropout_rows", ptr(x) if vec is None else None, ldx, ptr(vec), ptr(resid), ldr, ptr(out), ldo if ldo is not None else N,
              M, N, key, p, _lib.stream())
    return out


def _ln_fwd(x, add, g, b, rows, eps, want_y=True, d=128):
    f32 = _f32(x.device)
    y = torch.empty((rows, d), **f32) if want_y else None
    xh = torch.empty((rows, d), **f32)
    rs = torch.empty((rows,), **f32)
    _lib.call("paths_layernorm_fwd_stats" if d == 128 else "paths_layernorm_fwd_stats_any", P(x), P(add), P(g), P(b), P(y), P(xh), P(rs),
              rows, d, eps, _lib.stream())
    return y, xh, rs


def _ln_bwd(dy, xh, rs, g, rows, d=128):
    f32 = _f32(dy.device)
    dx = torch.empty((rows, d), **f32)
    dyx = torch.empty((rows, d), **f32)
    _lib.call("paths_layernorm_bwd" if d == 128 else "paths_layernorm_bwd_any", P(dy), P(xh), P(rs), P(g), P(dx), P(dyx), rows, d, _lib.stream())
    return dx, dyx


def _ln_bwd_sums(dy, xh, rs, g, rows, d=128):
    """LayerNorm backward + affine gradients + column sums of dx in two launches: (dx, dgamma, dbeta, colsum(dx))."""
    f32 = _f32(dy.device)
    dx = torch.empty((rows, d), **f32)
    rpb = 64
    nblk = (rows + rpb - 1) // rpb
    slabs = torch.empty((nblk, 3 * d), **f32)
    _lib.call("paths_layernorm_bwd_sums" if d == 128 else "paths_layernorm_bwd_sums_any", P(dy), P(xh), P(rs), P(g), P(dx), P(slabs), rows, d, rpb,
              _lib.stream())
    gb = torch.empty((3 * d,), **f32)
    _lib.call("paths_reduce_slabs_f32", P(slabs), nblk, 3 * d, P(gb), 0, _lib.stream())
    _keep_slabs(slabs)
    return dx, gb[:d], gb[d:2 * d], gb[2 * d:]


def chain_forward(w, x_in_ptr: int, ldx: int, attn_ptr: int, lda: int, M: int, dev, drop: Optional[Drop] = None, layer: int = 0,
                  want_out: bool = True):
    """Row chain of one decoder layer (out_proj .. norm3) with the generic kernels, keeping what its backward needs.
    x_in / attn are raw pointers with row strides (token-0 rows of the last layer are strided).  With ``drop`` the four row-wise
    dropout sites of the layer are applied (dropout1, dropout2 on the broadcast cross-attention bias, the feed-forward's inner
    dropout, dropout3); without it this is the plain chain.  Mask element index = row * width + column of the [M, width] matrix."""
    f32 = _f32(dev)
    eps = w["eps"]
    d, F = w["wo"].shape[0], w["w1"].shape[0]            # trans_dim, dim_feedforward (= 4 trans_dim, reference model/aggregator.py:30)
    di = w["wo"].shape[1]                                # inner width of the attention (= d, or H x the padded head width: ops.padded_head_dim)
    c: Dict[str, object] = {}
    if drop is None:
        u1 = torch.empty((M, d), **f32)
        gemm_nt(attn_ptr, lda, w["wo"], u1, d, M, d, di, bias=w["bo"], residual=x_in_ptr, ldr=ldx)
        n1, c["xh1"], c["rs1"] = _ln_fwd(u1, None, w["ln1g"], w["ln1b"], M, eps, d=d)
        n2, c["xh2"], c["rs2"] = _ln_fwd(n1, w["cab"], w["ln2g"], w["ln2b"], M, eps, d=d)
        hid = torch.empty((M, F), **f32)
        gemm_nt(n2, d, w["w1"], hid, F, M, F, d, bias=w["b1"], act=1)
        u3 = torch.empty((M, d), **f32)
        gemm_nt(hid, F, w["w2"], u3, d, M, d, F, bias=w["b2"], residual=n2, ldr=d)
        c["hid"], c["hid_used"] = hid, hid
    else:
        p = drop.p
        sa = torch.empty((M, d), **f32)
        gemm_nt(attn_ptr, lda, w["wo"], sa, d, M, d, di, bias=w["bo"])
        u1 = dropout_rows(sa, d, M, d, drop.key(layer, Drop.SA_OUT), p, out=sa, ldo=d, resid=x_in_ptr, ldr=ldx)       # x + dropout1(sa)
        n1, c["xh1"], c["rs1"] = _ln_fwd(u1, None, w["ln1g"], w["ln1b"], M, eps, d=d)
        u2 = dropout_rows(None, 0, M, d, drop.key(layer, Drop.CA_OUT), p, resid=n1, ldr=d, vec=w["cab"])                 # n1 + dropout2(cab)
        n2, c["xh2"], c["rs2"] = _ln_fwd(u2, None, w["ln2g"], w["ln2b"], M, eps, d=d)
        hid = torch.empty((M, F), **f32)
        gemm_nt(n2, d, w["w1"], hid, F, M, F, d, bias=w["b1"], act=1)
        hd = dropout_rows(hid, F, M, F, drop.key(layer, Drop.FF_INNER), p)                                             # dropout(relu(linear1))
        ffo = torch.empty((M, d), **f32)
        gemm_nt(hd, F, w["w2"], ffo, d, M, d, F, bias=w["b2"])
        u3 = dropout_rows(ffo, d, M, d, drop.key(layer, Drop.FF_OUT), p, out=ffo, ldo=d, resid=n2, ldr=d)           # n2 + dropout3(ffo)
        c["hid"], c["hid_used"] = hid, hd
    c["n2"] = n2
    x3, c["xh3"], c["rs3"] = _ln_fwd(u3, None, w["ln3g"], w["ln3b"], M, eps, want_y=want_out, d=d)
    c["x3"] = x3
    return c


def chain_backward(w, x_in_ptr: int, ldx: int, attn_ptr: int, lda: int, M: int, dx_out: torch.Tensor, dev,
                   drop: Optional[Drop] = None, layer: int = 0, saved: Optional[Dict[str, object]] = None):
    """Row chain of one decoder layer (out_proj .. norm3): recompute its intermediates with the generic kernels
    (:func:`chain_forward`, same dropout masks as the forward) unless the forward's own are handed in, then differentiate.
    Returns (grads, dx_in [M,128], dattn [M,128])."""
    f32 = _f32(dev)
    d, F = w["wo"].shape[0], w["w1"].shape[0]
    di = w["wo"].shape[1]                                # (see chain_forward)
    g: Dict[str, torch.Tensor] = {}
    # ``saved``: the dict the training forward's chain_forward returned (kept when the forward ran on these very kernels: dropout
    # on, or a shape-generic geometry); otherwise the fused forward kept nothing and the chain is recomputed here
    c = saved if saved is not None else chain_forward(w, x_in_ptr, ldx, attn_ptr, lda, M, dev, drop, layer, want_out=False)
    hid, hd, n2 = c["hid"], c["hid_used"], c["n2"]
    p = drop.p if drop is not None else 0.0
    # ---- backward
    du3, g["ln3g"], g["ln3b"], cs3 = _ln_bwd_sums(dx_out, c["xh3"], c["rs3"], w["ln3g"], M, d)
    if drop is None:
        dffo, g["b2"] = du3, cs3
    else:
        dffo = dropout_rows(du3, d, M, d, drop.key(layer, Drop.FF_OUT), p)
        g["b2"] = colsum(dffo, d, M, d)
    dhid = torch.empty((M, F), **f32)
    gemm_nt(dffo, d, Transposed(w["w2"], d, F), dhid, F, M, F, d, mask=hid, ldm=F)
    if drop is not None:
        dropout_rows(dhid, F, M, F, drop.key(layer, Drop.FF_INNER), p, out=dhid, ldo=F)
    with side_stream(dev, dffo, hd, dhid, n2):
        g["w2"] = torch.empty((d, F), **f32)
        gemm_tn(dffo, d, hd, F, g["w2"], M, d, F)
        g["w1"] = torch.empty((F, d), **f32)
        gemm_tn(dhid, F, n2, d, g["w1"], M, F, d)
        g["b1"] = colsum(dhid, F, M, F)
    dn2 = torch.empty((M, d), **f32)
    gemm_nt(dhid, F, Transposed(w["w1"], F, d), dn2, d, M, d, F, residual=du3, ldr=d)
    du2, g["ln2g"], g["ln2b"], cs2 = _ln_bwd_sums(dn2, c["xh2"], c["rs2"], w["ln2g"], M, d)
    g["cab"] = cs2 if drop is None else colsum(dropout_rows(du2, d, M, d, drop.key(layer, Drop.CA_OUT), p), d, M, d)
    du1, g["ln1g"], g["ln1b"], cs1 = _ln_bwd_sums(du2, c["xh1"], c["rs1"], w["ln1g"], M, d)
    if drop is None:
        dsa, g["bo"] = du1, cs1
    else:
        dsa = dropout_rows(du1, d, M, d, drop.key(layer, Drop.SA_OUT), p)
        g["bo"] = colsum(dsa, d, M, d)
    dattn = torch.empty((M, di), **f32)
    gemm_nt(dsa, d, Transposed(w["wo"], d, di), dattn, di, M, di, d)
    with side_stream(dev, dsa):
        g["wo"] = torch.empty((d, di), **f32)
        gemm_tn(dsa, d, attn_ptr, lda, g["wo"], M, d, di)
    return g, du1, dattn


def qkv_backward(w, x_in: torch.Tensor, dqkv: torch.Tensor, M: int, qscale: float, dx_accum: torch.Tensor, fold_qscale: bool = True):
    """in_proj backward.  dqkv [M,3d] = [dq | dk | dv]; dx_accum [M,d] += dqkv W_in.  fold_qscale (the 128-wide kernels, whose
    q is stored pre-scaled): dq is the gradient of the SCALED q, so d(q_scaled)/d(q) is folded into the weight copy and into the
    q rows of the weight / bias gradients; the shape-generic attention backward returns the gradient of the unscaled q."""
    f32 = _f32(x_in.device)
    d, di = w["wo"].shape                                     # model width, inner width of the attention (ops.padded_head_dim)
    g: Dict[str, torch.Tensor] = {}
    wt = transpose(w["wqkv"], 3 * di, d)                      # [d, 3 di]
    if fold_qscale:
        wt[:, :di] *= qscale
    gemm_nt(dqkv, 3 * di, wt, dx_accum, d, M, d, 3 * di, accumulate=True)
    with side_stream(x_in.device, dqkv, x_in):
        g["wqkv"] = torch.empty((3 * di, d), **f32)
        gemm_tn(dqkv, 3 * di, x_in, d, g["wqkv"], M, 3 * di, d)
        g["bqkv"] = colsum(dqkv, 3 * di, M, 3 * di)
        if fold_qscale:
            after_reductions(lambda w_=g["wqkv"], b_=g["bqkv"]: (w_[:d].mul_(qscale), b_[:d].mul_(qscale)))
    return g


def attention(q, k, v, out, lse, num_ims, B, T, H, hd, max_queries, drop_key: int = 0, drop_p: float = 0.0):
    """Masked self-attention forward (+ log2-domain lse for the backward kernels): split-bf16 MFMA kernel (exact fp32
    products) unless PATHS_GEMM_MODE=f32.  drop_p > 0: dropout on the attention probabilities (site key drop_key)."""
    st = _lib.stream()
    if ops.GEMM_MODE != "f32":
        TP = ops.TRAIN_FWD_PLANES
        ws = torch.empty((int(_lib.load().paths_attention_x6_workspace(B, T, H, hd, TP)),), device=q.device, dtype=torch.uint8)
        if drop_p > 0:
            _lib.call("paths_attention_x6_dropout", P(q), P(k), P(v), P(out), P(lse), P(num_ims), B, T, H, hd, max_queries, P(ws), TP,
                      drop_key, drop_p, st)
        else:
            _lib.call("paths_attention_x6", P(q), P(k), P(v), P(out), P(lse), P(num_ims), B, T, H, hd, max_queries, P(ws), TP, 0, st)
    elif drop_p > 0:
        raise NotImplementedError("dropout needs the split-operand attention kernel (PATHS_GEMM_MODE h3 or x6)")
    else:
        _lib.call("paths_attention_f32", P(q), P(k), P(v), P(out), P(lse), P(num_ims), B, T, H, hd, max_queries, st)


def attention_token0(q, k, v, num_ims, B, T, H, hd, drop_key: int = 0, drop_p: float = 0.0):
    """Token-0 attention of the last layer (csrc/attn_token0.hip, keys split over workgroups): (a0 [B, H*hd], lse0 [B, H])."""
    f32 = _f32(q.device)
    a0, lse0 = torch.empty((B, H * hd), **f32), torch.empty((B, H), **f32)
    ws = torch.empty((int(_lib.load().paths_attention_token0_workspace(B, T, H)),), **f32)
    _lib.call("paths_attention_token0_fwd", P(q), P(k), P(v), P(num_ims), P(a0), P(lse0), P(ws), B, T, H, hd, drop_key, drop_p, _lib.stream())
    return a0, lse0


def transformer_forward_train(mc, lvl_pack, tokens, num_ims, ctx_prev, drop: Optional[Drop] = None,
                              ctx_all: Optional[torch.Tensor] = None):
    """Forward of the aggregator with the per-layer tensors the backward needs (q,k,v, lse, attention output).
    ``drop`` (train mode with dropout > 0): the fused row-chain kernels have no dropout sites, so the chain of every layer runs
    on the generic kernels of :func:`chain_forward` - the very sequence the backward recomputes - with the masks applied."""
    if not ops.fast_path(mc):
        return _transformer_forward_train_generic(mc, lvl_pack, tokens, num_ims, ctx_prev, drop, ctx_all)
    B, T, d = tokens.shape
    H, L = mc.trans_heads, mc.trans_layers
    hd = d // H
    f32 = _f32(tokens.device)
    st = _lib.stream()
    qscale = LOG2E / math.sqrt(hd)
    layers = lvl_pack["layers"]
    # ctx_prev [B,128]: slide_ctx_mode "residual" (added to the slide feature); ctx_all [B,depth,128] contiguous: "concat" (the
    # classifier reads cat(flatten(ctx_all), slide feature), reference model/paths.py:134-137); at most one of the two is given
    assert ctx_prev is None or ctx_all is None
    cdepth = ctx_all.shape[1] if ctx_all is not None else 0
    cat_ptr = P(ctx_all) if cdepth > 0 else None
    sv = {"layers": [], "num_ims": num_ims, "tokens": tokens, "ctx_prev": ctx_prev, "ctx_all": ctx_all if cdepth > 0 else None}

    def token_layer(x_in, x_out, post, nxt, attn, q, k, v):
        w = post or nxt
        gg = lambda dct, key: P(dct[key]) if dct is not None else None
        _lib.call("paths_token_layer_f32", P(x_in), P(attn) if post else None, P(x_out) if post else None,
                  gg(post, "wo"), gg(post, "bo"), gg(post, "ln1g"), gg(post, "ln1b"), gg(post, "cab"), gg(post, "ln2g"), gg(post, "ln2b"),
                  gg(post, "w1"), gg(post, "b1"), gg(post, "w2"), gg(post, "b2"), gg(post, "ln3g"), gg(post, "ln3b"),
                  gg(nxt, "wqkv"), gg(nxt, "bqkv"), P(q), P(k), P(v), P(num_ims), B, T, d, H,
                  1 if post else 0, 1 if nxt else 0, 0, qscale, w["eps"], 0, st)

    x = tokens
    q, k, v = (torch.empty((B, H, T, hd), **f32) for _ in range(3))
    token_layer(x, None, None, layers[0], None, q, k, v)
    sv["drop"] = drop
    for l in range(L - 1):
        attn, lse = zeros_group(q.device, (B, T, d), (B, H, T))
        q2, k2, v2 = (torch.empty((B, H, T, hd), **f32) for _ in range(3))
        if drop is None:
            attention(q, k, v, attn, lse, num_ims, B, T, H, hd, 0)
            x_out = torch.empty((B, T, d), **f32)
            token_layer(x, x_out, layers[l], layers[l + 1], attn, q2, k2, v2)
        else:
            attention(q, k, v, attn, lse, num_ims, B, T, H, hd, 0, drop.key(l, Drop.ATTN), drop.p)
            chain = chain_forward(layers[l], x.data_ptr(), d, attn.data_ptr(), d, B * T, tokens.device, drop, l)
            x_out = chain["x3"].view(B, T, d)
            token_layer(x_out, None, None, layers[l + 1], None, q2, k2, v2)
        sv["layers"].append({"x_in": x, "q": q, "k": k, "v": v, "attn": attn, "lse": lse, "chain": chain if drop is not None else None})
        x, q, k, v = x_out, q2, k2, v2
    # last layer at token 0 (+ decoder.norm, residual, classifier): one fused launch
    w = layers[L - 1]
    nlog = lvl_pack["wcls"].shape[0]
    ctx_out = torch.empty((B, d), **f32)
    logits = torch.empty((B, nlog), **f32)
    if drop is not None:
        # the fused token-0 tail has no dropout sites either: single-query attention (rows > 0 of the output are not needed), the
        # row chain on the B token-0 rows, decoder.norm, slide-context residual and classifier with the generic kernels
        a0, lse0 = attention_token0(q, k, v, num_ims, B, T, H, hd, drop.key(L - 1, Drop.ATTN), drop.p)
        chain0 = chain_forward(w, x.data_ptr(), T * d, a0.data_ptr(), d, B, tokens.device, drop, L - 1)
        x3 = chain0["x3"]
        _lib.call("paths_final_head", P(x3), d, P(lvl_pack["lnfg"]), P(lvl_pack["lnfb"]), P(ctx_prev),
                  ctx_prev.stride(0) if ctx_prev is not None else 0, cat_ptr, cdepth, P(lvl_pack["wcls"]), P(lvl_pack["bcls"]), nlog,
                  lvl_pack["wcls"].shape[1], P(ctx_out), P(logits), B, d, lvl_pack["lnf_eps"], st)
        sv["last"] = {"x_in": x, "q": q, "k": k, "v": v, "a0": a0, "lse0": lse0, "chain": chain0}
        sv["ctx_out"], sv["logits"] = ctx_out, logits
        return sv
    ws = torch.empty((B * H * 16 * 36,), **f32)
    _lib.call("paths_token0_tail", P(x), P(q), P(k), P(v), P(num_ims), P(w["wo"]), P(w["bo"]), P(w["ln1g"]), P(w["ln1b"]),
              P(w["cab"]), P(w["ln2g"]), P(w["ln2b"]), P(w["w1"]), P(w["b1"]), P(w["w2"]), P(w["b2"]), P(w["ln3g"]), P(w["ln3b"]),
              P(lvl_pack["lnfg"]), P(lvl_pack["lnfb"]), P(ctx_prev), ctx_prev.stride(0) if ctx_prev is not None else 0, cat_ptr, cdepth,
              P(lvl_pack["wcls"]), P(lvl_pack["bcls"]), nlog, lvl_pack["wcls"].shape[1], P(ctx_out), P(logits), P(ws),
              B, T, d, H, w["eps"], lvl_pack["lnf_eps"], st)
    sv["last"] = {"x_in": x, "q": q, "k": k, "v": v}
    sv["ctx_out"], sv["logits"] = ctx_out, logits
    return sv


def _attention_generic(qkv, attn, lse, num_ims, B, T, H, hd, qscale, max_queries, drop: Optional[Drop], layer: int):
    key, p = (drop.key(layer, Drop.ATTN), drop.p) if drop is not None else (0, 0.0)
    if ops.wide_head(hd):            # head_dim > 64: csrc/attn_wide.hip (qkv carries 128 spare rows, see in_proj below)
        ws = torch.empty((int(_lib.load().paths_attention_wide_workspace(T, hd)),), **_f32(attn.device))
        _lib.call("paths_attention_wide_fwd", P(qkv), qkv.stride(0), P(attn), P(lse), P(num_ims), B, T, H, hd, qscale, max_queries, key, p,
                  P(ws), _lib.stream())
        return
    _lib.call("paths_attention_any_train", P(qkv), qkv.stride(0), P(attn), P(lse), P(num_ims), B, T, H, hd, qscale, max_queries, key, p,
              _lib.stream())


def _attention_bwd_generic(qkv, o, d_o, lse, num_ims, dqkv, ws_dsum, B, T, H, hd, d, qscale, max_queries, dkey):
    """dqkv (zero on entry) from the token-major qkv of the generic training forward: the flash-style kernels of csrc/generic_bwd.hip up
    to head_dim 64, the three-step form of csrc/attn_wide.hip above."""
    if ops.wide_head(hd):
        ws = torch.empty((int(_lib.load().paths_attention_wide_workspace(T, hd)),), **_f32(dqkv.device))
        _lib.call("paths_attention_wide_bwd", P(qkv), qkv.stride(0), P(o), P(d_o), P(lse), P(num_ims), P(dqkv), B, T, H, hd, qscale, max_queries,
                  *dkey, P(ws), _lib.stream())
        return
    _lib.call("paths_attention_bwd_any", P(qkv), qkv.stride(0), P(o), P(d_o), P(lse), P(num_ims), P(dqkv), P(ws_dsum), B, T, H, hd, qscale, max_queries,
              *dkey, _lib.stream())


def _transformer_forward_train_generic(mc, lvl_pack, tokens, num_ims, ctx_prev, drop: Optional[Drop], ctx_all: Optional[torch.Tensor]):
    """:func:`transformer_forward_train` for any (trans_dim, heads): every product on the generic GEMMs, attention on the shape-generic
    kernel (csrc/generic.hip, TRAIN form: lse + dropout), the row chains on :func:`chain_forward` - the sequence the backward
    recomputes.  q, k, v stay token-major ([B*T, 3d], q unscaled: the attention kernels scale it on load)."""
    B, T, d = tokens.shape
    H, L = mc.trans_heads, mc.trans_layers
    hd = ops.padded_head_dim(d // H)        # the width the kernels run: narrower heads are zero-padded in the pack (ops.padded_head_dim)
    di = H * hd
    dev = tokens.device
    f32 = _f32(dev)
    qscale = LOG2E / math.sqrt(d // H)
    layers = lvl_pack["layers"]
    M = B * T
    assert ctx_prev is None or ctx_all is None
    cdepth = ctx_all.shape[1] if ctx_all is not None else 0
    sv = {"layers": [], "num_ims": num_ims, "tokens": tokens, "ctx_prev": ctx_prev, "ctx_all": ctx_all if cdepth > 0 else None,
          "drop": drop}

    spare = 128 if ops.wide_head(hd) else 0          # wide heads: the score products read whole 128-row tiles of k / v

    def in_proj(x, w):
        qkv = torch.empty((M + spare, 3 * di), **f32)
        if spare:
            qkv[M:].zero_()
        gemm_nt(x.view(M, d), d, w["wqkv"], qkv, 3 * di, M, 3 * di, d, bias=w["bqkv"])
        return qkv

    x = tokens
    qkv = in_proj(x, layers[0])
    for l in range(L - 1):
        attn, lse = zeros_group(dev, (B, T, di), (B, H, T))
        _attention_generic(qkv, attn, lse, num_ims, B, T, H, hd, qscale, 0, drop, l)
        chain = chain_forward(layers[l], x.data_ptr(), d, attn.data_ptr(), di, M, dev, drop, l)
        x_out = chain["x3"].view(B, T, d)
        sv["layers"].append({"x_in": x, "qkv": qkv, "attn": attn, "lse": lse, "chain": chain})
        x, qkv = x_out, in_proj(x_out, layers[l + 1])
    # last layer: only token 0 of its output is read (reference model/aggregator.py:75)
    w = layers[L - 1]
    attn0, lse0 = zeros_group(dev, (B, T, di), (B, H, T))
    _attention_generic(qkv, attn0, lse0, num_ims, B, T, H, hd, qscale, 1, drop, L - 1)
    chain0 = chain_forward(w, x.data_ptr(), T * d, attn0.data_ptr(), T * di, B, dev, drop, L - 1)
    x3 = chain0["x3"]
    nlog = lvl_pack["wcls"].shape[0]
    ctx_out = torch.empty((B, d), **f32)
    logits = torch.empty((B, nlog), **f32)
    _lib.call("paths_final_head_any", P(x3), d, P(lvl_pack["lnfg"]), P(lvl_pack["lnfb"]), P(ctx_prev),
              ctx_prev.stride(0) if ctx_prev is not None else 0, P(ctx_all) if cdepth > 0 else None, cdepth, P(lvl_pack["wcls"]),
              P(lvl_pack["bcls"]), nlog, lvl_pack["wcls"].shape[1], P(ctx_out), P(logits), B, d, lvl_pack["lnf_eps"], _lib.stream())
    sv["last"] = {"x_in": x, "qkv": qkv, "attn0": attn0, "lse0": lse0, "chain": chain0}
    sv["ctx_out"], sv["logits"] = ctx_out, logits
    return sv


def transformer_backward(mc, lvl_pack, sv, d_logits: Optional[torch.Tensor], d_ctx_out: Optional[torch.Tensor]):
    """Returns (grads, d_tokens [B,T,d], d_ctx): d_ctx = gradient of ctx_prev [B,d] (residual mode), of ctx_all [B,depth,d]
    (concat mode) or None.  grads: {"layers": [per-layer dict], "lnfg", "lnfb", "wcls", "bcls"}.  d_logits / d_ctx_out may be
    None (zero).  The shipped geometry (ops.fast_path) differentiates its attention with the 128-wide kernels (q, k, v head-major,
    q pre-scaled); every other geometry with csrc/generic_bwd.hip (token-major qkv)."""
    tokens, num_ims, ctx_prev = sv["tokens"], sv["num_ims"], sv["ctx_prev"]
    B, T, d = tokens.shape
    H, L = mc.trans_heads, mc.trans_layers
    hd = ops.padded_head_dim(d // H)        # (the shipped geometry: 32 = its own width)
    di = H * hd
    dev = tokens.device
    f32 = _f32(dev)
    st = _lib.stream()
    qscale = LOG2E / math.sqrt(d // H)
    layers = lvl_pack["layers"]
    grads = {"layers": [None] * L}
    nlog = lvl_pack["wcls"].shape[0]
    fast = ops.fast_path(mc)
    assert nlog <= 128

    # ---- head: logits = F Wcls^T + bcls, F = decoder.norm(x3) (+ ctx_prev)
    wl = layers[L - 1]
    last = sv["last"]
    x_last = last["x_in"]
    drop: Optional[Drop] = sv.get("drop")
    dk = lambda l: (drop.key(l, Drop.ATTN), drop.p) if drop is not None else (0, 0.0)
    # token 0 of the last layer up to x3: kept by the forward when it ran on the generic kernels (dropout on, or a shape-generic
    # geometry); the fused token-0 tail keeps nothing, so its attention output and row chain are recomputed
    chain0 = last.get("chain")
    if fast:
        a0, lse0 = last.get("a0"), last.get("lse0")           # [B, d] / [B, H]
        if a0 is None:
            a0, lse0 = attention_token0(last["q"], last["k"], last["v"], num_ims, B, T, H, hd, *dk(L - 1))
        a0_ptr, a0_ld = a0.data_ptr(), d
    else:
        attn0 = last["attn0"]                                  # [B, T, di], row 0 of every slide filled
        a0_ptr, a0_ld = attn0.data_ptr(), T * di
    if chain0 is None:
        chain0 = chain_forward(wl, x_last.data_ptr(), T * d, a0_ptr, a0_ld, B, dev, drop, L - 1)
    x3 = chain0["x3"]
    xf, xhf, rsf = _ln_fwd(x3, None, lvl_pack["lnfg"], lvl_pack["lnfb"], B, lvl_pack["lnf_eps"], d=d)
    feat = xf + ctx_prev if ctx_prev is not None else xf                     # [B,d] (8 rows: bookkeeping)
    ctx_all = sv.get("ctx_all")
    cdepth = ctx_all.shape[1] if ctx_all is not None else 0
    dF = torch.zeros((B, d), **f32)
    d_ctx_all = torch.zeros((B, cdepth, d), **f32) if cdepth > 0 else None
    if d_ctx_out is not None:
        dF += d_ctx_out
    if d_logits is not None:
        dl = torch.zeros((B, 128), **f32)                                    # logits padded to one 128-wide k panel
        dl[:, :nlog] = d_logits
        wcls = lvl_pack["wcls"]                                              # [nlog, (cdepth + 1) * d]: blocks = [ctx levels | F]
        dpad = (d + 127) // 128 * 128
        for kb in range(cdepth + 1):
            wc = torch.zeros((dpad, 128), **f32)
            wc[:d, :nlog] = wcls[:, kb * d:(kb + 1) * d].t()                 # [in = d (padded rows), out padded]
            if kb == cdepth:
                gemm_nt(dl, 128, wc, dF, d, B, d, 128, accumulate=True)      # dF += dlogits Wcls[:, F block]
            else:
                blk = torch.empty((B, d), **f32)
                gemm_nt(dl, 128, wc, blk, d, B, d, 128)                      # gradient of the concatenated context of level kb
                d_ctx_all[:, kb] = blk
        cat_in = torch.cat((ctx_all.reshape(B, cdepth * d), feat), dim=1).contiguous() if cdepth > 0 else feat
        gw = torch.empty((128, (cdepth + 1) * d), **f32)
        gemm_tn(dl, 128, cat_in, (cdepth + 1) * d, gw, B, 128, (cdepth + 1) * d)         # (dlogits^T [ctx | F]), rows >= nlog are zero
        grads["wcls"] = gw[:nlog].contiguous()
        grads["bcls"] = colsum(dl, 128, B, 128)[:nlog].contiguous()
    else:
        grads["wcls"] = torch.zeros_like(lvl_pack["wcls"])
        grads["bcls"] = torch.zeros_like(lvl_pack["bcls"])
    d_ctx_prev = dF.clone() if ctx_prev is not None else d_ctx_all
    dx3, dyxf = _ln_bwd(dF, xhf, rsf, lvl_pack["lnfg"], B, d)
    grads["lnfg"], grads["lnfb"] = colsum(dyxf, d, B, d), colsum(dF, d, B, d)

    # ---- last layer, token 0 only
    g, dx0, da0 = chain_backward(wl, x_last.data_ptr(), T * d, a0_ptr, a0_ld, B, dx3, dev, drop, L - 1, saved=chain0)
    dqkv, dx = zeros_group(dev, (B, T, 3 * di), (B, T, d))                      # (dx: gradient of the last layer's input)
    if fast:
        ws = torch.empty((int(_lib.load().paths_attention_token0_workspace(B, T, H)),), **f32)
        _lib.call("paths_attention_token0_bwd", P(last["q"]), P(last["k"]), P(last["v"]), P(a0), P(da0), P(lse0), P(num_ims), P(dqkv), P(ws),
                  B, T, H, hd, *dk(L - 1), st)
    else:
        d_o = torch.zeros((B, T, di), **f32)                                 # only token 0 carries an output gradient
        d_o[:, 0, :] = da0
        ws = torch.empty((B * H * T,), **f32)
        _attention_bwd_generic(last["qkv"], attn0, d_o, last["lse0"], num_ims, dqkv, ws, B, T, H, hd, d, qscale, 1, dk(L - 1))
    dx[:, 0, :] = dx0
    g.update(qkv_backward(wl, x_last, dqkv, B * T, qscale, dx, fold_qscale=fast))
    grads["layers"][L - 1] = g

    # ---- full layers
    for l in range(L - 2, -1, -1):
        lv = sv["layers"][l]
        w = layers[l]
        M = B * T
        g, dx_in, dattn = chain_backward(w, lv["x_in"].data_ptr(), d, lv["attn"].data_ptr(), di, M, dx.view(M, d), dev, drop, l,
                                         saved=lv.get("chain"))
        dqkv = torch.zeros((B, T, 3 * di), **f32)
        ws = torch.empty((B * H * T,), **f32)
        if not fast:
            _attention_bwd_generic(lv["qkv"], lv["attn"], dattn, lv["lse"], num_ims, dqkv, ws, B, T, H, hd, d, qscale, 0, dk(l))
        elif ATTN_BWD_MODE == "x6q":    # dQ, dK and dV on the split-bf16 kernels (csrc/attn_bwd_x6.hip; PATHS_ATTN_BWD_KV_X6=0 in the C library keeps dK / dV on the f32 MFMA)
            img = torch.empty((int(_lib.load().paths_attention_bwd_x6_workspace(B, T, H, hd)),), device=dqkv.device, dtype=torch.uint8)
            # operand split as the other gradient products of the step (ops.TRAIN_PLANES: 4 = two bf16 planes, 3 = the exact three)
            _lib.call("paths_attention_bwd_x6_planes", P(lv["q"]), P(lv["k"]), P(lv["v"]), P(lv["attn"]), P(dattn), P(lv["lse"]),
                      P(num_ims), P(dqkv), P(ws), P(img), B, T, H, hd, *dk(l), 2 if ops.TRAIN_PLANES == 4 else 3, st)
        else:
            _lib.call("paths_attention_bwd_f32_dropout", P(lv["q"]), P(lv["k"]), P(lv["v"]), P(lv["attn"]), P(dattn), P(lv["lse"]),
                      P(num_ims), P(dqkv), P(ws), B, T, H, hd, *dk(l), st)
        g.update(qkv_backward(w, lv["x_in"], dqkv, M, qscale, dx_in, fold_qscale=fast))
        grads["layers"][l] = g
        dx = dx_in.view(B, T, d)
    side_join(dev)                     # the weight / bias gradients issued on the side stream (chain_backward, qkv_backward)
    # zero-padded heads (ops.padded_head_dim): gradients of the padded in_proj / out_proj images back to the parameters' own rows /
    # columns (once the deferred slab reductions that produce them have run)
    for l in range(L):
        if "_unpad" in layers[l]:
            def unpad(g_=grads["layers"][l], idx=layers[l]["_unpad"]):
                row, col = idx
                g_["wqkv"] = g_["wqkv"].index_select(0, row)
                g_["bqkv"] = g_["bqkv"].index_select(0, row)
                g_["wo"] = g_["wo"].index_select(1, col)
            after_reductions(unpad)
    return grads, dx, d_ctx_prev
