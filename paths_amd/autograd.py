"""torch.autograd glue for training (reference train.py:59-68: ``loss.backward(); opt.step()``).

torch's autograd ENGINE only orders the calls and accumulates ``.grad``; every derivative is computed by the HIP
launch sequences of paths_amd/backward.py:

  * :class:`LevelFn`  — one magnification level (``PATHSProcessor.process``): forward = the training forward that keeps
    the tensors the backward needs, backward = transformer_backward + selection_backward;
  * :class:`GatherFn` — the child gather between levels (``PreprocessedSlide.iter``): backward = paths_gather_rows_bwd.

Training covers lstm = true / false and slide_ctx_mode residual / concat / none.  Dropout > 0 in train mode runs the
transformer's row chain on the generic kernels with regenerated masks (paths_amd/backward.py:Drop).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import _lib, backward as bw, ops

LSTM_ORDER = ["forget_gate.0.weight", "forget_gate.0.bias", "remember_gate.0.weight", "remember_gate.0.bias",
              "remember_map.0.weight", "remember_map.0.bias", "out_select_gate.0.weight", "out_select_gate.0.bias",
              "mem_to_out.0.weight", "mem_to_out.0.bias"]
LAYER_ORDER = [("self_attn.in_proj_weight", "wqkv"), ("self_attn.in_proj_bias", "bqkv"), ("self_attn.out_proj.weight", "wo"),
               ("self_attn.out_proj.bias", "bo"), ("multihead_attn.out_proj.bias", "cab"), ("norm1.weight", "ln1g"),
               ("norm1.bias", "ln1b"), ("norm2.weight", "ln2g"), ("norm2.bias", "ln2b"), ("norm3.weight", "ln3g"),
               ("norm3.bias", "ln3b"), ("linear1.weight", "w1"), ("linear1.bias", "b1"), ("linear2.weight", "w2"),
               ("linear2.bias", "b2")]


def lstm_params(lstm) -> List[torch.nn.Parameter]:
    cached = lstm.__dict__.get("_paths_lstm_params")          # (Parameter objects are stable; the module walk was 0.1 ms per call)
    if cached is None:
        sd = dict(lstm.named_parameters())
        cached = [sd[k] for k in LSTM_ORDER]
        object.__setattr__(lstm, "_paths_lstm_params", cached)
    return list(cached)


def level_params(proc) -> List[torch.nn.Parameter]:
    """Live parameters of one level in the order LevelFn returns their gradients (cached on the module: walking named_parameters()
    of every decoder layer was 0.5 ms of host time per training step)."""
    cached = proc.__dict__.get("_paths_level_params")
    if cached is not None:
        return list(cached)
    out = _level_params(proc)
    object.__setattr__(proc, "_paths_level_params", out)
    return list(out)


def _level_params(proc) -> List[torch.nn.Parameter]:
    out = [proc.importance_mlp[0].weight, proc.importance_mlp[0].bias, proc.importance_mlp[2].weight, proc.importance_mlp[2].bias,
           proc.global_agg.proj_in.weight, proc.global_agg.proj_in.bias, proc.global_agg.special_token]
    dec = proc.global_agg.transformer.decoder
    for lyr in dec.layers:
        sd = dict(lyr.named_parameters())
        out += [sd[name] for name, _ in LAYER_ORDER]
    out += [dec.norm.weight, dec.norm.bias, proc.classification_layer.weight, proc.classification_layer.bias]
    return out


def dead_params(model) -> List[torch.nn.Parameter]:
    """Parameters that exist for checkpoint compatibility but never influence an output: the whole nn.Transformer
    encoder and the cross-attention matrices.  The reference gives them ZERO gradients (so AdamW still
    applies weight decay to them); :func:`fill_dead_grads` reproduces that."""
    cached = getattr(model, "_paths_dead_params", None)      # (the module walk below is ~1 ms of host time per training step)
    if cached is not None:
        return cached
    out = []
    for proc in model.procs:
        tr = proc.global_agg.transformer
        out += list(tr.encoder.parameters())
        for lyr in tr.decoder.layers:
            out += [lyr.multihead_attn.in_proj_weight, lyr.multihead_attn.in_proj_bias, lyr.multihead_attn.out_proj.weight]
    object.__setattr__(model, "_paths_dead_params", out)     # Parameter objects are stable (load_state_dict / .to() work in place)
    return out


def live_grad_params(model, num_levels: Optional[int] = None) -> List[torch.nn.Parameter]:
    """The parameters that receive a real gradient from one training step, in a FIXED order that is the same on every
    rank whatever slides it holds: the shared LSTM + the live parameters of levels 0..num_levels-1, without the classifiers
    of the non-final levels (their logits are unused: grad stays None, reference SURVEY 3.2).  This is the list the gradient
    all-reduce walks (paths_amd/distributed.py) and the list a rank without slides fills with zeros."""
    L = len(model.procs) if num_levels is None else min(int(num_levels), len(model.procs))
    out = list(lstm_params(model.lstm)) if model.use_lstm else []
    for i in range(L):
        mc = model.procs[i].config
        no_agg = i < L - 1 and mc.slide_ctx_mode == "none"          # nothing of this level's aggregator reaches the loss
        if model.use_lstm:
            lp = level_params(model.procs[i])
            if no_agg:
                lp = []
            elif mc.importance_mode != "mul":
                lp = lp[4:]
        else:
            # lstm = false: hctx_mlp of level 0 never runs (no previous state); the importance MLP only with importance_mode "mul"
            lp = level_params_nolstm(model.procs[i])
            mc = model.procs[i].config
            drop = set()
            if i == 0 or not mc.hierarchical_ctx:
                drop |= {4, 5, 6, 7}
            if mc.importance_mode != "mul":
                drop |= {0, 1, 2, 3}
            if no_agg:
                drop |= set(range(8, len(lp)))
            lp = [p for j, p in enumerate(lp) if j not in drop]
        out += lp if (i == L - 1 or no_agg) else lp[:-2]
    return out


def zero_live_grads(model, num_levels: Optional[int] = None):
    """A rank that holds no slide of a (short) global batch: exactly the gradient set of an active rank, all zeros."""
    for p in live_grad_params(model, num_levels):
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        else:
            p.grad.zero_()
    fill_dead_grads(model)


def fill_dead_grads(model):
    """Zero gradients for the dead parameters.  They are views of ONE zero buffer kept on the model (160 tensors per step
    would otherwise cost 160 fill launches); nothing downstream writes a non-zero into a gradient that is exactly zero
    (AdamW only reads it, an all-reduce / clipping of zeros yields zeros)."""
    dead = dead_params(model)
    todo = [p for p in dead if p.grad is None]
    if not todo:
        return
    need = max(p.numel() for p in dead)
    zero = getattr(model, "_paths_dead_zero", None)
    if zero is None or zero.numel() < need or zero.device != todo[0].device:
        zero = torch.zeros((need,), device=todo[0].device, dtype=torch.float32)
        object.__setattr__(model, "_paths_dead_zero", zero)
    for p in todo:
        p.grad = zero[:p.numel()].view_as(p)


def clear_grads(model, optimizer):
    """``optimizer.zero_grad(set_to_none=True)`` for one training step of this model - except that the DEAD parameters keep the
    views of the shared zero buffer :func:`fill_dead_grads` gave them (they are exactly zero and nothing ever writes them: dropping
    and re-creating 160 views per step was 0.9 ms of host time next to a 13-ms step)."""
    dead = getattr(model, "_paths_dead_ids", None)
    if dead is None:
        dead = frozenset(id(p) for p in dead_params(model))
        object.__setattr__(model, "_paths_dead_ids", dead)
    zero = getattr(model, "_paths_dead_zero", None)
    lo = zero.data_ptr() if zero is not None else 0
    hi = lo + 4 * zero.numel() if zero is not None else 0
    for group in optimizer.param_groups:
        for p in group["params"]:
            g = p.grad
            if g is not None and not (id(p) in dead and lo <= g.data_ptr() < hi):
                p.grad = None


def next_dropout_seed(device) -> int:
    """A fresh 64-bit seed for one level's dropout masks, taken from the DEVICE's default generator - the stream the reference's
    dropout consumes when it trains on a GPU - by reading (seed, philox offset) on the host and advancing the offset: no device
    sync, ``torch.manual_seed`` restarts the sequence, and the CPU generator is untouched, so the DataLoader-compatible shuffle
    order (paths_amd/train.py:epoch_permutation) does not depend on whether dropout is on."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    gen = torch.cuda.default_generators[idx]
    seed, off = int(gen.initial_seed()), int(gen.get_offset())
    gen.set_offset(off + 4)
    import os
    rank = int(os.environ.get("RANK", "0"))      # ranks share torch.manual_seed(config.seed): fold the rank in, or every rank would
    z = (seed * 0x9E3779B97F4A7C15 + (off + 1) * 0xD1B54A32D192ED03 + rank * 0xA24BAED4963EE407) & 0xFFFFFFFFFFFFFFFF   # draw the same masks
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 29)


def check_dropout_supported(proc):
    """Every entry to the differentiable path passes here (LevelFn.forward): a config with dropout > 0 in train mode is
    never silently run with its five dropout sites per layer as identity (reference model/aggregator.py:25-33)."""
    if proc.training and proc.config.dropout > 0 and not DROPOUT_IMPLEMENTED:
        raise NotImplementedError("dropout > 0 in train mode is not implemented on the HIP path: set model_config.dropout = 0 "
                                  "or call model.eval()")


DROPOUT_IMPLEMENTED = True


def _detached_saves(sel: dict, tr: dict):
    """The dicts kept on ctx for the backward, with the tensors that forward() RETURNS replaced by detached aliases.  A returned
    tensor gets grad_fn = this node, the node owns ctx, and ctx -> dict -> that tensor would close a reference cycle through C++
    shared pointers that Python's collector cannot see: every level's saved activations (2.6 GiB per step at the K = 2048
    shape) leaked until the device was full.  The backward drops the dicts as well, so the activations die with the step."""
    sel, tr = dict(sel), dict(tr)
    for d, keys in ((sel, ("state_out", "importance")), (tr, ("logits", "ctx_out"))):
        for k in keys:
            if d.get(k) is not None:
                d[k] = d[k].detach()
    return sel, tr


class LevelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, proc, lstm, fts, locs, num_ims, state_prev, ctx_prev, *params):
        """ctx_prev: [B,d] (slide_ctx_mode "residual": the previous level's slide context) or [B,depth,d] (mode "concat": all
        previous levels' slide contexts, reference model/paths.py:134-137) or None."""
        mc = proc.config
        check_dropout_supported(proc)
        lp, vp = ops.pack_lstm(lstm), ops.pack_level(proc)
        sel = bw.selection_forward_train(mc, lp, vp, fts, locs.contiguous(), num_ims.contiguous(), state_prev)
        res = ctx_prev if (mc.slide_ctx_mode == "residual" and ctx_prev is not None) else None
        cat = ctx_prev.contiguous() if (mc.slide_ctx_mode == "concat" and ctx_prev is not None and ctx_prev.shape[1] > 0) else None
        assert res is None or res.dim() == 2
        assert cat is None or cat.dim() == 3
        drop = None
        if proc.training and mc.dropout > 0:
            drop = bw.Drop(mc.dropout, next_dropout_seed(fts.device), proc.depth)       # (torch.manual_seed controls the masks)
        tr = bw.transformer_forward_train(mc, vp, sel["tokens"], sel["num_ims"], res, drop, cat)
        ctx.proc, ctx.lstm = proc, lstm
        ctx.sel, ctx.tr = _detached_saves(sel, tr)
        ctx.has_state, ctx.has_ctx = state_prev is not None, (res is not None or cat is not None)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(sel["importance"])
        return tr["logits"], tr["ctx_out"], sel["state_out"], sel["importance"]

    @staticmethod
    def backward(ctx, d_logits, d_ctx_out, d_state_out, _d_imp):
        proc, lstm, sel, tr = ctx.proc, ctx.lstm, ctx.sel, ctx.tr
        ctx.sel = ctx.tr = None                     # the saved activations die with this call (see _detached_saves)
        mc = proc.config
        lp, vp = ops.pack_lstm(lstm), ops.pack_level(proc)
        cont = lambda t: t.contiguous() if t is not None else None
        # Parameters without a path to the loss get NO gradient (None, as in the reference: AdamW then leaves them alone, weight
        # decay included).  no_agg: neither this level's logits nor its slide context are used downstream (non-final levels
        # under slide_ctx_mode "none"): the whole aggregator, proj_in and - its only consumer being the tokens - the
        # importance MLP are cut off; importance_mode != "mul": the importance only drives the (non-differentiable) top-K.
        no_agg = d_logits is None and d_ctx_out is None
        with bw.deferred_reductions():               # the ~35 slab reductions of this level's parameter gradients: one launch
            if no_agg:
                tg, d_ctx_prev = None, None
                d_tok = torch.zeros_like(tr["tokens"])
            else:
                tg, d_tok, d_ctx_prev = bw.transformer_backward(mc, vp, tr, cont(d_logits), cont(d_ctx_out))
            sg, d_state_prev = bw.selection_backward(mc, lp, vp, sel, d_tok, cont(d_state_out))
        grads = _level_grads(mc, lstm, sg, tg, no_agg, d_logits is not None)
        return (None, None, None, None, None, d_state_prev if ctx.has_state else None,
                d_ctx_prev if ctx.has_ctx else None, *grads)


class LevelParentFn(torch.autograd.Function):
    """:class:`LevelFn` in the device recursion's once-per-parent form: instead of a per-child copy of the parent state the level takes
    the children's inherited memory cell ``c0`` [B,N,Hc], the kept parents' h rows ``h_kept`` [B*cap, D] and the child -> kept-slot map
    (paths_amd/backward.py:selection_forward_train ``parent=``); gradients come back for ``c0`` and ``h_kept``."""

    @staticmethod
    def forward(ctx, proc, lstm, fts, locs, num_ims, c0, h_kept, hp_row, child_pos, keep_count, cap, ctx_prev, *params):
        mc = proc.config
        check_dropout_supported(proc)
        lp, vp = ops.pack_lstm(lstm), ops.pack_level(proc)
        sel = bw.selection_forward_train(mc, lp, vp, fts, locs.contiguous(), num_ims.contiguous(), None,
                                         parent={"c0": c0, "h_kept": h_kept, "hp_row": hp_row, "child_pos": child_pos,
                                                 "keep_count": keep_count, "cap": int(cap)})
        res = ctx_prev if (mc.slide_ctx_mode == "residual" and ctx_prev is not None) else None
        cat = ctx_prev.contiguous() if (mc.slide_ctx_mode == "concat" and ctx_prev is not None and ctx_prev.shape[1] > 0) else None
        drop = None
        if proc.training and mc.dropout > 0:
            drop = bw.Drop(mc.dropout, next_dropout_seed(fts.device), proc.depth)
        tr = bw.transformer_forward_train(mc, vp, sel["tokens"], sel["num_ims"], res, drop, cat)
        ctx.proc, ctx.lstm = proc, lstm
        ctx.sel, ctx.tr = _detached_saves(sel, tr)
        ctx.has_ctx = res is not None or cat is not None
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(sel["importance"])
        return tr["logits"], tr["ctx_out"], sel["state_out"], sel["importance"]

    @staticmethod
    def backward(ctx, d_logits, d_ctx_out, d_state_out, _d_imp):
        proc, lstm, sel, tr = ctx.proc, ctx.lstm, ctx.sel, ctx.tr
        ctx.sel = ctx.tr = None
        mc = proc.config
        lp, vp = ops.pack_lstm(lstm), ops.pack_level(proc)
        cont = lambda t: t.contiguous() if t is not None else None
        no_agg = d_logits is None and d_ctx_out is None
        with bw.deferred_reductions():
            if no_agg:
                tg, d_ctx_prev = None, None
                d_tok = torch.zeros_like(tr["tokens"])
            else:
                tg, d_tok, d_ctx_prev = bw.transformer_backward(mc, vp, tr, cont(d_logits), cont(d_ctx_out))
            sg, (d_c0, d_hk) = bw.selection_backward(mc, lp, vp, sel, d_tok, cont(d_state_out))
        grads = _level_grads(mc, lstm, sg, tg, no_agg, d_logits is not None)
        return (None, None, None, None, None, d_c0, d_hk, None, None, None, None, d_ctx_prev if ctx.has_ctx else None, *grads)


def _level_grads(mc, lstm, sg, tg, no_agg: bool, has_logits: bool):
    """Parameter gradients of one level in the order of lstm_params + level_params (None where there is no path to the loss)."""
    lg = bw.unpack_lstm_grads(lstm, sg)
    grads = [lg[k] for k in LSTM_ORDER]
    Hi = mc.importance_mlp_hidden_dim
    imp_live = mc.importance_mode == "mul" and not no_agg
    grads += [sg["w_ip"][:Hi], sg["b1"], sg["w2"].view(1, -1), sg["b2"]] if imp_live else [None] * 4
    if no_agg:
        grads += [None] * (3 + len(LAYER_ORDER) * mc.trans_layers + 4)
    else:
        grads += [sg["w_ip"][Hi:], sg["bp"], sg["special"]]
        for l, g in enumerate(tg["layers"]):
            grads += [g[key] for _, key in LAYER_ORDER]
        grads += [tg["lnfg"], tg["lnfb"]]
        grads += [tg["wcls"], tg["bcls"]] if has_logits else [None, None]
    return grads


def aggregator_params(agg) -> List[torch.nn.Parameter]:
    """Live parameters of a standalone TransformerAggregator in the order AggregatorFn returns their gradients."""
    dec = agg.transformer.decoder
    out = []
    for lyr in dec.layers:
        sd = dict(lyr.named_parameters())
        out += [sd[name] for name, _ in LAYER_ORDER]
    return out + [dec.norm.weight, dec.norm.bias]


class AggregatorFn(torch.autograd.Function):
    """``TransformerAggregator.forward`` on its own (reference model/aggregator.py:58-76) under autograd: tokens [B, T, d] (special
    token already prepended) -> token 0 of the decoder stack's output [B, d]; forward / backward are the launch sequences the levels
    use (paths_amd/backward.py:transformer_forward_train / transformer_backward)."""

    @staticmethod
    def forward(ctx, agg, tokens, num_ims, *params):
        mc, vp = agg._geometry(), ops.pack_aggregator(agg)
        drop = None
        if agg.training and agg.dropout_p > 0:
            drop = bw.Drop(agg.dropout_p, next_dropout_seed(tokens.device), 0)
        tr = bw.transformer_forward_train(mc, vp, tokens.detach(), num_ims, None, drop, None)
        ctx.agg = agg
        _, ctx.tr = _detached_saves({}, tr)
        ctx.set_materialize_grads(False)
        return tr["ctx_out"]

    @staticmethod
    def backward(ctx, d_out):
        agg, tr = ctx.agg, ctx.tr
        ctx.tr = None
        if d_out is None:
            return (None,) * (3 + len(aggregator_params(agg)))
        mc, vp = agg._geometry(), ops.pack_aggregator(agg)
        with bw.deferred_reductions():
            tg, d_tok, _ = bw.transformer_backward(mc, vp, tr, None, d_out.contiguous())
        grads = []
        for g in tg["layers"]:
            grads += [g[key] for _, key in LAYER_ORDER]
        grads += [tg["lnfg"], tg["lnfb"]]
        return (None, d_tok, None, *grads)


class GatherParentFn(torch.autograd.Function):
    """Children of the kept patches in the once-per-parent form: gathered feature rows (not differentiable), the children's inherited
    memory cell c0 [B,n_next,Hc] and the kept parents' h rows h_kept [B*cap, D].  Backward: the parents' state gradient =
    (scatter of d_h_kept | sum over the surviving children of d_c0), fixed order."""

    @staticmethod
    def forward(ctx, state_cur, grid_ptrs, src_cell, src_row, num_next, keep_idx, keep_count, child_pos, D, n_next):
        B, n_cur, Dp = state_cur.shape
        Hc = Dp - D
        cap = keep_idx.shape[1]
        f32 = dict(device=state_cur.device, dtype=torch.float32)
        fts_next = torch.empty((B, n_next, D), **f32)
        c0 = torch.empty((B, n_next, Hc), **f32)
        h_kept = torch.empty((B * cap, D), **f32)
        p = _lib.ptr
        _lib.call("paths_gather_rows", p(grid_ptrs), p(src_cell), D, state_cur.data_ptr() + 4 * D, n_cur, Dp, p(src_row), Hc, p(num_next), B,
                  n_next, p(fts_next), p(c0), 1, None, None, _lib.stream())
        _lib.call("paths_gather_kept_rows", p(state_cur), n_cur, Dp, p(keep_idx), cap, p(keep_count), D, B, p(h_kept), _lib.stream())
        ctx.meta = (keep_idx, keep_count, child_pos, n_cur, n_next, Dp, D, B)
        ctx.mark_non_differentiable(fts_next)
        return fts_next, c0, h_kept

    @staticmethod
    def backward(ctx, _d_fts, d_c0, d_hk):
        keep_idx, keep_count, child_pos, n_cur, n_next, Dp, D, B = ctx.meta
        cap = keep_idx.shape[1]
        dev = (d_c0 if d_c0 is not None else d_hk).device
        d_cur = torch.zeros((B, n_cur, Dp), device=dev, dtype=torch.float32)
        p = _lib.ptr
        if d_c0 is not None:
            _lib.call("paths_sibling_sum", p(keep_idx), cap, p(keep_count), p(child_pos), p(d_c0.contiguous()), n_next, Dp - D, Dp - D,
                      d_cur.data_ptr() + 4 * D, n_cur, Dp, B, _lib.stream())
        if d_hk is not None:
            _lib.call("paths_scatter_kept_rows", p(d_hk.contiguous()), cap, D, p(keep_idx), p(keep_count), p(d_cur), n_cur, Dp, D, B, _lib.stream())
        return (d_cur,) + (None,) * 9


def level_params_nolstm(proc) -> List[torch.nn.Parameter]:
    """lstm = false: the level's live parameters incl. its hctx_mlp, in the order LevelFnNoLstm returns their gradients."""
    return [proc.importance_mlp[0].weight, proc.importance_mlp[0].bias, proc.importance_mlp[2].weight, proc.importance_mlp[2].bias,
            proc.hctx_mlp[0].weight, proc.hctx_mlp[0].bias, proc.hctx_mlp[2].weight, proc.hctx_mlp[2].bias] + level_params(proc)[4:]


class LevelFnNoLstm(torch.autograd.Function):
    """One level of the lstm = false variant (reference model/paths.py:95-109): same transformer as LevelFn, the selection chain
    is alpha * X + hctx_mlp(previous Z) (paths_amd/backward.py:selection_forward_train_nolstm)."""

    @staticmethod
    def forward(ctx, proc, fts, locs, num_ims, state_prev, ctx_prev, *params):
        mc = proc.config
        check_dropout_supported(proc)
        vp = ops.pack_level(proc)
        if state_prev is not None:
            state_prev = state_prev.contiguous()
        sel = bw.selection_forward_train_nolstm(mc, vp, fts, locs.contiguous(), num_ims.contiguous(), state_prev)
        res = ctx_prev if (mc.slide_ctx_mode == "residual" and ctx_prev is not None) else None
        cat = ctx_prev.contiguous() if (mc.slide_ctx_mode == "concat" and ctx_prev is not None and ctx_prev.shape[1] > 0) else None
        drop = None
        if proc.training and mc.dropout > 0:
            drop = bw.Drop(mc.dropout, next_dropout_seed(fts.device), proc.depth)
        tr = bw.transformer_forward_train(mc, vp, sel["tokens"], sel["num_ims"], res, drop, cat)
        ctx.proc = proc
        ctx.sel, ctx.tr = _detached_saves(sel, tr)
        ctx.has_state, ctx.has_ctx = state_prev is not None, (res is not None or cat is not None)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(sel["importance"])
        return tr["logits"], tr["ctx_out"], sel["state_out"], sel["importance"]

    @staticmethod
    def backward(ctx, d_logits, d_ctx_out, d_state_out, _d_imp):
        proc, sel, tr = ctx.proc, ctx.sel, ctx.tr
        ctx.sel = ctx.tr = None
        mc = proc.config
        vp = ops.pack_level(proc)
        cont = lambda t: t.contiguous() if t is not None else None
        no_agg = d_logits is None and d_ctx_out is None          # (see LevelFn.backward; here Z = alpha X + hctx still reaches the next level)
        with bw.deferred_reductions():
            if no_agg:
                tg, d_ctx_prev = None, None
                d_tok = torch.zeros_like(tr["tokens"])
            else:
                tg, d_tok, d_ctx_prev = bw.transformer_backward(mc, vp, tr, cont(d_logits), cont(d_ctx_out))
            sg, d_state_prev = bw.selection_backward_nolstm(mc, vp, sel, d_tok, cont(d_state_out))
        v = lambda t, shape: t.view(shape) if t is not None else None
        grads = [sg["w1"], sg["b1"], v(sg["w2"], (1, -1)), sg["b2"], sg["wh1"], sg["bh1"], sg["wh2"], sg["bh2"]]
        if no_agg:
            grads += [None] * (3 + len(LAYER_ORDER) * mc.trans_layers + 4)
        else:
            grads += [sg["wp"], sg["bp"], sg["special"]]
            for l, g in enumerate(tg["layers"]):
                grads += [g[key] for _, key in LAYER_ORDER]
            grads += [tg["lnfg"], tg["lnfb"]]
            grads += [tg["wcls"], tg["bcls"]] if d_logits is not None else [None, None]
        return (None, None, None, None, d_state_prev if (ctx.has_state and d_state_prev is not None) else None,
                d_ctx_prev if ctx.has_ctx else None, *grads)


def level_apply(proc, lstm, fts, locs, num_ims, state_prev, ctx_prev):
    """Differentiable ``process``: returns (logits, ctx_slide, ctx_patch, importance)."""
    if not proc.config.lstm:
        return LevelFnNoLstm.apply(proc, fts, locs, num_ims, state_prev, ctx_prev, *level_params_nolstm(proc))
    return LevelFn.apply(proc, lstm, fts, locs, num_ims, state_prev, ctx_prev, *lstm_params(lstm), *level_params(proc))


class GatherFn(torch.autograd.Function):
    """fts_next, state_next = gather(next-level grid rows, parent state rows)  (paths_gather_rows, zero padded)."""

    @staticmethod
    def forward(ctx, state_cur, grid_ptrs, src_cell, src_row, num_next, keep_idx, keep_count, child_pos, D, n_next):
        B, n_cur, Dp = state_cur.shape
        f32 = dict(device=state_cur.device, dtype=torch.float32)
        fts_next = torch.empty((B, n_next, D), **f32)
        state_next = torch.empty((B, n_next, Dp), **f32)
        p = _lib.ptr
        _lib.call("paths_gather_rows", p(grid_ptrs), p(src_cell), D, p(state_cur), n_cur, Dp, p(src_row), Dp, p(num_next), B,
                  n_next, p(fts_next), p(state_next), 1, None, None, _lib.stream())
        ctx.meta = (keep_idx, keep_count, child_pos, n_cur, n_next, Dp, B)
        ctx.mark_non_differentiable(fts_next)
        return fts_next, state_next

    @staticmethod
    def backward(ctx, _d_fts, d_state_next):
        keep_idx, keep_count, child_pos, n_cur, n_next, Dp, B = ctx.meta
        d_cur = torch.zeros((B, n_cur, Dp), device=d_state_next.device, dtype=torch.float32)
        p = _lib.ptr
        _lib.call("paths_gather_rows_bwd", p(keep_idx), keep_idx.shape[1], p(keep_count), p(child_pos),
                  p(d_state_next.contiguous()), n_next, Dp, p(d_cur), n_cur, B, _lib.stream())
        return (d_cur,) + (None,) * 9
