"""ORACLE — test infrastructure only.  NOT part of the product path.

A CPU restatement, in plain fp32 ``torch`` CPU ops, of the PATHS hot path (one magnification level
of ``PATHSProcessor.process`` and the recursive top-K driver ``inference_end2end``).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
``paths_amd`` never imports anything under ``oracle/``.

Pinning: the reference ships no tests or golden vectors for this path, so this file
is pinned against outputs of the reference itself, imported in the build container by
``tools/make_goldens.py``; the captured vectors live in ``tests/golden/*.npz`` and are checked by
``tests/test_oracle_golden.py``.

Every function cites the reference file:line (relative to the reference repository root) it
restates.  The arithmetic of attention / LayerNorm / top-k lives in the third-party dependency
``torch`` (reference pins pytorch=2.1.0, environment.yml:11); the documented semantics of
``nn.TransformerDecoderLayer`` (post-LN, relu, eps=1e-5, in_proj rows = [Wq;Wk;Wv], scale
1/sqrt(head_dim)) are restated explicitly below.

Parameters are a flat ``dict`` keyed exactly like the reference ``state_dict``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class OracleConfig:
    """The subset of config.json the path reads (reference config.py:19-37, 41-79)."""
    patch_embed_dim: int = 1024
    trans_dim: int = 128
    trans_heads: int = 4
    trans_layers: int = 2
    importance_mlp_hidden_dim: int = 128
    hierarchical_ctx_mlp_hidden_dim: int = 256
    patch_size: int = 256
    lstm: bool = True
    hierarchical_ctx: bool = True
    slide_ctx_mode: str = "residual"
    importance_mode: str = "mul"
    pos_encoding_mode: str = "2d"
    num_levels: int = 5
    top_k_patches: List[int] = field(default_factory=lambda: [20, 20, 20, 20])
    task: str = "survival"
    nbins: int = 4
    num_subtypes: int = 2


# ----------------------------------------------------------------------------------------------
# LSTM cell over depth                                             reference model/interface.py:31-58
# ----------------------------------------------------------------------------------------------
def lstm_cell(p: Dict[str, Tensor], xs: Tensor, hs: Tensor, cs: Tensor):
    xh = torch.cat((xs, hs), dim=-1)                                           # interface.py:49

    def lin(name):
        return F.linear(xh, p[f"lstm.{name}.0.weight"], p[f"lstm.{name}.0.bias"])

    cs = cs * torch.sigmoid(lin("forget_gate"))                                # interface.py:52
    cs = cs + torch.sigmoid(lin("remember_gate")) * torch.tanh(lin("remember_map"))   # :53
    hs = torch.sigmoid(lin("out_select_gate")) * torch.tanh(
        F.linear(cs, p["lstm.mem_to_out.0.weight"], p["lstm.mem_to_out.0.bias"]))     # :56
    return hs, cs


# ----------------------------------------------------------------------------------------------
# positional encodings                                                      reference utils.py:16-23, 47-67
# ----------------------------------------------------------------------------------------------
def positional_encoding(length: int, dim: int, k: float = 10000.0) -> Tensor:
    position = torch.arange(length).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2) * (-math.log(k) / dim))
    pe = torch.zeros(length, dim)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def positional_encoding_2d_from_pos(xpos: Tensor, ypos: Tensor, dim: int, k: float = 10000.0) -> Tensor:
    n = xpos.shape[0]
    div_term = torch.exp(torch.arange(0, dim // 2, 2) * (-math.log(k) / dim))[None]
    pe = torch.zeros(n, dim)
    pe[:, 0:dim // 2:2] = torch.sin(xpos[:, None] * div_term)
    pe[:, 1:dim // 2:2] = torch.cos(xpos[:, None] * div_term)
    pe[:, dim // 2::2] = torch.sin(ypos[:, None] * div_term)
    pe[:, dim // 2 + 1::2] = torch.cos(ypos[:, None] * div_term)
    return pe


# ----------------------------------------------------------------------------------------------
# nn.Transformer decoder stack over a length-0 memory           reference model/aggregator.py:25-33,70-72
# ----------------------------------------------------------------------------------------------
def _self_attention(S: Tensor, w_in: Tensor, b_in: Tensor, w_out: Tensor, b_out: Tensor,
                    nhead: int, key_pad: Tensor) -> Tensor:
    B, T, d = S.shape
    hd = d // nhead
    qkv = F.linear(S, w_in, b_in)                                  # rows of w_in = [Wq; Wk; Wv]
    q, k, v = qkv.split(d, dim=-1)
    q = q.view(B, T, nhead, hd).transpose(1, 2)
    k = k.view(B, T, nhead, hd).transpose(1, 2)
    v = v.view(B, T, nhead, hd).transpose(1, 2)
    scores = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))     # [B,h,T,T]
    scores = scores.masked_fill(key_pad[:, None, None, :], float("-inf"))
    attn = torch.softmax(scores, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, T, d)
    return F.linear(out, w_out, b_out)


def decoder_stack(p: Dict[str, Tensor], prefix: str, S: Tensor, key_pad: Tensor, nhead: int,
                  layers: int, eps: float = 1e-5) -> Tensor:
    """Post-LN ``nn.TransformerDecoder`` (+ final norm) with an EMPTY memory sequence.

    Cross-attention over zero keys yields a zero vector per query, so the block reduces to adding
    ``multihead_attn.out_proj.bias`` (verified against the reference import).
    """
    d = S.shape[-1]
    for l in range(layers):
        q = f"{prefix}.decoder.layers.{l}."
        sa = _self_attention(S, p[q + "self_attn.in_proj_weight"], p[q + "self_attn.in_proj_bias"],
                             p[q + "self_attn.out_proj.weight"], p[q + "self_attn.out_proj.bias"],
                             nhead, key_pad)
        S = F.layer_norm(S + sa, (d,), p[q + "norm1.weight"], p[q + "norm1.bias"], eps)
        S = F.layer_norm(S + p[q + "multihead_attn.out_proj.bias"], (d,),
                         p[q + "norm2.weight"], p[q + "norm2.bias"], eps)
        ff = F.linear(torch.relu(F.linear(S, p[q + "linear1.weight"], p[q + "linear1.bias"])),
                      p[q + "linear2.weight"], p[q + "linear2.bias"])
        S = F.layer_norm(S + ff, (d,), p[q + "norm3.weight"], p[q + "norm3.bias"], eps)
    return F.layer_norm(S, (d,), p[f"{prefix}.decoder.norm.weight"], p[f"{prefix}.decoder.norm.bias"], eps)


# ----------------------------------------------------------------------------------------------
# one magnification level                                         reference model/paths.py:66-146
# ----------------------------------------------------------------------------------------------
def process_level(p: Dict[str, Tensor], cfg: OracleConfig, depth: int, fts: Tensor, locs: Tensor,
                  num_ims: Tensor, ctx_slide: Tensor, ctx_patch: Tensor, probe: Optional[dict] = None) -> Dict[str, Tensor]:
    """``RecursiveModel.forward(depth, PatchBatch)`` (interface.py:96-99 → paths.py:66-146).

    fts [B,N,D] (padded rows zero), locs [B,N,2] int64 pixel coords, num_ims [B] int64,
    ctx_slide [B,depth,d], ctx_patch [B,N,depth,Dp].
    """
    B, N, D = fts.shape
    d = cfg.trans_dim
    pre = f"procs.{depth}."
    valid = torch.arange(N)[None, :] < num_ims[:, None]            # patch_batch.py:64-68
    X = fts
    if cfg.lstm:
        Hc = cfg.hierarchical_ctx_mlp_hidden_dim
        if depth == 0:                                             # paths.py:78-80
            hs = torch.zeros(B, N, D)
            cs = torch.zeros(B, N, Hc)
        else:                                                      # paths.py:83-86
            st = ctx_patch[:, :, -1]
            hs, cs = st[..., :D], st[..., D:]
        hs, cs = lstm_cell(p, X, hs, cs)                           # paths.py:88
        Y = X + hs                                                 # paths.py:89
        patch_ctx = torch.cat((hs, cs), dim=-1)                    # paths.py:91
    else:
        Y = X

    # importance: MLP + sigmoid on valid rows, 0 on padding       paths.py:95, utils.py:106-115
    a = F.linear(torch.relu(F.linear(Y[valid], p[pre + "importance_mlp.0.weight"], p[pre + "importance_mlp.0.bias"])),
                 p[pre + "importance_mlp.2.weight"], p[pre + "importance_mlp.2.bias"])
    imp = torch.zeros(B, N)
    imp[valid] = torch.sigmoid(a)[:, 0]
    Z = Y * imp[..., None] if cfg.importance_mode == "mul" else Y  # paths.py:96-98

    if not cfg.lstm:                                               # paths.py:101-109
        if depth > 0 and cfg.hierarchical_ctx:
            hctx_in = ctx_patch[:, :, -1]
            h = torch.zeros(B, N, D)
            h[valid] = F.linear(torch.relu(F.linear(hctx_in[valid], p[pre + "hctx_mlp.0.weight"], p[pre + "hctx_mlp.0.bias"])),
                                p[pre + "hctx_mlp.2.weight"], p[pre + "hctx_mlp.2.bias"])
            Z = Z + h
        patch_ctx = Z

    # proj_in + positional encoding                               paths.py:119-124, aggregator.py:37-56
    g = pre + "global_agg."
    if cfg.pos_encoding_mode == "1d":
        t = F.linear(Z, p[g + "proj_in.weight"], p[g + "proj_in.bias"]) + positional_encoding(N, d)[None]
    elif cfg.pos_encoding_mode == "2d":
        pl = torch.div(locs, cfg.patch_size, rounding_mode="floor")
        t = F.linear(Z, p[g + "proj_in.weight"], p[g + "proj_in.bias"])
        t = t + positional_encoding_2d_from_pos(pl[..., 0].reshape(-1), pl[..., 1].reshape(-1), d).view(B, N, d)
    else:
        raise RuntimeError("reference raises a size mismatch for other pos_encoding_mode values (SURVEY §3.3)")

    if probe is not None:                                          # (tests: the aggregator's input sequence / its raw output)
        probe["xs"] = t
    # special token + key padding mask + decoder stack            aggregator.py:58-76, utils.py:97-103
    S = torch.cat((p[g + "special_token"].view(1, 1, -1).repeat(B, 1, 1), t), dim=1)
    key_pad = torch.arange(N + 1)[None, :] >= (num_ims + 1)[:, None]
    S = decoder_stack(p, g + "transformer", S, key_pad, cfg.trans_heads, cfg.trans_layers)
    agg = S[:, 0]                                                  # aggregator.py:75
    if probe is not None:
        probe["agg"] = agg

    if cfg.slide_ctx_mode == "residual" and ctx_slide.shape[1] > 0:   # paths.py:130-131
        agg = agg + ctx_slide[:, -1]
    if cfg.slide_ctx_mode == "concat":                             # paths.py:134-137
        ft = torch.cat((torch.flatten(ctx_slide, start_dim=1), agg), dim=1)
    else:
        ft = agg
    logits = F.linear(ft, p[pre + "classification_layer.weight"], p[pre + "classification_layer.bias"])
    return {"logits": logits, "ctx_slide": agg, "ctx_patch": patch_ctx, "importance": imp}


# ----------------------------------------------------------------------------------------------
# top-K + child expansion + gather for ONE slide                  reference data_utils/slide.py:277-360
# ----------------------------------------------------------------------------------------------
class DenseGrids:
    """Per-level dense feature grids [X,Y,D] (the reference's in-RAM representation)."""

    def __init__(self, grids: Sequence[Tensor]):
        self.grids = list(grids)

    def shape(self, level: int):
        return self.grids[level].shape[0], self.grids[level].shape[1]

    def rows(self, level: int, x: Tensor, y: Tensor) -> Tensor:
        return self.grids[level][x, y]


class LazyGrids:
    """Grids evaluated on demand from a ``paths_amd.synthetic.SyntheticSlide`` (no dense storage)."""

    def __init__(self, slide):
        self.slide = slide

    def shape(self, level: int):
        return self.slide.shape(level)

    def rows(self, level: int, x: Tensor, y: Tensor) -> Tensor:
        return torch.from_numpy(self.slide.rows(level, x.numpy(), y.numpy()))


def topk_indices(imp: Tensor, count: int) -> Tensor:
    """slide.py:298 — ``torch.topk(imp_cpu, count).indices`` (descending value)."""
    return torch.topk(imp, count).indices


def iter_slide(grids, level: int, npatches: int, locs_px: Tensor, ctx_slide: Tensor, ctx_patch: Tensor,
               new_ctx_slide: Tensor, new_ctx_patch: Tensor, imp: Tensor, keep: int, patch_size: int):
    """One slide, level ``level`` → ``level+1``.  Returns the next-level item dict + kept indices."""
    locs = torch.div(locs_px, patch_size, rounding_mode="floor")[:npatches]   # slide.py:280,288
    ctx_patch = ctx_patch[:npatches]
    new_ctx_patch = new_ctx_patch[:npatches]
    imp = imp[:npatches]
    ctx_slide = torch.cat((ctx_slide, new_ctx_slide[None]), dim=0)            # slide.py:291
    ctx_patch = torch.cat((ctx_patch, new_ctx_patch[:, None]), dim=1)         # slide.py:292
    keep_inds = torch.arange(npatches)
    if keep != -1:                                                            # slide.py:294-301
        count = min(npatches, keep)
        keep_inds = topk_indices(imp, count)
        ctx_patch = ctx_patch[keep_inds]
        locs = locs[keep_inds]
    n = locs.shape[0]
    base = locs * 2                                                           # slide.py:307
    parent_inds = torch.arange(n).repeat(4)                                   # slide.py:311
    new_locs = torch.cat((base, base + torch.tensor([0, 1]), base + torch.tensor([1, 0]),
                          base + torch.tensor([1, 1])), dim=0)                # slide.py:312-315
    ctx_patch = torch.cat((ctx_patch,) * 4, dim=0)                            # slide.py:318
    X, Y = grids.shape(level + 1)
    in_bound = (new_locs[:, 0] < X) & (new_locs[:, 1] < Y)                    # slide.py:322
    safe = new_locs * in_bound[:, None]
    rows = grids.rows(level + 1, safe[:, 0], safe[:, 1])
    flt = in_bound & (rows.sum(dim=1) != 0)                                   # slide.py:324-325
    new_locs, parent_inds, ctx_patch, new_fts = new_locs[flt], parent_inds[flt], ctx_patch[flt], rows[flt]
    fallback = new_locs.shape[0] == 0
    if fallback:                                                              # slide.py:336-352 (rare fallback)
        gx, gy = torch.meshgrid(torch.arange(X), torch.arange(Y), indexing="ij")
        new_locs = torch.stack((gx.reshape(-1), gy.reshape(-1)), dim=1)
        rows = grids.rows(level + 1, new_locs[:, 0], new_locs[:, 1])
        flt = rows.sum(dim=1) != 0
        if int(flt.count_nonzero()) == 0:
            flt[:] = True
        ctx_patch = torch.zeros((X * Y, ctx_patch.shape[1], ctx_patch.shape[2]))[flt]
        parent_inds = torch.arange(X * Y)[flt]
        new_locs, new_fts = new_locs[flt], rows[flt]
    item = {"fts": new_fts, "ctx_patch": ctx_patch, "ctx_slide": ctx_slide,
            "locs": new_locs * patch_size, "parent_inds": parent_inds,
            "fallback": fallback}         # (trace only: parent_inds are then CELL indices of the new level, not indices into the kept list)
    return item, keep_inds


def collate(items: List[Dict[str, Tensor]]) -> Dict[str, Tensor]:
    """Zero-pad variable-length fields to the batch max (reference data_utils/dataset.py:206-243)."""
    num = [it["locs"].shape[0] for it in items]
    mx = max(num)

    def padrows(t):
        out = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype)
        out[: t.shape[0]] = t
        return out

    return {
        "fts": torch.stack([padrows(it["fts"]) for it in items]),
        "locs": torch.stack([padrows(it["locs"]) for it in items]),
        "parent_inds": torch.stack([padrows(it["parent_inds"]) for it in items]),
        "ctx_patch": torch.stack([padrows(it["ctx_patch"]) for it in items]),
        "ctx_slide": torch.stack([it["ctx_slide"] for it in items]),
        "num_ims": torch.tensor(num, dtype=torch.int64),
        "fallback": [bool(it.get("fallback", False)) for it in items],
    }


def initial_item(grids, cfg: OracleConfig) -> Dict[str, Tensor]:
    """Level-0 item: every grid cell in row-major order (slide.py:257-269, 362-381); no bg filter."""
    X, Y = grids.shape(0)
    gx, gy = torch.meshgrid(torch.arange(X), torch.arange(Y), indexing="ij")
    locs = torch.stack((gx.reshape(-1), gy.reshape(-1)), dim=1)
    Dp = cfg.patch_embed_dim + (cfg.hierarchical_ctx_mlp_hidden_dim if cfg.lstm else 0)
    return {"fts": grids.rows(0, locs[:, 0], locs[:, 1]), "locs": locs * cfg.patch_size,
            "parent_inds": torch.arange(X * Y), "ctx_patch": torch.zeros(X * Y, 0, Dp),
            "ctx_slide": torch.zeros(0, cfg.trans_dim)}


# ----------------------------------------------------------------------------------------------
# loss                                                                        reference utils.py:283-305
# ----------------------------------------------------------------------------------------------
def nll_loss(hazards: Tensor, y: Tensor, c: Tensor, alpha: float = 0.4, eps: float = 1e-7) -> Tensor:
    B = hazards.shape[0]
    surv = torch.cumprod(1 - hazards, dim=1)
    surv_pad = torch.cat([torch.ones(B, 1, dtype=surv.dtype), surv], dim=1)
    r = torch.arange(B)
    unc = -(1 - c) * (torch.log(surv_pad[r, y].clamp(min=eps)) + torch.log(hazards[r, y].clamp(min=eps)))
    cen = -c * torch.log(surv_pad[r, y + 1].clamp(min=eps))
    return ((1 - alpha) * (cen + unc) + alpha * unc).mean()


# ----------------------------------------------------------------------------------------------
# the recursion                                                               reference utils.py:228-279
# ----------------------------------------------------------------------------------------------
def inference_end2end(p: Dict[str, Tensor], cfg: OracleConfig, slides_grids: Sequence, labels=None,
                      trace: Optional[list] = None):
    """Returns (hazards-or-logits [B,*], loss or None).  ``trace`` (if a list) receives per-level
    dicts with num_ims / locs / parent_inds / keep_inds / importance / logits for parity tests."""
    batch = collate([initial_item(g, cfg) for g in slides_grids])
    out = None
    for i in range(cfg.num_levels):
        out = process_level(p, cfg, i, batch["fts"], batch["locs"], batch["num_ims"],
                            batch["ctx_slide"], batch["ctx_patch"])
        rec = {"num_ims": batch["num_ims"].clone(), "locs": batch["locs"].clone(),
               "parent_inds": batch["parent_inds"].clone(), "importance": out["importance"].clone(),
               "logits": out["logits"].clone(), "ctx_slide": out["ctx_slide"].clone(), "keep_inds": [],
               "fallback": list(batch.get("fallback", [False] * len(slides_grids)))}
        if i != cfg.num_levels - 1:
            items = []
            for j, g in enumerate(slides_grids):
                item, keep_inds = iter_slide(g, i, int(batch["num_ims"][j]), batch["locs"][j], batch["ctx_slide"][j],
                                             batch["ctx_patch"][j], out["ctx_slide"][j], out["ctx_patch"][j],
                                             out["importance"][j], cfg.top_k_patches[i], cfg.patch_size)
                items.append(item)
                rec["keep_inds"].append(keep_inds)
            batch = collate(items)
        if trace is not None:
            trace.append(rec)
    logits = out["logits"]
    if cfg.task == "survival":
        hazards = torch.sigmoid(logits)
        loss = None
        if labels is not None:
            loss = nll_loss(hazards, labels["survival_bin"], labels["censored"])
        return hazards, loss
    loss = None
    if labels is not None:
        loss = F.cross_entropy(logits, labels["subtype"])
    return logits, loss


def state_dict_shapes(cfg: OracleConfig) -> Dict[str, tuple]:
    """Shapes of every tensor of the reference ``state_dict`` (SURVEY.md §8b checkpoint surface),
    including the dead encoder / cross-attention tensors that no kernel reads."""
    D, d, Hi, Hc, L = (cfg.patch_embed_dim, cfg.trans_dim, cfg.importance_mlp_hidden_dim,
                       cfg.hierarchical_ctx_mlp_hidden_dim, cfg.trans_layers)
    nlog = cfg.nbins if cfg.task == "survival" else cfg.num_subtypes
    s: Dict[str, tuple] = {}
    for i in range(cfg.num_levels):
        pre = f"procs.{i}."
        cin = d * (i + 1) if cfg.slide_ctx_mode == "concat" else d
        s[pre + "classification_layer.weight"] = (nlog, cin)
        s[pre + "classification_layer.bias"] = (nlog,)
        s[pre + "importance_mlp.0.weight"] = (Hi, D)
        s[pre + "importance_mlp.0.bias"] = (Hi,)
        s[pre + "importance_mlp.2.weight"] = (1, Hi)
        s[pre + "importance_mlp.2.bias"] = (1,)
        if not cfg.lstm:
            s[pre + "hctx_mlp.0.weight"] = (Hc, D)
            s[pre + "hctx_mlp.0.bias"] = (Hc,)
            s[pre + "hctx_mlp.2.weight"] = (D, Hc)
            s[pre + "hctx_mlp.2.bias"] = (D,)
        g = pre + "global_agg."
        s[g + "special_token"] = (d,)
        s[g + "proj_in.weight"] = (d, D)
        s[g + "proj_in.bias"] = (d,)
        t = g + "transformer."

        def attn(q):
            s[q + "in_proj_weight"] = (3 * d, d)
            s[q + "in_proj_bias"] = (3 * d,)
            s[q + "out_proj.weight"] = (d, d)
            s[q + "out_proj.bias"] = (d,)

        def ffn_norms(q, n):
            s[q + "linear1.weight"] = (4 * d, d)
            s[q + "linear1.bias"] = (4 * d,)
            s[q + "linear2.weight"] = (d, 4 * d)
            s[q + "linear2.bias"] = (d,)
            for j in range(1, n + 1):
                s[q + f"norm{j}.weight"] = (d,)
                s[q + f"norm{j}.bias"] = (d,)

        for l in range(L):
            q = t + f"encoder.layers.{l}."
            attn(q + "self_attn.")
            ffn_norms(q, 2)
        s[t + "encoder.norm.weight"] = (d,)
        s[t + "encoder.norm.bias"] = (d,)
        for l in range(L):
            q = t + f"decoder.layers.{l}."
            attn(q + "self_attn.")
            attn(q + "multihead_attn.")
            ffn_norms(q, 3)
        s[t + "decoder.norm.weight"] = (d,)
        s[t + "decoder.norm.bias"] = (d,)
    if cfg.lstm:
        for gate, (o, i_) in {"forget_gate": (Hc, 2 * D), "remember_gate": (Hc, 2 * D), "remember_map": (Hc, 2 * D),
                              "out_select_gate": (D, 2 * D), "mem_to_out": (D, Hc)}.items():
            s[f"lstm.{gate}.0.weight"] = (o, i_)
            s[f"lstm.{gate}.0.bias"] = (o,)
    return s
