"""Shared test helpers: rebuild fixture inputs from seeds (paths_amd.synthetic) and drive the oracle."""
from __future__ import annotations

import numpy as np
import torch

from oracle import paths_oracle as orc
from paths_amd import synthetic as syn


def oracle_config(cfg_over=None, **kw) -> orc.OracleConfig:
    """models/sample/config.json (reference models/sample/config.json:26-43) + overrides."""
    c = orc.OracleConfig()
    over = dict(cfg_over or {})
    for k, v in over.pop("model_config", {}).items():
        setattr(c, k, v)
    if "filter_to_subtypes" in over:
        c.num_subtypes = len(over.pop("filter_to_subtypes"))
    for k, v in over.items():
        setattr(c, k, v)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def oracle_params(cfg: orc.OracleConfig, wseed: int):
    shapes = orc.state_dict_shapes(cfg)
    sd = syn.make_state_dict(wseed, shapes)
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def level_inputs(dseed, depth, B, N, num_ims, D, d, pd_dim, patch_size):
    """Same construction as tools/make_goldens.py:level_inputs (inputs are functions of seeds only)."""
    fts = np.zeros((B, N, D), np.float32)
    locs = np.zeros((B, N, 2), np.int64)
    ctx_patch = np.zeros((B, N, depth, pd_dim), np.float32)
    side = int(np.ceil(np.sqrt(N))) << depth
    for b in range(B):
        n = int(num_ims[b])
        order = np.argsort(syn.fmix32(np.arange(side * side, dtype=np.uint64) + np.uint64(dseed * 977 + b)), kind="stable")[:n]
        x, y = order // side, order % side
        fts[b, :n] = syn.cell_features(dseed, b, depth, x, y, D, 0.0)
        locs[b, :n, 0], locs[b, :n, 1] = x * patch_size, y * patch_size
        if depth:
            ctx_patch[b, :n] = syn.uniform_tensor(dseed, f"ctx_patch.{b}", (n, depth, pd_dim), 0.5)
    ctx_slide = syn.uniform_tensor(dseed, "ctx_slide", (B, depth, d), 1.0)
    return {"fts": fts, "locs": locs, "num_ims": np.asarray(num_ims, np.int64),
            "parent_inds": np.zeros((B, N), np.int64), "ctx_slide": ctx_slide, "ctx_patch": ctx_patch}


def single_level_inputs(info, cfg):
    pd_dim = cfg.patch_embed_dim + (cfg.hierarchical_ctx_mlp_hidden_dim if cfg.lstm else 0)
    return level_inputs(info["dseed"], info["depth"], info["B"], info["N"], info["num_ims"],
                        cfg.patch_embed_dim, cfg.trans_dim, pd_dim, cfg.patch_size)


def synthetic_slides(info, cfg):
    return [syn.SyntheticSlide(info["dseed"], sid, tuple(info["base_shape"]), cfg.patch_embed_dim,
                               cfg.num_levels, info["p_bg"]) for sid in info["slide_ids"]]


def set_agreement(a, b):
    return np.array_equal(np.sort(np.asarray(a)), np.sort(np.asarray(b)))


def sequence_inversions(keep_a, keep_b, scores):
    """Largest score gap among positions where two index sequences differ (0.0 if identical)."""
    keep_a, keep_b = np.asarray(keep_a), np.asarray(keep_b)
    diff = keep_a != keep_b
    if not diff.any():
        return 0.0
    s = np.asarray(scores, dtype=np.float64)
    return float(np.abs(s[keep_a[diff]] - s[keep_b[diff]]).max())


def golden_trace(g, B, L):
    """The reference's per-level records of a recursion fixture (G3/G4) in the oracle's trace format (oracle/compare.py)."""
    tr = []
    for l in range(L):
        tr.append({"num_ims": g[f"L{l}_num_ims"], "locs": g[f"L{l}_locs"], "parent_inds": g[f"L{l}_parent_inds"],
                   "importance": g[f"L{l}_importance"], "logits": g[f"L{l}_logits"],
                   "keep_inds": [g[f"L{l}_keep_{j}"] for j in range(B)] if l < L - 1 else []})
    return tr


import contextlib


@contextlib.contextmanager
def spy_calls():
    """Names of the C entry points (paths_amd._lib.call) a block of code goes through."""
    from paths_amd import _lib
    calls = []
    orig = _lib.call

    def spy(cname, *a):
        calls.append(cname)
        return orig(cname, *a)

    _lib.call = spy
    try:
        yield calls
    finally:
        _lib.call = orig
