"""``PATHSProcessor`` — one magnification level P_i (reference model/paths.py:12-151), MI355X-native.

Same constructor / ``process(data, lstm=None)`` / ``ctx_dim()`` surface and state_dict keys as the reference;
``process`` issues the HIP launch sequence of paths_amd/ops.py:level_forward and returns the reference's
dict {"logits", "ctx_slide", "ctx_patch", "importance"} (model/paths.py:141-146).

Limits (rejected loudly, never silently approximated): the aggregator kernels are built for trans_dim=128 / 4 heads / importance hidden
128 (paths_amd/ops.py:check_supported).  Dropout > 0 is active in train mode (counter-based masks, paths_amd/backward.py:Drop).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
from torch import nn

from .. import ops
from .aggregator import TransformerAggregator
from .interface import Processor


def _lib_absmax(fts: torch.Tensor, state_prev) -> float:
    from .. import _lib
    m = _lib.absmax(fts)
    if state_prev is not None and state_prev.numel():
        sp = state_prev.detach().float().contiguous()
        m = max(m, _lib.absmax(sp))
    return m


class PATHSProcessor(nn.Module, Processor):
    def __init__(self, config, train_config, depth: int):
        super().__init__()
        self.depth = depth
        self.config = config
        self.train_config = train_config
        num_logits = train_config.nbins if train_config.task == "survival" else len(train_config.filter_to_subtypes)
        self.dim = config.patch_embed_dim
        self.slide_ctx_dim = config.trans_dim
        cls_in = self.slide_ctx_dim * (depth + 1) if config.slide_ctx_mode == "concat" else self.slide_ctx_dim
        self.classification_layer = nn.Linear(cls_in, num_logits)
        self.importance_mlp = nn.Sequential(nn.Linear(self.dim, config.importance_mlp_hidden_dim), nn.ReLU(),
                                            nn.Linear(config.importance_mlp_hidden_dim, 1))
        if config.lstm:
            self.hdim = config.hierarchical_ctx_mlp_hidden_dim
        else:
            self.hctx_mlp = nn.Sequential(nn.Linear(self.dim, config.hierarchical_ctx_mlp_hidden_dim), nn.ReLU(),
                                          nn.Linear(config.hierarchical_ctx_mlp_hidden_dim, self.dim))
        self.global_agg = TransformerAggregator(input_dim=self.dim, model_dim=config.trans_dim, output_dim=self.dim,
                                                nhead=config.trans_heads, layers=config.trans_layers, dropout=config.dropout)

    def process(self, data, lstm=None, skip_padding: bool = False) -> Dict[str, torch.Tensor]:
        mc = self.config
        ops.check_supported(mc)
        assert lstm is not None or not mc.lstm, "lstm=True needs the shared LSTMCell (RecursiveModel passes it)"
        if self.training and mc.dropout > 0 and not torch.is_grad_enabled():
            raise NotImplementedError("dropout > 0 in train mode is implemented on the differentiable path only: call model.eval() "
                                      "for inference, or run under autograd (torch.enable_grad) for training")
        if mc.patch_embed_dim % 4 == 0 and (ops.GEMM_MODE == "h3" or ops.TRAIN_FWD_PLANES == 2):
            # drop-in batches come from the caller: reduce max|x| on entry (one host sync; the reference's own PatchBatch
            # constructor syncs on num_ims.max(), data_utils/patch_batch.py:50) and keep out-of-range data off the fp16 split
            f32 = data.fts if (data.fts.dtype == torch.float32 and data.fts.is_contiguous()) else data.fts.float().contiguous()
            amax = _lib_absmax(f32, data.ctx_patch[:, :, -1] if self.depth > 0 else None)
            with ops.range_guard(amax):
                return self._process(data, lstm, skip_padding)
        return self._process(data, lstm, skip_padding)

    def _process(self, data, lstm, skip_padding: bool) -> Dict[str, torch.Tensor]:
        mc = self.config
        if torch.is_grad_enabled():
            # differentiable path (training): same kernels + saved activations, backward in HIP (paths_amd/autograd.py)
            from .. import autograd as pag
            fts = data.fts.float().contiguous()
            state_prev = data.ctx_patch[:, :, -1] if self.depth > 0 else None
            if state_prev is not None and (state_prev.stride(2) != 1 or state_prev.stride(1) % 4 or
                                           state_prev.stride(0) != fts.shape[1] * state_prev.stride(1) or state_prev.data_ptr() % 16):
                state_prev = state_prev.contiguous()
            ctx_prev = data.ctx_slide[:, -1] if (mc.slide_ctx_mode == "residual" and data.ctx_depth > 0) else None
            if ctx_prev is not None and ctx_prev.stride(1) != 1:
                ctx_prev = ctx_prev.contiguous()
            if mc.slide_ctx_mode == "concat" and data.ctx_depth > 0:
                ctx_prev = data.ctx_slide.float().contiguous()        # [B, depth, d]: all previous slide contexts
            logits, ctx_slide, ctx_patch, importance = pag.level_apply(self, lstm, fts, data.locs, data.num_ims, state_prev, ctx_prev)
            return {"logits": logits, "ctx_slide": ctx_slide, "ctx_patch": ctx_patch, "importance": importance}
        fts = data.fts
        if fts.dtype != torch.float32 or not fts.is_contiguous():
            fts = fts.float().contiguous()
        B, N, D = fts.shape
        assert D == self.dim
        state_prev = None
        if self.depth > 0:
            assert data.ctx_patch.dim() == 4 and data.ctx_patch.shape[-1] == self.ctx_dim()[1]
            state_prev = data.ctx_patch[:, :, -1]                     # strided view, read in place by the kernel
            if state_prev.stride(2) != 1 or state_prev.dtype != torch.float32 or state_prev.stride(1) % 4 or \
                    state_prev.stride(0) != N * state_prev.stride(1) or state_prev.data_ptr() % 16:
                state_prev = state_prev.float().contiguous()
        ctx_prev = data.ctx_slide[:, -1] if (mc.slide_ctx_mode == "residual" and data.ctx_depth > 0) else None
        if ctx_prev is not None and (ctx_prev.stride(1) != 1 or ctx_prev.dtype != torch.float32):
            ctx_prev = ctx_prev.float().contiguous()
        ctx_all = data.ctx_slide.float().contiguous() if mc.slide_ctx_mode == "concat" else None
        with torch.no_grad():
            return ops.level_forward(mc, ops.pack_lstm(lstm) if mc.lstm else None, ops.pack_level(self), fts, data.locs, data.num_ims,
                                     state_prev, ctx_prev, ctx_all, skip_padding)

    def ctx_dim(self) -> Tuple[int, int]:
        if self.config.lstm:
            return self.slide_ctx_dim, self.dim + self.hdim
        return self.slide_ctx_dim, self.dim
