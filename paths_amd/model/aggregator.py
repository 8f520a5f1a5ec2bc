"""``TransformerAggregator`` (reference model/aggregator.py:8-76), MI355X-native.

Keeps the reference's attribute names so the state_dict keys match
(``global_agg.{special_token, proj_in.*, transformer.{encoder,decoder}...}``).  ``nn.Transformer`` is
instantiated only as the owner of those tensors — including the dead encoder and cross-attention weights,
which must round-trip through checkpoints — its own forward is never called.

Inside ``PATHSProcessor.process`` the aggregation (proj_in + PE + special token + post-LN decoder stack over an
empty memory + token-0 read-out) is fused into the level's launch sequence (paths_amd/ops.py).  The module is
also callable on its own with the reference's signatures — ``forward(seq1, seq2, lengths1, lengths2)``,
``pos_encode_1d`` / ``pos_encode_2d`` — dispatching to the same HIP kernels (``ops._aggregator_forward``;
under autograd ``backward.transformer_forward_train`` / ``transformer_backward``).
"""
from __future__ import annotations

import types

import torch
from torch import nn


class TransformerAggregator(nn.Module):
    def __init__(self, input_dim: int, model_dim: int, output_dim: int, nhead: int, layers: int, dropout: float):
        super().__init__()
        self.dim = model_dim
        self.nhead = nhead
        self.num_layers = layers
        self.dropout_p = float(dropout)
        self.proj_in = nn.Linear(input_dim, model_dim)
        self.proj_out = nn.Identity()
        self.transformer = nn.Transformer(model_dim, nhead=nhead, num_encoder_layers=layers, num_decoder_layers=layers,
                                          dim_feedforward=model_dim * 4, dropout=dropout, batch_first=True)
        self.special_token = nn.Parameter(torch.randn(model_dim), requires_grad=True)

    # ---- geometry seen by the kernels' dispatcher (paths_amd/ops.py:fast_path / check_aggregator_geometry)
    def _geometry(self):
        return types.SimpleNamespace(trans_dim=self.dim, trans_heads=self.nhead, trans_layers=self.num_layers,
                                     importance_mlp_hidden_dim=128, slide_ctx_mode="none", dropout=self.dropout_p)

    def _encode(self, data: torch.Tensor, locs: torch.Tensor, pe_mode: int, project: bool) -> torch.Tensor:
        """proj_in (optional) + positional encoding on the HIP kernels: one GEMM + paths_tokens_assemble."""
        from .. import _lib, ops
        _lib.require_cuda(data, locs)
        if torch.is_grad_enabled() and (data.requires_grad or (project and self.proj_in.weight.requires_grad)):
            raise NotImplementedError("pos_encode_* is inference-only on its own; training runs through PATHSProcessor.process")
        B, N, Din = data.shape
        d = self.dim
        ops.check_aggregator_geometry(d, self.nhead)
        pk = ops.pack_aggregator(self)
        dev = data.device
        x = data.detach().float().contiguous().view(B * N, Din)
        with torch.no_grad():
            if project:
                assert Din == self.proj_in.in_features
                proj = torch.empty((B * N, d), device=dev, dtype=torch.float32)
                if "wp_pad" not in pk:
                    pk["wp_pad"] = ops._pad_rows(pk["wp"])
                ops.gemm_f32(x, Din, pk["wp_pad"], None, proj, d, B * N, d, Din)
                bias = pk["bp"]
            else:
                assert Din == d
                proj, bias = x, torch.zeros((d,), device=dev, dtype=torch.float32)
            tokens = torch.empty((B, N + 1, d), device=dev, dtype=torch.float32)
            ones = torch.ones((1,), device=dev, dtype=torch.float32)
            _lib.call("paths_tokens_assemble", proj.data_ptr(), d, ones.data_ptr(), 0, bias.data_ptr(), pk["special"].data_ptr(),
                      pk["div_2d" if pe_mode == 2 else "div_1d"].data_ptr(), locs.data_ptr(), N, 1, pe_mode, d, B, tokens.data_ptr(), _lib.stream())
        return tokens[:, 1:]

    def pos_encode_1d(self, xs, project=True):
        """reference model/aggregator.py:37-41: (proj_in(xs) if project) + positional_encoding(length, dim)."""
        B, N, _ = xs.shape
        locs = torch.zeros((B, N, 2), device=xs.device, dtype=torch.int64)
        return self._encode(xs, locs, 1, project)

    def pos_encode_2d(self, data, normalized_locs, project=True):
        """reference model/aggregator.py:43-56: ``normalized_locs`` are patch indices [B, S, 2] (x, y)."""
        assert normalized_locs.shape == (data.shape[0], data.shape[1], 2)
        return self._encode(data, normalized_locs.to(torch.int64).contiguous(), 2, project)

    def forward(self, seq1, seq2, lengths1, lengths2):
        """reference model/aggregator.py:58-76: prepend the special token to ``seq2`` [B, M, dim], mask keys beyond ``lengths2 + 1``,
        run the decoder stack over the memory ``encoder(seq1)`` and return token 0 [B, dim].  PATHS always passes an EMPTY ``seq1``
        (model/paths.py:113-115) - cross-attention over zero keys adds ``multihead_attn.out_proj.bias`` - and that is the form the
        HIP path implements; a non-empty condition sequence is rejected."""
        from .. import _lib, ops
        if seq1 is not None and seq1.dim() == 3 and seq1.shape[1] != 0:
            raise NotImplementedError("TransformerAggregator on the HIP path takes an empty condition sequence seq1 [B, 0, dim] (all PATHS "
                                      "models, reference model/paths.py:113-115); the encoder / cross-attention weights are dead parameters")
        _lib.require_cuda(seq2, lengths2)
        B, M, d = seq2.shape
        assert d == self.dim
        ops.check_aggregator_geometry(d, self.nhead)
        mc = self._geometry()
        num_ims = (lengths2.to(torch.int64).contiguous() if lengths2 is not None
                   else torch.full((B,), M, device=seq2.device, dtype=torch.int64))
        # the special-token row (aggregator.py:62-64): plumbing, and torch's own cat carries its gradient under autograd
        tokens = torch.cat((self.special_token.view(1, 1, -1).expand(B, 1, d).to(torch.float32), seq2.float()), dim=1).contiguous()
        if torch.is_grad_enabled() and tokens.requires_grad:
            from .. import autograd as pag
            return pag.AggregatorFn.apply(self, tokens, num_ims, *pag.aggregator_params(self))
        with torch.no_grad():
            status = torch.zeros((1,), device=seq2.device, dtype=torch.int32)
            out = ops._aggregator_forward(mc, ops.pack_aggregator(self), tokens, num_ims, None, None, status)
            if int(status.item()) & 4:
                raise _lib.PathsHipError("a bounded in-launch hand-off wait of the token-0 tail gave up (csrc/token0_ws.hip): results invalid")
        return out["ctx_slide"]
