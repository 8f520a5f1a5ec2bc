"""``TransformerAggregator`` parameter container (reference model/aggregator.py:8-76).

Keeps the reference's attribute names so the state_dict keys match
(``global_agg.{special_token, proj_in.*, transformer.{encoder,decoder}...}``).  ``nn.Transformer`` is
instantiated only as the owner of those tensors — including the dead encoder and cross-attention weights,
which must round-trip through checkpoints — its forward is never called.  The aggregation
itself (proj_in + PE + special token + post-LN decoder stack over an empty memory + token-0 read-out) runs in
paths_importance_proj / paths_attention_f32 / paths_token_layer_f32 / paths_final_head (paths_amd/ops.py).
"""
from __future__ import annotations

import torch
from torch import nn


class TransformerAggregator(nn.Module):
    def __init__(self, input_dim: int, model_dim: int, output_dim: int, nhead: int, layers: int, dropout: float):
        super().__init__()
        self.dim = model_dim
        self.nhead = nhead
        self.num_layers = layers
        self.proj_in = nn.Linear(input_dim, model_dim)
        self.proj_out = nn.Identity()
        self.transformer = nn.Transformer(model_dim, nhead=nhead, num_encoder_layers=layers, num_decoder_layers=layers,
                                          dim_feedforward=model_dim * 4, dropout=dropout, batch_first=True)
        self.special_token = nn.Parameter(torch.randn(model_dim), requires_grad=True)

    def forward(self, *args, **kwargs):
        raise RuntimeError("TransformerAggregator is fused into PATHSProcessor.process on the HIP path; "
                           "call the processor (model(depth, PatchBatch)) instead")
