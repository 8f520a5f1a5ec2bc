"""``LSTMCell`` / ``Processor`` / ``RecursiveModel`` with the reference's constructor signatures, attribute
names and state_dict keys (reference model/interface.py:10-99).

The modules only OWN parameters; the arithmetic runs in libpaths_hip.so (paths_amd/ops.py).  Construction
order of sub-modules follows the reference so that ``torch.manual_seed(s); Config.get_model()`` draws the
same initial weights.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Callable, Dict, Tuple

import torch
from torch import nn

from .. import _lib, ops


class LSTMCell(nn.Module):
    """One LSTM step over *depth* (not over a sequence): (x, h, c) -> (h', c')."""

    def __init__(self, input_dim: int, output_dim: int, hidden_dim: int):
        super().__init__()
        self.xdim, self.hdim, self.cdim = input_dim, output_dim, hidden_dim
        xh = input_dim + output_dim
        # key names lstm.<gate>.0.{weight,bias}
        self.forget_gate = nn.Sequential(nn.Linear(xh, hidden_dim), nn.Sigmoid())
        self.remember_gate = nn.Sequential(nn.Linear(xh, hidden_dim), nn.Sigmoid())
        self.remember_map = nn.Sequential(nn.Linear(xh, hidden_dim), nn.Tanh())
        self.out_select_gate = nn.Sequential(nn.Linear(xh, output_dim), nn.Sigmoid())
        self.mem_to_out = nn.Sequential(nn.Linear(hidden_dim, output_dim), nn.Tanh())

    def forward(self, xs: torch.Tensor, hs: torch.Tensor, cs: torch.Tensor):
        """(..., xdim), (..., hdim), (..., cdim) -> (hs', cs') on the HIP GEMM kernels (forward only)."""
        assert xs.shape[:-1] == hs.shape[:-1] == cs.shape[:-1], "Mismatching starting dimensions"
        assert xs.shape[-1] == self.xdim and hs.shape[-1] == self.hdim and cs.shape[-1] == self.cdim
        assert self.xdim == self.hdim, "PATHS uses input_dim == output_dim"
        _lib.require_cuda(xs, hs, cs)
        pk = ops.pack_lstm(self)
        lead = xs.shape[:-1]
        x = xs.reshape(-1, self.xdim).float().contiguous()
        M = x.shape[0]
        state = torch.cat((hs.reshape(M, -1), cs.reshape(M, -1)), dim=-1).float().contiguous()
        out = torch.empty_like(state)
        y = torch.empty_like(x)
        ws = torch.empty_like(x)
        p = _lib.ptr
        Dp = self.hdim + self.cdim
        _lib.call("paths_lstm_cell", p(x), self.xdim, state.data_ptr(), Dp, state.data_ptr() + 4 * self.hdim, Dp,
                  p(pk["w_gates"]), p(pk["b_gates"]), p(pk["w_mem"]), p(pk["b_mem"]), p(out), Dp, p(y), self.xdim,
                  p(ws), None, None, None, None, M, self.xdim, self.cdim, None, 1, 7, _lib.stream())
        return out[:, : self.hdim].reshape(*lead, self.hdim), out[:, self.hdim:].reshape(*lead, self.cdim)


class Processor(ABC):
    """One-magnification-level processor (reference model/interface.py:61-80)."""

    @abstractmethod
    def process(self, data) -> Dict:
        raise NotImplementedError

    @abstractmethod
    def ctx_dim(self) -> Tuple[int, int]:
        raise NotImplementedError


class RecursiveModel(nn.Module):
    """``num_levels`` processors + one shared LSTM cell (reference model/interface.py:83-99)."""

    def __init__(self, processor_constructor: Callable, config_, train_config, **kwargs):
        super().__init__()
        self.procs = nn.ModuleList([processor_constructor(config_, train_config, depth=i, **kwargs)
                                    for i in range(train_config.num_levels)])
        from ..config import PATHSProcessorConfig
        if isinstance(config_, PATHSProcessorConfig) and config_.lstm:
            self.lstm = LSTMCell(config_.patch_embed_dim, config_.patch_embed_dim, config_.hierarchical_ctx_mlp_hidden_dim)
            self.use_lstm = True
        else:
            self.use_lstm = False

    def forward(self, depth, *args, **kwargs):
        if self.use_lstm:
            kwargs["lstm"] = self.lstm
        return self.procs[depth].process(*args, **kwargs)
