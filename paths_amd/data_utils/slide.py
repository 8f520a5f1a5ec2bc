"""HBM-resident preprocessed slides (the device-side analogue of reference
data_utils/slide.py:PreprocessedSlide, :225-271).

A slide is its per-level feature grids ``[X, Y, D]`` fp32 (reference grid format: an all-zero row is a
background cell, preprocess/preprocess.py:89,172-175) kept RESIDENT in HBM, plus a one-byte-per-cell tissue
mask computed once on upload (``sum(dim=1) != 0``, reference data_utils/slide.py:324).  The per-level host
gather + H2D copy of the reference disappears: child rows are gathered on the device straight from these grids.
At K=2048 one slide is 2.9 GB (level-4 grid alone 2.1 GB); 288 GB of HBM holds ~90 of them.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib, synthetic


class DeviceSlide:
    def __init__(self, grids: Sequence[torch.Tensor], patch_size: int = 256, slide_id: str = "", subtype=None):
        assert len(grids) >= 1
        self.patch_size = patch_size
        self.slide_id = slide_id
        self.subtype = subtype
        self.grids: List[torch.Tensor] = []
        self.masks: List[torch.Tensor] = []
        self._absmax_bits = None          # fp32 bit pattern of max|feature| over all levels, written by the mask pass
        self._absmax: Optional[float] = None
        for g in grids:
            _lib.require_cuda(g)
            assert g.dim() == 3 and g.dtype == torch.float32
            g = g.contiguous()
            X, Y, D = g.shape
            m = torch.empty((X, Y), dtype=torch.uint8, device=g.device)
            if self._absmax_bits is None:
                self._absmax_bits = torch.zeros((1,), dtype=torch.int32, device=g.device)
            _lib.call("paths_tissue_mask_absmax", g.data_ptr(), X * Y, D, m.data_ptr(), self._absmax_bits.data_ptr(), _lib.stream())
            self.grids.append(g)
            self.masks.append(m)

    def feature_absmax(self) -> float:
        """max|x| over every grid of the slide (inf if any element is inf or NaN): the operand-range check of the default
        fp16-split GEMM mode (paths_amd/ops.py:h3_in_range).  One host sync on first use, cached."""
        if self._absmax is None:
            self._absmax = _lib.float_from_bits(int(self._absmax_bits.item()))
        return self._absmax

    @property
    def num_levels(self) -> int:
        return len(self.grids)

    def shape(self, level: int) -> Tuple[int, int]:
        return self.grids[level].shape[0], self.grids[level].shape[1]

    @property
    def dim(self) -> int:
        return self.grids[0].shape[2]

    @staticmethod
    def from_host(grids: Sequence, device, **kw) -> "DeviceSlide":
        """Upload host grids (numpy or CPU tensors, e.g. ``torch.load('<slide>_<power:.3f>.pt')``)."""
        return DeviceSlide([torch.as_tensor(g, dtype=torch.float32).to(device) for g in grids], **kw)

    @staticmethod
    def from_preprocessed(root: str, slide_id: str, powers: Sequence[float], device="cuda", patch_size: int = 256,
                          subtype=None) -> "DeviceSlide":
        """Load the reference's preprocessed-grid files ``<root>/<slide_id>_<power:.3f>.pt`` (one ``[X, Y, D]`` float
        tensor per magnification, all-zero row = background; written by reference preprocess/preprocess.py:89,134 and
        read by preprocess/loader.py:14-18 / data_utils/slide.py:247-253) and make them resident in HBM."""
        import os
        grids = []
        for power in powers:
            path = os.path.join(root, slide_id + f"_{power:.3f}.pt")
            assert os.path.isfile(path), f"Pre-process load: path '{path}' not found!"
            g = torch.load(path, map_location="cpu")
            assert g.dim() == 3, f"{path}: expected a [X, Y, D] grid, got {tuple(g.shape)}"
            grids.append(g.float())
        return DeviceSlide.from_host(grids, device, patch_size=patch_size, slide_id=slide_id, subtype=subtype)

    @staticmethod
    def synthetic(seed: int, slide: int, base_shape: Tuple[int, int], dim: int = 1024, num_levels: int = 5,
                  p_bg: float = 0.1, device="cuda", patch_size: int = 256) -> "DeviceSlide":
        """Generate the counter-based synthetic pyramid directly in HBM (paths_synth_grid)."""
        grids = []
        thr = synthetic.bg_threshold(p_bg)
        for l in range(num_levels):
            X, Y = base_shape[0] << l, base_shape[1] << l
            g = torch.empty((X, Y, dim), dtype=torch.float32, device=device)
            key = int(synthetic.slide_level_key(seed, slide, l))
            _lib.call("paths_synth_grid", g.data_ptr(), X, Y, dim, key, l, thr, _lib.stream())
            grids.append(g)
        s = DeviceSlide(grids, patch_size=patch_size, slide_id=f"synthetic-{seed}-{slide}")
        s.synthetic_spec = synthetic.SyntheticSlide(seed, slide, tuple(base_shape), dim, num_levels, p_bg)
        return s


class DeviceSlideBatch:
    """Per-batch device tables (grid / mask base pointers and grid dims per level) built ONCE.

    ``torch.tensor(list, device=...)`` is a blocking host->device copy that also waits for everything queued on
    the stream; building these tables inside every recursion call cost ~1.5 ms of idle GPU per step.
    """

    def __init__(self, slides):
        assert len(slides) > 0
        self.slides = list(slides)
        dev = self.slides[0].grids[0].device
        L = min(s.num_levels for s in self.slides)
        self.device, self.num_levels = dev, L
        self.dim = self.slides[0].dim
        assert all(s.dim == self.dim for s in self.slides)

        def table(fn, dtype):
            return [torch.tensor([fn(s, l) for s in self.slides], device=dev, dtype=dtype) for l in range(L)]

        self.grid_ptrs = table(lambda s, l: s.grids[l].data_ptr(), torch.int64)
        self.mask_ptrs = table(lambda s, l: s.masks[l].data_ptr(), torch.int64)
        self.gx = table(lambda s, l: s.shape(l)[0], torch.int32)
        self.gy = table(lambda s, l: s.shape(l)[1], torch.int32)
        self.n0 = max(s.shape(0)[0] * s.shape(0)[1] for s in self.slides)
        self.feat_absmax = max(s.feature_absmax() for s in self.slides)
        self.max_dim = [max(max(s.shape(l)) for s in self.slides) for l in range(L)]   # bound of locs // patch_size per level

    def __len__(self):
        return len(self.slides)

    def flat_tables(self) -> torch.Tensor:
        """Every per-level table in ONE byte buffer, level by level (grid pointers, mask pointers, gx, gy), built once per batch:
        binding a recorded launch tape to this batch is then a single small device-to-device copy."""
        flat = getattr(self, "_flat_tables", None)
        if flat is None:
            parts = []
            for l in range(self.num_levels):
                parts += [self.grid_ptrs[l].view(torch.uint8), self.mask_ptrs[l].view(torch.uint8), self.gx[l].view(torch.uint8), self.gy[l].view(torch.uint8)]
            flat = self._flat_tables = torch.cat(parts)
        return flat

    def clone_tables(self) -> "DeviceSlideBatch":
        """The same batch with PRIVATE copies of the table tensors (a recorded launch tape addresses these, and re-points them at
        other batches: paths_amd.utils.TapedRecursion.rebind); the slides themselves are shared.  The copies are views of one flat
        buffer laid out like :meth:`flat_tables`."""
        c = object.__new__(DeviceSlideBatch)
        c.__dict__.update(self.__dict__)
        flat = self.flat_tables().clone()
        B = len(self.slides)
        c._flat_tables = flat
        c.grid_ptrs, c.mask_ptrs, c.gx, c.gy = [], [], [], []
        off = 0
        for l in range(self.num_levels):
            c.grid_ptrs.append(flat[off:off + 8 * B].view(torch.int64)); off += 8 * B
            c.mask_ptrs.append(flat[off:off + 8 * B].view(torch.int64)); off += 8 * B
            c.gx.append(flat[off:off + 4 * B].view(torch.int32)); off += 4 * B
            c.gy.append(flat[off:off + 4 * B].view(torch.int32)); off += 4 * B
        c.max_dim = list(self.max_dim)
        return c
