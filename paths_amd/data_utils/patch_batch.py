"""``PatchBatch`` — the batch container at the model boundary (reference data_utils/patch_batch.py:13-75).

Same constructor arguments, attributes and shape checks as the reference.  One deliberate difference: the
reference asserts ``num_ims.max().item() == max_patches`` (patch_batch.py:50), which is a device->host sync;
here it is checked only when ``strict=True`` (default, drop-in behaviour).  The device-resident recursion
(paths_amd/utils.py) builds batches with ``strict=False`` because its padded length is a static capacity.
"""
from __future__ import annotations

from typing import Dict

import torch


class PatchBatch:
    def __init__(self, locs, num_ims, parent_inds, ctx_slide, ctx_patch, fts, strict: bool = True, **unused_kwargs):
        batch_size, max_patches, _ = fts.shape
        _, self.ctx_depth, self.ctx_dim1 = ctx_slide.shape
        self.ctx_dim2 = ctx_patch.shape[-1]
        assert locs.shape == (batch_size, max_patches, 2)
        assert num_ims.shape == (batch_size,)
        assert parent_inds.shape == (batch_size, max_patches)
        assert ctx_slide.shape == (batch_size, self.ctx_depth, self.ctx_dim1)
        assert ctx_patch.shape == (batch_size, max_patches, self.ctx_depth, self.ctx_dim2)
        if strict:
            assert num_ims.max().item() == max_patches
        self.device = fts.device
        assert all(t.device == self.device for t in (locs, num_ims, parent_inds, ctx_slide, ctx_patch))
        self.batch_size, self.max_patches = batch_size, max_patches
        self.fts, self.locs, self.num_ims = fts, locs, num_ims
        self.parent_inds, self.ctx_slide, self.ctx_patch = parent_inds, ctx_slide, ctx_patch
        self.valid_inds = torch.arange(max_patches, device=num_ims.device).expand(batch_size, -1) < num_ims[:, None]


def _todevice(x, device):
    if hasattr(x, "to"):
        return x.to(device)
    if isinstance(x, (list, tuple)):
        return [_todevice(i, device) for i in x]
    return x


def from_batch(batch: Dict, device) -> PatchBatch:
    """reference data_utils/patch_batch.py:73-75"""
    return PatchBatch(**{k: _todevice(v, device) for k, v in batch.items()})
