"""Importance heat-map export (the data consumed by reference heatmap_visualise.py:113-175; drawing is out of scope).

``hierarchy_from_trace`` pulls one slide's per-level patch locations / importances / selected indices out of the trace
of :func:`paths_amd.utils.recurse`; ``importance_map`` rasterises them exactly like the reference's overlay code:
every patch of depth d paints ``importance + 1e-4`` over its footprint, then deeper levels are folded upwards with
weight 1/2 wherever they exist (heatmap_visualise.py:147-171).  The raster is in units of the FINEST level's patches
(one cell = one patch of the last level), i.e. level-0 pixel space divided by ``patch_size / 2**(L-1)``.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np


def hierarchy_from_trace(trace: List[dict], slide: int) -> List[Dict[str, np.ndarray]]:
    out = []
    for lv in trace:
        n = int(lv["num_ims"][slide])
        d = {"locs": lv["locs"][slide, :n].cpu().numpy(), "importance": lv["importance"][slide, :n].cpu().numpy(),
             "parent_inds": lv["parent_inds"][slide, :n].cpu().numpy()}
        if "keep_idx" in lv:
            d["keep_inds"] = lv["keep_idx"][slide, : int(lv["keep_count"][slide])].cpu().numpy()
        out.append(d)
    return out


def importance_map(levels: List[Dict[str, np.ndarray]], base_grid, patch_size: int = 256, magnification_factor: int = 2) -> np.ndarray:
    """[X0 * f, Y0 * f] float map, f = magnification_factor**(L-1); 0 where no patch was visited."""
    L = len(levels)
    f = magnification_factor ** (L - 1)
    shape = (base_grid[0] * f, base_grid[1] * f)
    overall = np.zeros((L,) + shape, dtype=np.float64)
    for depth, lv in enumerate(levels):
        size = magnification_factor ** (L - 1 - depth)                       # footprint of one patch, in finest cells
        cells = lv["locs"] // patch_size
        for (cx, cy), imp in zip(cells, lv["importance"]):
            overall[depth, cx * size:(cx + 1) * size, cy * size:(cy + 1) * size] = imp + 1e-4
    for depth in range(L - 2, -1, -1):                                        # heatmap_visualise.py:167-169
        m = overall[depth + 1] != 0
        overall[depth][m] = overall[depth][m] + overall[depth + 1][m] * 0.5
    return overall[0]
