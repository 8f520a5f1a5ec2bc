"""ORACLE-side checker — test infrastructure only (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).

Compares one recursion of the HIP path (the ``trace`` list filled by ``paths_amd.utils.recurse``) with the oracle's trace
(``oracle.paths_oracle.inference_end2end(..., trace=...)``) on the same slides.  What is compared, per level and slide
(reference utils.py:228-279, data_utils/slide.py:294-331):

* ``num_ims``                                              bit-exact
* the patch LOCATION set of the level                     bit-exact  (row order inside a level may legally differ where two
                                                          importance scores nearly tie: SURVEY.md §7 hard part 1)
* the kept (top-K) patches, as a location set             bit-exact
* the kept patches as a SEQUENCE (torch.topk order)       identical up to permutations among scores closer than ``seq_tol``
                                                          (the order decides row / token positions downstream: 1-D positional
                                                          encoding, fp summation order); reported: positions that differ and
                                                          the largest score gap among them
* ``parent_inds`` as (child location -> parent location)  bit-exact  (fallback levels: as (cell location -> cell index))
* importance per location                                 <= ``imp_tol``
* final hazards                                           <= ``hazard_tol`` (north_star bar 1e-4)

A slide whose ORACLE boundary gap (score[k-1] - score[k]) at some level is below ``gap_screen`` is not required to select the
same set from that level on (the reference's own selection is thread-count dependent there); it is reported in
``near_tie_slides`` instead of failing.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def _locset(a) -> set:
    return {tuple(int(v) for v in r) for r in a}


def compare_recursion(gpu_trace: Sequence[dict], oracle_trace: Sequence[dict], gpu_hazards, oracle_hazards,
                      imp_tol: float = 5e-6, hazard_tol: float = 1e-4, gap_screen: float = 2e-6,
                      raise_on_mismatch: bool = True, seq_tol: float = 1e-6) -> Dict[str, object]:
    L = len(oracle_trace)
    assert len(gpu_trace) == L, (len(gpu_trace), L)
    B = int(_np(oracle_trace[0]["num_ims"]).shape[0])
    problems: List[str] = []
    screened = set()                    # slides past a near-tie boundary: later levels are not comparable
    min_gap = float("inf")
    max_imp = 0.0
    n_idx = 0
    seq_moved, seq_gap = 0, 0.0         # kept-sequence positions that differ from the oracle's, largest oracle-score gap among them
    for l in range(L):
        g, o = gpu_trace[l], oracle_trace[l]
        gn, on = _np(g["num_ims"]).astype(np.int64), _np(o["num_ims"]).astype(np.int64)
        gl, ol = _np(g["locs"]), _np(o["locs"])
        gi, oi = _np(g["importance"]), _np(o["importance"])
        gp, op_ = _np(g["parent_inds"]), _np(o["parent_inds"])
        for j in range(B):
            if j in screened:
                continue
            if gn[j] != on[j]:
                problems.append(f"L{l} slide {j}: num_ims {gn[j]} != {on[j]}")
                screened.add(j)
                continue
            n = int(on[j])
            if _locset(gl[j, :n]) != _locset(ol[j, :n]):
                problems.append(f"L{l} slide {j}: location sets differ")
                screened.add(j)
                continue
            # importance by location
            go, oo = np.lexsort(gl[j, :n].T), np.lexsort(ol[j, :n].T)
            d = float(np.abs(gi[j, :n][go] - oi[j, :n][oo]).max()) if n else 0.0
            max_imp = max(max_imp, d)
            if d > imp_tol:
                problems.append(f"L{l} slide {j}: importance differs by {d:.3g}")
            if (gi[j, n:] != 0).any():
                problems.append(f"L{l} slide {j}: importance of padding rows not 0")
            # parent_inds as (child loc -> parent loc) pairs; level 0: parent_inds = arange (slide.py:266)
            if l == 0:
                if not np.array_equal(gp[j, :n], np.arange(n)) or not np.array_equal(op_[j, :n], np.arange(n)):
                    problems.append(f"L0 slide {j}: parent_inds != arange")
            elif o.get("fallback", [False] * B)[j]:
                # the reference's all-cells fallback (slide.py:336-352): parent_inds are the cell indices of this level's grid
                pairs_g = {tuple(c) + (int(p_),) for c, p_ in zip(gl[j, :n].tolist(), gp[j, :n].tolist())}
                pairs_o = {tuple(c) + (int(p_),) for c, p_ in zip(ol[j, :n].tolist(), op_[j, :n].tolist())}
                if pairs_g != pairs_o:
                    problems.append(f"L{l} slide {j}: (cell -> parent_inds) pairs of the fallback differ")
            else:
                pg, po = gpu_trace[l - 1], oracle_trace[l - 1]
                kg = _np(pg["keep_idx"])[j, : int(_np(pg["keep_count"])[j])].astype(np.int64)
                ko = _np(po["keep_inds"][j]).astype(np.int64)
                par_g = _np(pg["locs"])[j][kg[gp[j, :n]]]
                par_o = _np(po["locs"])[j][ko[op_[j, :n]]]
                pairs_g = {tuple(c) + tuple(p) for c, p in zip(gl[j, :n].tolist(), par_g.tolist())}
                pairs_o = {tuple(c) + tuple(p) for c, p in zip(ol[j, :n].tolist(), par_o.tolist())}
                if pairs_g != pairs_o:
                    problems.append(f"L{l} slide {j}: (child -> parent) pairs differ")
            if l == L - 1:
                continue
            ko = _np(o["keep_inds"][j]).astype(np.int64)
            cnt = int(_np(g["keep_count"])[j])
            kg = _np(g["keep_idx"])[j, :cnt].astype(np.int64)
            srt = np.sort(oi[j, :n])[::-1]
            k = len(ko)
            gap = float(srt[k - 1] - srt[k]) if k < n else float("inf")
            min_gap = min(min_gap, gap)
            same = cnt == k and _locset(gl[j][kg]) == _locset(ol[j][ko])
            n_idx += k
            if not same:
                if gap < gap_screen:
                    screened.add(j)
                else:
                    problems.append(f"L{l} slide {j}: kept sets differ (boundary gap {gap:.3g})")
                    screened.add(j)
                continue
            # same set: the ORDER must agree too, except among (near-)tied scores
            seq_g = [tuple(int(v) for v in r) for r in gl[j][kg]]
            seq_o = [tuple(int(v) for v in r) for r in ol[j][ko]]
            if seq_g != seq_o:
                score = {tuple(int(v) for v in r): float(x) for r, x in zip(ol[j, :n], oi[j, :n])}
                moved = [i for i in range(k) if seq_g[i] != seq_o[i]]
                gmax = max(abs(score[seq_g[i]] - score[seq_o[i]]) for i in moved)
                seq_moved += len(moved)
                seq_gap = max(seq_gap, gmax)
                if gmax >= seq_tol:
                    problems.append(f"L{l} slide {j}: kept sequence differs at {len(moved)} positions, score gap {gmax:.3g}")
    gh, oh = _np(gpu_hazards), _np(oracle_hazards)
    ok_rows = [j for j in range(B) if j not in screened]
    hz = float(np.abs(gh[ok_rows] - oh[ok_rows]).max()) if ok_rows else 0.0
    if hz > hazard_tol:
        problems.append(f"hazards differ by {hz:.3g}")
    res = {"levels": L, "slides": B, "index_sets_identical": not any("sets differ" in s or "num_ims" in s for s in problems),
           "parent_pairs_identical": not any("pairs" in s or "parent_inds" in s for s in problems),
           "kept_indices_compared": n_idx, "max_importance_diff": max_imp, "max_hazard_diff": hz,
           "min_boundary_gap": None if min_gap == float("inf") else min_gap,
           "sequence_identical": seq_moved == 0, "sequence_positions_moved": seq_moved, "sequence_max_gap": seq_gap,
           "near_tie_slides": sorted(j for j in screened if not any(f"slide {j}:" in s for s in problems)),
           "problems": problems}
    if raise_on_mismatch and problems:
        raise AssertionError("HIP recursion != oracle: " + "; ".join(problems[:8]))
    return res
