"""Pin the oracle (oracle/paths_oracle.py) against vectors captured from the reference import
(tools/make_goldens.py).  CPU-only: runs under ``-m "not gpu"``.

Tolerances: the reference's own outputs move by ~1e-7 with thread count / MHA code path
(SURVEY.md §7 hard part 1, §8c "Version caveat"), so float outputs are compared at 2e-6 and index
SETS bit-exactly (sequence: identical up to permutations among scores closer than 1e-6).
"""
import numpy as np
import pytest
import torch

from oracle import paths_oracle as orc
from tests import helpers as H
from tests.conftest import load_golden

ATOL = 2e-6


def _run_single(name):
    g, info = load_golden(name)
    cfg = H.oracle_config(info["cfg_over"])
    p = H.oracle_params(cfg, info["wseed"])
    inp = {k: torch.from_numpy(v) for k, v in H.single_level_inputs(info, cfg).items()}
    with torch.no_grad():
        out = orc.process_level(p, cfg, info["depth"], inp["fts"], inp["locs"], inp["num_ims"],
                                inp["ctx_slide"], inp["ctx_patch"])
    return g, info, out


@pytest.mark.parametrize("name", ["g1_level0_b2_k256", "g2_level2_b2_k256"])
def test_single_level_default(name):
    g, info, out = _run_single(name)
    for k in ("logits", "ctx_slide", "importance", "ctx_patch"):
        np.testing.assert_allclose(out[k].numpy(), g[k], atol=ATOL, rtol=0, err_msg=k)
    # padded rows get exactly zero importance (reference utils.py:106-115)
    for b, n in enumerate(info["num_ims"]):
        assert (out["importance"][b, n:] == 0).all()


@pytest.mark.parametrize("tag", ["pe1d", "nolstm", "concat", "impnone", "subtype"])
def test_single_level_variants(tag):
    g, info, out = _run_single(f"g5_{tag}_level1")
    for k in ("logits", "ctx_slide", "importance", "ctx_patch"):
        np.testing.assert_allclose(out[k].numpy(), g[k], atol=ATOL, rtol=0, err_msg=k)


@pytest.mark.parametrize("tag", ["td192", "td192_pe1d", "h8_hi64", "td64_h2_l3"])
def test_single_level_other_aggregator_geometries(tag):
    """g12: the oracle on aggregator geometries other than the shipped 128 / 4 / 128 (the reference dataclass default trans_dim 192,
    head_dim 48; 8 heads with a 64-wide importance MLP; trans_dim 64 x 2 heads x 3 layers) against the reference's outputs."""
    g, info, out = _run_single(f"g12_{tag}_level1")
    for k in ("logits", "ctx_slide", "importance", "ctx_patch"):
        np.testing.assert_allclose(out[k].numpy(), g[k], atol=ATOL, rtol=0, err_msg=k)


@pytest.mark.parametrize("tag", ["td256_h2", "td1536_h4"])
def test_single_level_wide_heads(tag):
    """g14: the oracle at head_dim 128 (trans_dim 256 / 2 heads) and 384 (trans_dim 1536 / 4 heads: SURVEY 8(d)'s stress form) against the
    reference's outputs (reference model/aggregator.py:25-33 accepts any trans_dim % trans_heads == 0)."""
    g, info, out = _run_single(f"g14_{tag}_level1")
    for k in ("logits", "ctx_slide", "importance", "ctx_patch"):
        np.testing.assert_allclose(out[k].numpy(), g[k], atol=ATOL, rtol=0, err_msg=k)


@pytest.mark.parametrize("tag", ["td160_h4_hd40", "td320_h4_hd80", "td96_h4_hd24_pe1d"])
def test_single_level_odd_head_dims(tag):
    """g15: head dims that are neither 16 / 32 / 48 / 64 nor a multiple of 32 (40, 80, 24: reference model/aggregator.py:25-33 accepts any
    trans_dim % trans_heads == 0) - the oracle against the reference's outputs."""
    g, info, out = _run_single(f"g15_{tag}_level1")
    for k in ("logits", "ctx_slide", "importance", "ctx_patch"):
        np.testing.assert_allclose(out[k].numpy(), g[k], atol=ATOL, rtol=0, err_msg=k)


@pytest.mark.parametrize("name", ["g8_level0_b1_k2048", "g9_level1_b2_k2048"])
def test_single_level_k2048(name):
    g, info, out = _run_single(name)
    for k in ("logits", "ctx_slide", "importance"):
        np.testing.assert_allclose(out[k].numpy(), g[k], atol=ATOL, rtol=0, err_msg=k)
    idx = g["ctx_patch_probe_idx"]
    np.testing.assert_allclose(out["ctx_patch"].numpy()[tuple(idx.T)], g["ctx_patch_probe"], atol=ATOL, rtol=0)
    # top-512 of the oracle == top-512 of the reference's importance, as a set
    n = info["num_ims"][0]
    a = torch.topk(out["importance"][0, :n], 512).indices.numpy()
    b = torch.topk(torch.from_numpy(g["importance"][0, :n]), 512).indices.numpy()
    assert H.set_agreement(a, b)


@pytest.mark.parametrize("name", ["g3_recursion_6x7_top5", "g4_recursion_16x16_top64"])
def test_recursion(name):
    g, info = load_golden(name)
    cfg = H.oracle_config(info["cfg_over"], top_k_patches=[info["top_k"]] * 4)
    p = H.oracle_params(cfg, info["wseed"])
    slides = H.synthetic_slides(info, cfg)
    labels = {"survival_bin": torch.from_numpy(g["labels"][:, 0]), "censored": torch.from_numpy(g["labels"][:, 1])}
    trace = []
    with torch.no_grad():
        hazards, loss = orc.inference_end2end(p, cfg, [orc.LazyGrids(s) for s in slides], labels, trace)
    assert info["min_gap"] >= 2e-6
    for l, lv in enumerate(trace):
        np.testing.assert_array_equal(lv["num_ims"].numpy(), g[f"L{l}_num_ims"])
        if l < cfg.num_levels - 1:
            for j, ki in enumerate(lv["keep_inds"]):
                ref_ki = g[f"L{l}_keep_{j}"]
                assert H.set_agreement(ki.numpy(), ref_ki), (l, j)
                n = int(lv["num_ims"][j])
                assert H.sequence_inversions(ki.numpy(), ref_ki, g[f"L{l}_importance"][j, :n]) < 1e-6
        # same rows in the same order whenever the index sequences agree; always equal as multisets
        a = np.concatenate([lv["locs"].numpy(), lv["parent_inds"].numpy()[..., None]], -1)
        b = np.concatenate([g[f"L{l}_locs"], g[f"L{l}_parent_inds"][..., None]], -1)
        for j in range(a.shape[0]):
            n = int(lv["num_ims"][j])
            assert {tuple(r[:2]) for r in a[j, :n]} == {tuple(r[:2]) for r in b[j, :n]}
        np.testing.assert_allclose(lv["logits"].numpy(), g[f"L{l}_logits"], atol=5e-6, rtol=0)
    np.testing.assert_allclose(hazards.numpy(), g["hazards"], atol=5e-6, rtol=0)
    np.testing.assert_allclose(float(loss), float(g["loss"]), atol=5e-6, rtol=0)


def test_nll_known_answers():
    g, _ = load_golden("g7_nll")
    h, y, c = (torch.from_numpy(g[k]) for k in ("hazards", "y", "c"))
    np.testing.assert_allclose(float(orc.nll_loss(h, y, c)), float(g["loss"]), rtol=1e-6)
    for i in range(h.shape[0]):
        np.testing.assert_allclose(float(orc.nll_loss(h[i:i + 1], y[i:i + 1], c[i:i + 1])), g["per_sample"][i], rtol=1e-6, atol=1e-7)


def test_state_dict_surface():
    """Key names / count of the checkpoint surface (9 881 881 params verified there)."""
    cfg = H.oracle_config()
    shapes = orc.state_dict_shapes(cfg)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 9_881_881
    assert "procs.0.global_agg.transformer.encoder.layers.0.self_attn.in_proj_weight" in shapes
    assert "lstm.mem_to_out.0.weight" in shapes


def test_compare_recursion_checker_detects_differences():
    """oracle/compare.py (used by the GPU tests and bench.py's parity_checked): the oracle against itself passes; a swapped
    kept index, a wrong parent index and a hazard shift are each detected."""
    import copy
    from oracle import paths_oracle as orc
    from oracle.compare import compare_recursion
    from paths_amd import synthetic as syn
    ocfg = H.oracle_config(top_k_patches=[6] * 4)
    params = H.oracle_params(ocfg, 3)
    grids = [orc.LazyGrids(syn.SyntheticSlide(5, s, (5, 6), 1024, 5, 0.1)) for s in range(2)]
    tr = []
    with torch.no_grad():
        hz, _ = orc.inference_end2end(params, ocfg, grids, None, tr)

    def as_gpu(t):          # the HIP trace's field names (paths_amd/utils.py:_recurse_body)
        out = []
        for l, lv in enumerate(t):
            r = {k: lv[k].clone() for k in ("num_ims", "locs", "parent_inds", "importance", "logits")}
            if lv["keep_inds"]:
                cap = max(len(k) for k in lv["keep_inds"])
                ki = torch.zeros((len(lv["keep_inds"]), cap), dtype=torch.int32)
                for j, k in enumerate(lv["keep_inds"]):
                    ki[j, :len(k)] = k.to(torch.int32)
                r["keep_idx"], r["keep_count"] = ki, torch.tensor([len(k) for k in lv["keep_inds"]], dtype=torch.int32)
            out.append(r)
        return out

    res = compare_recursion(as_gpu(tr), tr, hz, hz)
    assert res["index_sets_identical"] and res["parent_pairs_identical"] and res["max_hazard_diff"] == 0 and res["levels"] == 5
    bad = as_gpu(tr)
    n0 = int(tr[0]["num_ims"][0])
    unkept = [i for i in range(n0) if i not in set(tr[0]["keep_inds"][0].tolist())][0]
    bad[0]["keep_idx"][0, 0] = unkept
    assert not compare_recursion(bad, tr, hz, hz, raise_on_mismatch=False)["index_sets_identical"]
    bad = as_gpu(tr)
    bad[2]["parent_inds"][1, 0] = (bad[2]["parent_inds"][1, 0] + 1) % 6
    assert not compare_recursion(bad, tr, hz, hz, raise_on_mismatch=False)["parent_pairs_identical"]
    with pytest.raises(AssertionError):
        compare_recursion(as_gpu(tr), tr, hz + 1e-3, hz)
    # the kept SEQUENCE: swapping two kept entries whose scores differ by more than 1e-6 keeps the set but is reported
    assert res["sequence_identical"] and res["sequence_positions_moved"] == 0
    bad = as_gpu(tr)
    a, b = int(bad[1]["keep_idx"][0, 0]), int(bad[1]["keep_idx"][0, 1])
    bad[1]["keep_idx"][0, 0], bad[1]["keep_idx"][0, 1] = b, a
    r2 = compare_recursion(bad, tr, hz, hz, raise_on_mismatch=False)
    gap = abs(float(tr[1]["importance"][0, a] - tr[1]["importance"][0, b]))
    assert gap > 1e-6 and r2["index_sets_identical"] and not r2["sequence_identical"] and r2["sequence_positions_moved"] == 2
    assert abs(r2["sequence_max_gap"] - gap) < 1e-12 and any("kept sequence" in p_ for p_ in r2["problems"])
    # ... while a swap inside the tolerance only shows in the counters
    r3 = compare_recursion(bad, tr, hz, hz, raise_on_mismatch=False, seq_tol=gap * 2)
    assert r3["sequence_positions_moved"] == 2 and not any("kept sequence" in p_ for p_ in r3["problems"])
    # (the next level's parent_inds of this hand-made trace still index the un-swapped order: only that is reported)
    assert all("(child -> parent)" in p_ for p_ in r3["problems"])


def test_importance_map_equals_reference_overlay_g11():
    """paths_amd.heatmap.importance_map against the array the REFERENCE's own overlay code (heatmap_visualise.py:147-171, run inside
    heatmap_camelyon17 by tools/make_goldens.py on the per-level locations / importances of fixture G3) handed to imshow."""
    from paths_amd.heatmap import importance_map
    g11, info = load_golden("g11_heatmap_g3_slide0")
    g3, info3 = load_golden(info["source"])
    j, L = info["slide"], info["levels"]
    levels = []
    for l in range(L):
        n = int(g3[f"L{l}_num_ims"][j])
        levels.append({"locs": g3[f"L{l}_locs"][j, :n], "importance": g3[f"L{l}_importance"][j, :n]})
    m = importance_map(levels, tuple(info["base_shape"]), patch_size=info["patch_size"])
    assert m.shape == g11["map"].shape == (96, 112)
    seen = m > 0
    assert np.array_equal(seen, g11["alpha"] > 0)                         # the same cells are covered (alpha = 0.5 where drawn)
    np.testing.assert_allclose(m[seen], g11["map"][seen], rtol=0, atol=1e-12)
    # the reference paints never-visited pixels with the smallest drawn value before plotting (heatmap_visualise.py:174)
    assert np.all(g11["map"][~seen] == g11["map"][seen].min()) if (~seen).any() else True
