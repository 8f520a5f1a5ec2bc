"""CPU tests of the train/eval harness pieces that need no GPU: metrics, checkpoint format, shuffle semantics."""
import itertools
import os
import pickle

import numpy as np
import pytest
import torch


def brute_cindex(event, time, risk):
    num = conc = 0.0
    n = len(time)
    for i, j in itertools.permutations(range(n), 2):
        if not event[i]:
            continue
        if time[i] < time[j] or (time[i] == time[j] and not event[j]):
            num += 1
            if abs(risk[i] - risk[j]) <= 1e-8:
                conc += 0.5
            elif risk[i] > risk[j]:
                conc += 1
    return conc / num


def test_concordance_index_matches_definition():
    from paths_amd.eval import concordance_index_censored
    rng = np.random.RandomState(0)
    for n in (5, 17, 60):
        event = rng.rand(n) < 0.6
        event[0] = True
        time = rng.randint(0, 8, n).astype(float)          # many tied times
        risk = np.round(rng.randn(n), 1)                   # some tied risks
        assert abs(concordance_index_censored(event, time, risk) - brute_cindex(event, time, risk)) < 1e-12
    # known answers: perfectly ordered / reversed
    e, t = np.ones(4, bool), np.array([1.0, 2, 3, 4])
    assert concordance_index_censored(e, t, np.array([4.0, 3, 2, 1])) == 1.0
    assert concordance_index_censored(e, t, np.array([1.0, 2, 3, 4])) == 0.0
    assert concordance_index_censored(e, t, np.zeros(4)) == 0.5


def test_concordance_index_hand_computed_sksurv_cases():
    """Known answers worked out by hand from the definition sksurv.metrics.concordance_index_censored documents (pair (i, j) is
    comparable iff i had an event and t_i < t_j, or t_i == t_j and j is censored; a pair counts 1 if risk_i > risk_j, 1/2 if the
    risks are within tied_tol; sksurv itself is not installed here)."""
    from paths_amd.eval import concordance_index_censored as ci
    # events at t = 1, 2, 4; censored at t = 2, 3.  Comparable: 0 -> {1, 2, 3, 4} (4 concordant); 1 -> {3 (conc), 4 (disc), 2 (same
    # time, censored: risk tie -> 1/2)}; the last event has nobody after it.  (4 + 1 + 0.5) / 7
    t = np.array([1.0, 2, 2, 3, 4]); e = np.array([1, 1, 0, 0, 1], bool); r = np.array([0.9, 0.5, 0.5, 0.1, 0.7])
    assert abs(ci(e, t, r) - 5.5 / 7.0) < 1e-15
    # tied times: two events at the same time are NOT comparable with each other; an event and a censored sample at that time are
    assert ci(np.array([1, 0], bool), np.array([5.0, 5.0]), np.array([0.2, 0.8])) == 0.0
    assert ci(np.array([1, 0], bool), np.array([5.0, 5.0]), np.array([0.8, 0.2])) == 1.0
    assert abs(ci(np.array([1, 1, 0], bool), np.array([5.0, 5.0, 7.0]), np.array([0.3, 0.9, 0.5])) - 0.5) < 1e-15      # pairs 0->2 (disc), 1->2 (conc)
    # a censored sample never opens a pair, even when it is the earliest
    assert ci(np.array([0, 1, 1], bool), np.array([1.0, 2.0, 3.0]), np.array([0.0, 0.9, 0.1])) == 1.0                  # only 1 -> 2
    # risk ties inside tied_tol count one half, outside it they are ordered
    e3, t3 = np.array([1, 1], bool), np.array([1.0, 2.0])
    assert ci(e3, t3, np.array([0.5, 0.5 + 5e-9])) == 0.5 and ci(e3, t3, np.array([0.5, 0.5 + 5e-8])) == 0.0
    assert ci(e3, t3, np.array([0.5, 0.5 + 5e-8]), tied_tol=1e-7) == 0.5
    # no comparable pair at all: all censored, or every event tied in time with only events
    for ev, tt in ((np.zeros(3, bool), np.array([1.0, 2, 3])), (np.ones(3, bool), np.array([5.0, 5, 5]))):
        with pytest.raises(ValueError):
            ci(ev, tt, np.array([0.1, 0.2, 0.3]))


def test_binary_auroc_matches_pair_counting():
    from paths_amd.eval import binary_auroc
    rng = np.random.RandomState(1)
    for n in (6, 40):
        s = np.round(rng.rand(n), 1)
        y = rng.rand(n) < 0.4
        y[0], y[1] = True, False
        pos, neg = s[y], s[~y]
        ref = ((pos[:, None] > neg[None, :]).sum() + 0.5 * (pos[:, None] == neg[None, :]).sum()) / (len(pos) * len(neg))
        assert abs(binary_auroc(s, y) - ref) < 1e-12
    assert binary_auroc(np.array([0.1, 0.2]), np.array([1, 1])) == 0.5


def test_survival_evaluator_and_train_stats():
    from paths_amd.eval import SurvivalEvaluator
    ev = SurvivalEvaluator("val")
    hz = torch.tensor([[0.9, 0.5, 0.5, 0.5], [0.1, 0.1, 0.1, 0.1], [0.5, 0.5, 0.5, 0.5]])
    batch = {"censored": torch.tensor([0, 0, 1]), "survival": torch.tensor([1.0, 9.0, 5.0])}
    ev.register(batch, hz, torch.tensor(0.7))
    stats = {"val_loss": {}, "val_c-index": {}}
    out = ev.calculate(stats, 3)
    assert out["val_c-index"] == 1.0 and abs(out["val_loss"] - 0.7) < 1e-6      # high hazard <-> short survival
    assert stats["val_c-index"][3] == 1.0


def test_checkpoint_format_roundtrip(tmp_path):
    """model.pt = state_dict, train_stats.pkl = pickled dict with 'epoch' (reference utils.py:169-198)."""
    from paths_amd.config import Config
    from paths_amd.train import load_state, save_state
    cfg = Config.load(os.path.join(os.path.dirname(__file__), "golden", "sample"), test_mode=True)
    torch.manual_seed(1)
    m1 = cfg.get_model()
    save_state(str(tmp_path), m1, {"epoch": 7, "train_loss": {1: 0.5}})
    assert sorted(os.listdir(tmp_path)) == ["model.pt", "train_stats.pkl"]
    assert pickle.load(open(tmp_path / "train_stats.pkl", "rb"))["epoch"] == 7
    torch.manual_seed(2)
    m2 = cfg.get_model()
    stats = load_state(str(tmp_path), m2, map_location="cpu")
    assert stats["epoch"] == 7 and all(torch.equal(a, b) for a, b in zip(m1.state_dict().values(), m2.state_dict().values()))
    assert load_state(str(tmp_path / "nope"), m2) == {"epoch": 1}


def test_epoch_permutation_equals_dataloader_shuffle():
    from paths_amd.train import epoch_permutation
    data = list(range(23))
    torch.manual_seed(123)
    ref = [int(x) for x in torch.utils.data.DataLoader(data, batch_size=1, shuffle=True)]
    ref2 = [int(x) for x in torch.utils.data.DataLoader(data, batch_size=1, shuffle=True)]
    torch.manual_seed(123)
    assert epoch_permutation(23, True) == ref and epoch_permutation(23, True) == ref2
    assert epoch_permutation(5, False) == [0, 1, 2, 3, 4]
    # a non-shuffling loader consumes one draw of the global generator per pass too (its iterator's base seed): the order of the
    # NEXT shuffled epoch depends on it
    torch.manual_seed(5)
    a = [int(x) for x in torch.utils.data.DataLoader(data, batch_size=1, shuffle=True)]
    _ = [x for x in torch.utils.data.DataLoader(data, batch_size=4, shuffle=False)]
    b = [int(x) for x in torch.utils.data.DataLoader(data, batch_size=1, shuffle=True)]
    torch.manual_seed(5)
    assert epoch_permutation(23, True) == a and epoch_permutation(23, False) == data and epoch_permutation(23, True) == b


def test_importance_map_weighting():
    """Two levels: a 2x1 level-0 grid, the second patch expanded into two children."""
    from paths_amd.heatmap import importance_map
    levels = [{"locs": np.array([[0, 0], [256, 0]]), "importance": np.array([0.2, 0.8])},
              {"locs": np.array([[512, 0], [768, 256]]), "importance": np.array([0.5, 0.1])}]
    m = importance_map(levels, (2, 1))
    assert m.shape == (4, 2)
    np.testing.assert_allclose(m[0:2], 0.2 + 1e-4)                                   # never expanded
    np.testing.assert_allclose(m[2, 0], 0.8 + 1e-4 + 0.5 * (0.5 + 1e-4))
    np.testing.assert_allclose(m[3, 1], 0.8 + 1e-4 + 0.5 * (0.1 + 1e-4))
    np.testing.assert_allclose(m[2, 1], 0.8 + 1e-4)                                  # child filtered out (background)


def test_lagged_scale_registry_resets():
    """The training forward's lagged fp16 weight scales are dropped when weights are replaced wholesale."""
    from paths_amd import ops
    a, b = torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)
    ops._lagged_store(a)["w"] = [2.0, None, None]
    assert "w" in ops._lagged_store(a) and "w" not in ops._lagged_store(b)      # kept on the module, not keyed by id()
    ops.reset_lagged_scales()
    assert "w" not in ops._lagged_store(a)
