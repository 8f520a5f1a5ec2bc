"""CPU-only tests (-m "not gpu"): the C-ABI library loads and exports every declared symbol, the reference-shaped
Python surface (config.json, state_dict keys, PatchBatch) holds, the synthetic generator is deterministic, and the
slide-sharding helpers are correct under a world_size-2 gloo job.  No kernel is launched here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLE = os.path.join(ROOT, "tests", "golden", "sample")


def header_symbols():
    text = open(os.path.join(ROOT, "include", "paths_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(paths_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from paths_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/paths_hip.h but not exported"
    # every ctypes signature corresponds to a declared symbol, with the same number of parameters
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "paths_hip.h")).read(), flags=re.S)
    for name, args in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\((.*?)\);" % name, text, flags=re.S)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip()]) == len(args), name
    lib.paths_abi_version.restype = ctypes.c_int
    assert lib.paths_abi_version() == 2


def test_invalid_arguments_are_reported_not_launched():
    """Host-side validation happens before any launch, so this is safe without a GPU."""
    from paths_amd import _lib
    lib = _lib.load()
    rc = lib.paths_attention_f32(None, None, None, None, None, None, 1, 16, 4, 64, 0, None)     # head_dim 64 unsupported
    assert rc == -1 and b"head_dim" in lib.paths_last_error()
    rc = lib.paths_topk(None, 0, None, 1, 100000, 5, None, 5, None, None)              # n_max too large
    assert rc == -1 and b"n_max" in lib.paths_last_error()
    with pytest.raises(_lib.PathsHipError):
        _lib.call("paths_layernorm_f32", None, None, None, None, 0, 128, 1e-5, None)


def test_config_surface_and_state_dict_keys():
    from oracle import paths_oracle as orc
    from paths_amd.config import Config, PATHSProcessorConfig
    cfg = Config.load(SAMPLE, test_mode=True)
    assert cfg.top_k_patches == [20] * 4 and cfg.batch_size == [32] * 5        # reference config.py:93-100
    assert cfg.power_levels() == [0.625, 1.25, 2.5, 5.0, 10.0]
    assert isinstance(cfg.model_config, PATHSProcessorConfig) and cfg.model_config.trans_dim == 128
    torch.manual_seed(0)
    model = cfg.get_model()
    sd = model.state_dict()
    shapes = orc.state_dict_shapes(orc.OracleConfig())
    assert set(sd.keys()) == set(shapes.keys())
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    assert sum(p.numel() for p in model.parameters()) == 9_881_881
    assert model.procs[0].ctx_dim() == (128, 1280)
    with pytest.raises(AssertionError):
        Config.from_dict({**{k: getattr(cfg, k) for k in ("base_power", "magnification_factor", "num_levels", "num_epochs",
                                                           "top_k_patches", "model_type", "wsi_dir", "csv_path")},
                          "model_config": {"lstm": True, "hierarchical_ctx": False}})


def test_initial_weights_match_reference_under_same_seed():
    """Sub-module construction order follows the reference, so manual_seed(s) + get_model() draws the same weights
    (digests captured from the reference import by tools/make_goldens.py)."""
    from tests.conftest import load_golden
    from paths_amd.config import Config
    try:
        g, info = load_golden("g0_init_digest")
    except FileNotFoundError:
        pytest.skip("g0 fixture not generated")
    cfg = Config.load(SAMPLE, test_mode=True)
    torch.manual_seed(info["seed"])
    sd = cfg.get_model().state_dict()
    for k in info["keys"]:
        a = sd[k].double()
        np.testing.assert_allclose([float(a.sum()), float(a.abs().sum())], g[k], rtol=1e-12, atol=1e-12, err_msg=k)


def test_patch_batch_contract():
    from paths_amd.data_utils.patch_batch import PatchBatch, from_batch
    B, N, D = 2, 5, 8
    kw = dict(locs=torch.zeros(B, N, 2, dtype=torch.int64), num_ims=torch.tensor([5, 3]), parent_inds=torch.zeros(B, N, dtype=torch.int64),
              ctx_slide=torch.zeros(B, 1, 4), ctx_patch=torch.zeros(B, N, 1, 6), fts=torch.zeros(B, N, D))
    pb = PatchBatch(**kw)
    assert pb.batch_size == 2 and pb.max_patches == 5 and pb.ctx_depth == 1
    assert pb.valid_inds.tolist() == [[True] * 5, [True] * 3 + [False] * 2]
    with pytest.raises(AssertionError):                       # reference patch_batch.py:50
        PatchBatch(**{**kw, "num_ims": torch.tensor([4, 3])})
    PatchBatch(**{**kw, "num_ims": torch.tensor([4, 3])}, strict=False)
    assert from_batch(kw, torch.device("cpu")).fts.shape == (B, N, D)


def test_pack_lstm_layout():
    from paths_amd import ops
    from paths_amd.model.interface import LSTMCell
    torch.manual_seed(1)
    cell = LSTMCell(128, 128, 64)
    pk = ops.pack_lstm(cell)
    assert pk["w_gates"].shape == (3 * 64 + 128, 256)
    # group j-block 1: rows 96..191 = forget[32:64] | remember[32:64] | map[32:64]
    assert torch.equal(pk["w_gates"][96:128], cell.forget_gate[0].weight[32:64])
    assert torch.equal(pk["w_gates"][128:160], cell.remember_gate[0].weight[32:64])
    assert torch.equal(pk["w_gates"][160:192], cell.remember_map[0].weight[32:64])
    assert torch.equal(pk["w_gates"][192:], cell.out_select_gate[0].weight)
    assert torch.equal(pk["b_gates"][128:160], cell.remember_gate[0].bias[32:64])
    assert ops.pack_lstm(cell) is pk                          # cached
    with torch.no_grad():
        cell.forget_gate[0].weight.add_(1.0)                  # version bump invalidates the cache
    assert ops.pack_lstm(cell) is not pk


def test_cpu_inputs_are_rejected_loudly():
    from paths_amd import _lib
    from paths_amd.model.interface import LSTMCell
    cell = LSTMCell(128, 128, 64)
    with pytest.raises(_lib.PathsHipError):
        cell(torch.zeros(2, 128), torch.zeros(2, 128), torch.zeros(2, 64))


def test_status_word_decoding():
    """paths_amd.utils.check_status_word: bit 0 asks for the careful re-run, bits 1 and 2 invalidate the results."""
    from paths_amd import utils as putils
    assert putils.check_status_word(0) is False and putils.check_status_word(torch.tensor([1])) is True
    assert putils.check_status_word(1, fallback_done=True) is False
    for code in (2, 4, 6, 3, 5):
        with pytest.raises(putils.RecursionError_):
            putils.check_status_word(code)


def test_synthetic_generator_properties():
    from paths_amd import synthetic as syn
    s = syn.SyntheticSlide(7, 3, (4, 5), dim=64, num_levels=3, p_bg=0.25)
    g1 = s.grid(1)
    assert g1.shape == (8, 10, 64) and g1.dtype == np.float32
    assert np.array_equal(g1, s.grid(1))
    x, y = np.array([3, 0, 7]), np.array([9, 0, 2])
    assert np.array_equal(s.rows(1, x, y), g1[x, y])
    bg = s.is_background(1, *np.meshgrid(np.arange(8), np.arange(10), indexing="ij"))
    assert np.array_equal(bg, g1.sum(-1) == 0) and 0 < bg.sum() < 80
    assert not s.is_background(0, np.arange(4), np.arange(4)).any()
    big = syn.cell_features(1, 0, 0, np.arange(64), np.arange(64), 1024, 0.0)
    assert abs(big.mean()) < 0.02 and abs(big.std() - 1.0) < 0.02 and np.abs(big).max() < 1.7320509
    assert int(syn.fmix32(np.uint64(1))) == 0x514E28B7          # murmur3 fmix32 known answer


GLOO_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["PATHS_ROOT"])
from paths_amd import distributed as pd
rank, world = pd.init("gloo")
n_total = 7
mine = pd.shard_range(n_total, rank, world)
local = torch.tensor([[float(i), float(i) * 2] for i in mine])
allrows = pd.gather_rows(local, n_total)
assert allrows.shape == (n_total, 2) and torch.equal(allrows[:, 0], torch.arange(n_total, dtype=torch.float32)), allrows
t = pd.max_over_ranks(1.0 + rank, torch.device("cpu"))
assert t == float(world), t
# flat-bucket gradient all-reduce over the FIXED live-parameter list: rank 0 is an active rank (real gradients, the unused
# classifiers of levels 0..3 have grad None), rank 1 holds no slide of the batch (zeros for exactly the same set)
from paths_amd import autograd as pag
from paths_amd.config import Config
cfg = Config.load(os.path.join(os.environ["PATHS_ROOT"], "tests", "golden", "sample"), test_mode=True)
torch.manual_seed(0)
model = cfg.get_model()
live = pag.live_grad_params(model, cfg.num_levels)
unused = [p for i in range(cfg.num_levels - 1) for p in model.procs[i].classification_layer.parameters()]
assert len({id(p) for p in live}) == len(live) and not ({id(p) for p in live} & {id(p) for p in unused})
assert sum(p.numel() for p in live) + sum(p.numel() for p in unused) + sum(p.numel() for p in pag.dead_params(model)) == 9881881
if rank == 0:
    for i, p in enumerate(live):
        p.grad = torch.full_like(p, 1.0 + (i % 3))
    pag.fill_dead_grads(model)
else:
    pag.zero_live_grads(model, cfg.num_levels)
pd.allreduce_gradients(model, num_levels=cfg.num_levels)
for i, p in enumerate(live):
    assert torch.equal(p.grad, torch.full_like(p, 1.0 + (i % 3))), i
# the reduced bucket is persistent storage whose 16-byte aligned slices ARE the gradients (float4 path of the one-launch AdamW)
flat = model._paths_grad_bucket[1]
assert all(p.grad.data_ptr() % 16 == 0 and p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in live)
pd.allreduce_gradients(model, num_levels=cfg.num_levels)       # second step: same storage, nothing re-allocated; values x world
assert model._paths_grad_bucket[1] is flat
for i, p in enumerate(live):
    assert torch.equal(p.grad, torch.full_like(p, 2.0 * (1.0 + (i % 3)))), i
    p.grad.div_(2.0)
assert all(p.grad is None for p in unused)
assert all(p.grad is not None and not p.grad.any() for p in pag.dead_params(model))
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
opt.step()
digest = float(sum(p.double().sum() for p in model.parameters()))
import torch.distributed as dist
objs = [None, None]; dist.all_gather_object(objs, digest)
assert objs[0] == objs[1], objs              # replicas stay identical after the step (same grads, same None set)
# evaluators on a rank that registered nothing (no val slide on this rank) still take part in the gathers
from paths_amd.eval import SurvivalEvaluator, SubtypeClassificationEvaluator
ev = SurvivalEvaluator("val"); ev2 = SubtypeClassificationEvaluator("val", 2)
if rank == 0:
    hz = torch.tensor([[0.9, 0.5, 0.5, 0.5], [0.1, 0.1, 0.1, 0.1], [0.5, 0.5, 0.5, 0.5]])
    ev.register({"censored": torch.tensor([0, 0, 1]), "survival": torch.tensor([1.0, 9.0, 5.0])}, hz, torch.tensor(0.7))
    ev2.register({"subtype": torch.tensor([0, 1, 1])}, torch.tensor([[2.0, 0.0], [0.0, 1.0], [0.5, 0.6]]), torch.tensor(0.3))
r1, r2 = ev.calculate(), ev2.calculate()
assert abs(r1["val_loss"] - 0.7) < 1e-6 and r1["val_c-index"] == 1.0 and r2["val_AUC"] == 1.0, (r1, r2)
pd.barrier()
print("rank", rank, "ok", list(mine))
'''


def test_slide_sharding_world_size_2_gloo(tmp_path):
    from paths_amd import distributed as pd
    for n in (1, 7, 8, 64):
        for w in (1, 2, 3, 8):
            parts = [list(pd.shard_range(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER)
    env = dict(os.environ, PATHS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "[0, 1, 2, 3]" in outs[0] and "[4, 5, 6]" in outs[1]


def test_clear_grads_keeps_the_dead_parameters_zero_views():
    """``autograd.clear_grads`` (what ``utils.train_step`` calls instead of ``optimizer.zero_grad(set_to_none=True)``): live parameters'
    gradients are dropped, the DEAD parameters (nn.Transformer's encoder and cross-attention matrices: reference SURVEY 3.3, zero
    gradients every step) keep their views of the shared zero buffer - unless somebody gave one of them a real gradient."""
    from paths_amd import autograd as pag
    from paths_amd.config import Config
    cfg = Config.load(SAMPLE, test_mode=True)
    torch.manual_seed(0)
    model = cfg.get_model()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    dead = pag.dead_params(model)
    dead_ids = {id(p) for p in dead}
    live = [p for p in model.parameters() if id(p) not in dead_ids]
    assert dead and live
    for p in live:
        p.grad = torch.ones_like(p)
    pag.fill_dead_grads(model)
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in dead)
    ptrs = [p.grad.data_ptr() for p in dead]
    pag.clear_grads(model, opt)
    assert all(p.grad is None for p in live)
    assert [p.grad.data_ptr() for p in dead] == ptrs                      # the same views, not re-created
    pag.fill_dead_grads(model)                                            # nothing to do now
    assert [p.grad.data_ptr() for p in dead] == ptrs
    dead[0].grad = torch.ones_like(dead[0])                               # a gradient that is NOT the shared zero view is cleared like any other
    pag.clear_grads(model, opt)
    assert dead[0].grad is None and dead[1].grad is not None
    pag.fill_dead_grads(model)
    assert float(dead[0].grad.abs().max()) == 0.0
