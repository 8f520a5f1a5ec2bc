"""N > 1 rehearsal on ONE GPU: two fresh processes, both on device 0, collectives over gloo
(RCCL refuses two ranks of one communicator on one device).  Everything except the transport is the production path:
bench.py's launch contract, slide sharding, the fixed-list flat gradient all-reduce, an idle rank in a short last batch, a rank
without validation slides.

This file sorts first so that its child processes are started BEFORE anything in the pytest process has touched the GPU
(the box forbids exec from a process that has initialised the GPU); nothing here calls torch.cuda in the parent.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_children(cmds_envs, timeout=900):
    """Start the child processes with stdout / stderr in temporary files (a full pipe cannot stall a rank that another rank is
    waiting for inside a collective), wait for all of them, and ALWAYS reap them: on a timeout or a failure the survivors are
    killed, so no orphan keeps the GPU."""
    import tempfile
    import time
    files = [(tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")) for _ in cmds_envs]
    procs = []
    try:
        for (cmd, env), (fo, fe) in zip(cmds_envs, files):
            procs.append(subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=fo, stderr=fe, text=True))
        deadline = time.time() + timeout
        while any(p.poll() is None for p in procs):
            if time.time() > deadline or any(p.poll() not in (None, 0) for p in procs):
                break
            time.sleep(0.1)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    outs = []
    for fo, fe in files:
        fo.seek(0); fe.seek(0)
        outs.append((fo.read(), fe.read()))
        fo.close(); fe.close()
    assert all(p.returncode == 0 for p in procs), "\n".join(f"rank {i} rc={p.returncode}\n" + o[1][-3000:] for i, (p, o) in enumerate(zip(procs, outs)))
    return outs


def _spawn(argv, port, extra_env=None, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", PATHS_DIST_BACKEND="gloo",
               PATHS_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    return _run_children([([sys.executable] + argv, dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)], timeout)


@pytest.mark.parametrize("mode", ["infer", "train"])
def test_bench_two_ranks_one_device(mode):
    outs = _spawn(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--slides-per-gpu", "2", "--k", "256",
                   "--no-cpu-baseline", "--mode", mode], 29741 + (mode == "train"))
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # rank 0 prints THE line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and rec["config"]["global_batch"] == 4 and rec["unit"] == "slides/s"
    assert abs(rec["value"] - 4 * 2 / (rec["ms_per_step"] * 2e-3)) / rec["value"] < 0.01       # whole-job aggregate over both ranks
    assert len(rec["per_rank_ms_per_step"]) == 2 and rec["rank_imbalance_max_over_min"] >= 1.0    # each rank's own step time (load imbalance)
    ids = []
    for r in range(2):
        tag = [l for l in outs[r][1].splitlines() if l.startswith(f"[bench rank {r}/2] slide ids ")]
        assert len(tag) == 1
        ids.append(set(json.loads(tag[0].split("slide ids ")[1])))
    assert ids[0] == {0, 1} and ids[1] == {2, 3} and not (ids[0] & ids[1])
    if mode == "infer":
        assert "roofline" in rec and "cpu_baseline" not in rec        # the CPU leg is rank 0 at N=1 only


WORKER = r'''
import os, sys, json, tempfile
sys.path.insert(0, os.environ["PATHS_ROOT"])
import numpy as np, torch
import torch.distributed as dist
from paths_amd import distributed as pd, utils as putils, autograd as pag
from paths_amd.config import Config
from paths_amd.data_utils.slide import DeviceSlideBatch
from paths_amd.train import synthetic_dataset, train_loop
rank, world, _ = pd.env_rank_world()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
pd.init("gloo")
cfg = Config.load(os.path.join(os.environ["PATHS_ROOT"], "tests", "golden", "sample"), test_mode=True)
cfg.model_config.dropout = 0.0
cfg.num_levels, cfg.top_k_patches, cfg.batch_size = 3, [8, 8], [4, 4, 4]
cfg.num_epochs, cfg.lr, cfg.early_stopping, cfg.eval_epochs, cfg.min_epochs = 2, 2e-4, True, 1, 0
torch.manual_seed(0)
model = cfg.get_model().to(dev).train()
ds = synthetic_dataset(8, (6, 6), 3, dev, seed=77)

def batch_of(items):
    return {"slide": DeviceSlideBatch([it["slide"] for it in items]), "survival_bin": torch.as_tensor([it["survival_bin"] for it in items]),
            "censored": torch.as_tensor([it["censored"] for it in items])}

# (1) 2-rank gradients through the REAL all-reduce == 1-rank gradients of the same global batch of 3 slides (ranks hold 2 + 1)
glob = ds[:3]
model.zero_grad(set_to_none=True)
putils.forward_backward(model, batch_of(glob), 3, cfg.top_k_patches, "survival", 3)
full = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
mine = [glob[i] for i in pd.shard_range(3, rank, world)]
model.zero_grad(set_to_none=True)
putils.forward_backward(model, batch_of(mine), 3, cfg.top_k_patches, "survival", 3)
pd.allreduce_gradients(model, num_levels=3)
got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
assert set(got) == set(full), set(got) ^ set(full)
worst = max(float((got[n] - full[n]).abs().max() / full[n].abs().max().clamp_min(1e-20)) for n in full)
assert worst < 2e-5, worst
# (2) the epoch loop: 5 training slides in batches of 4 -> in the last batch rank 1 holds NO slide; 1 validation slide -> rank 1
# evaluates nothing; both must stay in lock-step through every collective and end with identical replicas
logs = []
stats = train_loop(model, ds[3:8], ds[:1], ds[1:3], cfg, os.environ["PATHS_MODEL_DIR"], log=logs.append)   # shared model dir
digest = float(sum(p.double().sum() for p in model.parameters()))
objs = [None, None]
dist.all_gather_object(objs, (digest, stats["train_loss"], stats["val_c-index"]))
assert objs[0] == objs[1], objs
assert set(stats["train_loss"]) == {1, 2} and all(np.isfinite(v) for v in stats["train_loss"].values())
pd.barrier()
print(json.dumps({"rank": rank, "worst_grad_rel_err": worst, "digest": digest}))
'''


def test_two_rank_training_equals_one_rank_and_survives_idle_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    mdir = tmp_path / "model"
    mdir.mkdir()
    outs = _spawn([str(script)], 29745, {"PATHS_MODEL_DIR": str(mdir)})
    recs = [json.loads([l for l in o[0].splitlines() if l.startswith("{")][-1]) for o in outs]
    assert recs[0]["digest"] == recs[1]["digest"] and max(r["worst_grad_rel_err"] for r in recs) < 2e-5


def test_bench_self_launch_without_a_launcher():
    """``python bench.py --gpus 2`` with no RANK / WORLD_SIZE in the environment: the parent (which never touches the GPU) starts
    the two ranks itself, relays rank 0's single JSON line and exits 0; with one device the ranks rehearse over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "PATHS_DIST_BACKEND")}
    env.update(PATHS_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = _run_children([([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--slides-per-gpu", "2", "--k", "256",
                           "--no-cpu-baseline", "--train-steps", "1", "--sustain", "0", "--breakdown-steps", "0"], env)])[0]
    lines = [l for l in out[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["value"] > 0
    assert rec["train"]["allreduce_ms"] is not None and rec["train"]["ms_per_step"] > 0       # the gradient all-reduce ran and was timed
    # a WORLD_SIZE that contradicts --gpus is an error message, not a bare assert
    bad = subprocess.run([sys.executable, "bench.py", "--gpus", "8"], env=dict(env, WORLD_SIZE="1", RANK="0"), cwd=ROOT, capture_output=True, text=True)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def test_training_step_on_two_host_cores():
    """8 ranks share one host's cores in the driver's N = 8 run and the training step's host side (~13.5 ms of launch enqueue) is
    nearly as long as its device side (~16 ms): a rank pinned to TWO host cores (os.sched_setaffinity at process start, before
    anything touches the GPU: bench.py --cores-per-rank 2) must not run its step more than 10 % slower than with the box's whole
    share.  (the 1 -> 8 curve itself needs the node.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "PATHS_DIST_BACKEND")}
    env.update(PATHS_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    ms = {}
    for cores in (0, 2):
        out = _run_children([([sys.executable, "bench.py", "--mode", "train", "--steps", "8", "--warmup", "3", "--cores-per-rank", str(cores)], env)])[0]
        rec = json.loads([l for l in out[0].splitlines() if l.startswith("{")][-1])
        ms[cores] = rec["ms_per_step"]
        assert rec["per_rank_ms_per_step"] and rec["rank_imbalance_max_over_min"] == 1.0
        if cores:
            assert "pinned to host cores" in out[1] and rec["cores_per_rank"] == 2
    print("training step, whole CPU share vs two cores:", ms)
    assert ms[2] <= 1.10 * ms[0], ms


RCCL_WORKER = r'''
import os, sys, json, time
sys.path.insert(0, os.environ["PATHS_ROOT"])
import torch
import torch.distributed as dist
from paths_amd import distributed as pd, autograd as pag
from paths_amd.config import Config
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
rank, world = pd.init("nccl", dev, force=True)            # RCCL communicator of ONE rank on the one GPU
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
cfg = Config.load(os.path.join(os.environ["PATHS_ROOT"], "tests", "golden", "sample"), test_mode=True)
model = cfg.get_model().to(dev)
params = pag.live_grad_params(model, cfg.num_levels)
torch.manual_seed(1)
for p in params[::2]:                                        # every other gradient missing: must count as zeros
    p.grad = torch.randn_like(p)
want = {id(p): (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for p in params}
n = sum(p.numel() for p in params)
pd.TIME_ALLREDUCE = True
times = []
for i in range(5):
    pd.allreduce_gradients(model, num_levels=cfg.num_levels)
    torch.cuda.synchronize()
    a, b = pd.LAST_ALLREDUCE_EVENTS
    times.append(a.elapsed_time(b))
assert all(torch.equal(p.grad, want[id(p)]) for p in params)             # SUM over one rank = identity, zeros filled in
rows = pd.gather_rows(torch.arange(12, device=dev, dtype=torch.float32).view(3, 4), 3)
assert torch.equal(rows.cpu(), torch.arange(12, dtype=torch.float32).view(3, 4))
assert pd.max_over_ranks(1.25, dev) == 1.25
pd.barrier()
print(json.dumps({"bucket_mb": n * 4 / 2**20, "allreduce_ms": sorted(times)[len(times) // 2], "first_ms": times[0]}))
dist.destroy_process_group()
'''


def test_rccl_collectives_on_one_gpu(tmp_path):
    """The backend == "nccl" (RCCL) branches of paths_amd.distributed on a real device: a forced one-rank communicator runs the flat
    29 MB gradient all-reduce (missing gradients = zeros), the row all-gather, the MAX-reduce and the barrier; prints the all-reduce
    time.  (Two ranks of one communicator cannot share a device, so N > 1 over RCCL needs the multi-GPU node the driver has.)"""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(PATHS_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29751")
    out = _run_children([([sys.executable, str(script)], env)])[0]
    rec = json.loads([l for l in out[0].splitlines() if l.startswith("{")][-1])
    print("RCCL one-rank all-reduce:", rec)
    assert 25 < rec["bucket_mb"] < 32 and rec["allreduce_ms"] < 50
