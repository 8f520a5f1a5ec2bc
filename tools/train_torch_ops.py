#!/usr/bin/env python3
"""Which lines of paths_amd issue the torch-native kernels of a training step (fills, adds, copies, cats ...): torch.profiler with
stacks, aggregated by (aten op, innermost paths_amd frame).  Bench shape, 3 profiled steps."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from paths_amd import utils as putils
from paths_amd.data_utils.slide import DeviceSlide, DeviceSlideBatch
from paths_amd.optim import HipAdamW
dev = torch.device("cuda:0")
cfg, model, sd = bench.build_model(2048, dev)
slides = DeviceSlideBatch([DeviceSlide.synthetic(1234, i, (32, 64), device=dev) for i in range(8)])
model.train()
labels = np.asarray([s.synthetic_spec.label(4) for s in slides.slides], np.int64)
batch = {"slide": slides, "survival_bin": torch.from_numpy(labels[:, 0]), "censored": torch.from_numpy(labels[:, 1])}
opt = HipAdamW(model.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay)
for _ in range(4):
    putils.train_step(model, opt, batch, cfg.num_levels, cfg.top_k_patches)
torch.cuda.synchronize()
NSTEP = 3
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
agg = collections.Counter()
SKIP = ("aten.view", "aten.detach", "aten.empty", "aten.as_strided", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.slice", "aten.select", "aten.unsqueeze",
        "aten.squeeze", "aten.expand", "aten.alias", "aten.reshape", "aten.permute", "aten.split", "aten.unbind", "aten.new_empty", "aten.empty_like", "aten._reshape_alias",
        "aten.lift_fresh", "aten.narrow", "aten.unfold", "aten.diagonal", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.numel", "aten.result_type", "aten.empty_strided")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            fr = "?"
            for f in reversed(traceback.extract_stack(limit=14)):
                if "paths_amd" in f.filename and "_python_dispatch" not in f.filename:
                    fr = f"{os.path.relpath(f.filename, ROOT)}:{f.lineno} {f.line[:70] if f.line else ''}"
                    break
            agg[(name, fr)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    for _ in range(NSTEP):
        putils.train_step(model, opt, batch, cfg.num_levels, cfg.top_k_patches)
    torch.cuda.synchronize()
tot = 0
for (name, frame), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    tot += n
    if n >= NSTEP:
        print(f"{n / NSTEP:7.1f} per step  {name:34s} {frame[:130]}")
print("aten ops per step (views / allocations excluded):", tot / NSTEP)
