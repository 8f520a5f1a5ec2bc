#!/usr/bin/env python3
"""Soak: replay the recorded launch tape of the bench batch for N seconds and compare every replay's outputs (logits, slide context, the
kept index tables of every level) BITWISE with the first replay's - the tape works on the same resident slides, so any difference is a race
or a hardware fault.  usage: soak_replay.py [seconds] [trans_dim]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from paths_amd import utils as putils
from paths_amd.data_utils.slide import DeviceSlide, DeviceSlideBatch
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
td = int(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device("cuda:0")
cfg, model, sd = bench.build_model(2048, dev, None, trans_dim=td)
slides = DeviceSlideBatch([DeviceSlide.synthetic(1234, i, bench.BASE_SHAPES[2048], device=dev) for i in range(8)])
tape = putils.TapedRecursion(model, slides, cfg.top_k_patches, cfg.num_levels).record()
out = tape.replay(); torch.cuda.synchronize()
keys = [k for k, v in out.items() if torch.is_tensor(v)]
ref = {k: out[k].clone() for k in keys}
print("compared per replay:", {k: tuple(ref[k].shape) for k in keys}, flush=True)
bad = torch.zeros((), device=dev, dtype=torch.int64)
n, t0, last = 0, time.time(), time.time()
while time.time() - t0 < secs:
    for _ in range(50):
        out = tape.replay()
        for k in keys:
            a, b = out[k], ref[k]
            bad += (a.view(torch.int32) != b.view(torch.int32)).sum() if a.dtype == torch.float32 else (a != b).sum()
        n += 1
    torch.cuda.synchronize()
    if time.time() - last > 20:
        print(f"{n} replays, differing elements so far: {int(bad)}", flush=True); last = time.time()
print(f"soak done: {n} replays in {time.time() - t0:.1f} s (trans_dim {td or 128}), differing elements: {int(bad)}")
sys.exit(1 if int(bad) else 0)
