#!/usr/bin/env python3
"""Development aid: the aggregator on the weight-stationary token-layer path (csrc/tlayer_ws.hip) against the previous kernels on
the same random tokens, and event-timed per kernel group (single stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from paths_amd import _lib, ops

dev = torch.device("cuda:0")
cfg, model, sd = bench.build_model(2048, dev)
mc = model.procs[0].config
torch.manual_seed(0)


def run(B, T, lens, ws, reps=0):
    lvl = ops.pack_level(model.procs[1])
    g = torch.Generator(device=dev); g.manual_seed(T * 7 + B)
    tokens = torch.randn(B, T, 128, device=dev, generator=g)
    num_ims = torch.tensor([n - 1 for n in lens], device=dev, dtype=torch.int64)
    ctx_prev = torch.randn(B, 128, device=dev, generator=g)
    ops.TLAYER_WS = ws
    out = ops._aggregator_forward(mc, lvl, tokens, num_ims, ctx_prev, None)
    torch.cuda.synchronize()
    times = {}
    if reps:
        ev = []

        def timer(name, launch, meta):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = launch(); e1.record(); ev.append((name, e0, e1)); return r
        ops.KERNEL_TIMER, ops.TIMER_ALL = timer, True
        for _ in range(reps):
            ops._aggregator_forward(mc, lvl, tokens, num_ims, ctx_prev, None)
        torch.cuda.synchronize()
        ops.KERNEL_TIMER, ops.TIMER_ALL = None, False
        for n, e0, e1 in ev:
            times.setdefault(n, []).append(e0.elapsed_time(e1) * 1e3)
        times = {k: round(sorted(v)[len(v) // 2], 1) for k, v in times.items()}
        t0 = time.perf_counter()
        for _ in range(reps):
            ops._aggregator_forward(mc, lvl, tokens, num_ims, ctx_prev, None)
        torch.cuda.synchronize()
        times["span_wall_us"] = round((time.perf_counter() - t0) / reps * 1e6, 1)
    return out, times


ok = True
CASES = [] if "--time-only" in sys.argv else [(2, 65, [65, 30]), (3, 300, [300, 37, 129]), (8, 2049, [2049, 1844, 1850, 1790, 1900, 1844, 700, 1])]
for B, T, lens in CASES:
    o_old, _ = run(B, T, lens, False)
    o_new, _ = run(B, T, lens, True)
    for key in ("logits", "ctx_slide"):
        err = float((o_old[key] - o_new[key]).abs().max())
        good = err < 5e-6 and bool(torch.isfinite(o_new[key]).all())
        ok &= good
        print(f"B={B} T={T} {key:10s} max|new - old| = {err:.3e} {'OK' if good else 'FAIL'}", flush=True)
lens = [1845, 1850, 1838, 1860, 1841, 1849, 1852, 1844]
for ws in ((True,) if "--ws-only" in sys.argv else (False, True)):
    _, t = run(8, 2049, lens, ws, reps=20)
    print("ws" if ws else "old", t, flush=True)
print("ALL OK" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
