#!/usr/bin/env python3
"""Throughput of the headline recursion on slides WITHOUT background cells (p_bg = 0: every level keeps all 2,048 children: 2,049 tokens,
the worst case for the 64-token / 256-query tilings) against the bench's slides (p_bg = 0.1, ~1,850 valid patches per level)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from paths_amd import utils as putils
from paths_amd.data_utils.slide import DeviceSlide, DeviceSlideBatch
dev = torch.device("cuda:0")
cfg, model, sd = bench.build_model(2048, dev, None)
for p_bg in (0.1, 0.0, 0.1, 0.0):
    slides = DeviceSlideBatch([DeviceSlide.synthetic(1234, i, bench.BASE_SHAPES[2048], device=dev, p_bg=p_bg) for i in range(8)])
    tr = []
    with torch.no_grad():
        putils.recurse(model, slides, cfg.top_k_patches, cfg.num_levels, trace=tr, check_status=False)
    valid = [int(t["num_ims"].sum()) for t in tr]
    tape = putils.TapedRecursion(model, slides, cfg.top_k_patches, cfg.num_levels).record()
    for _ in range(5): tape.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): tape.replay()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"p_bg {p_bg}: valid patches per level {valid}: {el / 40 * 1e3:.3f} ms per 8-slide step = {8 * 40 / el:.0f} slides/s = {sum(valid) * 40 / el / 1e6:.2f} M patches/s", flush=True)
    tape.close(); del tape, slides
    torch.cuda.empty_cache()
