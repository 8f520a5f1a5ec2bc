#!/usr/bin/env python3
"""cProfile of the host side of the training step (bench shape): where the ~13 ms of enqueue time per step go."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
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
for _ in range(5):
    putils.train_step(model, opt, batch, cfg.num_levels, cfg.top_k_patches)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    putils.train_step(model, opt, batch, cfg.num_levels, cfg.top_k_patches)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
print(s.getvalue()[:8000])
