"""Host-side phase timing of the training step (K = 2048, 8 slides): how long the host takes to ENQUEUE the forward, the backward
and the optimizer, and how long it then waits for the GPU - i.e. which side bounds the step.  usage: python tools/train_host_phases.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from paths_amd import _lib, ops, utils as putils
from paths_amd.data_utils.slide import DeviceSlide, DeviceSlideBatch
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); _lib.load()
cfg, model, _ = bench.build_model(2048, dev, None); model.train()
slides = DeviceSlideBatch([DeviceSlide.synthetic(1234, i, bench.BASE_SHAPES[2048], device=dev) for i in range(8)])
labels = np.asarray([s.synthetic_spec.label(4) for s in slides.slides], np.int64)
batch = {"slide": slides, "survival_bin": torch.from_numpy(labels[:, 0]), "censored": torch.from_numpy(labels[:, 1])}
from paths_amd.optim import HipAdamW
opt = (torch.optim.AdamW if os.environ.get("PATHS_TORCH_ADAMW", "0") != "0" else HipAdamW)(model.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay)
for _ in range(4):
    putils.train_step(model, opt, batch, cfg.num_levels, cfg.top_k_patches, global_batch=8)
torch.cuda.synchronize()
# phase timing: forward enqueue, backward enqueue, optimizer, drain
import paths_amd.utils as U
from paths_amd import autograd as pag
tf = tb = to = td = 0.0
N = 10
for _ in range(N):
    opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = U.recurse_train(model, batch["slide"], cfg.top_k_patches, cfg.num_levels)
    outputs, loss = U.loss_from_logits(out["logits"], batch, "survival", 8)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    pag.fill_dead_grads(model)
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1; to += t3 - t2; td += t4 - t3
print(f"host: forward enqueue {tf/N*1e3:.2f} ms, backward enqueue {tb/N*1e3:.2f} ms, optimizer {to/N*1e3:.2f} ms, drain wait {td/N*1e3:.2f} ms, total {(tf+tb+to+td)/N*1e3:.2f} ms")
