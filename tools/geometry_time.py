"""Time the 5-level K=2048 recursion (8 resident slides, eval) at an aggregator geometry other than the shipped one
(shape-generic kernels) next to the shipped geometry, and the training step: tools/geometry_time.py [trans_dim heads hidden]."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paths_amd import synthetic as syn, utils as putils
from paths_amd.config import Config
from paths_amd.data_utils.slide import DeviceSlide, DeviceSlideBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
td, heads, hid = (int(a) for a in (sys.argv[1:4] + ["192", "4", "128"][len(sys.argv) - 1:]))
K, spg = 2048, 8


def build(td, heads, hid):
    cfg = Config.load(os.path.join(ROOT, "tests", "golden", "sample"), test_mode=True)
    cfg.model_config.trans_dim, cfg.model_config.trans_heads, cfg.model_config.importance_mlp_hidden_dim = td, heads, hid
    cfg.top_k_patches = [K // 4] * (cfg.num_levels - 1)
    model = cfg.get_model()
    sd = syn.make_state_dict(0, {k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return cfg, model.to(dev).eval()


slides = DeviceSlideBatch([DeviceSlide.synthetic(1234, i, (32, 64), device=dev) for i in range(spg)])
labels = np.asarray([s.synthetic_spec.label(4) for s in slides.slides], np.int64)
batch = {"slide": slides, "survival_bin": torch.from_numpy(labels[:, 0]), "censored": torch.from_numpy(labels[:, 1])}
for geo in (((td, heads, hid),) if os.environ.get("GEO_ONLY") else ((128, 4, 128), (td, heads, hid))):
    cfg, model = build(*geo)
    with torch.no_grad():
        for _ in range(3):
            putils.recurse(model, slides, cfg.top_k_patches, cfg.num_levels, check_status=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            putils.recurse(model, slides, cfg.top_k_patches, cfg.num_levels, check_status=False)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    with torch.no_grad():
        tp = putils.TapedRecursion(model, slides, cfg.top_k_patches, cfg.num_levels).record()
        for _ in range(3):
            tp.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            tp.replay()
        torch.cuda.synchronize()
        el_t = time.perf_counter() - t0
        tp.close()
    line = (f"trans_dim {geo[0]} / {geo[1]} heads / hidden {geo[2]}: inference {el / n * 1e3:.3f} ms per 8-slide step = {spg * n / el:.0f} slides/s (eager launches), "
            f"{spg * n / el_t:.0f} slides/s (launch tape)")
    if os.environ.get("GEO_NO_TRAIN"):
        print(line, flush=True)
        continue
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay)
    for _ in range(3):
        putils.train_step(model, opt, batch, cfg.num_levels, cfg.top_k_patches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        putils.train_step(model, opt, batch, cfg.num_levels, cfg.top_k_patches)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(line + f"; training {el / 5 * 1e3:.2f} ms per step = {spg * 5 / el:.0f} slides/s (dropout {cfg.model_config.dropout})", flush=True)
    del model, opt
    torch.cuda.empty_cache()
