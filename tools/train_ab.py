"""A/B the training step in ONE process: alternate settings of paths_amd.backward switches, 10 steps each, several rounds
(box-to-box and run-to-run clock differences are larger than most kernel-level gains).
usage: python tools/train_ab.py NAME=v1,v2 [NAME2=...]   e.g.  NT_X6_MIN_N=128,256"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import numpy as np
    from paths_amd import backward as bw, utils as putils
    sw = [a.split("=") for a in sys.argv[1:]]
    name, vals = sw[0][0], sw[0][1].split(",")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from paths_amd import _lib
    from paths_amd.data_utils.slide import DeviceSlide, DeviceSlideBatch
    _lib.load()
    cfg, model, _ = bench.build_model(2048, dev, None)
    model.train()
    slides = DeviceSlideBatch([DeviceSlide.synthetic(1234, i, bench.BASE_SHAPES[2048], device=dev) for i in range(8)])
    labels = np.asarray([s.synthetic_spec.label(4) for s in slides.slides], np.int64)
    batch = {"slide": slides, "survival_bin": torch.from_numpy(labels[:, 0]), "censored": torch.from_numpy(labels[:, 1])}
    opt = torch.optim.AdamW(model.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay)

    def run(n):
        for _ in range(n):
            putils.train_step(model, opt, batch, cfg.num_levels, cfg.top_k_patches, global_batch=8)
        torch.cuda.synchronize()

    run(3)
    res = {v: [] for v in vals}
    for rnd in range(4):
        for v in vals:
            cur = getattr(bw, name)
            setattr(bw, name, type(cur)(v))
            run(2)
            t0 = time.perf_counter()
            run(10)
            res[v].append((time.perf_counter() - t0) / 10 * 1e3)
    for v in vals:
        print(f"{name}={v}: ms/step {[round(x, 2) for x in res[v]]}", flush=True)


if __name__ == "__main__":
    main()
