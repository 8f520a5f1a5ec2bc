#!/usr/bin/env python3
"""In-kernel phase stamps of the token-0 tail INSIDE the recursion against the same launch repeated on its own (-DPATHS_T0_STAMPS build via
PATHS_HIP_LIB).  One-stream tape of the bench batch; levels 1 and 2."""
import ctypes, os, sys, statistics as st
os.environ["PATHS_OVERLAP_AGGREGATOR"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from paths_amd import _lib
from paths_amd import utils as putils
from paths_amd.data_utils.slide import DeviceSlide, DeviceSlideBatch
dev = torch.device("cuda:0")
lib = _lib.load()
assert hasattr(lib, "paths_t0_stamp_buffer")
cfg, model, sd = bench.build_model(2048, dev, None)
slides = DeviceSlideBatch([DeviceSlide.synthetic(1234, i, bench.BASE_SHAPES[2048], device=dev) for i in range(8)])
tr = putils.TapedRecursion(model, slides, cfg.top_k_patches, cfg.num_levels).record()
names = [n for _, _, n in tr.tape]
tails = [i for i, n in enumerate(names) if n == "paths_token0_tail_ws"]
stamps = torch.zeros((16 * 8, 16), device=dev, dtype=torch.int64)
lib.paths_t0_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.paths_t0_stamp_buffer(stamps.data_ptr())
W = [0, 1, 2, 3, 4, 5, 9]


def show(tag):
    s = stamps.cpu().numpy()
    live = s[:, 0] > 0
    t0 = s[live][:, 0].min()
    med = [int(st.median(s[live][:, j] - s[live][:, 0])) for j in W[:-1]]
    last = s[live][:, 9]; last = last[last > 0]
    print(f"{tag:32s} wgs {int(live.sum()):3d}  stamps {W[:-1]} (median, from the workgroup's own start): {med}   first start -> classifier done {int(last.max() - t0)}  start skew {int(s[live][:, 0].max() - t0)}")


for lvl in (1, 2):
    ti = tails[lvl]
    torch.cuda.synchronize(); stamps.zero_(); tr._play(tr.tape[:ti + 1]); torch.cuda.synchronize(); show(f"level {lvl} tail, in sequence")
    for _ in range(3):
        stamps.zero_(); tr._play(tr.tape[ti:ti + 1]); torch.cuda.synchronize()
    show(f"level {lvl} tail, repeated alone")
