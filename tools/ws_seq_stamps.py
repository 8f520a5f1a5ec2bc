#!/usr/bin/env python3
"""In-kernel phase stamps of the row-chain kernel (paths_token_layer_ws, POST) INSIDE the recursion against the same launch repeated
on its own: where the difference between its in-sequence time and its hot-loop time goes.  Needs a -DPATHS_WS_STAMPS build
(PATHS_HIP_LIB).  One-stream tape of the bench batch; level 1."""
import ctypes, os, sys, statistics as st
os.environ["PATHS_OVERLAP_AGGREGATOR"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from paths_amd import _lib, ops
from paths_amd import utils as putils
from paths_amd.data_utils.slide import DeviceSlide, DeviceSlideBatch

dev = torch.device("cuda:0")
lib = _lib.load()
assert hasattr(lib, "paths_ws_stamp_buffer"), "build tlayer_ws.hip with -DPATHS_WS_STAMPS and point PATHS_HIP_LIB at it"
cfg, model, sd = bench.build_model(2048, dev, None)
slides = DeviceSlideBatch([DeviceSlide.synthetic(1234, i, bench.BASE_SHAPES[2048], device=dev) for i in range(8)])
tr = putils.TapedRecursion(model, slides, cfg.top_k_patches, cfg.num_levels).record()
names = [n for _, _, n in tr.tape]
chains = [i for i, n in enumerate(names) if n == "paths_token_layer_ws"]
fins = [i for i, n in enumerate(names) if n == "paths_importance_qkv_x6"]
print("token_layer_ws calls at", chains, "importance_qkv_x6 at", fins)
stamps = torch.zeros((33 * 8, 16), device=dev, dtype=torch.int64)
lib.paths_ws_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.paths_ws_stamp_buffer(stamps.data_ptr())


def show(tag, which):
    s = stamps.cpu().numpy()
    live = s[:, 0] > 0
    rel = s[live][:, which] - s[live][:, [0]]
    med = [int(st.median(rel[:, j])) for j in range(len(which))]
    span = (s[live][:, which[-1]].max() - s[live][:, 0].min())
    print(f"{tag:34s} wgs {int(live.sum()):3d}  stamps {which}: {med}  first start -> last end {int(span)} cycles")


CH = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15]
FI = [0, 2, 3, 4, 1, 11, 12, 13, 14, 15]
for lvl in (1, 2):
    ci = chains[lvl]
    fi = [i for i in fins if i < ci][-1]                      # the tokens / in_proj finish of the same level
    torch.cuda.synchronize()
    stamps.zero_(); tr._play(tr.tape[:fi + 1]); torch.cuda.synchronize(); show(f"level {lvl} finish, in sequence", FI)
    for _ in range(3):
        stamps.zero_(); tr._play(tr.tape[fi:fi + 1]); torch.cuda.synchronize()
    show(f"level {lvl} finish, repeated alone", FI)
    stamps.zero_(); tr._play(tr.tape[fi + 1:ci + 1]); torch.cuda.synchronize(); show(f"level {lvl} row chain, in sequence", CH)
    for _ in range(3):
        stamps.zero_(); tr._play(tr.tape[ci:ci + 1]); torch.cuda.synchronize()
    show(f"level {lvl} row chain, repeated alone", CH)
