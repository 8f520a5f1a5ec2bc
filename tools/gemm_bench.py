#!/usr/bin/env python3
"""Micro-benchmark of the LSTM GEMM launches at the bench shape (development aid).
usage: gemm_bench.py [reps] [M]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from paths_amd import _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
D, Hc = 1024, 256
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
x = torch.rand(M, D, device=dev, generator=g) * 2 - 1
st0 = torch.rand(M, D + Hc, device=dev, generator=g) * 2 - 1
wg = (torch.rand(3 * Hc + D, 2 * D, device=dev, generator=g) * 2 - 1) / 45.0
bg = torch.rand(3 * Hc + D, device=dev, generator=g)
wm = (torch.rand(D, Hc, device=dev, generator=g) * 2 - 1) / 16.0
bm = torch.rand(D, device=dev, generator=g)
so = torch.empty(M, D + Hc, device=dev); y = torch.empty(M, D, device=dev); ws = torch.empty(M, D, device=dev)
p = _lib.ptr
def run(ph):
    _lib.call("paths_lstm_cell", p(x), D, st0.data_ptr(), D + Hc, st0.data_ptr() + 4 * D, D + Hc, p(wg), p(bg), p(wm), p(bm),
              p(so), D + Hc, p(y), D, p(ws), None, None, None, None, M, D, Hc, None, 1, ph, _lib.stream())
for ph, name, flop in ((1, "c-part", 2.0 * M * 2 * D * 3 * Hc), (2, "o-gate", 2.0 * M * 2 * D * D), (4, "h-part", 2.0 * M * Hc * D)):
    for _ in range(3): run(ph)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run(ph)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name}: {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s", flush=True)
