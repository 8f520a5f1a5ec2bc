#!/usr/bin/env python3
"""paths_gemm_nt_fp8 alone at the five shapes of the stress geometry's aggregator (M = 8 x 8193 token rows): TFLOP/s per shape, with the
fp32 output and with the e4m3 hand-over output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from paths_amd import _lib
dev = torch.device("cuda:0")
lib = _lib.load()
M = 8 * 8193
Mp = (M + 255) // 256 * 256
st = _lib.stream()
one = torch.ones(1, device=dev)
for name, N, K, res in (("in_proj", 4608, 1536, False), ("out_proj", 1536, 1536, True), ("linear1", 6144, 1536, False), ("linear2", 1536, 6144, True), ("kv_proj", 3072, 1536, False)):
    a8 = torch.randint(0, 120, (Mp, K), device=dev, dtype=torch.uint8)
    w8 = torch.randint(0, 120, ((N + 255) // 256 * 256, K), device=dev, dtype=torch.uint8)
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev)
    r = torch.zeros(M, N, device=dev) if res else None
    out8 = torch.zeros(Mp, N, device=dev, dtype=torch.uint8)
    amax = torch.zeros(1, device=dev, dtype=torch.int32)

    def f32():
        _lib.call("paths_gemm_nt_fp8", a8.data_ptr(), w8.data_ptr(), one.data_ptr(), one.data_ptr(), bias.data_ptr(), out.data_ptr(), N, M, N, K, 0,
                  r.data_ptr() if res else None, N if res else 0, st)

    def o8():
        _lib.call("paths_gemm_nt_fp8_out8", a8.data_ptr(), w8.data_ptr(), one.data_ptr(), one.data_ptr(), bias.data_ptr(), out8.data_ptr(), one.data_ptr(),
                  amax.data_ptr(), M, N, K, 1, st)
    line = f"{name:9s} N={N:5d} K={K:5d}"
    for tag, fn in (("fp32 out", f32), ("e4m3 out", o8)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record(); b.synchronize()
        us = a.elapsed_time(b) * 100
        line += f" | {tag}: {us:8.1f} us = {2 * M * N * K / us / 1e6:7.1f} TFLOP/s"
    print(line, flush=True)
    del a8, w8, out, r, out8
