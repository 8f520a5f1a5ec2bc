#!/usr/bin/env python3
"""Attention of the bench shape (8 slides, T = 2049 tokens) with ragged slides (~1,850 valid) against FULL slides (2,048 patches + the
special token = 2,049 = 8 x 256 + 1 queries): does the stray ninth query block of 256 cost a second round of workgroups?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from paths_amd import _lib
dev = torch.device("cuda:0")
B, H, T, hd, planes = 8, 4, 2049, 32, 2
g = torch.Generator(device=dev); g.manual_seed(0)
q, k, v = ((torch.rand(B, H, T, hd, device=dev, generator=g) * 2 - 1) for _ in range(3))
ws = torch.empty(int(_lib.load().paths_attention_x6_workspace(B, T, H, hd, planes)), device=dev, dtype=torch.uint8)
o = torch.zeros(B, T, H * hd, device=dev)
p, st = _lib.ptr, _lib.stream()
for name, lens in (("ragged", [1844, 1850, 1839, 1861, 1822, 1847, 1855, 1830]), ("full", [2048] * 8), ("2047", [2047] * 8), ("one full", [2048] + [1850] * 7)):
    num_ims = torch.tensor(lens, device=dev)
    run = lambda ready: _lib.call("paths_attention_x6", p(q), p(k), p(v), p(o), None, p(num_ims), B, T, H, hd, 0, p(ws), planes, ready, st)
    run(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run(1)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:9s} valid patches {lens[0]}..: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us per launch", flush=True)
