#!/usr/bin/env python3
"""Accuracy of paths_attention_x6 (planes = 2) against an fp64 softmax(q k^T) v on uniform and on peaked score distributions.
PATHS_HIP_LIB selects the library (attention variants: tools/mkvariant.sh)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from paths_amd import _lib
dev = torch.device("cuda:0")
p, st = _lib.ptr, _lib.stream()
H, hd = 4, 32
for T, lens, qscale, tag in [(2049, [2049, 1844, 700, 1], 1.5, "flat"), (2049, [2049, 1844, 700, 33], 12.0, "peaked"), (300, [300, 37], 5.0, "mid"), (2049, [2049, 1500], 40.0, "very peaked")]:
    B = len(lens)
    g = torch.Generator(device=dev); g.manual_seed(T)
    q = (torch.rand(B, H, T, hd, device=dev, generator=g) * 2 - 1) * qscale
    k = (torch.rand(B, H, T, hd, device=dev, generator=g) * 2 - 1) * 1.5
    v = (torch.rand(B, H, T, hd, device=dev, generator=g) * 2 - 1)
    num_ims = torch.tensor([n - 1 for n in lens], device=dev, dtype=torch.int64)
    ws = torch.empty((int(_lib.load().paths_attention_x6_workspace(B, T, H, hd, 2)),), device=dev, dtype=torch.uint8)
    o = torch.full((B, T, H * hd), float("nan"), device=dev)
    _lib.call("paths_attention_x6", p(q), p(k), p(v), p(o), None, p(num_ims), B, T, H, hd, 0, p(ws), 2, 0, st)
    worst, rms = 0.0, 0.0
    for b, n in enumerate(lens):
        s = (q[b, :, :n].double() @ k[b, :, :n].double().transpose(1, 2)) * np.log(2.0)
        pr = torch.softmax(s, dim=-1)
        ref = (pr @ v[b, :, :n].double()).permute(1, 0, 2).reshape(n, H * hd)
        e = (o[b, :n].double() - ref).abs()
        worst = max(worst, e.max().item()); rms = max(rms, e.pow(2).mean().sqrt().item())
        neff = (1.0 / pr.pow(2).sum(-1)).median().item()
    print(f"{os.path.basename(os.environ.get('PATHS_HIP_LIB', 'default')):18s} {tag:12s} T={T}: max|o - fp64| = {worst:.3e}  rms {rms:.3e}  (median effective keys {neff:.1f})", flush=True)
