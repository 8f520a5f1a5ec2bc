#!/usr/bin/env python3
"""Which multiply-adds of torch's foreach AdamW kernels are fused on this build?  Tries the 8 flavors of paths_adamw_multi against
torch.optim.AdamW(foreach=True) and prints the ones that reproduce it bit for bit (development aid for paths_amd/optim.py:FLAVOR)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paths_amd import optim as popt
dev = torch.device("cuda:0")
for flavor in range(8):
    popt.FLAVOR = flavor
    g = torch.Generator().manual_seed(1)
    base = [torch.randn(1000, 777, generator=g), torch.randn(4097, generator=g) * 1e-3]
    pa = [torch.nn.Parameter(b.clone().to(dev)) for b in base]; pb = [torch.nn.Parameter(b.clone().to(dev)) for b in base]
    oa, ob = torch.optim.AdamW(pa, lr=2e-3, weight_decay=0.01, foreach=True), popt.HipAdamW(pb, lr=2e-3, weight_decay=0.01)
    bad = 0
    for it in range(5):
        for a, b in zip(pa, pb):
            gr = (torch.randn(a.shape, generator=g) * 2.0 ** (it - 3)).to(dev)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
        bad += sum(int((a != b).sum()) + int((oa.state[a]["exp_avg"] != ob.state[b]["exp_avg"]).sum()) + int((oa.state[a]["exp_avg_sq"] != ob.state[b]["exp_avg_sq"]).sum()) for a, b in zip(pa, pb))
    print(f"flavor {flavor} (lerp fma {flavor & 1}, addcmul fma {(flavor >> 1) & 1}, addcdiv fma {(flavor >> 2) & 1}): {bad} differing elements")
