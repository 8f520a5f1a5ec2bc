#!/usr/bin/env python3
"""Time paths_attention_x6 (images written by the prep launch) at the bench shape.  PATHS_HIP_LIB selects the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from paths_amd import _lib
dev = torch.device("cuda:0")
B, H, T, hd, planes = int(os.environ.get('ATTN_B', '8')), 4, int(os.environ.get('ATTN_T', '2049')), 32, 2
g = torch.Generator(device=dev); g.manual_seed(0)
q, k, v = ((torch.rand(B, H, T, hd, device=dev, generator=g) * 2 - 1) for _ in range(3))
num_ims = torch.tensor(([1844, 1850, 1839, 1861, 1822, 1847, 1855, 1830] * 4)[:B], device=dev).clamp(max=T - 1)
ws = torch.empty(int(_lib.load().paths_attention_x6_workspace(B, T, H, hd, planes)), device=dev, dtype=torch.uint8)
o = torch.zeros(B, T, H * hd, device=dev)
valid = (torch.arange(T, device=dev)[None, :] <= num_ims[:, None])[:, :, None]        # (rows past a slide's length are never read)
p, st = _lib.ptr, _lib.stream()
run = lambda ready: _lib.call("paths_attention_x6", p(q), p(k), p(v), p(o), None, p(num_ims), B, T, H, hd, 0, p(ws), planes, ready, st)
run(0); torch.cuda.synchronize()
for ready in (1, 1, 0):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run(ready)
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('PATHS_HIP_LIB', 'default')}: images_ready={ready}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us  checksum {(o.double() * valid).sum().item():.6f}")
