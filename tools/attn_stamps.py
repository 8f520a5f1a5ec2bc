#!/usr/bin/env python3
"""Phase shares of a key step of attn_x6_kernel from a -DPATHS_ATTN_STAMPS=n build (PATHS_HIP_LIB selects it): cycles per step
and wave (wave 0 of every workgroup, median over workgroups).  Diagnostic only: shares, never run time."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from paths_amd import _lib
dev = torch.device("cuda:0")
B, H, T, hd, planes = int(os.environ.get("ATTN_B", "8")), 4, 2049, 32, 2
g = torch.Generator(device=dev); g.manual_seed(0)
q, k, v = ((torch.rand(B, H, T, hd, device=dev, generator=g) * 2 - 1) for _ in range(3))
num_ims = torch.tensor(([1844, 1850, 1839, 1861, 1822, 1847, 1855, 1830] * 4)[:B], device=dev)
ws = torch.empty(int(_lib.load().paths_attention_x6_workspace(B, T, H, hd, planes)), device=dev, dtype=torch.uint8)
o = torch.empty(B, T, H * hd, device=dev)
p, st = _lib.ptr, _lib.stream()
lib = _lib.load()
lib.paths_attn_debug_buffer.argtypes = [C.c_void_p]; lib.paths_attn_debug_buffer.restype = None
run = lambda ready: _lib.call("paths_attention_x6", p(q), p(k), p(v), p(o), None, p(num_ims), B, T, H, hd, 0, p(ws), planes, ready, st)
lib.paths_attn_debug_buffer(None)
for _ in range(5): run(0)
torch.cuda.synchronize()
nb = B * H * 17 + 64
dbg = torch.zeros(nb * 16, device=dev, dtype=torch.int64)
lib.paths_attn_debug_buffer(dbg.data_ptr())
for _ in range(20): run(1)
torch.cuda.synchronize()
lib.paths_attn_debug_buffer(None)
d = dbg.view(-1, 16).cpu()
d = d[d[:, 1] > 0]
life_us = (d[:, 1] - d[:, 0]).double() / 100.0
steps = d[:, 11].double()
names = ["barrier->step start", "loads + QK^T", "exp2 + sum + split", "revision path", "PV", "LDS writes", "barrier wait"]
print(f"{os.environ.get('PATHS_HIP_LIB', 'default')}: workgroups {len(d)}, life p50 {life_us.median():.1f} us, steps p50 {steps.median():.0f}")
tot = 0.0
for i, n in enumerate(names):
    per = (d[:, 3 + i].double() / steps).median().item()
    tot += per
    print(f"   {n:22s} {per:8.0f} cycles per step")
print(f"   {'sum':22s} {tot:8.0f}   (life / steps: {(life_us * 1e-6 / steps).median().item() * 1e9:.0f} ns per step)")
