#!/usr/bin/env python3
"""Segment shares of the wave-pair attention kernel from a -DPATHS_M32P_STAMPS=1 build (PATHS_HIP_LIB selects it): cycles per key
step and wave, median over the active waves of each group.  Diagnostic only: shares, never run time."""
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
lib = _lib.load()
ws = torch.empty(int(lib.paths_attention_x6_workspace(B, T, H, hd, planes)), device=dev, dtype=torch.uint8)
o = torch.empty(B, T, H * hd, device=dev)
p, st = _lib.ptr, _lib.stream()
lib.paths_attn_pair_debug_buffer.argtypes = [C.c_void_p]; lib.paths_attn_pair_debug_buffer.restype = None
run = lambda ready: _lib.call("paths_attention_x6", p(q), p(k), p(v), p(o), None, p(num_ims), B, T, H, hd, 0, p(ws), planes, ready, st)
for _ in range(5): run(0)
torch.cuda.synchronize()
nwg = 8 * ((B * H + 7) // 8) * ((T + 255) // 256)
dbg = torch.zeros(nwg * 8 * 8, device=dev, dtype=torch.int64)
lib.paths_attn_pair_debug_buffer(dbg.data_ptr())
for _ in range(5): run(1)
torch.cuda.synchronize()
lib.paths_attn_pair_debug_buffer(None)
d = dbg.view(nwg, 8, 8).cpu().double()
names = ["prologue (once)", "V segment", "barrier after V", "M segment (after DMA issue)", "DMA issue at the head of M"]
for grp in (0, 1):
    w = d[:, 4 * grp:4 * grp + 4].reshape(-1, 8)
    w = w[w[:, 7] > 0]
    print(f"   staging {(w[:, 5] / w[:, 6]).median():.0f}, exp2 + sums + split {(w[:, 7] / w[:, 6]).median():.0f}, rest of V (below) = check / revision")
    steps = w[:, 6]
    print(f"waves {4 * grp}-{4 * grp + 3}: {len(w)} active, steps p50 {steps.median():.0f}")
    for i, n in enumerate(names):
        per = (w[:, i] / (steps if i else 1)).median().item()
        print(f"   {n:18s} {per:8.0f} cycles" + (" per step" if i else ""))
