#!/usr/bin/env python3
"""Per-workgroup start / end times of attn_x6_kernel (debug build -DPATHS_ATTN_DEBUG; development aid)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from paths_amd import _lib
dev = torch.device("cuda:0")
B, H, T, hd, planes = 8, 4, 2049, 32, 2
g = torch.Generator(device=dev); g.manual_seed(0)
q, k, v = ((torch.rand(B, H, T, hd, device=dev, generator=g) * 2 - 1) for _ in range(3))
num_ims = torch.tensor([1844, 1850, 1839, 1861, 1822, 1847, 1855, 1830], device=dev)
ws = torch.empty(int(_lib.load().paths_attention_x6_workspace(B, T, H, hd, planes)), device=dev, dtype=torch.uint8)
o = torch.empty(B, T, H * hd, device=dev)
p, st = _lib.ptr, _lib.stream()
lib = _lib.load()
lib.paths_attn_debug_buffer.argtypes = [C.c_void_p]; lib.paths_attn_debug_buffer.restype = None
run = lambda ready: _lib.call("paths_attention_x6", p(q), p(k), p(v), p(o), None, p(num_ims), B, T, H, hd, 0, p(ws), planes, ready, st)
lib.paths_attn_debug_buffer(None)
for _ in range(5): run(0)
torch.cuda.synchronize()
nb = 8 * 4 * 17
dbg = torch.zeros(nb * 3, device=dev, dtype=torch.int64)
lib.paths_attn_debug_buffer(dbg.data_ptr())
run(1); torch.cuda.synchronize()
lib.paths_attn_debug_buffer(None)
d = dbg.view(-1, 3).cpu()
d = d[d[:, 1] > 0]
t0 = d[:, 0].min().item()
start, end = (d[:, 0] - t0).double() / 100.0, (d[:, 1] - t0).double() / 100.0
life = end - start
xcc = (d[:, 2] >> 32) & 0xF
hw = d[:, 2] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
cuid = xcc * 64 + se * 8 + sh * 16 + cu     # (an id that separates CUs; field layout is informative only)
print(f"workgroups with work: {len(d)}; kernel span {end.max().item():.1f} us")
print(f"start  p50 {start.median().item():.1f} p90 {start.quantile(0.9).item():.1f} max {start.max().item():.1f} us")
print(f"life   p10 {life.quantile(0.1).item():.1f} p50 {life.median().item():.1f} p90 {life.quantile(0.9).item():.1f} max {life.max().item():.1f} us")
import collections
per = collections.Counter(cuid.tolist())
print("distinct CU ids:", len(per), "workgroups per CU id histogram:", sorted(collections.Counter(per.values()).items()))
per_x = collections.Counter(xcc.tolist())
print("workgroups per XCC:", sorted(per_x.items()))
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  xcc {x}: life p50 {life[m].median().item():.1f} end max {end[m].max().item():.1f}")
# ---- which workgroups are slow?
import numpy as np
lin = torch.arange(dbg.numel() // 3)[(dbg.view(-1, 3).cpu()[:, 1] > 0)]
xg, jx = lin & 7, lin >> 3
pair = xg + 8 * (jx % 4); qb = jx // 4
life_n = life.numpy()
print("life by q-block:", [f"{qb_}:{life_n[(qb == qb_).numpy()].mean():.0f}" for qb_ in range(int(qb.max()) + 1)])
print("life by pair (first 8):", [f"{p_}:{life_n[(pair == p_).numpy()].mean():.0f}" for p_ in range(8)])
cu_np = cuid.numpy()
cnt = collections.Counter(cu_np.tolist())
slow = life_n > 55.0
print(f"slow (> 55 us): {slow.sum()} workgroups; their q-blocks: {sorted(collections.Counter(qb[torch.from_numpy(slow)].tolist()).items())}")
print("  their xcc:", sorted(collections.Counter(xcc[torch.from_numpy(slow)].tolist()).items()))
print("  workgroups on the slow ones' CU ids:", sorted(collections.Counter(cnt[c] for c in cu_np[slow]).items()))
fast = life_n < 40
print(f"fast (< 40 us): {fast.sum()}; on CU ids with n workgroups:", sorted(collections.Counter(cnt[c] for c in cu_np[fast]).items()))
print("hw_id fields sample:", [hex(int(x)) for x in hw[:6].tolist()])
