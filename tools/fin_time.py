#!/usr/bin/env python3
"""Time the importance / projection GEMM and its finishes at the bench shape (8 slides x 2048 rows, D = 1024): the round-4 pair
(paths_importance_proj_x6 split-K = GEMM + finish, then paths_token_layer_ws in_proj) against paths_importance_qkv_x6's phases
(1 GEMM, 2 importance-only finish, 4 tokens + in_proj finish).  With a PATHS_WS_STAMPS build (PATHS_HIP_LIB) the in-kernel
phase stamps of the fused finish are printed (median over workgroups)."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from paths_amd import _lib, ops
dev = torch.device("cuda:0")
B, N, D, d, H, hd = 8, 2048, 1024, 128, 4, 32
T = N + 1
M = B * N
g = torch.Generator(device=dev); g.manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
x, h1 = rnd(M, D) * 1.7, rnd(M, D + 256) * 0.5
lvl = {"w_ip_fwd": rnd(256, D) / 32, "_owner": None}
wip, wip_s = ops.x6_pack(lvl["w_ip_fwd"], planes=2)
b1, w2, b2, bp, sp = rnd(128) * 0.1, rnd(128) * 0.1, rnd(1) * 0.1, rnd(128) * 0.1, rnd(128)
layer = {"wo": rnd(d, d) / 11, "w1": rnd(512, d) / 11, "w2": rnd(d, 512) / 22, "wqkv": rnd(384, d) / 11}
bqkv = rnd(384) * 0.1
iq, sq = ops.tlayer_ws_images(layer, 1)
div = torch.exp(torch.arange(0, d // 2, 2) * (-math.log(10000.0) / d)).float().to(dev)
pe_tab = torch.empty((1024, d // 2), device=dev)
_lib.call("paths_pe_table", div.data_ptr(), 2, d, 1024, pe_tab.data_ptr(), _lib.stream())
locs = (torch.randint(0, 1000, (B, N, 2), device=dev, generator=g) * 256).to(torch.int64)
num_ims = torch.tensor([1844, 1850, 1839, 1861, 1822, 1847, 1855, 1830], device=dev)
imp = torch.zeros(B, N, device=dev)
tokens = torch.empty(B, T, d, device=dev)
ws = torch.empty(int(_lib.load().paths_importance_proj_x6_workspace(M)), device=dev, dtype=torch.uint8)
qkv = torch.empty(int(_lib.load().paths_attention_x6_workspace(B, T, H, hd, 2)), device=dev, dtype=torch.uint8)
KEEP = 512
keep_idx = torch.empty(B, KEEP, device=dev, dtype=torch.int32)
keep_count = torch.empty(B, device=dev, dtype=torch.int32)
kept_rows = torch.empty(B, KEEP, device=dev, dtype=torch.int64)
zero_row = torch.zeros(D, device=dev)
counters = torch.zeros(2 * B, device=dev, dtype=torch.int32)
p, st = _lib.ptr, _lib.stream()
qs = math.log2(math.e) / math.sqrt(hd)
lib = _lib.load()
stamps = None
if hasattr(lib, "paths_ws_stamp_buffer"):
    stamps = torch.zeros((33 * B, 16), device=dev, dtype=torch.int64)
    lib.paths_ws_stamp_buffer.argtypes = [ctypes.c_void_p]
    lib.paths_ws_stamp_buffer(stamps.data_ptr())


def fused(phases, afi=0):
    _lib.call("paths_importance_qkv_x6", p(x), D, None, p(h1), D + 256, p(wip), p(b1), p(w2), p(b2), p(bp), p(sp), p(pe_tab), 1024, p(locs),
              p(num_ims), B, N, 256, 2, 1, p(imp), p(tokens), D, 1, wip_s, 16.0, p(ws), p(iq), p(bqkv), sq[0], qs, p(qkv), phases, afi,
              *((KEEP, p(keep_idx), KEEP, p(keep_count), p(h1), D + 256, p(kept_rows), p(zero_row), p(counters), None) if phases & 8 else
                (0, None, 0, None, None, 0, None, None, None, None)), st)


def old_pair(which):
    if which & 1:
        _lib.call("paths_importance_proj_x6", p(x), D, None, p(h1), D + 256, p(wip), p(b1), p(w2), p(b2), p(bp), p(sp), p(div), p(pe_tab), 1024, p(locs),
                  p(num_ims), N, 256, 2, 1, p(imp), p(tokens), None, None, M, D, 128, d, 1, 2, wip_s, 16.0, p(ws), st)
    if which & 2:
        _lib.call("paths_token_layer_ws", p(tokens), None, None, None, None, p(iq), None, None, None, None, None, None, None, None, None, None,
                  p(bqkv), 1.0, 1.0, 1.0, sq[0], p(qkv), p(num_ims), B, T, d, H, 0, 1, 1, qs, 1e-5, None, 0, st)


def timeit(name, fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('PATHS_HIP_LIB', 'default')}: {name:46s} {e0.elapsed_time(e1) * 1e3 / n:7.1f} us", flush=True)


timeit("round 4: GEMM + finish", lambda: old_pair(1))
timeit("round 4: in_proj (token_layer_ws qkv)", lambda: old_pair(2))
timeit("round 4: GEMM + finish + in_proj", lambda: old_pair(3))
timeit("fused: GEMM only (phase 1)", lambda: fused(1))
timeit("fused: importance finish (phase 2)", lambda: fused(2))
timeit("fused: tokens + in_proj finish (phase 4)", lambda: fused(4))
timeit("fused: tokens + in_proj finish, alpha read (4)", lambda: fused(4, 1))
timeit("fused: GEMM + fused finish (5)", lambda: fused(5))
timeit("fused: GEMM + importance finish (3)", lambda: fused(3))
timeit("fused: importance + top-K finish (8)", lambda: fused(8))
timeit("fused: GEMM + importance + top-K finish (9)", lambda: fused(9))


def old_topk():
    _lib.call("paths_topk_rows", p(imp), N, p(num_ims), B, N, KEEP, p(keep_idx), KEEP, p(keep_count), p(h1), D + 256, N, p(kept_rows), p(zero_row), st)


timeit("round 4: paths_topk_rows", old_topk)
timeit("importance finish (2) + paths_topk_rows", lambda: (fused(2), old_topk()))
# the fused selection against the separate one
fused(2); old_topk(); torch.cuda.synchronize()
ref = (keep_idx.clone(), keep_count.clone(), kept_rows.clone(), imp.clone())
keep_idx.fill_(-7); keep_count.fill_(-7); kept_rows.fill_(-7); imp.zero_()
fused(8); torch.cuda.synchronize()
print("fused top-K == separate top-K:", bool(torch.equal(keep_idx, ref[0]) and torch.equal(keep_count, ref[1]) and torch.equal(kept_rows, ref[2]) and torch.equal(imp, ref[3])),
      "counters left zero:", bool((counters == 0).all()))
if stamps is not None:
    stamps.zero_(); fused(4); torch.cuda.synchronize()
    s = stamps.cpu()
    act = s[:, 0] > 0
    s = s[act]
    rel = (s - s[:, :1]).float()
    names = {0: "start", 2: "slabs in", 3: "logits", 4: "tokens st", 1: "image", 11: "q mm", 12: "q st", 13: "k mm", 14: "k st", 15: "end (v)"}
    med = rel.median(dim=0).values
    print("   workgroups", int(act.sum()), "start spread (cycles)", int(s[:, 0].max() - s[:, 0].min()), "end spread", int(s[:, 15].max() - s[:, 15].min()))
    prev = 0.0
    for i in (0, 2, 3, 4, 1, 11, 12, 13, 14, 15):
        if s[:, i].max() > 0:
            print(f"   {names[i]:10s} at {med[i]:9.0f}  (+{med[i] - prev:7.0f})")
            prev = float(med[i])
