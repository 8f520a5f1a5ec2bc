#!/usr/bin/env python3
"""Time paths_token_layer_ws at the bench shape (qkv only / post only / post + qkv); with a PATHS_WS_STAMPS build also print the
in-kernel phase stamps (median over workgroups).  PATHS_HIP_LIB selects the library."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from paths_amd import _lib, ops
dev = torch.device("cuda:0")
B, H, T, d, hd = int(os.environ.get("WS_B", "8")), 4, 2049, 128, 32
g = torch.Generator(device=dev); g.manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
layer = {"wo": rnd(d, d) / 11, "w1": rnd(512, d) / 11, "w2": rnd(d, 512) / 22, "wqkv": rnd(384, d) / 11}
vec = {k: rnd(n) * 0.1 for k, n in (("bo", d), ("ln1b", d), ("cab", d), ("ln2b", d), ("b1", 512), ("b2", d), ("ln3b", d), ("bqkv", 384))}
gam = {k: 1 + rnd(d) * 0.1 for k in ("ln1g", "ln2g", "ln3g")}
ip, sp = ops.tlayer_ws_images(layer, 0)
iq, sq = ops.tlayer_ws_images(layer, 1)
Tp = (T + 63) // 64 * 64
x, xo = rnd(B, T, d), torch.empty(B, T, d, device=dev)
aimg = (torch.randn(B * Tp * d * 2, device=dev, generator=g) * 0.5).half().view(torch.uint8)
ws = torch.empty(int(_lib.load().paths_attention_x6_workspace(B, T, H, hd, 2)), device=dev, dtype=torch.uint8)
num_ims = torch.tensor(([1844, 1850, 1839, 1861, 1822, 1847, 1855, 1830] * 8)[:B], device=dev)
p, st = _lib.ptr, _lib.stream()
lib = _lib.load()
stamps = None
if hasattr(lib, "paths_ws_stamp_buffer"):
    stamps = torch.zeros(((T + 63) // 64 * B, 16), device=dev, dtype=torch.int64)
    lib.paths_ws_stamp_buffer.argtypes = [ctypes.c_void_p]
    lib.paths_ws_stamp_buffer(stamps.data_ptr())
def run(post, qkv):
    _lib.call("paths_token_layer_ws", p(x), None, p(aimg) if post else None, p(xo) if post else None, p(ip) if post else None, p(iq) if qkv else None,
              p(vec["bo"]), p(gam["ln1g"]), p(vec["ln1b"]), p(vec["cab"]), p(gam["ln2g"]), p(vec["ln2b"]), p(vec["b1"]), p(vec["b2"]),
              p(gam["ln3g"]), p(vec["ln3b"]), p(vec["bqkv"]), sp[0], sp[1], sp[2], sq[0], p(ws) if qkv else None, p(num_ims), B, T, d, H,
              post, qkv, 1, math.log2(math.e) / math.sqrt(hd), 1e-5, None, 0, st)
for post, qkv in ((0, 1), (1, 0), (1, 1)):
    run(post, qkv); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run(post, qkv)
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('PATHS_HIP_LIB', 'default')}: post={post} qkv={qkv}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us", flush=True)
    if stamps is not None:
        stamps.zero_(); run(post, qkv); torch.cuda.synchronize()
        s = stamps.cpu()
        act = s[:, 0] > 0
        s = s[act]
        rel = (s - s[:, :1]).float()
        names = ["start", "prologue", "out_proj", "ln1+ln2", "put x1", "ffn0", "ffn1", "ffn2", "ffn3", "ln3", "x_out/put", "q mm", "q st", "k mm", "k st", "end"]
        med = rel.median(dim=0).values
        print("   workgroups", int(act.sum()), "start spread (cycles)", int(s[:, 0].max() - s[:, 0].min()), "end spread", int(s[:, 15].max() - s[:, 15].min()))
        prev = 0.0
        for i, n in enumerate(names):
            if s[:, i].max() > 0:
                print(f"   {n:10s} at {med[i]:9.0f}  (+{med[i] - prev:7.0f})")
                prev = float(med[i])

# ---- kernel-level comparison against paths_token_layer_h3 on the same random layer
if "--check" in sys.argv:
    attn = rnd(B, T, d)
    iph, sph = ops.tlayer_h3_images(layer, 0)
    iqh, sqh = ops.tlayer_h3_images(layer, 1)
    xo_old, xo_new = torch.zeros(B, T, d, device=dev), torch.zeros(B, T, d, device=dev)
    q, k, v = (torch.empty(B, H, T, hd, device=dev) for _ in range(3))
    ws_old = torch.zeros_like(ws); ws_new = torch.zeros_like(ws)
    qs = math.log2(math.e) / math.sqrt(hd)
    _lib.call("paths_token_layer_h3", p(x), p(attn), p(xo_old), p(iph), p(iqh),
              p(vec["bo"]), p(gam["ln1g"]), p(vec["ln1b"]), p(vec["cab"]), p(gam["ln2g"]), p(vec["ln2b"]), p(vec["b1"]), p(vec["b2"]),
              p(gam["ln3g"]), p(vec["ln3b"]), p(vec["bqkv"]), sph[0], sph[1], sph[2], sqh[0], p(q), p(k), p(v), p(num_ims), B, T, d, H,
              1, 1, 1, qs, 1e-5, 0, p(ws_old), st)
    for post, qkv, tag in ((1, 1, "post+qkv"), (1, 0, "post"), (0, 1, "qkv of x")):
        xo_new.zero_(); ws_new.zero_()
        _lib.call("paths_token_layer_ws", p(x), p(attn) if post else None, None, p(xo_new) if post else None, p(ip) if post else None, p(iq) if qkv else None,
                  p(vec["bo"]), p(gam["ln1g"]), p(vec["ln1b"]), p(vec["cab"]), p(gam["ln2g"]), p(vec["ln2b"]), p(vec["b1"]), p(vec["b2"]),
                  p(gam["ln3g"]), p(vec["ln3b"]), p(vec["bqkv"]), sp[0], sp[1], sp[2], sq[0], p(ws_new) if qkv else None, p(num_ims), B, T, d, H,
                  post, qkv, 1, qs, 1e-5, None, 0, st)
        torch.cuda.synchronize()
        if post:
            err = max(float((xo_old[b, :int(num_ims[b]) + 1] - xo_new[b, :int(num_ims[b]) + 1]).abs().max()) for b in range(B))
            print(f"   {tag}: x_out max|ws - h3| = {err:.3e}")
        if qkv and post:
            o_old, o_new = torch.zeros(B, T, d, device=dev), torch.zeros(B, T, d, device=dev)
            _lib.call("paths_attention_x6", None, None, None, p(o_old), None, p(num_ims), B, T, H, hd, 0, p(ws_old), 2, 1, st)
            _lib.call("paths_attention_x6", None, None, None, p(o_new), None, p(num_ims), B, T, H, hd, 0, p(ws_new), 2, 1, st)
            torch.cuda.synchronize()
            err = max(float((o_old[b, :int(num_ims[b]) + 1] - o_new[b, :int(num_ims[b]) + 1]).abs().max()) for b in range(B))
            print(f"   {tag}: attention(images) max|ws - h3| = {err:.3e}")
