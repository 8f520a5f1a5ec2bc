#!/usr/bin/env python3
"""Time paths_token_layer_h3 at the bench shape (post + in_proj, and in_proj alone).  PATHS_HIP_LIB selects the library."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from paths_amd import _lib, ops
dev = torch.device("cuda:0")
B, H, T, d, hd = 8, 4, 2049, 128, 32
g = torch.Generator(device=dev); g.manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
layer = {"wo": rnd(d, d) / 11, "w1": rnd(512, d) / 11, "w2": rnd(d, 512) / 22, "wqkv": rnd(384, d) / 11}
vec = {k: rnd(n) * 0.1 for k, n in (("bo", d), ("ln1b", d), ("cab", d), ("ln2b", d), ("b1", 512), ("b2", d), ("ln3b", d), ("bqkv", 384))}
gam = {k: 1 + rnd(d) * 0.1 for k in ("ln1g", "ln2g", "ln3g")}
ip, sp = ops.tlayer_h3_images(layer, 0)
iq, sq = ops.tlayer_h3_images(layer, 1)
x, attn, xo = rnd(B, T, d), rnd(B, T, d), torch.empty(B, T, d, device=dev)
q, k, v = (torch.empty(B, H, T, hd, device=dev) for _ in range(3))
ws = torch.empty(int(_lib.load().paths_attention_x6_workspace(B, T, H, hd, 2)), device=dev, dtype=torch.uint8)
num_ims = torch.tensor([1844, 1850, 1839, 1861, 1822, 1847, 1855, 1830], device=dev)
p, st = _lib.ptr, _lib.stream()
def run(post, images):
    _lib.call("paths_token_layer_h3", p(x), p(attn) if post else None, p(xo) if post else None, p(ip) if post else None, p(iq),
              p(vec["bo"]), p(gam["ln1g"]), p(vec["ln1b"]), p(vec["cab"]), p(gam["ln2g"]), p(vec["ln2b"]), p(vec["b1"]), p(vec["b2"]),
              p(gam["ln3g"]), p(vec["ln3b"]), p(vec["bqkv"]), sp[0], sp[1], sp[2], sq[0], p(q), p(k), p(v), p(num_ims), B, T, d, H,
              1 if post else 0, 1, 1, math.log2(math.e) / math.sqrt(hd), 1e-5, 0, p(ws) if images else None, st)
for post, images in ((1, 0), (1, 0), (0, 1), (0, 0)):
    run(post, images); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run(post, images)
    e1.record(); torch.cuda.synchronize()
    cs = (xo[:, :1800].double().sum().item() if post else 0.0) + (0.0 if images else q[:, :, :1800].double().sum().item())
    print(f"{os.environ.get('PATHS_HIP_LIB', 'default')}: post={post} images={images}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us  checksum {cs:.6f}")
