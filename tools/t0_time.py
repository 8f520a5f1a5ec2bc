#!/usr/bin/env python3
"""Time paths_token0_tail_ws at the bench shape; with a PATHS_T0_STAMPS build print the in-kernel phase stamps."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from paths_amd import _lib, ops
dev = torch.device("cuda:0")
cfg, model, sd = bench.build_model(2048, dev)
lvl = ops.pack_level(model.procs[1])
w = lvl["layers"][1]
B, T, d, H = 8, 2049, 128, 4
g = torch.Generator(device=dev); g.manual_seed(0)
x1 = torch.randn(B, T, d, device=dev, generator=g)
num_ims = torch.tensor([1844, 1850, 1839, 1861, 1822, 1847, 1855, 1830], device=dev)
res = torch.randn(B, d, device=dev, generator=g)
p, st = _lib.ptr, _lib.stream()
lib = _lib.load()
qscale = ops.LOG2E / math.sqrt(32)
img = ops.token0_ws_image(w, qscale)
part = torch.empty(int(lib.paths_token0_ws_partials(B, T)), device=dev)
cnt = ops.token0_counters(dev, B)
ctx_out, logits = torch.empty(B, d, device=dev), torch.empty(B, 4, device=dev)
nblk = 4 * max(1, min(16, (T + 511) // 512)) * B
stamps = None
if hasattr(lib, "paths_t0_stamp_buffer"):
    stamps = torch.zeros((4 * 16 * B, 16), device=dev, dtype=torch.int64)
    lib.paths_t0_stamp_buffer.argtypes = [ctypes.c_void_p]
    lib.paths_t0_stamp_buffer(stamps.data_ptr())
def run():
    _lib.call("paths_token0_tail_ws", p(x1), p(num_ims), p(img), w["bqkv"].data_ptr() + 8 * d, p(w["bo"]), p(w["ln1g"]), p(w["ln1b"]),
              p(w["cab"]), p(w["ln2g"]), p(w["ln2b"]), p(w["b1"]), p(w["b2"]), p(w["ln3g"]), p(w["ln3b"]), p(lvl["lnfg"]), p(lvl["lnfb"]),
              p(res), res.stride(0), None, 0, p(lvl["wcls"]), p(lvl["bcls"]), 4, 128, p(ctx_out), p(logits), p(part), p(cnt), None, B, T, d, H,
              w["eps"], lvl["lnf_eps"], 0, st)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"{os.environ.get('PATHS_HIP_LIB', 'default')}: token0_tail_ws {e0.elapsed_time(e1) * 1e3 / 50:.1f} us  logits {logits[0].tolist()}", flush=True)
if stamps is not None:
    stamps.zero_(); run(); torch.cuda.synchronize()
    s = stamps.cpu(); s = s[s[:, 0] > 0]
    names = ["start", "phase0 qt", "phase1", "ticket1", "x ready", "ticket2", "-", "-", "-", "end"] if os.environ.get("PATHS_T0_DIST", "1") != "0" else ["start", "phase0 qt", "phase1", "publish", "merge", "o", "outproj+ln", "ffn1", "ffn2", "end"]
    last = s[s[:, 9] > 0]
    print("   workgroups", len(s), "last arrivers", len(last))
    for i, n in enumerate(names):
        rows = last if (i > 3 and os.environ.get("PATHS_T0_DIST", "1") == "0") or i == 9 else s
        if rows[:, i].max() == 0: continue
        rel = (rows[:, i] - rows[:, 0]).float()
        print(f"   {n:12s} median +{int(rel.median()):7d} cycles   max +{int(rel.max()):7d}")
