#!/usr/bin/env python3
"""In-kernel timing of the split-bf16 GEMM launches (development aid).

Builds a -DPATHS_X6_DEBUG copy of the library (tools/_bin/libpaths_hip_dbg.so, hipcc needed) unless it exists, then for each
launch of the bench shapes prints wall time, and per wave (median over waves): init / main-loop / epilogue shader cycles, cycles
per k16 stage against the MFMA floor, and the in-kernel clock.
usage: x6_stages.py [M_children] [M_parents]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "tools", "_bin", "libpaths_hip_dbg.so")
if not os.path.exists(DBG) or "--rebuild" in sys.argv:
    os.makedirs(os.path.dirname(DBG), exist_ok=True)
    srcs = [os.path.join(ROOT, "paths_amd", "csrc", f) for f in sorted(os.listdir(os.path.join(ROOT, "paths_amd", "csrc"))) if f.endswith(".hip")]
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-DPATHS_X6_DEBUG"] + [a for a in sys.argv if a.startswith("-D")] + ["-o", DBG] + srcs)
    if "--build-only" in sys.argv:
        sys.exit(0)
os.environ["PATHS_HIP_LIB"] = DBG
sys.path.insert(0, ROOT)
import torch
from paths_amd import _lib, ops
args = [a for a in sys.argv[1:] if not a.startswith("-")]
M = int(args[0]) if len(args) > 0 else 14746
MP = int(args[1]) if len(args) > 1 else 4096
D, Hc, G = 1024, 256, 1792
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
x, c0 = rnd(M, D), rnd(M, Hc)
wg, bg, wm, bm = rnd(G, 2 * D) / 45, rnd(G), rnd(D, Hc) / 16, rnd(D)
(wg6, wgs), (wm6, wms) = ops.x6_pack(wg), ops.x6_pack(wm)
PL, AS = ops.split_planes(), ops.a_scale()
wip, wip6 = rnd(256, D) / 32, None
wip6, wips = ops.x6_pack(wip)
hk = rnd(MP, D)
hp = torch.empty(MP, G, device=dev)
hp_row = (torch.arange(M, device=dev, dtype=torch.int32) // 4) % MP
so, y, ws = torch.empty(M, D + Hc, device=dev), torch.empty(M, D, device=dev), torch.empty((M + 255) // 256 * 256, D, device=dev)
B = 8; N = (M + B - 1) // B
Mi = B * N
yi = rnd(Mi, D)
num_ims = torch.full((B,), N, device=dev, dtype=torch.int64)
locs = torch.randint(0, 1024, (Mi, 2), device=dev, dtype=torch.int64, generator=g) * 256
petab = ops.pe_table({"div_2d": (torch.rand(32, device=dev, generator=g))}, 2, 128, 1024)
imp, tok = torch.empty(Mi, device=dev), torch.empty(B, N + 1, 128, device=dev)
b1, w2, bp, sp, div = rnd(128), rnd(128), rnd(128), rnd(128), rnd(32)
b2s = torch.full((1,), 0.1, device=dev)
p, st = _lib.ptr, _lib.stream
lib = _lib.load()
lib.paths_x6_debug_buffer.argtypes = [C.c_void_p]; lib.paths_x6_debug_buffer.restype = None

def lstm(ph):
    _lib.call("paths_lstm_cell_x6", p(x), D, None, None, 0, p(c0), Hc, p(wg6), p(bg), p(wm6), p(bm), p(so), D + Hc, None, D, p(ws), None, None,
              p(hp), p(hp_row), M, D, Hc, None, 1, ph, PL, wgs, wms, AS, st())
def parent():
    _lib.call("paths_gemm_nt_x6", p(hk), D, p(wg6), 2 * D, D, None, p(hp), G, MP, G, G, D, 0, None, 0, None, 0, 0, PL, wgs, AS, st())
def impproj():
    _lib.call("paths_importance_proj_x6", p(yi), D, None, p(yadd) if USE_ADD else None, D, p(wip6), p(b1), p(w2), p(b2s), p(bp), p(sp), p(div), p(petab) if USE_TAB else None, petab.shape[0] if USE_TAB else 0, p(locs), p(num_ims), N, 256, 2, 1,
              p(imp), p(tok), None, None, Mi, D, 128, 128, 1, PL, wips, AS, None, st())

USE_TAB = True
USE_ADD = True
yadd = rnd(Mi, D)
NPR = 6 if PL == 3 else 3
CASES = [("parent partials  <2,4> K=1024", parent, (MP + 127) // 128 * 7, 64, 2 * 4 * NPR * 32, 2.0 * MP * G * D),
         ("gate c-part      <4,3> K=1024", lambda: lstm(1), (M + 255) // 256 * 4, 64, 4 * 3 * NPR * 32, 2.0 * M * 768 * D),
         ("gate o-part      <4,4> K=1024", lambda: lstm(2), (M + 255) // 256 * 4, 64, 4 * 4 * NPR * 32, 2.0 * M * D * D),
         ("mem_to_out       <4,4> K=256 ", lambda: lstm(4), (M + 255) // 256 * 4, 16, 4 * 4 * NPR * 32, 2.0 * M * D * Hc),
         ("importance+proj  <2,4> K=1024", impproj, (Mi + 127) // 128, 64, 2 * 4 * NPR * 32, 2.0 * Mi * 256 * D)]
for name, fn, blocks, stages, floor, flop in CASES:
    lib.paths_x6_debug_buffer(None)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    dbg = torch.zeros(blocks * 4 * 9, device=dev, dtype=torch.int64)
    lib.paths_x6_debug_buffer(dbg.data_ptr())
    fn(); torch.cuda.synchronize()
    lib.paths_x6_debug_buffer(None)
    d = dbg.view(-1, 9).double()
    d = d[d[:, 3] > 0]
    t_first, t_last = d[:, 4].min().item(), d[:, 5].max().item()
    starts = (d[:, 4] - t_first) / 100.0        # us
    life = d[:, 3] / 100.0
    med = d.median(dim=0).values
    tot = d[:, :3].sum(dim=1)
    ghz = (tot / (d[:, 3] * 10.0)).median().item()
    print(f"{name}: {us:7.1f} us {flop / us / 1e6:6.1f} TF | blocks {blocks:4d} | init {med[0]:8.0f} loop {med[1]:8.0f} epi {med[2]:8.0f} cyc | "
          f"{med[1] / stages:6.0f} cyc/stage (MFMA floor {floor}) | {ghz:.2f} GHz\n"
          f"      in-kernel span {(t_last - t_first) / 100.0:6.1f} us; wave start p50 {starts.median().item():5.1f} max {starts.max().item():5.1f} us; "
          f"wave life p50 {life.median().item():6.1f} max {life.max().item():6.1f} us\n"
          f"      stage 10: pre-barrier {d[:, 6].median().item():6.0f}  barrier wait {d[:, 7].median().item():6.0f}  post-barrier {d[:, 8].median().item():6.0f} cycles", flush=True)
