#!/usr/bin/env python3
"""Kernel-by-kernel sanity check on a real GPU (development aid; the judged tests are tests/ -m gpu)."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F
from paths_amd import _lib, synthetic as syn

dev = torch.device("cuda:0")
p = _lib.ptr
st = _lib.stream
torch.manual_seed(0)

def report(name, got, ref, tol):
    err = float((got.double().cpu() - ref.double()).abs().max())
    print(f"{name:28s} max|diff| = {err:.3e}  {'OK' if err <= tol else 'FAIL'}", flush=True)
    return err <= tol

ok = True
# --- linear
M, N, K = 300, 200, 256
a = torch.randn(M, K); w = torch.randn(N, K) / math.sqrt(K); b = torch.randn(N)
wp = torch.zeros(256, K); wp[:N] = w
out = torch.empty(M, N, device=dev)
a_d, wp_d, b_d = a.to(dev), wp.to(dev), b.to(dev)
_lib.call("paths_linear_f32", p(a_d), K, p(wp_d), p(b_d), p(out), N, M, N, 256, K, 1, st())
torch.cuda.synchronize()
ok &= report("linear_f32(relu)", out, torch.relu(F.linear(a, w, b)), 5e-6)

# --- synth grid vs numpy
g = torch.empty(6, 10, 1024, device=dev)
key = int(syn.slide_level_key(3, 2, 1))
_lib.call("paths_synth_grid", p(g), 6, 10, 1024, key, 1, syn.bg_threshold(0.3), st())
ref = syn.SyntheticSlide(3, 2, (3, 5), 1024, 2, 0.3).grid(1)
torch.cuda.synchronize()
print("synth_grid bit-exact:", bool((g.cpu().numpy() == ref).all()), "bg cells:", int((ref.sum(-1) == 0).sum()), flush=True)
ok &= bool((g.cpu().numpy() == ref).all())
m = torch.empty(60, dtype=torch.uint8, device=dev)
_lib.call("paths_tissue_mask", p(g), 60, 1024, p(m), st())
mk = bool((m.cpu().numpy() == (ref.reshape(60, -1).sum(-1) != 0)).all()); ok &= mk; print("tissue mask ok", mk, flush=True)

# --- layernorm
x = torch.randn(1000, 128); gm = torch.randn(128); bt = torch.randn(128)
y = torch.empty(1000, 128, device=dev)
x_d, gm_d, bt_d = x.to(dev), gm.to(dev), bt.to(dev)
_lib.call("paths_layernorm_f32", p(x_d), p(gm_d), p(bt_d), p(y), 1000, 128, 1e-5, st())
torch.cuda.synchronize()
ok &= report("layernorm", y, F.layer_norm(x, (128,), gm, bt, 1e-5), 5e-6)

# --- attention
B, H, T = 2, 4, 300
nim = torch.tensor([299, 150])
q = torch.randn(B, H, T, 32); k = torch.randn(B, H, T, 32); v = torch.randn(B, H, T, 32)
o = torch.zeros(B, T, 128, device=dev)
qs = q * (1.4426950408889634 / math.sqrt(32))
q_d, k_d, v_d, nim_d = qs.to(dev), k.to(dev), v.to(dev), nim.to(dev)
_lib.call("paths_attention_f32", p(q_d), p(k_d), p(v_d), p(o), None, p(nim_d), B, T, H, 32, 0, st())
torch.cuda.synchronize()
sc = (q @ k.transpose(-1, -2)) / math.sqrt(32)
mask = torch.arange(T)[None, :] >= (nim + 1)[:, None]
sc = sc.masked_fill(mask[:, None, None, :], float("-inf"))
ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B, T, 128)
for bb in range(B):
    n = int(nim[bb]) + 1
    ok &= report(f"attention[b={bb}]", o[bb, :n], ref[bb, :n], 5e-6)

# --- topk
sc = torch.rand(3, 700); sc[1, 5] = sc[1, 77]  # tie
nim = torch.tensor([700, 650, 10])
ki = torch.full((3, 64), -1, dtype=torch.int32, device=dev); kc = torch.zeros(3, dtype=torch.int32, device=dev)
sc_d, nim_d = sc.to(dev), nim.to(dev)
_lib.call("paths_topk", p(sc_d), 700, p(nim_d), 3, 700, 64, p(ki), 64, p(kc), st())
torch.cuda.synchronize()
for bb in range(3):
    n = int(nim[bb]); c = min(n, 64)
    order = np.lexsort((np.arange(n), -sc[bb, :n].numpy()))[:c]
    good = int(kc[bb]) == c and np.array_equal(ki[bb, :c].cpu().numpy(), order)
    print(f"topk[b={bb}] {'OK' if good else 'FAIL'}", flush=True); ok &= good
print("ALL OK" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
