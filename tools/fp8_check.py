"""Check / time the e4m3 GEMM (csrc/gemm_fp8.hip) against float64 on the SAME quantised operands, and the fp8 attention at head_dim 64."""
import sys, time, math
import torch
sys.path.insert(0, ".")
from paths_amd import _lib

dev = torch.device("cuda:0")
st = _lib.stream()
p = _lib.ptr


def q8(x, s):
    return (x * s).clamp(-448, 448).to(torch.float8_e4m3fn).double()


def run(M, N, K, act=0, res=True, reps=0):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * 1.7
    w = torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
    rd = r.to(dev) if res else None
    scratch = torch.zeros(1, dtype=torch.int32, device=dev)
    sa, sw = torch.empty(1, device=dev), torch.empty(1, device=dev)
    w8 = torch.empty(((N + 255) // 256 * 256, K), dtype=torch.uint8, device=dev)
    _lib.call("paths_fp8_pack_weight", p(wd), K, N, K, p(w8), p(sw), p(scratch), st)
    _lib.call("paths_fp8_scale", p(ad), K, M, K, p(sa), p(scratch), None, 0, st)
    out = torch.full((M, N), 7.0, device=dev)
    a8 = torch.empty(((M + 255) // 256 * 256, K), dtype=torch.uint8, device=dev)
    _lib.call("paths_fp8_quantize", p(ad), K, M, K, p(sa), p(a8), st)
    args = (p(a8), p(w8), p(sa), p(sw), p(bd), p(out), N, M, N, K, act, p(rd) if res else None, N if res else 0, st)
    _lib.call("paths_gemm_nt_fp8", *args)
    torch.cuda.synchronize()
    fsa, fsw = float(sa), float(sw)
    assert abs(fsa - 448 / float(a.abs().max())) < 1e-3 * fsa and abs(fsw - 448 / float(w.abs().max())) < 1e-3 * fsw, (fsa, fsw)
    ref = q8(a, fsa) @ q8(w, fsw).t() / (fsa * fsw) + bias.double()
    if act:
        ref = torch.relu(ref)
    if res:
        ref = ref + r.double()
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    exact = a.double() @ w.double().t() + bias.double()
    if act:
        exact = torch.relu(exact)
    if res:
        exact = exact + r.double()
    qerr = float((out.cpu().double() - exact).abs().max() / exact.abs().max())
    msg = f"M={M} N={N} K={K} act={act} res={res}: vs same-quantised fp64 {err:.2e}, vs exact {qerr:.2e}"
    if reps:
        for _ in range(3):
            _lib.call("paths_gemm_nt_fp8", *args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.Event(enable_timing=True).record()
        e0.record()
        for _ in range(reps):
            _lib.call("paths_gemm_nt_fp8", *args)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        msg += f"  {us:.1f} us = {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s"
    print(msg, flush=True)
    return err


ok = True
for shp in [(256, 256, 128), (1000, 640, 256), (300, 100, 128), (4097, 1536, 1536)]:
    ok &= run(*shp) < 1e-4
ok &= run(513, 384, 512, act=1, res=False) < 1e-4
print("layout", "OK" if ok else "MISMATCH", flush=True)
if ok and len(sys.argv) > 1:
    run(65544, 4608, 1536, res=False, reps=10)
    run(65544, 6144, 1536, act=1, res=False, reps=10)
    run(65544, 1536, 6144, reps=10)
    run(16392, 384, 128, reps=20)
    run(16392, 512, 128, act=1, res=False, reps=20)
