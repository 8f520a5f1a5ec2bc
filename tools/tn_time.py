"""Time the weight-gradient GEMM (split-bf16 kernel vs f32-MFMA kernel) on the training step's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paths_amd import backward as bw  # noqa: E402

SHAPES = [(16384, 1792, 2048, 1024), (16384, 1792, 1024, 0), (16384, 1024, 256, 0), (16384, 256, 1024, 0), (16392, 128, 512, 0),
          (16392, 512, 128, 0), (16392, 384, 128, 0), (16392, 128, 128, 0)]


def main():
    dev = torch.device("cuda:0")
    for M, N1, N2, nb0 in SHAPES:
        a = torch.randn(M, N1, device=dev)
        b = torch.randn(M, N2, device=dev)
        out = torch.empty(N1, N2, device=dev)
        kw = dict(b1=b[:, nb0:].data_ptr(), ldb1=N2, nb0=nb0) if nb0 else {}
        res = {}
        for mode in ("f32", "x6"):
            bw.TN_MODE = mode
            for _ in range(3):
                bw.gemm_tn(a, N1, b, N2, out, M, N1, N2, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                bw.gemm_tn(a, N1, b, N2, out, M, N1, N2, **kw)
            e1.record()
            torch.cuda.synchronize()
            res[mode] = e0.elapsed_time(e1) / 20 * 1e3
        gf = 2.0 * M * N1 * N2 / 1e9
        print(f"M={M} N1={N1} N2={N2}: f32 {res['f32']:.1f} us ({gf / res['f32'] * 1e3:.0f} TF)   x6 {res['x6']:.1f} us ({gf / res['x6'] * 1e3:.0f} TF)"
              f"   splits {bw._splits_x6(M, N1, N2, nb0)}", flush=True)


if __name__ == "__main__":
    main()
