import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paths_amd import _lib, ops
lib = _lib.load()
torch.zeros(1, device="cuda")
for B, T in ((2, 257), (4, 2049), (8, 2049), (2, 65), (24, 2049), (25, 2049)):
    print(B, T, "supported 192:", lib.paths_token0_ws_supported(B, T, 192, 4), "128:", lib.paths_token0_ws_supported(B, T, 128, 4), "partials", lib.paths_token0_ws_partials_d(B, T, 192))
print(ops.TAIL_WS, ops.TAIL_WS_192, ops.WS_CHAIN_192, ops.GENERIC_SPLIT, ops.GEMM_MODE)
