import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import paths_amd.ops as O
import test_gpu_parity as T
dev = torch.device("cuda:0")
for name in sys.argv[1:]:
    calls = []
    orig = O._lib.call
    O._lib.call = lambda cname, *a: (calls.append(cname), orig(cname, *a))[1]
    try:
        g, info, out = T.run_single(dev, name)
    finally:
        O._lib.call = orig
    print(name, info.get("cfg_over"), calls)
