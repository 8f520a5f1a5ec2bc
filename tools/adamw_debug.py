import os, sys, copy
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from paths_amd import synthetic as syn, train as ptrain, utils as putils
from paths_amd.config import Config
from paths_amd.optim import HipAdamW
dev = torch.device("cuda:0")
cfg = Config.load(os.path.join("tests", "golden", "sample"), test_mode=True)
cfg.model_config.dropout = 0.0
cfg.num_levels, cfg.top_k_patches = 3, [8] * 2
model = cfg.get_model()
sd = syn.make_state_dict(3, {k: tuple(v.shape) for k, v in model.state_dict().items()})
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
ma = model.to(dev).train(); mb = copy.deepcopy(ma).train()
ds = ptrain.synthetic_dataset(6, (6, 6), 3, dev, seed=11)
oa = torch.optim.AdamW(ma.parameters(), lr=2e-3, weight_decay=cfg.weight_decay)
ob = HipAdamW(mb.parameters(), lr=2e-3, weight_decay=cfg.weight_decay)
from paths_amd.data_utils.slide import DeviceSlideBatch
for step in range(6):
    items = [ds[(2 * step) % 6], ds[(2 * step + 1) % 6]]
    batch = {"slide": DeviceSlideBatch([it["slide"] for it in items]), "survival_bin": torch.tensor([it["survival_bin"] for it in items]), "censored": torch.tensor([it["censored"] for it in items])}
    la = putils.train_step(ma, oa, batch, 3, cfg.top_k_patches)
    # gradient comparison needs grads before stepping: recompute on mb
    lb = putils.train_step(mb, ob, batch, 3, cfg.top_k_patches)
    bad = []
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        if not torch.equal(pa, pb):
            bad.append((n, float((pa - pb).abs().max()), pa.grad is None, pb.grad is None,
                        None if pa.grad is None or pb.grad is None else float((pa.grad - pb.grad).abs().max()),
                        None if pa.grad is None else (tuple(pa.grad.stride()), pa.grad.is_contiguous(), pa.grad.data_ptr() % 16)))
    print("step", step, float(la), float(lb), "differing params:", len(bad))
    for b in bad[:12]:
        print("   ", b)
    if bad:
        break
