"""Two (or more) FULL 8-slide batches replayed concurrently on their own stream triples against the same batches replayed one after
the other: does inter-batch pipelining fill the selection chain's latency bubbles?  tools/lanes_time.py [lanes]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paths_amd import synthetic as syn, utils as putils
from paths_amd.config import Config
from paths_amd.data_utils.slide import DeviceSlide, DeviceSlideBatch
ROOT = sys.path[0]
dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K, spg = 2048, 8
cfg = Config.load(os.path.join(ROOT, "tests", "golden", "sample"), test_mode=True)
cfg.top_k_patches = [K // 4] * (cfg.num_levels - 1)
model = cfg.get_model()
sd = syn.make_state_dict(0, {k: tuple(v.shape) for k, v in model.state_dict().items()})
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to(dev).eval()
batches = [DeviceSlideBatch([DeviceSlide.synthetic(1234, 100000 * r + i, (32, 64), device=dev) for i in range(spg)]) for r in range(L)]
seq = [putils.TapedRecursion(model, b, cfg.top_k_patches, cfg.num_levels).record() for b in batches]
par = [putils.TapedRecursion(model, b, cfg.top_k_patches, cfg.num_levels, lane=r).record() for r, b in enumerate(batches)]
for a, b in zip(seq, par):
    assert torch.equal(a.out["logits"], b.out["logits"])


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def run_seq():
    for t in seq:
        t.replay()


def run_par():
    for t in par:
        t.replay(join=False)
    for t in par:
        t.join()


for rep in range(2):
    ts, tp = timed(run_seq), timed(run_par)
    print(f"{L} batches x {spg} slides: sequential {ts * 1e3:.3f} ms = {L * spg / ts:.0f} slides/s; concurrent lanes {tp * 1e3:.3f} ms = {L * spg / tp:.0f} slides/s", flush=True)
for a, b in zip(seq, par):
    assert torch.equal(a.out["logits"], b.out["logits"]) and torch.equal(a.out["importance"], b.out["importance"])
