#!/usr/bin/env python3
"""What a launch boundary costs between two given kernels of the recursion.  Records the one-stream launch tape of the bench batch
(8 slides, K = 2048), takes the C calls of level 1 and replays single calls, pairs and runs of them between two events:
boundary(a, b) = t(a; b) - t(a) - t(b) + t(empty bracket).  The LSTM cell is recorded as three calls (phases 1, 2, 4).
usage: pair_gap.py [reps]"""
import os, sys, statistics as st
os.environ["PATHS_OVERLAP_AGGREGATOR"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from paths_amd import _lib, ops
from paths_amd import utils as putils
from paths_amd.data_utils.slide import DeviceSlide, DeviceSlideBatch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
_lib.load()
cfg, model, sd = bench.build_model(2048, dev, None)
slides = DeviceSlideBatch([DeviceSlide.synthetic(1234, i, bench.BASE_SHAPES[2048], device=dev) for i in range(8)])
ops.KERNEL_TIMER, ops.TIMER_ALL = (lambda name, fn, meta: fn()), True      # (splits the LSTM cell call into its three phases)
tr = putils.TapedRecursion(model, slides, cfg.top_k_patches, cfg.num_levels).record()
ops.KERNEL_TIMER, ops.TIMER_ALL = None, False
names = [n for _, _, n in tr.tape]
# level 1 = the calls between the second and the third paths_lstm_cell_x6 group
idx = [i for i, n in enumerate(names) if n == "paths_lstm_cell_x6"]
lo, hi = idx[3], idx[6]
# Only a PREFIX of the tape is ever replayed: the recorded pass re-uses the blocks of dead intermediates at later levels, so after
# a whole replay the level-1 tables (row-pointer tables among them) hold later levels' data - a level-1 call replayed then reads
# garbage addresses.  After tape[:hi] every input of level 1 is valid and stays so while only level-1 calls run.
level = [c for c in tr.tape[lo:hi] if c[2] not in ("paths_expand_children", "paths_gather_rows")]
print("level-1 calls:", [n for _, _, n in level])
torch.cuda.synchronize()
tr._play(tr.tape[:hi]); torch.cuda.synchronize()


def t_of(calls):
    ts = []
    for _ in range(reps):
        torch.cuda.Event(enable_timing=True).record()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        tr._play(calls)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return st.median(ts)


empty = t_of([])
print(f"empty bracket {empty:.2f} us")
single = [t_of([c]) - empty for c in level]
for c, t in zip(level, single):
    print(f"  {c[2]:36s} alone {t:8.2f} us")
print("pairs (consecutive calls of the level): together - sum of the two alone = boundary")
for i in range(len(level) - 1):
    both = t_of(level[i:i + 2]) - empty
    print(f"  {level[i][2]:32s} -> {level[i + 1][2]:32s} together {both:8.2f}  boundary {both - single[i] - single[i + 1]:7.2f}")
whole = t_of(level) - empty
print(f"whole level {whole:.2f} us, sum of singles {sum(single):.2f}, boundaries {whole - sum(single):.2f}")
# same kernel twice: is the cost a property of the successor or of the pair?
for i, c in enumerate(level):
    two = t_of([c, c]) - empty
    print(f"  {c[2]:36s} twice {two:8.2f}  boundary {two - 2 * single[i]:7.2f}")
