"""npp_reachability launch time: with the keys of the previous call still valid (no recomputation: staging + copy only) and right after a
step (the envs whose cell / switch key changed recompute)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from nclone_amd.engine import NppBatch
from nclone_amd.levels import door_levels

n = 8192
levels, _ = door_levels()
b = NppBatch(n, autoreset=True, outputs=["reachability_features", "mine_sdf_features"])
b.load_levels(levels)
b.assign_levels((np.arange(n) // 64) % len(levels))
acts = torch.from_numpy(np.random.default_rng(0).integers(0, 6, size=(400, n)).astype(np.uint8)).cuda()
for t in range(300):
    b.step(acts[t])
    b.reachability()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record(b.stream)
for _ in range(50):
    b.reachability()
ev[1].record(b.stream)
torch.cuda.synchronize()
print("keys unchanged: %.1f us per launch" % (ev[0].elapsed_time(ev[1]) * 20))
tot = 0.0
for t in range(300, 400):
    b.step(acts[t])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(b.stream)
    b.reachability()
    e1.record(b.stream)
    torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
print("after a step: %.1f us per launch" % (tot * 10))
