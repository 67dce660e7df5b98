#!/usr/bin/env python3
"""global_view launch time on a workload (8192 envs, 300 random steps first): NPP_AMD_LIB=... python tools/gv_time.py [workload]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nclone_amd import levels as level_sets
from nclone_amd.engine import NppBatch

wl = sys.argv[1] if len(sys.argv) > 1 else "doors"
levels, tags = {"c0": level_sets.curriculum0_levels, "mines": level_sets.mine_levels, "doors": level_sets.door_levels,
                "zoo": level_sets.zoo_levels, "c3mixed": level_sets.c3_mixed_levels}[wl]()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
b = NppBatch(n, autoreset=True, outputs=("global_view",))
b.load_levels(levels)
b.assign_levels((np.arange(n) // 64) % len(levels))
acts = torch.from_numpy(np.random.default_rng(0).integers(0, 6, size=(300, n)).astype(np.uint8)).cuda()
for t in range(300):
    b.step(acts[t])
b.render_global_view()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record(b.stream)
for _ in range(20):
    b.render_global_view()
ev[1].record(b.stream)
torch.cuda.synchronize()
print("%s: global_view %.1f us per launch (%s)" % (wl, ev[0].elapsed_time(ev[1]) * 50, os.environ.get("NPP_AMD_LIB", "shipped")))
