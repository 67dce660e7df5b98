#!/usr/bin/env python3
"""Probe for the depenetration iteration cost: 64 envs resting in the V crease of curriculum-0 level 29 (114 iterations per
tick, an exact fixed point under NOOP), G = 16 -> 16 wavefronts that do nothing but iterate.  Run under rocprofv3 --pmc to get
instructions and cycles per iteration (456 iterations per env-step)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nclone_amd.engine import NppBatch
from nclone_amd.levels import curriculum0_levels

levels, _ = curriculum0_levels()
n = 64
b = NppBatch(n, autoreset=True)
b.load_levels([levels[29]])
b.set_launch_geometry(16, 4)
b.assign_levels(np.zeros(n, dtype=np.int64))
acts = torch.zeros((n,), dtype=torch.uint8, device="cuda")
for k in range(60):          # settle into the crease
    b.step(acts, 4, want_terminal=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 200
e0.record()
for k in range(K):
    b.step(acts, 4, want_terminal=False)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / K
f, i = b.dump_state()
print("level 29 x 64 envs resting: %.1f us/step, %.3f us per iteration (456 per step); pos %s state %d" % (us, us / 456, f[0, :2], i[0, 0]))
