#!/usr/bin/env python3
"""Render-kernel probe for rocprofv3: 8192 envs on the mine level set, 60 random steps (so the ninjas have spread), then the
player_frame kernel N times.  Prints the mean kernel time from HIP events.

    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES ... -- python3 tools/render_probe.py [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from nclone_amd.engine import NppBatch  # noqa: E402
from nclone_amd.levels import mine_levels  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
levels, _ = mine_levels()
n = 8192
b = NppBatch(n, autoreset=True, outputs=("player_frame",))
b.load_levels(levels)
b.assign_levels((np.arange(n) // 64) % len(levels))
acts = torch.from_numpy(np.random.default_rng(1).integers(0, 6, size=(60, n)).astype(np.uint8)).cuda()
for s in range(60):
    b.step(acts[s])
b.render_player_frame()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(b.stream)
for _ in range(reps):
    b.render_player_frame()
e1.record(b.stream)
torch.cuda.synchronize()
f, _ = b.dump_state()
print("render: %.1f us per launch over %d launches; %.0f %% of the frames are all padding (player_x > 642)"
      % (e0.elapsed_time(e1) * 1e3 / reps, reps, 100.0 * float((f[:, 0] - 42 >= 600).mean())))
