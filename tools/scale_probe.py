#!/usr/bin/env python3
"""Probe: step time vs number of envs (fixed G) and vs action pattern, to separate per-wave base cost from the tail."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nclone_amd.engine import NppBatch
from nclone_amd.levels import curriculum0_levels

levels, _ = curriculum0_levels()
K, W = 300, 100
for n in (256, 1024, 2048, 4096, 8192, 16384):
    for pattern in ("random", "noop"):
        rng = np.random.default_rng(0)
        a = rng.integers(0, 6, size=(K + W, n)).astype(np.uint8) if pattern == "random" else np.zeros((K + W, n), dtype=np.uint8)
        acts = torch.from_numpy(a).cuda()
        b = NppBatch(n, autoreset=True)
        b.load_levels(levels)
        b.set_launch_geometry(16, 4)
        b.assign_levels((np.arange(n) // 64) % len(levels))
        for k in range(W):
            b.step(acts[k], 4, want_terminal=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(W, W + K):
            b.step(acts[k], 4, want_terminal=False)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / K
        print("n=%6d %-6s G=16  %7.1f us/step  %6.2f M env-steps/s" % (n, pattern, us, n / us), flush=True)
        b.close()
