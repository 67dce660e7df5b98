#!/usr/bin/env python3
"""Time npp_step for several launch geometries at the bench workload (run on the GPU box)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nclone_amd.engine import NppBatch  # noqa: E402
from nclone_amd.levels import curriculum0_levels, mine_levels  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    which = sys.argv[2] if len(sys.argv) > 2 else "c0"
    levels, _ = curriculum0_levels() if which == "c0" else mine_levels()
    K, W = 300, 100
    rng = np.random.default_rng(0)
    acts = torch.from_numpy(rng.integers(0, 6, size=(K + W, n)).astype(np.uint8)).cuda()
    for g in (1, 2, 4, 8, 16, 32, 64):
        for wpb in (1, 2, 4):
            b = NppBatch(n, autoreset=True)
            b.load_levels(levels)
            b.set_launch_geometry(g, wpb)
            b.assign_levels((np.arange(n) // 64) % len(levels))
            for k in range(W):
                b.step(acts[k], 4, want_terminal=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(W, W + K):
                b.step(acts[k], 4, want_terminal=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("n=%d %s G=%2d wpb=%d  %8.1f us/step  %7.2f M env-steps/s" % (n, which, g, wpb, dt / K * 1e6, n * K / dt / 1e6), flush=True)
            b.close()


if __name__ == "__main__":
    main()
