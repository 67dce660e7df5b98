#!/usr/bin/env python3
"""A/B of two builds of the library (NPP_AMD_LIB) at several batch sizes: python tools/occupancy_ab.py <n_envs> [steps].
Prints us/step and env-steps/s with the automatic launch geometry (G = 16 / 8 / 4 by batch size)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from nclone_amd.engine import NppBatch  # noqa: E402
from nclone_amd.levels import curriculum0_levels  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 150
W = 300
levels, _ = curriculum0_levels()
acts = torch.from_numpy(np.random.default_rng(0).integers(0, 6, size=(K + W, n)).astype(np.uint8)).cuda()
b = NppBatch(n, autoreset=True)
b.load_levels(levels)
b.assign_levels((np.arange(n) // 64) % len(levels))
for k in range(W):
    b.step(acts[k], 4, want_terminal=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(W, W + K):
    b.step(acts[k], 4, want_terminal=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s n=%6d geometry %s  %8.1f us/step  %7.2f M env-steps/s" % (os.path.basename(os.environ.get("NPP_AMD_LIB", "libnpp_amd.so")), n,
                                                                      b.launch_geometry(), dt / K * 1e6, n * K / dt / 1e6), flush=True)
