import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nclone_amd.engine import NppBatch
from nclone_amd.levels import curriculum0_levels
levels, _ = curriculum0_levels()
n = 8192; K, W = 600, 100
rng = np.random.default_rng(0)
acts = torch.from_numpy(rng.integers(0, 6, size=(K + W, n)).astype(np.uint8)).cuda()
maps = {"blocks of 64": (np.arange(n) // 64) % 128, "blocks of 16": (np.arange(n) // 16) % 128, "blocks of 4": (np.arange(n) // 4) % 128,
        "interleaved": np.arange(n) % 128}
for name, lv in maps.items():
    b = NppBatch(n, autoreset=True); b.load_levels(levels); b.assign_levels(lv)
    for k in range(W): b.step(acts[k], 4, want_terminal=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(W, W + K): b.step(acts[k], 4, want_terminal=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-14s geometry %s  %7.1f us/step  %6.2f M env-steps/s" % (name, b.launch_geometry(), dt / K * 1e6, n * K / dt / 1e6), flush=True)
    b.close()
