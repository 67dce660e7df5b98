import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nclone_amd.engine import NppBatch
from nclone_amd.levels import curriculum0_levels
levels, _ = curriculum0_levels()
K, W = 100, 30
for n in (16384, 32768, 65536):
    rng = np.random.default_rng(0)
    acts = torch.from_numpy(rng.integers(0, 6, size=(K + W, n)).astype(np.uint8)).cuda()
    for g in (0, 2, 4, 8, 16):
        b = NppBatch(n, autoreset=True)
        b.load_levels(levels)
        b.set_launch_geometry(g, 0)
        b.assign_levels((np.arange(n) // 64) % len(levels))
        for k in range(W): b.step(acts[k], 4, want_terminal=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(W, W + K): b.step(acts[k], 4, want_terminal=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("n=%6d G=%2s (ran %s)  %8.1f us/step  %7.2f M env-steps/s" % (n, g or "auto", b.launch_geometry(), dt / K * 1e6, n * K / dt / 1e6), flush=True)
        b.close()
