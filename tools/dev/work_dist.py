"""Distribution of the per-env depenetration iteration counts of a step over envs / wavefronts (4 envs) / workgroups (16 envs):
how much of the step's long chains is a few envs dragging their lane-mates along."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from nclone_amd import levels as level_sets
from nclone_amd.engine import NppBatch

wl = sys.argv[1] if len(sys.argv) > 1 else "doors"
levels, tags = {"c0": level_sets.curriculum0_levels, "mines": level_sets.mine_levels, "doors": level_sets.door_levels,
                "c3mixed": level_sets.c3_mixed_levels}[wl]()
n, steps = 8192, 1300
b = NppBatch(n, autoreset=True, outputs=["work"])
b.load_levels(levels)
b.assign_levels((np.arange(n) // 64) % len(levels))
acts = torch.from_numpy(np.random.default_rng(0).integers(0, 6, size=(steps, n)).astype(np.uint8)).cuda()
work = torch.zeros((100, n), dtype=torch.int16, device="cuda")
for t in range(steps):
    b.step(acts[t], 4, want_terminal=False, work_out=work[t - (steps - 100)] if t >= steps - 100 else None)
torch.cuda.synchronize()
w = work.cpu().numpy().astype(np.int64) & 0xffff
for thr in (16, 32, 64, 128, 256):
    e = (w >= thr).mean()
    wv = (w.reshape(100, n // 4, 4).max(axis=2) >= thr).mean()
    wg = (w.reshape(100, n // 16, 16).max(axis=2) >= thr).mean()
    print("%s: iterations >= %3d: envs %.4f  wavefronts(4) %.4f  workgroups(16) %.4f" % (wl, thr, e, wv, wg))
print("mean iterations: env %.2f, wavefront max %.2f, workgroup max %.2f" % (w.mean(), w.reshape(100, n // 4, 4).max(axis=2).mean(), w.reshape(100, n // 16, 16).max(axis=2).mean()))
# persistence: does an env heavy at step t stay heavy at t + 16?
for lag in (1, 2, 4, 8, 16):
    h0, h1 = w[:-lag] >= 64, w[lag:] >= 64
    g0 = w.reshape(100, n // 16, 16).max(axis=2)
    k0, k1 = g0[:-lag] >= 64, g0[lag:] >= 64
    print("lag %2d: P(env heavy again) = %.3f (base %.4f); P(workgroup heavy again) = %.3f (base %.4f)" % (
        lag, (h0 & h1).sum() / max(1, h0.sum()), h0.mean(), (k0 & k1).sum() / max(1, k0.sum()), k0.mean()))
