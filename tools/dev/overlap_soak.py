"""Soak run of the observation overlap: N steps of the full observation Dict with the overlap on and off, outputs hashed every step
(a rare ordering bug between the streams would show as one differing step among thousands)."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from nclone_amd.engine import NppBatch
from nclone_amd import levels as level_sets

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n = 4096
wl = sys.argv[2] if len(sys.argv) > 2 else "doors"
levels, _ = {"doors": level_sets.door_levels, "zoo": level_sets.zoo_levels, "mines": level_sets.mine_levels}[wl]()
outs = ("positions", "spatial_context", "switch_states", "player_frame", "global_view", "reachability_features", "mine_sdf_features", "reach_status")
acts = torch.from_numpy(np.random.default_rng(3).integers(0, 6, size=(steps, n)).astype(np.uint8)).cuda()
digests = []
for cuts in (0, 50, (10, 30, 60)):
    b = NppBatch(n, autoreset=True, outputs=outs)
    b.load_levels(levels)
    b.assign_levels((np.arange(n) // 64) % len(levels))
    b.set_step_variant(0)
    b.set_obs_overlap(cuts)
    b.reset()
    hs = []
    for t in range(steps):
        b.step(acts[t])
        b.render_player_frame()
        b.render_global_view()
        b.reachability(with_switch_states=True)
        b.join()
        if t % 4 == 0:   # the whole output block as one device-side checksum (no host copy in the loop)
            hs.append(b.out.dev.view(torch.int32).to(torch.int64).sum())
    torch.cuda.synchronize()
    f, di = b.dump_state()
    hs = torch.stack(hs).cpu().numpy()
    digests.append((hs, hashlib.sha256(f.tobytes() + di.tobytes()).hexdigest()))
    print(wl, "cuts", cuts, "steps", steps, "state", digests[-1][1][:16], flush=True)
    del b
for hs, st in digests[1:]:
    bad = np.flatnonzero(hs != digests[0][0])
    assert st == digests[0][1] and len(bad) == 0, ("first differing sampled step", bad[:5] * 4)
print("identical")
