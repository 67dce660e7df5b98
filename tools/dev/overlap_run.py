"""A short doors --full-obs run with npp_set_obs_overlap for a rocprofv3 kernel trace (tools/dev/overlap_timeline.py reads it)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from nclone_amd.engine import NppBatch
from nclone_amd.levels import door_levels

pct = [int(c) for c in sys.argv[1].split('+')] if len(sys.argv) > 1 else [40]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 8192
levels, _ = door_levels()
b = NppBatch(n, autoreset=True, outputs=["work", "spatial_context", "switch_states", "player_frame", "global_view",
                                         "reachability_features", "mine_sdf_features"])
b.load_levels(levels)
b.assign_levels((np.arange(n) // 64) % len(levels))
b.set_step_variant(0)
b.set_obs_overlap(pct)
acts = torch.from_numpy(np.random.default_rng(0).integers(0, 6, size=(steps, n)).astype(np.uint8)).cuda()
for t in range(steps):
    b.step(acts[t], 4, want_terminal=False)
    b.switch_states()
    b.render_player_frame()
    b.render_global_view()
    b.reachability()
    b.join()
torch.cuda.synchronize()
