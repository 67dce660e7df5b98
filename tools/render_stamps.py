#!/usr/bin/env python3
"""Phase timing of the player_frame kernel (diagnostic -DNPP_RENDER_STAMPS build, NPP_AMD_LIB=...): per workgroup the shader
clocks spent in (0) env / level / position loads, (1) wavefront 0's draw-list build, (2) the barrier wait of thread 0,
(3) shading + stores; plus the draw-list length and the level's entity count."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from nclone_amd.engine import NppBatch  # noqa: E402
from nclone_amd.levels import mine_levels  # noqa: E402

levels, _ = mine_levels()
n = 8192
b = NppBatch(n, autoreset=True, outputs=("player_frame",))
b.load_levels(levels)
b.assign_levels((np.arange(n) // 64) % len(levels))
acts = torch.from_numpy(np.random.default_rng(1).integers(0, 6, size=(60, n)).astype(np.uint8)).cuda()
for s in range(60):
    b.step(acts[s])
for _ in range(3):
    b.render_player_frame()
torch.cuda.synchronize()
fr = b.out.t["player_frame"].cpu().numpy().reshape(n, -1)[:, :64].copy().view(np.uint32)
f, _ = b.dump_state()
live = f[:, 0] - 42 < 600
st = fr[live].astype(np.float64)
print("live frames %d of %d" % (live.sum(), n))
for k, name in enumerate(["loads", "list build (wave 0)", "barrier", "shade+store"]):
    print("%-22s mean %8.0f  p50 %8.0f  p95 %8.0f clocks" % (name, st[:, k].mean(), np.median(st[:, k]), np.percentile(st[:, k], 95)))
if st.shape[1] > 7:
    print("pass 1 (classify + plain spans) mean %.0f clocks; queued spans mean %.0f p95 %.0f max %d of 1764"
          % (st[:, 7].mean(), st[:, 6].mean(), np.percentile(st[:, 6], 95), st[:, 6].max()))
if st.shape[1] > 12:
    for k, name in zip(range(8, 13), ["build: header / pointers", "build: first record + state word", "build: drawables of round 0",
                                      "build: appends + further rounds", "build: row masks"]):
        print("%-36s mean %8.0f  p50 %8.0f clocks" % (name, st[:, k].mean(), np.median(st[:, k])))
print("draw list length mean %.1f max %d; level entities mean %.0f" % (st[:, 4].mean(), st[:, 4].max(), st[:, 5].mean()))
