#!/usr/bin/env python3
"""Phase stamps of the global_view kernel (diagnostic build -DNPP_GV_STATS):
    python -m nclone_amd.build_native --out build_ab/libnpp_gvstats.so -- -DNPP_GV_STATS
    NPP_AMD_LIB=build_ab/libnpp_gvstats.so python tools/gv_stats.py [workload]
Prints per-env means of the four phases (copy, draw list + dirty boxes, mark, queue + recompute; s_memtime ticks of 10 ns) and of
the counts (drawables, dirty boxes, dirty cells, records)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nclone_amd import levels as level_sets
from nclone_amd.engine import NppBatch

wl = sys.argv[1] if len(sys.argv) > 1 else "doors"
levels, tags = {"c0": level_sets.curriculum0_levels, "mines": level_sets.mine_levels, "doors": level_sets.door_levels,
                "zoo": level_sets.zoo_levels, "c3mixed": level_sets.c3_mixed_levels}[wl]()
n = 8192
b = NppBatch(n, autoreset=True, outputs=("global_view",))
b.load_levels(levels)
b.assign_levels((np.arange(n) // 64) % len(levels))
acts = torch.from_numpy(np.random.default_rng(0).integers(0, 6, size=(300, n)).astype(np.uint8)).cuda()
for t in range(300):
    b.step(acts[t])
b.render_global_view()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record(b.stream)
for _ in range(10):
    b.render_global_view()
ev[1].record(b.stream)
torch.cuda.synchronize()
print("launch us", ev[0].elapsed_time(ev[1]) * 100)
d = b.out.t["global_view"].reshape(n, -1)[:, :64].contiguous().cpu().numpy().view(np.uint32)
names = ["copy", "list+patches", "mark", "cells/export", "nd", "nb", "nq", "records", "list only", "K2 nq", "K2 in-place flag", "K2 patch bytes"]
for i, k in enumerate(names):
    col = d[:, i].astype(np.float64)
    print("%-8s mean %10.1f  p50 %8.0f  p95 %8.0f  max %8.0f" % (k, col.mean(), np.percentile(col, 50), np.percentile(col, 95), col.max()))
lv = (np.arange(n) // 64) % len(levels)
tot = d[:, :4].sum(axis=1).astype(np.float64)
worst = np.argsort(-np.array([tot[lv == i].mean() for i in range(len(levels))]))[:5]
k2 = d[:, 8].astype(np.float64)
top = np.argsort(-k2)[:8]
print("slowest cell-pass wavefronts:", [(int(e), tags[lv[e]], int(k2[e]), "nq", int(d[e, 9]), "inl", int(d[e, 10]), "nb", int(d[e, 5])) for e in top])
for i in worst:
    print("level", tags[i], "ticks", tot[lv == i].mean(), "nq", d[lv == i, 6].mean(), "nb", d[lv == i, 5].mean(), "nd", d[lv == i, 4].mean())
