#!/usr/bin/env python3
"""GPU vs mul-oracle, tick for tick, on the 26 zoo replays: prints the first divergence of every replay."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nclone_amd.engine import NppBatch
from oracle import oracle as om

g = os.path.join(ROOT, "tests", "golden")
c = np.load(os.path.join(g, "corpus.npz"))
z = np.load(os.path.join(g, "zoo.npz"))
idx = list(z["idx"])
if len(sys.argv) > 1:
    idx = [int(a) for a in sys.argv[1].split(",")]
n = len(idx)
b = NppBatch(n, autoreset=False)
b.load_levels([c["m%d" % i] for i in idx])
b.assign_levels(np.arange(n))
print("geometry", b.launch_geometry())
sims = []
for i in idx:
    o = om.Oracle("mul"); o.load(c["m%d" % i].astype(np.float64)); sims.append(o)
T = [len(z["t%d" % i]) for i in idx]
tmax = max(T)
inputs = np.zeros((tmax, n), dtype=np.uint8)
for k, i in enumerate(idx):
    inputs[:T[k], k] = c["in%d" % i][:T[k]]
d_in = torch.from_numpy(inputs).cuda()
first = {}
for tick in range(tmax):
    b.tick(d_in[tick:tick + 1])
    f, di = b.dump_state()
    cs = b.entity_checksum()
    for k in range(n):
        if tick >= T[k] or k in first:
            continue
        h, j = om.controls(int(inputs[tick, k]))
        sims[k].tick(h, j)
        of, od = sims[k].core()
        oc = sims[k].entity_checksum()
        okf = np.array_equal(f[k], of); okd = np.array_equal(di[k, :22], od[:22]); okc = np.array_equal(cs[k], oc)
        if not (okf and okd and okc):
            first[k] = tick
            print("replay %d DIVERGES at tick %d: f %s d %s c %s" % (idx[k], tick, okf, okd, okc))
            if not okf: print("   f gpu", f[k], "\n   f ora", of)
            if not okd: print("   d gpu", di[k, :22], "\n   d ora", od[:22])
            if not okc: print("   c gpu", cs[k], "\n   c ora", oc, "\n   diff", cs[k] - oc)
print("replays %d, divergent %d" % (n, len(first)))
