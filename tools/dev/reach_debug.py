import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_reach as T
from nclone_amd import _native as nat
from nclone_amd.engine import NppBatch, compile_level_entities
z = np.load("tests/golden/reach.npz"); names = bytes(z["names"]).decode().split("\n")
sup = [k for k, n in enumerate(names) if n not in T.UNSUPPORTED]
N, steps = 8192, 40
lib = nat.lib()
levels = [np.ascontiguousarray(z["m%d" % k]) for k in sup]
level_ids = (np.arange(N) // 64) % len(levels)
b = NppBatch(N, autoreset=True, outputs=T.OUT, fast_reset=True)
b.load_levels(levels); b.assign_levels(level_ids); b.reset(); b.observe()
acts = torch.from_numpy(np.random.default_rng(11).integers(0, 6, size=(steps, N)).astype(np.uint8)).cuda()
is_mine = [compile_level_entities(m)[:, 0] == 1 for m in levels]
total = np.array([int(v.sum()) for v in is_mine])
key = np.full((N, 3), -1, dtype=np.int64); cached = np.zeros((N, 38), np.float32)
for t in range(steps + 1):
    if t: b.step(acts[t - 1])
    b.reachability()
    h = b.to_host(T.OUT + ("flags",))
    pos = h["positions"][:, :2]; sw = (h["flags"] & 4) != 0
    k = np.stack([np.floor_divide(pos[:, 0], 24).astype(np.int64), np.floor_divide(pos[:, 1], 24).astype(np.int64), sw.astype(np.int64)], axis=1)
    changed = (k != key).any(axis=1); oldkey = key.copy(); key[changed] = k[changed]
    feats = h["reachability_features"]
    tot = total[level_ids]; deadly = np.rint(feats[:, 11].astype(np.float64) * tot).astype(np.int32)
    for li in np.unique(level_ids[changed]):
        sel = np.flatnonzero(changed & (level_ids == li))
        out, sd, st = T._host_features(lib, levels[li], pos[sel], np.stack([tot[sel], deadly[sel]], axis=1))
        cached[sel] = out
    bad = np.flatnonzero((feats != cached).any(axis=1))
    if len(bad):
        print("step", t, "bad envs", len(bad))
        for e in bad[:6]:
            c = np.flatnonzero(feats[e] != cached[e])
            print(" env", e, "level", names[sup[level_ids[e]]], "pos", pos[e], "flags", h["flags"][e], "key", k[e], "old", oldkey[e], "changed", changed[e], "cols", c, feats[e][c], cached[e][c], 'tot', tot[e], 'deadly', deadly[e])
        break
