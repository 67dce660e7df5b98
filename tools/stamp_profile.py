#!/usr/bin/env python3
"""Diagnostic: build a -DNPP_STAMPS copy of the library, run the bench workload and print per-phase shader-clock
totals (lane 0 of every wavefront).  Never used by the product, tests or bench."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ["load+stage", "tick loop total", "think_mines", "integrate+gather", "4 substeps", "post_collision", "ninja_think",
          "flags+obs", "store", "sweep+setup (in substeps)", "depen iterations fast (count)", "depen iterations LDS (count)"]


def main():
    import torch
    from nclone_amd import _native as nat

    # built in the build container (hipcc cross-compiles; ~4 min for the single translation unit) and shipped to the GPU box in build_ab/
    out = os.path.join(ROOT, "build_ab", "libnpp_stamps.so")
    csrc = os.path.join(ROOT, "nclone_amd", "csrc")
    if not os.path.isfile(out) or "--build" in sys.argv:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call([sys.executable, "-m", "nclone_amd.build_native", "--out", out, "--", "-DNPP_STAMPS", "-Wno-unused-value",
                               "-Wno-unused-result"], cwd=ROOT)
        if "--build" in sys.argv:
            return
    nat.LIB_PATH = out
    from nclone_amd.engine import NppBatch
    from nclone_amd import levels as level_sets

    lib = nat.lib()
    lib.npp_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
    n = 8192
    workload = os.environ.get("NPP_STAMP_WORKLOAD", "c0")
    levels, _ = {"c0": level_sets.curriculum0_levels, "zoo": level_sets.zoo_levels, "mines": level_sets.mine_levels, "doors": level_sets.door_levels}[workload]()
    print("workload", workload, len(levels), "levels")
    rng = np.random.default_rng(0)
    K, W = 100, 100
    acts = torch.from_numpy(rng.integers(0, 6, size=(K + W, n)).astype(np.uint8)).cuda()
    for g in [int(x) for x in ([v for v in sys.argv[1:] if not v.startswith("-")] or ["64", "16"])]:
        b = NppBatch(n, autoreset=True)
        b.load_levels(levels)
        b.set_launch_geometry(g, 4 if g <= 16 else 1)
        b.set_step_variant(int(os.environ.get("NPP_STAMP_VARIANT", "1")))
        b.assign_levels((np.arange(n) // 64) % len(levels))
        for k in range(W):
            b.step(acts[k], 4, want_terminal=False)
        lib.npp_debug_stamps(None, 0, 1)
        for k in range(W, W + K):
            b.step(acts[k], 4, want_terminal=False)
        waves = n * g // 64
        buf = (C.c_ulonglong * (32 + 16384))()
        lib.npp_debug_stamps(buf, waves, 1)
        print("G=%d: %d waves/launch, %d launches; shader-clock cycles per wave per launch (100 MHz s_memtime ticks x?)" % (g, waves, K))
        for i, name in enumerate(PHASES):
            print("  %-18s %12.1f" % (name, buf[i] / (waves * K)))
        # per-launch distribution of per-wave totals
        tot = []
        for k in range(W + K, W + K + 10):
            b.step(acts[k % (W + K)], 4, want_terminal=False)
            lib.npp_debug_stamps(buf, waves, 1)
            tot.append(np.array(buf[12:12 + waves], dtype=np.float64))
            if k == W + K:
                print("  slowest wave breakdown:", {PHASES[i]: int(buf[12 + waves + i]) for i in range(12)})
        tot = np.stack(tot)
        print("  per-wave total cycles per launch: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f (max per launch: %s)" % (
            tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(),
            ",".join("%.0f" % v for v in tot.max(axis=1))))
        b.close()


if __name__ == "__main__":
    main()
