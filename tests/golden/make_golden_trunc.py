#!/usr/bin/env python3
"""Golden values of the reference's dynamic truncation limit (gym_environment/truncation_calculator.py:19-57), produced by
CALLING the reference function (build container only; same import arrangement as make_golden_reach.py: the package
`nclone.gym_environment` needs gymnasium, so a bare package object stands in for its __init__ and the numpy-only module is
loaded under its normal dotted name).

    python3 tests/golden/make_golden_trunc.py      # -> trunc.npz

  area  i32[k]  surface areas: every value 0..4000 plus the `sa<k>` of the levels in reach.npz / reach2.npz
  limit i32[k]  calculate_truncation_limit(area, 0)  (the env passes reachable_mine_count = 0, npp_environment.py:1238-1256)
  limit_mines i32[k, 4] the same with 1, 2, 5, 20 mines (the Python helper nclone_amd.vec_env.calculate_truncation_limit)
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference"


def main():
    sys.path.insert(0, SRC)
    pkg = types.ModuleType("nclone.gym_environment")
    pkg.__path__ = [os.path.join(SRC, "nclone", "gym_environment")]
    sys.modules["nclone.gym_environment"] = pkg
    from nclone.gym_environment.truncation_calculator import calculate_truncation_limit

    areas = list(range(0, 4001))
    for fx in ("reach.npz", "reach2.npz"):
        z = np.load(os.path.join(HERE, fx))
        n = len(bytes(z["names"]).decode().split("\n"))
        areas += [int(z["sa%d" % k][0]) for k in range(n)]
    area = np.array(areas, dtype=np.int32)
    limit = np.array([calculate_truncation_limit(float(a), 0) for a in area], dtype=np.int32)
    lm = np.array([[calculate_truncation_limit(float(a), m) for m in (1, 2, 5, 20)] for a in area], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "trunc.npz"), area=area, limit=limit, limit_mines=lm)
    print("trunc.npz", len(area), "areas; limits", int(limit.min()), "..", int(limit.max()))


if __name__ == "__main__":
    main()
