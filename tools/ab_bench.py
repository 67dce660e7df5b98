#!/usr/bin/env python3
"""A/B timing of several builds of libnpp_amd.so on the bench workload (8192 envs, c0 levels), ctypes only.

    python tools/ab_bench.py build_ab/libnpp_X.so nclone_amd/libnpp_amd.so ...
"""
import ctypes as C
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: one HIP runtime for everybody)

from nclone_amd.levels import curriculum0_levels  # noqa: E402


class StepOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("gs", "mask", "epos", "flags", "reward", "frames", "term", "sc", "pos", "work")]


def run(path, steps=600, warmup=60, n=8192, with_sc_field=True):
    L = C.CDLL(os.path.join(ROOT, path))
    H = C.c_void_p
    L.npp_create.argtypes = [C.c_int, C.c_int, C.c_uint, C.POINTER(H)]
    L.npp_load_levels.argtypes = [H, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    L.npp_assign_levels.argtypes = [H, C.c_void_p, C.POINTER(C.c_int32), C.c_int]
    L.npp_step.argtypes = [H, C.c_void_p, C.c_int, C.c_void_p]
    L.npp_sync.argtypes = [H]
    L.npp_destroy.argtypes = [H]
    h = H()
    assert L.npp_create(n, 0, 1, C.byref(h)) == 0
    levels, _ = curriculum0_levels()
    arrs = [np.ascontiguousarray(np.asarray(m, dtype=np.float64)) for m in levels]
    offs = np.zeros(len(arrs) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(a) for a in arrs])
    blob = np.concatenate(arrs)
    assert L.npp_load_levels(h, blob.ctypes.data_as(C.POINTER(C.c_double)), offs.ctypes.data_as(C.POINTER(C.c_int64)), len(arrs)) == 0
    ids = ((np.arange(n) // 64) % len(arrs)).astype(np.int32)
    assert L.npp_assign_levels(h, None, ids.ctypes.data_as(C.POINTER(C.c_int32)), n) == 0
    gs = torch.zeros((n, 41), dtype=torch.float32, device="cuda")
    mask = torch.zeros((n, 6), dtype=torch.int8, device="cuda")
    epos = torch.zeros((n, 6), dtype=torch.float32, device="cuda")
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    rew = torch.zeros(n, dtype=torch.float32, device="cuda")
    fr = torch.zeros(n, dtype=torch.int16, device="cuda")
    so = StepOut(gs.data_ptr(), mask.data_ptr(), epos.data_ptr(), flags.data_ptr(), rew.data_ptr(), fr.data_ptr(), None, None, None, None)
    acts = torch.from_numpy(np.random.default_rng(0).integers(0, 6, size=(steps + warmup, n)).astype(np.uint8)).cuda()
    for k in range(warmup):
        L.npp_step(h, acts[k].data_ptr(), 4, C.byref(so))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(warmup, warmup + steps):
        L.npp_step(h, acts[k].data_ptr(), 4, C.byref(so))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    chk = float(gs.double().sum().item())
    L.npp_destroy(h)
    return us, chk


if __name__ == "__main__":
    for p in sys.argv[1:]:
        for rep in range(2):
            us, chk = run(p)
            print("%-40s %.1f us/step  %.1f M env-steps/s  checksum %.6f" % (p, us, 8192 / us, chk), flush=True)
