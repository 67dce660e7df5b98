"""Child process of tests/test_host_sanitized.py: runs under LD_PRELOAD=libasan with nclone_amd/libnpp_host_asan.so (the GPU-free
part of the library built with -fsanitize=address,undefined).  Exercises every host-only entry point on the fixture levels and on
malformed maps; any sanitizer report aborts the process (non-zero exit)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RN = 84 * 46


def main():
    lib = C.CDLL(sys.argv[1])
    D = C.POINTER(C.c_double)
    V = C.c_void_p
    lib.npp_compile_level_segments.argtypes = [D, C.c_int64, V, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
    lib.npp_compile_level_entities.argtypes = [D, C.c_int64, V, C.c_int, C.POINTER(C.c_int)]
    lib.npp_compile_level_zoo.argtypes = [D, C.c_int64, V, V, C.c_int, C.POINTER(C.c_int)]
    lib.npp_plan_zoo_block.argtypes = [D, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.npp_reach_compile.argtypes = [D, C.c_int64] + [V] * 12
    lib.npp_reach_compile_miss.argtypes = [D, C.c_int64] + [V] * 4
    lib.npp_reach_features_host.argtypes = [D, C.c_int64, V, V, C.c_int, V, V, V]
    lib.npp_reach_rollout_host.argtypes = [D, C.c_int64, V, V, V, C.c_int, V, V, V]
    lib.npp_level_truncation_limit.argtypes = [D, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    ptr = lambda a: a.ctypes.data_as(V)

    seg = np.zeros((16384, 8), np.int16)
    ent = np.zeros((4096, 6), np.float64)
    edges = np.zeros(2 * 89 * 51, np.int32)
    mov = np.zeros((1024, 4), np.float64)
    info = np.zeros(32, np.int32)
    bufs = dict(base_in=np.zeros(RN, np.uint8), base_adj=np.zeros(RN, np.uint8), phys=np.zeros(RN, np.uint8), inn=np.zeros(RN, np.uint8),
                adj=np.zeros(RN, np.uint8), dist=np.zeros((2, RN)), hop=np.zeros((2, RN), np.int16), mh=np.zeros((2, RN, 2)),
                sdf=np.ones((50, 88), np.float32), grad=np.zeros((50, 88, 2), np.float32), sc=np.zeros(8))
    cg = np.zeros(RN, np.uint8)
    ast = np.zeros((16, RN))
    mm = np.zeros(RN)

    def everything(m, positions=None):
        m = np.ascontiguousarray(np.asarray(m, dtype=np.float64))
        mp = m.ctypes.data_as(D)
        n, uns = C.c_int(0), C.c_uint32(0)
        rcs = [lib.npp_compile_level_segments(mp, m.size, ptr(seg), len(seg), C.byref(n), C.byref(uns)),
               lib.npp_compile_level_entities(mp, m.size, ptr(ent), len(ent), C.byref(n)),
               lib.npp_compile_level_zoo(mp, m.size, ptr(edges), ptr(mov), len(mov), C.byref(n)),
               lib.npp_reach_compile(mp, m.size, ptr(info), *[ptr(bufs[k]) for k in ("base_in", "base_adj", "phys", "inn", "adj", "dist", "hop",
                                                                                     "mh", "sdf", "grad", "sc")]),
               lib.npp_reach_compile_miss(mp, m.size, ptr(info), ptr(cg), ptr(ast), ptr(mm))]
        lim, area = C.c_int32(0), C.c_int32(0)
        rcs.append(lib.npp_level_truncation_limit(mp, m.size, C.byref(lim), C.byref(area)))
        if positions is not None and len(positions):
            pos = np.ascontiguousarray(positions, dtype=np.float64)
            k = len(pos)
            out = np.zeros((k, 38), np.float32)
            sd = np.zeros((k, 3), np.float32)
            st = np.zeros(k, np.int32)
            raw = np.zeros(k)
            mines = np.zeros((k, 2), np.int32)
            ne = (np.arange(k) % 7 == 0).astype(np.uint8)
            rcs.append(lib.npp_reach_features_host(mp, m.size, ptr(pos), ptr(mines), k, ptr(out), ptr(sd), ptr(st)))
            rcs.append(lib.npp_reach_rollout_host(mp, m.size, ptr(pos), ptr(mines), ptr(ne), k, ptr(out), ptr(st), ptr(raw)))
        return rcs

    from nclone_amd.levels import c3_mixed_levels, door_levels, zoo_levels

    rng = np.random.default_rng(2024)
    levels = zoo_levels()[0] + door_levels()[0] + c3_mixed_levels()[0][::9]
    z = np.load(os.path.join(ROOT, "tests", "golden", "reach_miss.npz"))
    levels += [z["m%d" % k] for k in range(7)] + [z["xm%d" % k] for k in range(14)]
    ok = 0
    for m in levels:
        pos = np.stack([rng.uniform(-30, 1090, size=40), rng.uniform(-30, 630, size=40)], axis=1)   # also outside the map
        rcs = everything(m, pos)
        assert all(rc == 0 for rc in rcs), rcs
        ok += 1
    # the zoo block plan over a whole set
    blob = np.concatenate([np.asarray(m, np.float64) for m in levels])
    offs = np.zeros(len(levels) + 1, np.int64)
    offs[1:] = np.cumsum([len(m) for m in levels])
    d, mv, w = C.c_int(0), C.c_int(0), C.c_int(0)
    assert lib.npp_plan_zoo_block(blob.ctypes.data_as(D), offs.ctypes.data_as(C.POINTER(C.c_int64)), len(levels), C.byref(d), C.byref(mv), C.byref(w)) == 0
    # ---- malformed maps: every call must return (OK or an error code), never touch memory it does not own
    bad = 0
    base = [np.asarray(m, np.float64) for m in levels[:12] + levels[26:32]]
    for it in range(900):
        m = base[it % len(base)].copy()
        kind = it % 9
        if kind == 0:      # truncated anywhere
            m = m[:int(rng.integers(0, len(m)))]
        elif kind == 1:    # tile ids out of range (negative, huge, fractional, nan)
            idx = rng.integers(184, 1150, size=40)
            m[idx] = rng.choice([-1, 33, 34, 37, 38, 200, 255, 256, 0.5, -65535, 65535], size=40)
        elif kind == 2:    # entity coordinates far outside / nan
            for i in range(1230, len(m) - 4, 5):
                if rng.random() < 0.5:
                    m[i + 1], m[i + 2] = rng.choice([-65535, 65535, -1, 0, 176.5, 500, 1000, 10922.6], size=2)
        elif kind == 3:    # entity types that do not exist / door records cut short
            for i in range(1235, len(m) - 4, 5):
                if rng.random() < 0.3:
                    m[i] = rng.choice([-3, 0, 3, 4, 6, 8, 15, 23, 29, 99, 255, 6.5])
        elif kind == 4:    # counts that lie
            m[1156] = rng.choice([-5, 0, 2, 7, 300, 60000, 1.5])
            m[1200] = rng.choice([-5, 0, 2, 300, 60000, 0.5])
        elif kind == 5:    # spawn outside / nan
            m[1231], m[1232] = rng.choice([-65535, -1, 0, 200, 10000, 65535, 7.25], size=2)
        elif kind == 6:    # random garbage of random length
            m = rng.uniform(-300, 300, size=int(rng.integers(0, 2500)))
        elif kind == 7:    # random bytes, like a corrupted file
            m = rng.integers(0, 256, size=int(rng.integers(1, 2500))).astype(np.float64)
        else:              # values no map can hold: refused by the compiler's range check
            m[rng.integers(0, len(m), size=5)] = rng.choice([np.nan, np.inf, -np.inf, 1e9, -1e300], size=5)
        pos = np.stack([rng.uniform(-1e4, 1e4, size=8), rng.uniform(-1e4, 1e4, size=8)], axis=1)
        pos[0] = [np.nan, np.inf]
        rcs = everything(m, pos)
        bad += any(rc != 0 for rc in rcs)
    print("sanitized host run: %d levels, 900 malformed maps (%d rejected)" % (ok, bad))


if __name__ == "__main__":
    main()
