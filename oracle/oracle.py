"""ctypes front-end for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (nclone_amd/) never does; it fails loudly when its HIP library is missing.

`Oracle(variant="pow")` squares like CPython's x**2 (libm pow) and matches the reference's
bits in the build container; `variant="mul"` squares by multiplication and is the bit-exact
twin of the HIP kernel.  See nsim_oracle.c for the reference file:line citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

ACTIONS = [(0, 0), (-1, 0), (1, 0), (0, 1), (-1, 1), (1, 1)]
N_DISC = 22


def build(force=False):
    """Compile both oracle variants with gcc (seconds)."""
    need = force or not all(
        os.path.isfile(os.path.join(_HERE, f)) for f in ("libnsim_oracle.so", "libnsim_oracle_mul.so")
    )
    src = os.path.join(_HERE, "nsim_oracle.c")
    if not need:
        for f in ("libnsim_oracle.so", "libnsim_oracle_mul.so"):
            if os.path.getmtime(os.path.join(_HERE, f)) < os.path.getmtime(src):
                need = True
    if need:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "all"])


def _lib(variant):
    if variant in _LIBS:
        return _LIBS[variant]
    name = {"pow": "libnsim_oracle.so", "mul": "libnsim_oracle_mul.so"}[variant]
    path = os.path.join(_HERE, name)
    if not os.path.isfile(path):
        build()
    lib = C.CDLL(path)
    P = C.c_void_p
    lib.osim_create.restype = P
    lib.osim_destroy.argtypes = [P]
    lib.osim_load.argtypes = [P, C.POINTER(C.c_double), C.c_int]
    lib.osim_load.restype = C.c_int
    lib.osim_reset.argtypes = [P]
    lib.osim_fast_reset.argtypes = [P]
    lib.osim_entity_states_dic.argtypes = [P, C.POINTER(C.c_int), C.c_int]
    lib.osim_entity_states_dic.restype = C.c_int
    lib.osim_tick.argtypes = [P, C.c_int, C.c_int]
    lib.osim_get_ninja_state.argtypes = [P, C.POINTER(C.c_double)]
    lib.osim_spatial_context.argtypes = [P, C.POINTER(C.c_float)]
    lib.osim_action_mask.argtypes = [P]
    lib.osim_action_mask.restype = C.c_int
    lib.osim_frame.argtypes = [P]
    lib.osim_frame.restype = C.c_int
    lib.osim_get_core.argtypes = [P, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.osim_dump_csr.argtypes = [P, C.POINTER(C.c_int16), C.c_int]
    lib.osim_dump_csr.restype = C.c_int
    lib.osim_dump_entities.argtypes = [P, C.POINTER(C.c_double), C.c_int]
    lib.osim_dump_entities.restype = C.c_int
    lib.osim_entity_states.argtypes = [P, C.POINTER(C.c_int), C.c_int]
    lib.osim_entity_states.restype = C.c_int
    lib.osim_set_entity_pos.argtypes = [P, C.c_int, C.c_double, C.c_double]
    lib.osim_dump_draw.argtypes = [P, C.POINTER(C.c_double), C.c_int]
    lib.osim_dump_draw.restype = C.c_int
    lib.osim_dump_edges.argtypes = [P, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.osim_entity_checksum.argtypes = [P, C.POINTER(C.c_double)]
    lib.osim_env_step.argtypes = [P, C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.osim_env_step.restype = C.c_int
    lib.osim_run_batch.argtypes = [C.POINTER(P), C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.osim_run_batch.restype = C.c_longlong
    _LIBS[variant] = lib
    return lib


class Oracle:
    """One simulator instance (counterpart of the reference's Simulator, nsim.py:11)."""

    def __init__(self, variant="pow"):
        self.lib = _lib(variant)
        self.h = self.lib.osim_create()
        self.unsupported = 0

    def __del__(self):
        try:
            self.lib.osim_destroy(self.h)
        except Exception:
            pass

    def load(self, map_data):
        m = np.ascontiguousarray(np.asarray(map_data, dtype=np.float64))
        self._map = m
        r = self.lib.osim_load(self.h, m.ctypes.data_as(C.POINTER(C.c_double)), len(m))
        if r < 0:
            raise ValueError("map_data too short")
        self.unsupported = r
        return r

    def reset(self):
        """Simulator.reset (nsim.py:62-76)."""
        self.lib.osim_reset(self.h)

    def fast_reset(self):
        """Simulator.fast_reset (nsim.py:78-140)."""
        self.lib.osim_fast_reset(self.h)

    def entity_states_dic(self):
        buf = np.zeros(4096, dtype=np.int32)
        n = self.lib.osim_entity_states_dic(self.h, buf.ctypes.data_as(C.POINTER(C.c_int)), len(buf))
        return buf[:n].copy()

    def tick(self, hor, jump):
        self.lib.osim_tick(self.h, int(hor), int(jump))

    def env_step(self, action, frame_skip=4):
        fl = C.c_int(0)
        k = self.lib.osim_env_step(self.h, int(action), frame_skip, C.byref(fl))
        return k, fl.value

    @property
    def frame(self):
        return self.lib.osim_frame(self.h)

    def core(self):
        f = np.zeros(12, dtype=np.float64)
        d = np.zeros(N_DISC, dtype=np.int32)
        self.lib.osim_get_core(self.h, f.ctypes.data_as(C.POINTER(C.c_double)), d.ctypes.data_as(C.POINTER(C.c_int)))
        return f, d

    def ninja_state(self):
        o = np.zeros(40, dtype=np.float64)
        self.lib.osim_get_ninja_state(self.h, o.ctypes.data_as(C.POINTER(C.c_double)))
        return o

    def spatial_context(self):
        o = np.zeros(112, dtype=np.float32)
        self.lib.osim_spatial_context(self.h, o.ctypes.data_as(C.POINTER(C.c_float)))
        return o

    def action_mask(self):
        return self.lib.osim_action_mask(self.h)

    def dump_csr(self):
        buf = np.zeros((16384, 8), dtype=np.int16)
        n = self.lib.osim_dump_csr(self.h, buf.ctypes.data_as(C.POINTER(C.c_int16)), len(buf))
        assert n >= 0
        return buf[:n].copy()

    def dump_entities(self):
        buf = np.zeros((4096, 8), dtype=np.float64)
        n = self.lib.osim_dump_entities(self.h, buf.ctypes.data_as(C.POINTER(C.c_double)), len(buf))
        assert n >= 0
        return buf[:n].copy()

    def set_entity_pos(self, kind, x, y):
        """Curriculum repositioning of the exit switch (0) / exit door (1); survives resets."""
        self.lib.osim_set_entity_pos(self.h, int(kind), float(x), float(y))

    def draw_list(self):
        """[n, 13] rows of what the entity layer would draw, entity_dic order (see osim_dump_draw)."""
        buf = np.zeros((4096, 13), dtype=np.float64)
        n = self.lib.osim_dump_draw(self.h, buf.ctypes.data_as(C.POINTER(C.c_double)), len(buf))
        return buf[:n].copy()

    def tiles(self):
        """[44, 25] tile ids incl. the border (map_loader.py:22-37)."""
        t = np.ones((44, 25), dtype=np.int64)
        m = np.asarray(self._map)
        for x in range(42):
            for y in range(23):
                t[x + 1, y + 1] = int(m[184 + x + 42 * y])
        return t

    def edges(self):
        """(hor, ver) int arrays [89, 51]: the grid-edge counters drones and thwumps test."""
        hor = np.zeros(89 * 51, dtype=np.int32)
        ver = np.zeros(89 * 51, dtype=np.int32)
        self.lib.osim_dump_edges(self.h, hor.ctypes.data_as(C.POINTER(C.c_int)), ver.ctypes.data_as(C.POINTER(C.c_int)))
        return hor.reshape(89, 51), ver.reshape(89, 51)

    def entity_checksum(self):
        o = np.zeros(6, dtype=np.float64)
        self.lib.osim_entity_checksum(self.h, o.ctypes.data_as(C.POINTER(C.c_double)))
        return o

    def entity_states(self):
        buf = np.zeros(4096, dtype=np.int32)
        n = self.lib.osim_entity_states(self.h, buf.ctypes.data_as(C.POINTER(C.c_int)), len(buf))
        return buf[:n].copy()


def run_batch(sims, actions, frame_skip=4, max_frames=10000, threads=1):
    """actions: uint8 [n_steps, n_envs]. Returns total ticks executed (auto-reset on termination)."""
    lib = sims[0].lib
    n = len(sims)
    arr = (C.c_void_p * n)(*[s.h for s in sims])
    a = np.ascontiguousarray(actions, dtype=np.uint8)
    assert a.shape[1] == n
    return lib.osim_run_batch(arr, n, a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[0], frame_skip, max_frames, threads)


def controls(b):
    """Replay input byte -> (hor, jump). Restates replay/replay_executor.py:61-84."""
    j = b & 1
    r = (b >> 1) & 1
    l = (b >> 2) & 1
    h = 0 if (l and r) else (-1 if l else (1 if r else 0))
    return h, j
