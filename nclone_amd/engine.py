"""NppBatch: the thin host object over one native handle (N environments on one GPU).

PyTorch is used only as plumbing: device buffers for actions/observations and the HIP stream.  All
simulation work happens in the HIP kernels behind the C ABI (include/npp_amd.h).
"""
import ctypes as C

import numpy as np
import torch

from . import _native as nat


class _DeviceStream:
    """torch.cuda.device + torch.cuda.stream in one context manager."""

    def __init__(self, device, stream):
        self._d = torch.cuda.device(device)
        self._s = torch.cuda.stream(stream)

    def __enter__(self):
        self._d.__enter__()
        self._s.__enter__()

    def __exit__(self, *a):
        self._s.__exit__(*a)
        self._d.__exit__(*a)


def _as_f64_blob(levels):
    """levels: sequence of 1-D arrays/lists/bytes of raw map_data values -> (blob f64, offsets i64)."""
    arrs = []
    for m in levels:
        if isinstance(m, (bytes, bytearray)):
            m = np.frombuffer(bytes(m), dtype=np.uint8)
        arrs.append(np.ascontiguousarray(np.asarray(m, dtype=np.float64).ravel()))
    offsets = np.zeros(len(arrs) + 1, dtype=np.int64)
    for i, a in enumerate(arrs):
        offsets[i + 1] = offsets[i] + len(a)
    blob = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.float64)
    return blob, offsets


# every output the kernels can produce: name -> (per-env shape, dtype).  The first six are the packed observation block
# of BASELINE.json config 4 (what a learner on another GPU needs each step); they sit first and contiguous in the block.
_FIELDS = {
    "game_state": ((41,), torch.float32),
    "entity_pos": ((6,), torch.float32),
    "reward": ((), torch.float32),
    "frames": ((), torch.int16),
    "action_mask": ((6,), torch.int8),
    "flags": ((), torch.uint8),
    "terminal_state": ((41,), torch.float32),
    "spatial_context": ((112,), torch.float32),
    "positions": ((6,), torch.float64),
    "work": ((), torch.int16),
    "switch_states": ((25,), torch.float32),
    "player_frame": ((84, 84, 1), torch.uint8),
    "global_view": ((176, 100, 1), torch.uint8),
    "reachability_features": ((38,), torch.float32),
    "mine_sdf_features": ((3,), torch.float32),
    "reach_status": ((), torch.int32),
}
_ALWAYS = ("game_state", "entity_pos", "reward", "frames", "action_mask", "flags", "terminal_state")
_PACKED = ("game_state", "entity_pos", "reward", "frames", "action_mask", "flags")
_OPTIONAL = ("spatial_context", "positions", "work", "switch_states", "player_frame", "global_view", "reachability_features",
             "mine_sdf_features", "reach_status")


class OutputBlock:
    """All enabled outputs of one handle in ONE contiguous device allocation (each field 256-byte aligned), plus two pinned
    host mirrors.  One block means: one RCCL all_gather moves the whole packed observation of a rank (config 4), and one
    asynchronous device-to-host copy + one synchronisation serves a numpy training loop (`to_host`)."""

    def __init__(self, n, device, names):
        self.n = int(n)
        self.names = [k for k in _FIELDS if k in names]   # canonical order: packed observation first
        self.offsets = {}
        off = 0
        for k in self.names:
            shape, dt = _FIELDS[k]
            nbytes = self.n * int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dt).element_size()
            self.offsets[k] = (off, nbytes)
            off = (off + nbytes + 255) // 256 * 256
            if k == _PACKED[-1]:
                self.packed_bytes = off
        self.nbytes = off
        self.dev = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.t = {k: self._view(self.dev, k) for k in self.names}
        self._host = [None, None]
        self._flip = 0

    def _view(self, base, k):
        off, nbytes = self.offsets[k]
        shape, dt = _FIELDS[k]
        return base[off:off + nbytes].view(dt).view((self.n,) + tuple(shape))

    def packed(self):
        """uint8 view of the packed observation (game_state, entity_pos, reward, frames, action_mask, flags)."""
        return self.dev[:self.packed_bytes]

    def split_packed(self, gathered, world):
        """Views into an all-gathered [world * packed_bytes] uint8 tensor: {name: [world, n, ...]}."""
        g = gathered.view(world, self.packed_bytes)
        out = {}
        for k in _PACKED:
            off, nbytes = self.offsets[k]
            shape, dt = _FIELDS[k]
            out[k] = g[:, off:off + nbytes].contiguous().view(dt).view((world, self.n) + tuple(shape))
        return out

    def to_host(self, stream, names=None):
        """One async copy of the block (up to the last requested field) into a pinned mirror on `stream`, one
        synchronisation; returns {name: numpy view}.  Two mirrors alternate, so the arrays of the previous call stay valid
        until the call after this one (obs_t and obs_t+1 can be held together)."""
        names = self.names if names is None else [k for k in self.names if k in names]
        end = max(self.offsets[k][0] + self.offsets[k][1] for k in names)
        if self._host[self._flip] is None:
            self._host[self._flip] = torch.empty(self.nbytes, dtype=torch.uint8, pin_memory=True)
        host = self._host[self._flip]
        self._flip ^= 1
        with torch.cuda.stream(stream):
            host[:end].copy_(self.dev[:end], non_blocking=True)
        stream.synchronize()
        return {k: self._view(host, k).numpy() for k in names}


class NppBatch:
    """N environments stepped in lock-step on one GPU.

    Counterpart of N instances of the reference's NPlayHeadless (nclone/nplay_headless.py:28).
    """

    def __init__(self, n_envs, device=0, autoreset=True, allow_unsupported=False, frame_centered=False, stream=None,
                 outputs=(), fast_reset=False):
        self.lib = nat.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("nclone_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.n = int(n_envs)
        self.device = torch.device("cuda", int(device))
        flags = ((nat.FLAG_AUTORESET if autoreset else 0) | (nat.FLAG_ALLOW_UNSUPPORTED if allow_unsupported else 0)
                 | (nat.FLAG_FRAME_CENTERED if frame_centered else 0) | (nat.FLAG_FAST_RESET if fast_reset else 0))
        h = C.c_void_p()
        nat.check(None, self.lib.npp_create(self.n, int(device), flags, C.byref(h)))
        self.h = h
        self.n_levels = 0
        with torch.cuda.device(self.device):
            # every launch of this handle is ordered on ONE HIP stream; handles on different streams overlap on the GPU
            self.stream = stream if stream is not None else torch.cuda.current_stream()
            nat.check(self.h, self.lib.npp_set_stream(self.h, C.c_void_p(self.stream.cuda_stream)))
        self._enabled = set(_ALWAYS)
        for k in outputs:
            if k not in _OPTIONAL:
                raise ValueError("unknown output %r (optional outputs: %s)" % (k, ", ".join(_OPTIONAL)))
            self._enabled.add(k)
        self._build_block()

    def _build_block(self):
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            self.out = OutputBlock(self.n, self.device, self._enabled)
        t = self.out.t
        self.game_state, self.action_mask, self.entity_pos = t["game_state"], t["action_mask"], t["entity_pos"]
        self.flags, self.reward, self.frames, self.terminal_state = t["flags"], t["reward"], t["frames"], t["terminal_state"]
        self.spatial_context = t.get("spatial_context")
        self.positions = t.get("positions")
        self.work = t.get("work")

        def ptr(k):
            return t[k].data_ptr() if k in t else None

        self._out = nat.StepOut(ptr("game_state"), ptr("action_mask"), ptr("entity_pos"), ptr("flags"), ptr("reward"),
                                ptr("frames"), ptr("terminal_state"), ptr("spatial_context"), ptr("positions"), ptr("work"))
        self._out_min = nat.StepOut(ptr("game_state"), ptr("action_mask"), ptr("entity_pos"), ptr("flags"), ptr("reward"),
                                    ptr("frames"), None, ptr("spatial_context"), ptr("positions"), ptr("work"))

    def enable_outputs(self, *names):
        """Add optional outputs (spatial_context, positions, work, switch_states, player_frame, global_view, reachability_features, mine_sdf_features, reach_status); the output
        block is re-allocated, so tensors obtained earlier are stale."""
        new = [k for k in names if k not in self._enabled]
        for k in new:
            if k not in _OPTIONAL:
                raise ValueError("unknown output %r" % (k,))
            self._enabled.add(k)
        if new:
            self._build_block()

    def enable_spatial_context(self):
        """Also produce the 112-float spatial_context observation (8x8 tile categories + 8 nearest mines)."""
        self.enable_outputs("spatial_context")

    def _ctx(self):
        """Launches, copies and allocations of this handle run with its device current and on its stream."""
        return _DeviceStream(self.device, self.stream)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.npp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- levels -----------------------------------------------------------------------------------------------
    def load_levels(self, levels):
        blob, offsets = _as_f64_blob(levels)
        nat.check(self.h, self.lib.npp_load_levels(
            self.h, blob.ctypes.data_as(C.POINTER(C.c_double)), offsets.ctypes.data_as(C.POINTER(C.c_int64)), len(offsets) - 1))
        self.n_levels = len(offsets) - 1

    def assign_levels(self, level_ids, env_ids=None):
        lv = np.ascontiguousarray(level_ids, dtype=np.int32)
        if env_ids is None:
            nat.check(self.h, self.lib.npp_assign_levels(self.h, None, lv.ctypes.data_as(C.POINTER(C.c_int32)), len(lv)))
        else:
            ev = np.ascontiguousarray(env_ids, dtype=np.int32)
            nat.check(self.h, self.lib.npp_assign_levels(
                self.h, ev.ctypes.data_as(C.POINTER(C.c_int32)), lv.ctypes.data_as(C.POINTER(C.c_int32)), len(lv)))

    def set_truncation_limit(self, limit):
        if np.isscalar(limit):
            nat.check(self.h, self.lib.npp_set_truncation_limit(self.h, None, int(limit)))
        else:
            lim = np.ascontiguousarray(limit, dtype=np.int32)
            assert len(lim) == self.n
            nat.check(self.h, self.lib.npp_set_truncation_limit(self.h, lim.ctypes.data_as(C.POINTER(C.c_int32)), 0))

    def set_dynamic_truncation(self, enable=True):
        """The reference env's per-level limit: int(clip(sqrt(reachable surface area) * 500, 1200, 10000)) frames
        (truncation_calculator.py:19-57), re-applied at every level (re)assignment."""
        nat.check(self.h, self.lib.npp_set_dynamic_truncation(self.h, 1 if enable else 0))

    def set_launch_geometry(self, lanes_per_env=0, waves_per_block=0):
        nat.check(self.h, self.lib.npp_set_launch_geometry(self.h, int(lanes_per_env), int(waves_per_block)))

    def set_step_variant(self, variant=-1):
        """Build variant of the step kernel: -1 = autotune on this handle's workload (default), 0..2 pin one (same bits either way)."""
        nat.check(self.h, self.lib.npp_set_step_variant(self.h, int(variant)))

    def set_obs_overlap(self, cuts=0):
        """Observation overlap (include/npp_amd.h npp_set_obs_overlap_parts): cut the step's heavy-first workgroup order at `cuts`
        (one percentage or up to three ascending ones) and run the pieces, and the observation kernels behind each, on streams of
        their own; 0 / () switches it off.  Same bits; call join() (to_host() does) before consuming the outputs."""
        cuts = [int(cuts)] if np.isscalar(cuts) else [int(c) for c in cuts]
        cuts = [c for c in cuts if c > 0]
        arr = (C.c_int * max(1, len(cuts)))(*cuts)
        nat.check(self.h, self.lib.npp_set_obs_overlap_parts(self.h, arr, len(cuts)))

    def join(self):
        nat.check(self.h, self.lib.npp_join(self.h))

    def step_variant(self):
        """(variant npp_step launches now, True once the autotuner has decided or a variant is pinned)"""
        v, t = C.c_int(0), C.c_int(0)
        nat.check(self.h, self.lib.npp_get_step_variant(self.h, C.byref(v), C.byref(t)))
        return v.value, bool(t.value)

    def launch_geometry(self):
        g, w = C.c_int(0), C.c_int(0)
        nat.check(self.h, self.lib.npp_get_launch_geometry(self.h, C.byref(g), C.byref(w)))
        return g.value, w.value

    # ---- stepping ---------------------------------------------------------------------------------------------
    def reset(self, mask=None, mode="default"):
        """mode: "default" (the handle's NPP_FLAG_FAST_RESET decides), "full" (Simulator.reset), "fast" (fast_reset)."""
        code = {"default": 0, "full": 1, "fast": 2}[mode]
        if mask is None:
            nat.check(self.h, self.lib.npp_reset_ex(self.h, None, code))
        else:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert len(m) == self.n
            nat.check(self.h, self.lib.npp_reset_ex(self.h, m.ctypes.data_as(C.POINTER(C.c_uint8)), code))

    def step(self, actions, frame_skip=4, want_terminal=True, work_out=None):
        """actions: uint8 CUDA tensor [N] with values 0..5.  Asynchronous; outputs land in self.game_state etc.
        work_out: optional int16 CUDA tensor [N] that receives this step's per-env depenetration iteration counts (instead
        of the block's `work` field) -- lets a profiler keep one row per step."""
        assert actions.dtype == torch.uint8 and actions.is_cuda and actions.numel() == self.n
        out = self._out if want_terminal else self._out_min
        if work_out is not None:
            assert work_out.dtype == torch.int16 and work_out.is_cuda and work_out.numel() == self.n and work_out.is_contiguous()
            tmp = nat.StepOut()
            C.memmove(C.byref(tmp), C.byref(out), C.sizeof(nat.StepOut))
            tmp.d_work = work_out.data_ptr()
            out = tmp
        nat.check(self.h, self.lib.npp_step(self.h, C.c_void_p(actions.data_ptr()), int(frame_skip), C.byref(out)))

    def step_many(self, actions, frame_skip=4):
        """actions: uint8 CUDA tensor [K, N].  K Gymnasium steps in one launch (open-loop sequences: checkpoint replay, fixed
        plans).  Returns (flags u8 [K, N], reward f32 [K, N], frames i16 [K, N]); observations of the last step land in
        self.game_state etc.; with auto-reset, envs that terminate mid-sequence restart on the spot."""
        assert actions.dtype == torch.uint8 and actions.is_cuda and actions.dim() == 2 and actions.shape[1] == self.n
        K = int(actions.shape[0])
        with self._ctx():   # allocations, the launch and the copies below are all ordered on the handle's stream
            actions = actions.contiguous()
            flags = torch.zeros((K, self.n), dtype=torch.uint8, device=self.device)
            reward = torch.zeros((K, self.n), dtype=torch.float32, device=self.device)
            frames = torch.zeros((K, self.n), dtype=torch.int16, device=self.device)
            out = nat.StepOut(self.game_state.data_ptr(), self.action_mask.data_ptr(), self.entity_pos.data_ptr(), flags.data_ptr(),
                              reward.data_ptr(), frames.data_ptr(), None,
                              self.spatial_context.data_ptr() if self.spatial_context is not None else None,
                              self.positions.data_ptr() if self.positions is not None else None, None)
            nat.check(self.h, self.lib.npp_step_many(self.h, C.c_void_p(actions.data_ptr()), K, int(frame_skip), C.byref(out)))
            self._keep = actions
            self.flags.copy_(flags[-1]); self.reward.copy_(reward[-1]); self.frames.copy_(frames[-1])
        return flags, reward, frames

    def tick(self, inputs):
        """inputs: uint8 CUDA tensor [T, N] of replay input bytes (bit0 jump, bit1 right, bit2 left)."""
        assert inputs.dtype == torch.uint8 and inputs.is_cuda and inputs.dim() == 2 and inputs.shape[1] == self.n
        inputs = inputs.contiguous()
        nat.check(self.h, self.lib.npp_tick(self.h, C.c_void_p(inputs.data_ptr()), int(inputs.shape[0])))
        self._keep = inputs

    def observe(self):
        nat.check(self.h, self.lib.npp_observe(self.h, C.byref(self._out_min)))

    def render_player_frame(self, out=None):
        """out: uint8 CUDA tensor [N, 84, 84] (or [N, 84, 84, 1]) filled with the player_frame of every env; default: the
        output block's player_frame field (enable_outputs("player_frame"))."""
        if out is None:
            out = self.out.t["player_frame"]
        assert out.dtype == torch.uint8 and out.is_cuda and out.numel() == self.n * 84 * 84 and out.is_contiguous()
        nat.check(self.h, self.lib.npp_render_player_frame(self.h, C.c_void_p(out.data_ptr())))

    def set_entity_pos(self, env, kind, x, y):
        """Move the exit switch (kind 0) or exit door (kind 1) of one env (curriculum repositioning); NaN clears."""
        nat.check(self.h, self.lib.npp_set_entity_pos(self.h, int(env), int(kind), float(x), float(y)))

    def switch_states(self, out=None):
        """float32 CUDA tensor [N, 25]: the reference's switch_states observation (5 locked doors x 5 features)."""
        if out is None:
            out = self.out.t.get("switch_states")
        if out is None:
            with self._ctx():
                out = torch.zeros((self.n, 25), dtype=torch.float32, device=self.device)
        assert out.dtype == torch.float32 and out.is_cuda and out.numel() == self.n * 25 and out.is_contiguous()
        nat.check(self.h, self.lib.npp_switch_states(self.h, C.c_void_p(out.data_ptr())))
        return out

    def reachability(self, with_switch_states=False):
        """Fill the block's reachability_features [N, 38] / mine_sdf_features [N, 3] / reach_status [N] (whichever are
        enabled) from the current state.  Call once per observation: the 38 floats follow the reference's cache rule
        (recomputed when the ninja's 24-px cell or exit_switch_activated changed since the previous call).
        with_switch_states: also fill the block's switch_states from the same launch (instead of a switch_states() call)."""
        t = self.out.t
        ptr = [C.c_void_p(t[k].data_ptr()) if k in t else None for k in ("reachability_features", "mine_sdf_features", "reach_status")]
        if ptr[0] is None and ptr[1] is None:
            raise RuntimeError('enable_outputs("reachability_features") and / or "mine_sdf_features" first')
        sw = C.c_void_p(t["switch_states"].data_ptr()) if with_switch_states else None
        nat.check(self.h, self.lib.npp_reachability_ex(self.h, *ptr, sw))

    def render_frame(self, env0=0, count=1):
        """uint8 CUDA tensor [count, 600, 1056, 1]: the whole gray frame (the reference's render() array) of some envs."""
        with self._ctx():
            out = torch.zeros((count, 600, 1056, 1), dtype=torch.uint8, device=self.device)
        nat.check(self.h, self.lib.npp_render_frame(self.h, int(env0), int(count), C.c_void_p(out.data_ptr())))
        return out

    def render_global_view(self, out=None):
        """out: uint8 CUDA tensor [N, 176, 100] (or [N, 176, 100, 1]): the reference's global_view of every env."""
        if out is None:
            out = self.out.t["global_view"]
        assert out.dtype == torch.uint8 and out.is_cuda and out.numel() == self.n * 176 * 100 and out.is_contiguous()
        nat.check(self.h, self.lib.npp_render_global_view(self.h, C.c_void_p(out.data_ptr())))

    def entity_checksum(self, env0=0, count=None):
        """[count, 6] f64: per-env sums over all entities in entity_dic order (see npp_entity_checksum)."""
        count = self.n - env0 if count is None else count
        o = np.zeros((count, 6), dtype=np.float64)
        nat.check(self.h, self.lib.npp_entity_checksum(self.h, env0, count, o.ctypes.data_as(C.POINTER(C.c_double))))
        return o

    def snapshot(self):
        """Checkpoint the state of every env on the device (one slot)."""
        nat.check(self.h, self.lib.npp_snapshot(self.h))

    def restore(self, mask=None):
        """Put the checkpointed state back (for the masked envs; None = all)."""
        if mask is None:
            nat.check(self.h, self.lib.npp_restore(self.h, None))
        else:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert len(m) == self.n
            nat.check(self.h, self.lib.npp_restore(self.h, m.ctypes.data_as(C.POINTER(C.c_uint8))))

    def to_host(self, names=None):
        """{name: numpy array} of the enabled outputs through ONE async device-to-host copy of the output block into pinned
        memory on the handle's stream + one synchronisation.  The arrays are views of a pinned staging block; two blocks
        alternate, so they stay valid until the to_host() call after the next one."""
        self.join()
        return self.out.to_host(self.stream, names)

    def sync(self):
        nat.check(self.h, self.lib.npp_sync(self.h))

    # ---- parity hooks -----------------------------------------------------------------------------------------
    def dump_state(self, env0=0, count=None):
        count = self.n - env0 if count is None else count
        f = np.zeros((count, nat.DUMP_F64), dtype=np.float64)
        i = np.zeros((count, nat.DUMP_I32), dtype=np.int32)
        nat.check(self.h, self.lib.npp_dump_state(
            self.h, env0, count, f.ctypes.data_as(C.POINTER(C.c_double)), i.ctypes.data_as(C.POINTER(C.c_int32))))
        return f, i

    def dump_entities(self, env):
        buf = np.zeros(4096, dtype=np.int32)
        n = C.c_int(0)
        nat.check(self.h, self.lib.npp_dump_entities(self.h, env, buf.ctypes.data_as(C.POINTER(C.c_int32)), len(buf), C.byref(n)))
        return buf[: n.value].copy()

    def dump_level_segments(self, level):
        buf = np.zeros((16384, 8), dtype=np.int16)
        n = C.c_int(0)
        nat.check(self.h, self.lib.npp_dump_level_segments(self.h, level, buf.ctypes.data_as(C.POINTER(C.c_int16)), len(buf), C.byref(n)))
        return buf[: n.value].copy()


def compile_level_segments(map_data):
    """Host-only: run the native level compiler, return (rows int16 [n,8], unsupported_mask)."""
    L = nat.lib()
    m = np.ascontiguousarray(np.asarray(map_data, dtype=np.float64))
    buf = np.zeros((16384, 8), dtype=np.int16)
    n = C.c_int(0)
    uns = C.c_uint32(0)
    nat.check(None, L.npp_compile_level_segments(
        m.ctypes.data_as(C.POINTER(C.c_double)), len(m), buf.ctypes.data_as(C.POINTER(C.c_int16)), len(buf), C.byref(n), C.byref(uns)))
    return buf[: n.value].copy(), uns.value


def compile_level_entities(map_data):
    """Host-only: rows of (kind, x, y, cell x, cell y, init) in map order."""
    L = nat.lib()
    m = np.ascontiguousarray(np.asarray(map_data, dtype=np.float64))
    buf = np.zeros((4096, 6), dtype=np.float64)
    n = C.c_int(0)
    nat.check(None, L.npp_compile_level_entities(
        m.ctypes.data_as(C.POINTER(C.c_double)), len(m), buf.ctypes.data_as(C.POINTER(C.c_double)), len(buf), C.byref(n)))
    return buf[: n.value].copy()


def reach_level_info(map_data):
    """Host-only: what the reachability table builder makes of one level: {"supported": the level is inside the restated part
    of the reference's reachability code, "nodes": len(adjacency), "mines": toggle mines, "surface_area": nodes reachable
    from the spawn (feature scale)}."""
    L = nat.lib()
    m = np.ascontiguousarray(np.asarray(map_data, dtype=np.float64))
    info = np.zeros(16, dtype=np.int32)
    nat.check(None, L.npp_reach_compile(m.ctypes.data_as(C.POINTER(C.c_double)), len(m), info.ctypes.data_as(C.c_void_p),
                                        *([None] * 11)))
    return {"supported": bool(info[0]), "nodes": int(info[1]), "mines": int(info[11]), "surface_area": int(info[13])}


def level_truncation_limit(map_data):
    """Host-only: (dynamic truncation limit in frames, reachable surface area in graph nodes) of one level."""
    L = nat.lib()
    m = np.ascontiguousarray(np.asarray(map_data, dtype=np.float64))
    lim, area = C.c_int32(0), C.c_int32(0)
    nat.check(None, L.npp_level_truncation_limit(m.ctypes.data_as(C.POINTER(C.c_double)), len(m), C.byref(lim), C.byref(area)))
    return lim.value, area.value


def compile_level_zoo(map_data):
    """Host-only: (hor [89, 51], ver [89, 51] grid-edge counters at load, movers [n, 4] = type, x, y, creation order)."""
    lib = nat.lib()
    m = np.ascontiguousarray(np.asarray(map_data, dtype=np.float64))
    edges = np.zeros(2 * 89 * 51, dtype=np.int32)
    mov = np.zeros((1024, 4), dtype=np.float64)
    n = C.c_int(0)
    nat.check(None, lib.npp_compile_level_zoo(m.ctypes.data_as(C.POINTER(C.c_double)), len(m), edges.ctypes.data_as(C.POINTER(C.c_int32)),
                                              mov.ctypes.data_as(C.POINTER(C.c_double)), len(mov), C.byref(n)))
    return edges[: 89 * 51].reshape(89, 51), edges[89 * 51 :].reshape(89, 51), mov[: n.value].copy()
