"""NppBatch: the thin host object over one native handle (N environments on one GPU).

PyTorch is used only as plumbing: device buffers for actions/observations and the HIP stream.  All
simulation work happens in the HIP kernels behind the C ABI (include/npp_amd.h).
"""
import ctypes as C

import numpy as np
import torch

from . import _native as nat


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


class NppBatch:
    """N environments stepped in lock-step on one GPU.

    Counterpart of N instances of the reference's NPlayHeadless (nclone/nplay_headless.py:28).
    """

    def __init__(self, n_envs, device=0, autoreset=True, allow_unsupported=False, frame_centered=False, stream=None):
        self.lib = nat.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("nclone_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.n = int(n_envs)
        self.device = torch.device("cuda", int(device))
        flags = ((nat.FLAG_AUTORESET if autoreset else 0) | (nat.FLAG_ALLOW_UNSUPPORTED if allow_unsupported else 0)
                 | (nat.FLAG_FRAME_CENTERED if frame_centered else 0))
        h = C.c_void_p()
        nat.check(None, self.lib.npp_create(self.n, int(device), flags, C.byref(h)))
        self.h = h
        self.n_levels = 0
        with torch.cuda.device(self.device):
            # every launch of this handle is ordered on ONE HIP stream; handles on different streams overlap on the GPU
            self.stream = stream if stream is not None else torch.cuda.current_stream()
            nat.check(self.h, self.lib.npp_set_stream(self.h, C.c_void_p(self.stream.cuda_stream)))
            N = self.n
            self.game_state = torch.zeros((N, 41), dtype=torch.float32, device=self.device)
            self.action_mask = torch.zeros((N, 6), dtype=torch.int8, device=self.device)
            self.entity_pos = torch.zeros((N, 6), dtype=torch.float32, device=self.device)
            self.flags = torch.zeros((N,), dtype=torch.uint8, device=self.device)
            self.reward = torch.zeros((N,), dtype=torch.float32, device=self.device)
            self.frames = torch.zeros((N,), dtype=torch.int16, device=self.device)
            self.terminal_state = torch.zeros((N, 41), dtype=torch.float32, device=self.device)
            self.spatial_context = None   # allocated by enable_spatial_context()
        self._out = nat.StepOut(
            self.game_state.data_ptr(), self.action_mask.data_ptr(), self.entity_pos.data_ptr(), self.flags.data_ptr(),
            self.reward.data_ptr(), self.frames.data_ptr(), self.terminal_state.data_ptr(), None,
        )
        self._out_min = nat.StepOut(self.game_state.data_ptr(), self.action_mask.data_ptr(), self.entity_pos.data_ptr(),
                                    self.flags.data_ptr(), self.reward.data_ptr(), self.frames.data_ptr(), None, None)

    def enable_spatial_context(self):
        """Also produce the 112-float spatial_context observation (8x8 tile categories + 8 nearest mines)."""
        if self.spatial_context is None:
            self.spatial_context = torch.zeros((self.n, 112), dtype=torch.float32, device=self.device)
            self._out.d_spatial_context = self.spatial_context.data_ptr()
            self._out_min.d_spatial_context = self.spatial_context.data_ptr()

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

    def set_launch_geometry(self, lanes_per_env=0, waves_per_block=0):
        nat.check(self.h, self.lib.npp_set_launch_geometry(self.h, int(lanes_per_env), int(waves_per_block)))

    def launch_geometry(self):
        g, w = C.c_int(0), C.c_int(0)
        nat.check(self.h, self.lib.npp_get_launch_geometry(self.h, C.byref(g), C.byref(w)))
        return g.value, w.value

    # ---- stepping ---------------------------------------------------------------------------------------------
    def reset(self, mask=None):
        if mask is None:
            nat.check(self.h, self.lib.npp_reset(self.h, None))
        else:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert len(m) == self.n
            nat.check(self.h, self.lib.npp_reset(self.h, m.ctypes.data_as(C.POINTER(C.c_uint8))))

    def step(self, actions, frame_skip=4, want_terminal=True):
        """actions: uint8 CUDA tensor [N] with values 0..5.  Asynchronous; outputs land in self.game_state etc."""
        assert actions.dtype == torch.uint8 and actions.is_cuda and actions.numel() == self.n
        out = self._out if want_terminal else self._out_min
        nat.check(self.h, self.lib.npp_step(self.h, C.c_void_p(actions.data_ptr()), int(frame_skip), C.byref(out)))

    def step_many(self, actions, frame_skip=4):
        """actions: uint8 CUDA tensor [K, N].  K Gymnasium steps in one launch (open-loop sequences: checkpoint replay, fixed
        plans).  Returns (flags u8 [K, N], reward f32 [K, N], frames i16 [K, N]); observations of the last step land in
        self.game_state etc.; with auto-reset, envs that terminate mid-sequence restart on the spot."""
        assert actions.dtype == torch.uint8 and actions.is_cuda and actions.dim() == 2 and actions.shape[1] == self.n
        actions = actions.contiguous()
        K = int(actions.shape[0])
        flags = torch.zeros((K, self.n), dtype=torch.uint8, device=self.device)
        reward = torch.zeros((K, self.n), dtype=torch.float32, device=self.device)
        frames = torch.zeros((K, self.n), dtype=torch.int16, device=self.device)
        out = nat.StepOut(self.game_state.data_ptr(), self.action_mask.data_ptr(), self.entity_pos.data_ptr(), flags.data_ptr(),
                          reward.data_ptr(), frames.data_ptr(), None,
                          self.spatial_context.data_ptr() if self.spatial_context is not None else None)
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

    def render_player_frame(self, out):
        """out: uint8 CUDA tensor [N, 84, 84] (or [N, 84, 84, 1]) filled with the player_frame of every env."""
        assert out.dtype == torch.uint8 and out.is_cuda and out.numel() == self.n * 84 * 84 and out.is_contiguous()
        nat.check(self.h, self.lib.npp_render_player_frame(self.h, C.c_void_p(out.data_ptr())))

    def set_entity_pos(self, env, kind, x, y):
        """Move the exit switch (kind 0) or exit door (kind 1) of one env (curriculum repositioning); NaN clears."""
        nat.check(self.h, self.lib.npp_set_entity_pos(self.h, int(env), int(kind), float(x), float(y)))

    def switch_states(self, out=None):
        """float32 CUDA tensor [N, 25]: the reference's switch_states observation (5 locked doors x 5 features)."""
        if out is None:
            out = torch.zeros((self.n, 25), dtype=torch.float32, device=self.device)
        assert out.dtype == torch.float32 and out.is_cuda and out.numel() == self.n * 25 and out.is_contiguous()
        nat.check(self.h, self.lib.npp_switch_states(self.h, C.c_void_p(out.data_ptr())))
        return out

    def render_frame(self, env0=0, count=1):
        """uint8 CUDA tensor [count, 600, 1056, 1]: the whole gray frame (the reference's render() array) of some envs."""
        out = torch.zeros((count, 600, 1056, 1), dtype=torch.uint8, device=self.device)
        nat.check(self.h, self.lib.npp_render_frame(self.h, int(env0), int(count), C.c_void_p(out.data_ptr())))
        return out

    def render_global_view(self, out):
        """out: uint8 CUDA tensor [N, 176, 100] (or [N, 176, 100, 1]): the reference's global_view of every env."""
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
