"""ctypes binding of libnpp_amd.so (the C ABI in include/npp_amd.h).

There is NO CPU fallback: if the library is missing this module raises, and if no GPU is present
`npp_create` fails with a clear message.  The oracle under oracle/ is test infrastructure and is never
imported from here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NPP_AMD_LIB selects another build of the same library (tools/ab_bench.py-style A/B runs of kernel variants)
LIB_PATH = os.environ.get("NPP_AMD_LIB") or os.path.join(_HERE, "libnpp_amd.so")

NPP_OK = 0
NPP_ERR_INVALID, NPP_ERR_HIP, NPP_ERR_UNSUPPORTED, NPP_ERR_STATE = 1, 2, 3, 4
FLAG_AUTORESET = 1
FLAG_ALLOW_UNSUPPORTED = 2
FLAG_FRAME_CENTERED = 4
FLAG_FAST_RESET = 8
F_WON, F_DEAD, F_SWITCH, F_TRUNCATED, F_CAUSE_MINE, F_CAUSE_IMPACT = 1, 2, 4, 8, 16, 32
GAME_STATE_DIM = 41
DUMP_F64 = 12
DUMP_I32 = 32

EXPORTS = [
    "npp_create", "npp_destroy", "npp_last_error", "npp_set_stream", "npp_sync", "npp_load_levels",
    "npp_assign_levels", "npp_reset", "npp_set_truncation_limit", "npp_step", "npp_tick", "npp_observe",
    "npp_render_player_frame", "npp_dump_state", "npp_dump_entities", "npp_dump_level_segments",
    "npp_compile_level_segments", "npp_compile_level_entities", "npp_set_step_variant", "npp_get_step_variant", "npp_num_envs", "npp_num_levels",
    "npp_set_launch_geometry", "npp_get_launch_geometry", "npp_snapshot", "npp_restore", "npp_entity_checksum", "npp_compile_level_zoo", "npp_render_global_view", "npp_switch_states", "npp_set_entity_pos", "npp_step_many", "npp_render_frame", "npp_plan_zoo_block", "npp_reset_ex",
    "npp_reachability", "npp_reachability_ex", "npp_reach_compile", "npp_reach_features_host", "npp_reach_compile_miss", "npp_reach_rollout_host", "npp_set_dynamic_truncation", "npp_level_truncation_limit",
    "npp_set_obs_overlap", "npp_set_obs_overlap_parts", "npp_join",
]


class StepOut(C.Structure):
    _fields_ = [
        ("d_game_state", C.c_void_p),
        ("d_action_mask", C.c_void_p),
        ("d_entity_pos", C.c_void_p),
        ("d_flags", C.c_void_p),
        ("d_reward", C.c_void_p),
        ("d_frames", C.c_void_p),
        ("d_terminal_state", C.c_void_p),
        ("d_spatial_context", C.c_void_p),
        ("d_positions", C.c_void_p),
        ("d_work", C.c_void_p),
    ]


class NativeLibraryMissing(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the native library; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so) and publishes its symbols globally; it must be
    # loaded BEFORE this library so that both share ONE runtime (streams and device pointers are exchanged).
    import torch  # noqa: F401

    if not os.path.isfile(LIB_PATH):
        raise NativeLibraryMissing(
            "nclone_amd: %s not found. Build it with `python -m nclone_amd.build_native` (needs hipcc). "
            "There is no CPU fallback for the accelerated path." % LIB_PATH
        )
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    L.npp_create.argtypes = [C.c_int, C.c_int, C.c_uint, C.POINTER(H)]
    L.npp_destroy.argtypes = [H]
    L.npp_last_error.argtypes = [H]
    L.npp_last_error.restype = C.c_char_p
    L.npp_set_stream.argtypes = [H, C.c_void_p]
    L.npp_sync.argtypes = [H]
    L.npp_load_levels.argtypes = [H, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    L.npp_assign_levels.argtypes = [H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]
    L.npp_reset.argtypes = [H, C.POINTER(C.c_uint8)]
    L.npp_reset_ex.argtypes = [H, C.POINTER(C.c_uint8), C.c_int]
    L.npp_set_truncation_limit.argtypes = [H, C.POINTER(C.c_int32), C.c_int32]
    L.npp_step.argtypes = [H, C.c_void_p, C.c_int, C.POINTER(StepOut)]
    L.npp_tick.argtypes = [H, C.c_void_p, C.c_int]
    L.npp_observe.argtypes = [H, C.POINTER(StepOut)]
    L.npp_render_player_frame.argtypes = [H, C.c_void_p]
    L.npp_dump_state.argtypes = [H, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.npp_dump_entities.argtypes = [H, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)]
    L.npp_dump_level_segments.argtypes = [H, C.c_int, C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int)]
    L.npp_compile_level_segments.argtypes = [
        C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint32)
    ]
    L.npp_compile_level_entities.argtypes = [C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    L.npp_set_launch_geometry.argtypes = [H, C.c_int, C.c_int]
    L.npp_set_step_variant.argtypes = [H, C.c_int]
    L.npp_set_obs_overlap.argtypes = [H, C.c_int]
    L.npp_join.argtypes = [H]
    L.npp_set_obs_overlap_parts.argtypes = [H, C.POINTER(C.c_int), C.c_int]
    L.npp_get_step_variant.argtypes = [H, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.npp_get_launch_geometry.argtypes = [H, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.npp_entity_checksum.argtypes = [H, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.npp_compile_level_zoo.argtypes = [C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    L.npp_render_global_view.argtypes = [H, C.c_void_p]
    L.npp_switch_states.argtypes = [H, C.c_void_p]
    L.npp_set_entity_pos.argtypes = [H, C.c_int, C.c_int, C.c_double, C.c_double]
    L.npp_step_many.argtypes = [H, C.c_void_p, C.c_int, C.c_int, C.POINTER(StepOut)]
    L.npp_render_frame.argtypes = [H, C.c_int, C.c_int, C.c_void_p]
    L.npp_plan_zoo_block.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.npp_reachability.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    L.npp_reachability_ex.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.npp_reach_compile.argtypes = [C.POINTER(C.c_double), C.c_int64] + [C.c_void_p] * 12
    L.npp_reach_features_host.argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p]
    L.npp_reach_compile_miss.argtypes = [C.POINTER(C.c_double), C.c_int64] + [C.c_void_p] * 4
    L.npp_reach_rollout_host.argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.npp_set_dynamic_truncation.argtypes = [H, C.c_int]
    L.npp_level_truncation_limit.argtypes = [C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.npp_snapshot.argtypes = [H]
    L.npp_restore.argtypes = [H, C.POINTER(C.c_uint8)]
    L.npp_num_envs.argtypes = [H]
    L.npp_num_levels.argtypes = [H]
    for name in EXPORTS:
        if name != "npp_last_error":
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


class NppError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("npp error %d: %s" % (code, msg))
        self.code = code


def check(handle, code):
    if code != NPP_OK:
        msg = lib().npp_last_error(handle)
        raise NppError(code, msg.decode() if msg else "?")
