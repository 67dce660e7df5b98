"""Gymnasium-shaped host classes over the native stepper.

  NppVecEnvironment  N environments per object; the same observation keys as the reference's NppEnvironment with a
                     leading batch dimension (nclone/gym_environment/npp_environment.py:504 reset,
                     base_environment.py:483 step).
  NppEnvironment     one environment, literal drop-in signatures: reset() -> (obs, info),
                     step(int) -> (obs, float, bool, bool, dict).
  NPlayHeadless      the facade the reference's tools drive (nclone/nplay_headless.py:28): load_map_from_map_data,
                     reset, tick(h, j), ninja_has_won/died, ninja_position, get_ninja_state, ...

Everything executes in the HIP kernels; nothing here falls back to the CPU.
"""
import numpy as np
import torch

from . import _native as nat
from . import spaces
from .engine import NppBatch

ACTION_TABLE = [(0, 0), (-1, 0), (1, 0), (0, 1), (-1, 1), (1, 1)]  # base_environment.py:366-402
DEATH_CAUSES = {0: None, 1: "mine", 2: "terminal_impact", 3: None}   # 3: drone / thwump / death ball / crush: kill() without a cause
MAX_TIME_IN_FRAMES = 10000   # gym_environment/constants.py:8-10


def calculate_truncation_limit(surface_area, reachable_mine_count=0):
    """Dynamic truncation limit of the reference (gym_environment/truncation_calculator.py:19-57):
    clip((sqrt(surface_area) * 20 + mines * 75) * 25, 1200, MAX_TIME_IN_FRAMES).  `surface_area` = reachable graph nodes
    (engine.level_truncation_limit computes both for a level).  The native path applies it per level by itself:
    NppVecEnvironment(truncation_limit="dynamic"), the default, = npp_set_dynamic_truncation."""
    v = (np.sqrt(surface_area) * 20.0 + reachable_mine_count * 75.0) * 25
    return int(np.clip(v, 1200, MAX_TIME_IN_FRAMES))


def controls_to_input_byte(hor, jump):
    """(hor, jump) -> replay input byte (bit0 jump, bit1 right, bit2 left; replay/replay_executor.py:61-84)."""
    return (1 if jump else 0) | (2 if hor > 0 else 0) | (4 if hor < 0 else 0)


class NppVecEnvironment:
    """N N++ environments stepped in lock-step on one MI355X.

    levels      sequence of raw map_data arrays (ints/floats/bytes)
    level_ids   which level each env plays (default: env i plays level (i // 64) % n_levels, so every 64-env block
                shares a level and the kernel stages it in LDS)
    output      "torch" (CUDA tensors, zero copies) or "numpy" (host copies; drop-in for numpy training loops)
    truncation_limit  "dynamic" (default; the reference env's per-level limit from the reachable surface area,
                npp_environment.py:1238-1256) or a number of frames for every env (10000 = the reference's fallback)
    The step's `reward` carries only the sparse terminal constants -- reward parity: NONE (the reference's PBRS reward
    calculator is out of scope); compute the reward from the observations / info flags.
    """

    metadata = {"render_modes": []}

    def __init__(self, levels, num_envs, level_ids=None, frame_skip=4, device=0, enable_visual_observations=False,
                 truncation_limit="dynamic", output="torch", autoreset=True, enable_spatial_context=False,
                 enable_switch_states=False, fast_reset=True, stream=None, enable_reachability=False, obs_overlap=0):
        assert output in ("torch", "numpy")
        self.num_envs = int(num_envs)
        self.frame_skip = int(frame_skip)
        self.output = output
        self.enable_visual_observations = bool(enable_visual_observations)
        self.single_action_space = spaces.action_space()
        self.single_observation_space = spaces.observation_space(self.enable_visual_observations,
                                                                 spatial_context=bool(enable_spatial_context),
                                                                 switch_states=bool(enable_switch_states),
                                                                 reachability=bool(enable_reachability))
        self.action_space = self.single_action_space
        self.observation_space = self.single_observation_space
        outputs = ["positions"]
        if enable_spatial_context:
            outputs.append("spatial_context")
        if enable_switch_states:
            outputs.append("switch_states")
        if self.enable_visual_observations:
            outputs += ["player_frame", "global_view"]
        if enable_reachability:   # reachability_features + mine_sdf_features (npp_environment.py observation keys)
            outputs += ["reachability_features", "mine_sdf_features"]
        # same-level resets are Simulator.fast_reset in the reference's env (npp_environment.py:541-557): the default here
        self._b = NppBatch(self.num_envs, device=device, autoreset=autoreset, outputs=outputs, fast_reset=fast_reset,
                           stream=stream)
        self._b.load_levels(levels)
        if level_ids is None:
            level_ids = (np.arange(self.num_envs) // 64) % len(levels)
        self._b.assign_levels(level_ids)
        if isinstance(truncation_limit, str):
            if truncation_limit != "dynamic":
                raise ValueError('truncation_limit: a number of frames or "dynamic"')
            self._b.set_dynamic_truncation(True)
        else:
            self._b.set_truncation_limit(truncation_limit)
        if obs_overlap:   # speed knob (same bits): observation kernels of the cheap envs beside the step's expensive tail
            self._b.set_obs_overlap(int(obs_overlap))
        self._rng = np.random.default_rng()
        with self._b._ctx():
            self._actions = torch.zeros(self.num_envs, dtype=torch.uint8, device=self._b.device)
        self._reset_bits = 11 if autoreset else 0
        self._obs_names = ["game_state", "action_mask", "entity_pos", "positions", "flags"] + outputs[1:]

    # -- helpers ------------------------------------------------------------------------------------------------
    def _produce(self):
        """Launch the secondary observation kernels (frames, switch_states) on the handle's stream."""
        b = self._b
        fused = "switch_states" in b.out.t and "reachability_features" in b.out.t   # one launch writes both
        if "switch_states" in b.out.t and not fused:
            b.switch_states()
        if "player_frame" in b.out.t:
            b.render_player_frame()
            b.render_global_view()
        if "reachability_features" in b.out.t:
            b.reachability(with_switch_states=fused)
        b.join()   # (obs_overlap) the handle's stream waits for the kernels that went to the second stream

    def _obs(self, src):
        """src: {name: tensor-or-array} (device tensors, or the host views of ONE staged copy)."""
        pos = src["positions"]
        obs = {
            "game_state": src["game_state"],
            "action_mask": src["action_mask"],
            "entity_positions": src["entity_pos"],
            # pass-through scalars of the raw observation (observation_processor.py:374-399), unrounded fp64
            "player_x": pos[:, 0], "player_y": pos[:, 1], "switch_x": pos[:, 2], "switch_y": pos[:, 3],
            "exit_door_x": pos[:, 4], "exit_door_y": pos[:, 5],
            # flags describe the step that just ran; an env that was auto-reset (terminated / truncated) returns the spawn
            # observation, whose switch is not yet hit
            "switch_activated": ((src["flags"] & 4) != 0) & ((src["flags"] & self._reset_bits) == 0),
        }
        for k in ("spatial_context", "switch_states", "player_frame", "global_view", "reachability_features", "mine_sdf_features"):
            if k in src:
                obs[k] = src[k]
        return obs

    # -- Gymnasium surface ----------------------------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        """Reset every env to its level's spawn state (npp_environment.py:504).  The first reset of a level assignment is
        Simulator.reset (nsim.py:62); later ones are Simulator.fast_reset (nsim.py:78) unless fast_reset=False.

        seed     seeds `action_space_sample()` only: the reference's seed picks the next map (env_map_loader.py), here the
                 levels are assigned explicitly, and the simulation itself has no randomness.
        options  {"checkpoint": c} restores a Go-Explore checkpoint (base_environment.py:1769-1789, _reset_to_checkpoint):
                 c = "snapshot" puts back the state saved by `snapshot()` (a device-side copy: what the replay below would
                 reproduce, bit for bit); otherwise c (or c["action_sequence"] / c.action_sequence) is the action sequence to
                 replay from the spawn -- one sequence for every env, or an [N, K] array -- frame_skip ticks per action like
                 ActionReplayer.replay_to_checkpoint; c.source_frame_skip / c["source_frame_skip"] overrides the tick count.
                 Other keys of the reference (skip_map_load, new_level, map_name) concern its map loader and are ignored."""
        if seed is not None:
            self._rng = np.random.default_rng(seed)
        ckpt = (options or {}).get("checkpoint")
        info = {}
        if isinstance(ckpt, str):
            if ckpt != "snapshot":
                raise ValueError('options["checkpoint"]: "snapshot", an action sequence, or an object with .action_sequence')
            self._b.restore()
            info = {"checkpoint_replay": False, "restored_snapshot": True}
        else:
            self._b.reset()
            if ckpt is not None:
                seq = ckpt.get("action_sequence") if isinstance(ckpt, dict) else getattr(ckpt, "action_sequence", ckpt)
                fs = ckpt.get("source_frame_skip") if isinstance(ckpt, dict) else getattr(ckpt, "source_frame_skip", None)
                seq = np.asarray(seq if seq is not None else [], dtype=np.uint8)
                if seq.ndim == 1:
                    seq = np.broadcast_to(seq[None, :], (self.num_envs, len(seq)))
                if seq.ndim != 2 or seq.shape[0] != self.num_envs:
                    raise ValueError("checkpoint action_sequence: [K] or [num_envs, K]")
                K = int(seq.shape[1])
                if K:
                    fs = self.frame_skip if fs is None else int(fs)
                    with self._b._ctx():
                        acts = torch.from_numpy(np.ascontiguousarray(seq.T)).to(self._b.device)
                    flags, _rew, frames = self._b.step_many(acts, fs)
                    info = {"checkpoint_replay": True, "replay_frames": frames.to(torch.int64).sum(dim=0),
                            "replay_terminated": (flags & 3).ne(0).any(dim=0)}
                else:
                    info = {"checkpoint_replay": False, "replay_frames": 0}
        self._b.observe()
        self._produce()
        if self.output == "torch":
            return self._obs(self._b.out.t), info
        return self._obs(self._b.to_host(self._obs_names)), info

    def snapshot(self):
        """Save the state of every env on the device (one slot); reset(options={"checkpoint": "snapshot"}) restores it."""
        self._b.snapshot()

    def action_space_sample(self):
        """uint8 [N] uniform actions from the generator reset(seed=...) seeds."""
        return self._rng.integers(0, 6, size=self.num_envs).astype(np.uint8)

    def step_async(self, actions):
        """Enqueue the step (action upload, step kernel, observation kernels) on the handle's stream; returns at once."""
        b = self._b
        with b._ctx():
            if isinstance(actions, torch.Tensor):
                self._actions.copy_(actions.to(torch.uint8), non_blocking=True)
            else:
                self._actions.copy_(torch.as_tensor(np.asarray(actions, dtype=np.uint8)), non_blocking=True)
        b.step(self._actions, self.frame_skip, want_terminal=True)
        self._produce()

    def step_wait(self):
        """Results of the step enqueued by step_async: (obs, reward, terminated, truncated, info) with batch dims.
        output="torch": device tensors, no synchronisation (consume them on the handle's stream);
        output="numpy": ONE async copy of the output block into pinned memory + one synchronisation."""
        b = self._b
        if self.output == "torch":
            src = b.out.t
        else:
            src = b.to_host(self._obs_names + ["reward", "frames", "terminal_state"])
        flags = src["flags"]
        info = {
            "player_won": (flags & 1) != 0,
            "player_dead": (flags & 2) != 0,
            "switch_activated": (flags & 4) != 0,
            "death_cause_code": (flags >> 4) & 3,
            "frames_executed": src["frames"],
            "terminal_observation": src["terminal_state"],
            "frame_skip_stats": {"skip_value": self.frame_skip},
        }
        return self._obs(src), src["reward"], (flags & 3) != 0, (flags & 8) != 0, info

    def step(self, actions):
        """actions: int array/tensor [N] in 0..5.  Returns (obs, reward, terminated, truncated, info) with batch dims."""
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self._b.close()

    @property
    def batch(self):
        return self._b


class NppEnvironment:
    """Single-environment adapter with the reference's exact call signatures (base_environment.py:483,
    npp_environment.py:504).  One GPU lane group does the work of one Python simulator; use NppVecEnvironment for
    throughput."""

    def __init__(self, map_data=None, custom_map_path=None, frame_skip=4, device=0, enable_visual_observations=False,
                 truncation_limit="dynamic", fast_reset=True, enable_spatial_context=False, enable_switch_states=False,
                 enable_reachability=False):
        if map_data is None:
            if custom_map_path is None:
                raise ValueError("NppEnvironment needs map_data or custom_map_path")
            with open(custom_map_path, "rb") as f:
                map_data = np.frombuffer(f.read(), dtype=np.uint8)
        self._v = NppVecEnvironment([map_data], 1, level_ids=[0], frame_skip=frame_skip, device=device,
                                    enable_visual_observations=enable_visual_observations,
                                    truncation_limit=truncation_limit, output="numpy", autoreset=False,
                                    fast_reset=fast_reset, enable_spatial_context=enable_spatial_context,
                                    enable_switch_states=enable_switch_states, enable_reachability=enable_reachability)
        self.action_space = self._v.single_action_space
        self.observation_space = self._v.single_observation_space
        self.frame_skip = frame_skip

    @staticmethod
    def _unbatch(obs):
        out = {k: v[0].copy() if isinstance(v[0], np.ndarray) else v[0] for k, v in obs.items()}
        for k in ("player_x", "player_y", "switch_x", "switch_y", "exit_door_x", "exit_door_y"):
            out[k] = float(out[k])
        out["switch_activated"] = bool(out["switch_activated"])
        return out

    def reset(self, seed=None, options=None):
        obs, info = self._v.reset(seed=seed, options=options)
        return self._unbatch(obs), info

    def step(self, action):
        obs, rew, term, trunc, info = self._v.step(np.array([int(action)], dtype=np.uint8))
        cause = DEATH_CAUSES[int(info["death_cause_code"][0])]
        out_info = {
            "player_won": bool(info["player_won"][0]),
            "player_dead": bool(info["player_dead"][0]),
            "switch_activated": bool(info["switch_activated"][0]),
            "death_cause": cause,
            "frame_skip_stats": {"skip_value": self.frame_skip, "frames_executed": int(info["frames_executed"][0])},
        }
        return self._unbatch(obs), float(rew[0]), bool(term[0]), bool(trunc[0]), out_info

    def close(self):
        self._v.close()


class _SimView:
    """`.sim.frame` of the facade."""

    def __init__(self, owner):
        self._o = owner

    @property
    def frame(self):
        return int(self._o._state()[1][0, 22])


class NPlayHeadless:
    """The reference's headless facade for one simulator, backed by the GPU stepper (nplay_headless.py:28).

    Method names, argument meaning and return conventions follow the reference so that harnesses such as
    tools/test_replay_playback.py read the same."""

    def __init__(self, device=0, enable_rendering=False, **_ignored):
        self._device = device
        self._b = None
        self.sim = _SimView(self)
        self.current_map_data = None

    def load_map_from_map_data(self, map_data):
        if self._b is not None:
            self._b.close()
        self._b = NppBatch(1, device=self._device, autoreset=False, outputs=("positions",))
        self._b.load_levels([map_data])
        with self._b._ctx():
            self._in = torch.zeros((1, 1), dtype=torch.uint8, device=self._b.device)
        self.current_map_data = map_data
        self._cache = {}

    def load_map(self, map_path):
        with open(map_path, "rb") as f:
            self.load_map_from_map_data(np.frombuffer(f.read(), dtype=np.uint8))

    def reset(self):
        """Simulator.reset (nsim.py:62-76): entities re-created."""
        self._b.reset(mode="full")
        self._cache = {}

    def fast_reset(self):
        """Simulator.fast_reset (nsim.py:78-140): entities reset in place (key-ordered cell lists, movers keep going)."""
        self._b.reset(mode="fast")
        self._cache = {}

    def tick(self, horizontal_input, jump_input):
        with self._b._ctx():
            self._in.fill_(controls_to_input_byte(horizontal_input, jump_input))
        self._b.tick(self._in)
        self._cache = {}

    # accessors share ONE state dump / ONE observation per tick (they used to cost a device round trip each)
    def _state(self):
        if "state" not in self._cache:
            self._cache["state"] = self._b.dump_state(0, 1)
        return self._cache["state"]

    def _observed(self):
        if "obs" not in self._cache:
            self._b.observe()
            self._cache["obs"] = self._b.to_host(["game_state", "action_mask", "entity_pos", "positions"])
        return self._cache["obs"]

    def ninja_has_won(self):
        return int(self._state()[1][0, 0]) == 8

    def ninja_has_died(self):
        return int(self._state()[1][0, 0]) in (6, 7)

    def ninja_death_cause(self):
        return DEATH_CAUSES[int(self._state()[1][0, 20])]

    def ninja_position(self):
        f = self._state()[0][0]
        return float(f[0]), float(f[1])

    def ninja_velocity(self):
        f = self._state()[0][0]
        return float(f[2]), float(f[3])

    def ninja_velocity_old(self):
        f = self._state()[0][0]
        return float(f[8]), float(f[9])

    def get_ninja_terminal_impact(self):
        return bool(self._state()[1][0, 21])

    def get_ninja_state(self):
        return [float(v) for v in self._observed()["game_state"][0, :40]]

    def get_action_mask(self):
        return [bool(v) for v in self._observed()["action_mask"][0]]

    def exit_switch_activated(self):
        return int(self._state()[1][0, 13]) != 1

    def exit_switch_position(self):
        p = self._observed()["positions"][0]
        return float(p[2]), float(p[3])

    def exit_door_position(self):
        p = self._observed()["positions"][0]
        return float(p[4]), float(p[5])

    def render(self):
        """The gray frame, numpy uint8 (600, 1056, 1), as the reference's render() returns it (nplay_headless.py:144-156)."""
        return self._b.render_frame(0, 1)[0].cpu().numpy()

    def _entities(self):
        """(compiled rows [kind, x, y, cx, cy, init], live 2-bit states) of the loaded level, map order."""
        from .engine import compile_level_entities

        return compile_level_entities(self.current_map_data), self._b.dump_entities(0)

    def locked_doors(self):
        """Locked-door entities (entity_dic[6]) as simple records: the entity sits at its switch (xpos, ypos == sw_xpos,
        sw_ypos), `active` until the switch is collected, `closed` likewise (nplay_headless.py:714-716)."""
        from types import SimpleNamespace

        rows, st = self._entities()
        return [SimpleNamespace(type=6, xpos=float(r[1]), ypos=float(r[2]), sw_xpos=float(r[1]), sw_ypos=float(r[2]),
                                active=bool(st[i] & 1), closed=bool(st[i] & 1))
                for i, r in enumerate(rows) if int(r[0]) == 6]

    def get_mine_entities(self):
        """(toggle mines of type 1, of type 21) with xpos, ypos, state (0 toggled / deadly, 1 untoggled, 2 toggling)."""
        from types import SimpleNamespace

        rows, st = self._entities()
        m1, m21 = [], []
        for i, r in enumerate(rows):
            if int(r[0]) == 1:
                (m1 if int(r[5]) == 0 else m21).append(SimpleNamespace(xpos=float(r[1]), ypos=float(r[2]), state=int(st[i]), active=True))
        return m1, m21

    def exit(self):
        if self._b is not None:
            self._b.close()
            self._b = None
