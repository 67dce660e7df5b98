"""Asynchronous vector environment: S independent sub-batches on S HIP streams.

What it is for: a learner that consumes sub-batches as they arrive, like the reference's own trainers, which are asynchronous
per env (`SubprocVecEnv`, one process per env: nclone/gym_environment/environment_factory.py:112-153).  The N envs are split
into S sub-batches, each with its own native handle and HIP stream: `step_async` enqueues the sub-batches without waiting and
`step_wait(k)` hands back sub-batch k as soon as ITS stream has drained -- a learner can act on sub-batch k while the others are
still stepping.

What it is NOT (any more): a way to more env-steps/s.  Round 1 measured +15 % over the synchronous step (the tail of one
sub-batch overlapped the bulk of the others); since the heavy-first launch order and the build-variant autotuner of round 2 the
synchronous step is the faster one (BENCH_r02: 90.1 M env-steps/s against 60.4 M for 4 x 2048 envs) -- each sub-batch fills a
quarter of the chip, orders and tunes itself on a quarter of the statistics, and four launches cost four times the host work
(DESIGN.md section 6).  Use NppVecEnvironment for throughput.

Results are bit-identical to the synchronous NppVecEnvironment on the same envs, levels and actions (envs are independent;
tests/test_gpu_round2.py::test_async_vec_env_matches_sync).
"""
import numpy as np
import torch

from .engine import NppBatch


class AsyncBatches:
    """S NppBatch handles over consecutive env blocks of n // S envs, one HIP stream each (the engine under
    NppAsyncVecEnvironment; bench.py's async_subbatches figure uses it directly)."""

    def __init__(self, n_envs, n_streams, device=0, autoreset=True, outputs=(), fast_reset=False):
        assert n_envs % n_streams == 0, "n_envs must be a multiple of the number of streams"
        self.n, self.S, self.sub = int(n_envs), int(n_streams), int(n_envs) // int(n_streams)
        self.device = torch.device("cuda", int(device))
        with torch.cuda.device(self.device):
            self.streams = [torch.cuda.Stream() for _ in range(self.S)]
        self.batches = [NppBatch(self.sub, device=device, autoreset=autoreset, stream=self.streams[k], outputs=outputs,
                                 fast_reset=fast_reset) for k in range(self.S)]

    def load_levels(self, levels):
        for b in self.batches:
            b.load_levels(levels)

    def assign_levels(self, level_ids):
        level_ids = np.asarray(level_ids)
        assert len(level_ids) == self.n
        for k, b in enumerate(self.batches):
            b.assign_levels(level_ids[k * self.sub:(k + 1) * self.sub])

    def set_truncation_limit(self, limit):
        for k, b in enumerate(self.batches):
            b.set_truncation_limit(limit if np.isscalar(limit) else np.asarray(limit)[k * self.sub:(k + 1) * self.sub])

    def reset(self):
        for b in self.batches:
            b.reset()
            b.observe()

    def step_async(self, actions, frame_skip=4, only=None):
        """actions: list of S uint8 CUDA tensors [n // S] (already on the device).  Enqueues and returns."""
        for k, b in enumerate(self.batches):
            if only is None or k in only:
                b.step(actions[k], frame_skip, want_terminal=True)

    def wait(self, k=None):
        if k is None:
            for s in self.streams:
                s.synchronize()
        else:
            self.streams[k].synchronize()

    def close(self):
        for b in self.batches:
            b.close()


class NppAsyncVecEnvironment:
    """Gymnasium-style async vector env (the `step_async` / `step_wait` protocol of gymnasium.vector / SB3 VecEnv) over
    AsyncBatches.  Observation keys are those of NppVecEnvironment (game_state, action_mask, entity_positions, the
    pass-through position scalars, ...), as numpy arrays (pinned staging, one copy per sub-batch) or CUDA tensors.

    step_async(actions)            enqueue all S sub-batches
    step_wait()                    the whole batch, in env order (drop-in for a synchronous consumer)
    step_wait_partial(k)           just sub-batch k: rows [k * n/S, (k + 1) * n/S) -- lets a learner overlap
    step_async_partial(k, actions) re-enqueue just sub-batch k
    """

    def __init__(self, levels, num_envs, n_streams=4, level_ids=None, frame_skip=4, device=0, truncation_limit="dynamic",
                 output="numpy", autoreset=True, fast_reset=True):
        assert output in ("torch", "numpy")
        self.num_envs, self.frame_skip, self.output = int(num_envs), int(frame_skip), output
        self.ab = AsyncBatches(num_envs, n_streams, device=device, autoreset=autoreset, outputs=("positions",),
                               fast_reset=fast_reset)
        self.ab.load_levels(levels)
        if level_ids is None:
            level_ids = (np.arange(self.num_envs) // 64) % len(levels)
        self.ab.assign_levels(level_ids)
        if isinstance(truncation_limit, str):   # "dynamic": the reference env's per-level limit, as in NppVecEnvironment
            if truncation_limit != "dynamic":
                raise ValueError('truncation_limit: a number of frames or "dynamic"')
            for b in self.ab.batches:
                b.set_dynamic_truncation(True)
        else:
            self.ab.set_truncation_limit(truncation_limit)
        self._reset_bits = 11 if autoreset else 0   # won | dead | truncated: the row holds the spawn observation
        self._acts = []
        for b in self.ab.batches:
            with b._ctx():
                self._acts.append(torch.zeros(self.ab.sub, dtype=torch.uint8, device=b.device))
        self._names = ["game_state", "action_mask", "entity_pos", "positions", "flags", "reward", "frames", "terminal_state"]

    @property
    def n_streams(self):
        return self.ab.S

    def _upload(self, k, actions):
        b = self.ab.batches[k]
        with b._ctx():
            if isinstance(actions, torch.Tensor):
                self._acts[k].copy_(actions.to(torch.uint8), non_blocking=True)
            else:
                self._acts[k].copy_(torch.as_tensor(np.ascontiguousarray(actions, dtype=np.uint8)), non_blocking=True)

    def _result(self, k):
        b = self.ab.batches[k]
        if self.output == "torch":
            self.ab.wait(k)
            src = b.out.t
        else:
            src = b.to_host(self._names)   # one async copy on the sub-batch's stream + one synchronisation of that stream
        flags, pos = src["flags"], src["positions"]
        obs = {"game_state": src["game_state"], "action_mask": src["action_mask"], "entity_positions": src["entity_pos"],
               "player_x": pos[:, 0], "player_y": pos[:, 1], "switch_x": pos[:, 2], "switch_y": pos[:, 3],
               "exit_door_x": pos[:, 4], "exit_door_y": pos[:, 5],
               # as in NppVecEnvironment: an auto-reset row returns the spawn observation, whose switch is not yet hit
               "switch_activated": ((flags & 4) != 0) & ((flags & self._reset_bits) == 0)}
        info = {"player_won": (flags & 1) != 0, "player_dead": (flags & 2) != 0, "death_cause_code": (flags >> 4) & 3,
                "frames_executed": src["frames"], "terminal_observation": src["terminal_state"]}
        return obs, src["reward"], (flags & 3) != 0, (flags & 8) != 0, info

    @staticmethod
    def _cat(parts):
        if isinstance(parts[0], dict):
            return {k: NppAsyncVecEnvironment._cat([p[k] for p in parts]) for k in parts[0]}
        if isinstance(parts[0], torch.Tensor):
            return torch.cat(parts)
        return np.concatenate(parts)

    def reset(self, seed=None, options=None):
        self.ab.reset()
        res = [self._result(k) for k in range(self.ab.S)]
        return self._cat([r[0] for r in res]), {}

    def step_async(self, actions):
        sub = self.ab.sub
        for k in range(self.ab.S):
            self._upload(k, actions[k * sub:(k + 1) * sub])
        self.ab.step_async(self._acts, self.frame_skip)

    def step_async_partial(self, k, actions):
        self._upload(k, actions)
        self.ab.step_async(self._acts, self.frame_skip, only=(k,))

    def step_wait_partial(self, k):
        return self._result(k)

    def step_wait(self):
        res = [self._result(k) for k in range(self.ab.S)]
        return tuple(self._cat([r[i] for r in res]) for i in range(5))

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.ab.close()
