"""Replay file format + batched replay validation on the GPU.

Format restated from nclone/replay/gameplay_recorder.py:67-129 (V0: 8-byte header map_len, input_len;
V1: 12-byte header version, map_len, input_len + 1 success byte) and input decoding from
nclone/replay/replay_executor.py:61-84.  `validate_replays` is the batched counterpart of
scripts/validate_replays.py:62-140 / tools/test_replay_playback.py:14-80: load -> one tick per input byte ->
stop at won/died.
"""
import struct

import numpy as np
import torch

from .engine import NppBatch


class CompactReplay:
    def __init__(self, map_data, input_sequence, success=True, version=0):
        self.map_data = bytes(map_data)
        self.input_sequence = list(input_sequence)
        self.success = bool(success)
        self.version = version

    @classmethod
    def from_binary(cls, data):
        first = struct.unpack("<I", data[0:4])[0]
        if first <= 100:
            version, map_len, in_len = struct.unpack("<III", data[0:12])
            success = bool(data[12])
            m = data[13:13 + map_len]
            i = data[13 + map_len:13 + map_len + in_len]
        else:
            version = 0
            map_len, in_len = struct.unpack("<II", data[0:8])
            success = True
            m = data[8:8 + map_len]
            i = data[8 + map_len:8 + map_len + in_len]
        return cls(m, i, success, version)

    def to_binary(self):
        return struct.pack("<III", 1, len(self.map_data), len(self.input_sequence)) + struct.pack(
            "<B", 1 if self.success else 0) + self.map_data + bytes(self.input_sequence)


def decode_input_to_controls(input_byte):
    """Replay input byte -> (horizontal, jump): bit 0 jump, bit 1 right, bit 2 left; left + right cancel
    (the recorder's format, replay/gameplay_recorder.py:67-129)."""
    b = int(input_byte)
    return ((b >> 1) & 1) - ((b >> 2) & 1), b & 1


def validate_replays(replays, device=0):
    """Run all replays at once (one env per replay).  Returns a list of dicts: ticks run until termination
    (or all inputs), won, died, final position."""
    n = len(replays)
    b = NppBatch(n, device=device, autoreset=False)
    b.load_levels([np.frombuffer(r.map_data, dtype=np.uint8) for r in replays])
    b.assign_levels(np.arange(n))
    tmax = max(len(r.input_sequence) for r in replays)
    inputs = np.zeros((tmax, n), dtype=np.uint8)
    for k, r in enumerate(replays):
        inputs[: len(r.input_sequence), k] = r.input_sequence
    d_in = torch.from_numpy(inputs).to(b.device)
    res = [None] * n
    for t in range(tmax):
        b.tick(d_in[t:t + 1])
        f, i = b.dump_state()
        for k, r in enumerate(replays):
            if res[k] is not None:
                continue
            done = int(i[k, 0]) in (6, 7, 8)
            if done or t == len(r.input_sequence) - 1:
                res[k] = {"ticks": t + 1, "won": int(i[k, 0]) == 8, "died": int(i[k, 0]) in (6, 7),
                          "x": float(f[k, 0]), "y": float(f[k, 1])}
    b.close()
    return res
