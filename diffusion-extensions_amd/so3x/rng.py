"""Counter-based RNG bookkeeping for the in-kernel Philox4x32-10 streams.

A draw is identified by (seed; global sample index, offset).  Every sampling call
consumes a fresh range of offsets, so results are reproducible after manual_seed()
and independent of launch geometry and of the number of GPUs (the sample index is
global: shard base + local index)."""
import torch

_state = {"seed": None, "offset": 0}


def manual_seed(seed: int):
    _state["seed"] = int(seed)
    _state["offset"] = 0


def seed() -> int:
    if _state["seed"] is None:
        _state["seed"] = int(torch.initial_seed())  # follow torch.manual_seed() by default (host value, no sync)
    return _state["seed"]


def next_offset(count: int = 1) -> int:
    o = _state["offset"]
    _state["offset"] = o + int(count)
    return o
