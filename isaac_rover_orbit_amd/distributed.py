"""Multi-GPU plumbing: one process per GPU, envs range-sharded, RCCL over xGMI only for rollout exchange.

The reference is single-process / single-GPU (no NCCL, no torch.distributed anywhere).  Envs never
interact, the terrain is read-only and replicated, and the reset RNG is keyed by the GLOBAL env id, so the step path
needs NO collective: every rank runs ``RoverEnv`` on its contiguous id range.  What a trainer may want from the other
ranks is the finished rollout; ``RolloutGatherer`` moves it with one ``all_gather_into_tensor`` per rollout
(backend "nccl" == RCCL on ROCm; "gloo" on CPU for the tests), optionally on a side stream so that it overlaps the
next rollout's simulation.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class Shard:
    rank: int
    world_size: int
    local_num_envs: int
    env_id_offset: int
    global_num_envs: int


def shard_envs(global_num_envs: int, rank: int, world_size: int) -> Shard:
    """Contiguous, balanced id ranges: rank r owns [offset, offset + local)."""
    if not (0 <= rank < world_size) or global_num_envs < world_size:
        raise ValueError("bad rank / world size / env count")
    base, rem = divmod(global_num_envs, world_size)
    local = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return Shard(rank, world_size, local, offset, global_num_envs)


def weak_shard(envs_per_rank: int, rank: int, world_size: int) -> Shard:
    """Weak scaling (BASELINE config 3): a fixed number of envs per GPU."""
    return Shard(rank, world_size, envs_per_rank, rank * envs_per_rank, envs_per_rank * world_size)


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            # ROVER_DIST_BACKEND=gloo: rehearse the multi-process path where the ranks cannot each own a GPU
            backend = os.environ.get("ROVER_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class RolloutGatherer:
    """all_gather of per-rank rollout shards: (T, n_local, ...) -> (world, T, n_local, ...), equal shards only."""

    def __init__(self, group=None, side_stream: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._stream = torch.cuda.Stream() if (side_stream and torch.cuda.is_available()) else None
        self._pending = None

    def gather(self, shard: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        if out is None:
            out = torch.empty((self.world,) + tuple(shard.shape), dtype=shard.dtype, device=shard.device)
        if self.world == 1:
            out[0].copy_(shard)
            return out
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                dist.all_gather_into_tensor(out.view(-1), shard.contiguous().view(-1), group=self.group)
            self._pending = out
        else:
            dist.all_gather_into_tensor(out.view(-1), shard.contiguous().view(-1), group=self.group)
        return out

    def wait(self):
        if self._stream is not None and self._pending is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
            self._pending = None


def reduce_log(log: torch.Tensor, group=None) -> torch.Tensor:
    """Global extras["log"]: count-weighted means of the per-rank log vectors (include/rover_hip.h layout)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return log.clone()
    cnt = log[13].clone()
    w = log.clone()
    w[0:7] *= cnt
    w[11:13] *= cnt
    dist.all_reduce(w, op=dist.ReduceOp.SUM, group=group)
    tot = w[13].clamp(min=1.0)
    w[0:7] /= tot
    w[11:13] /= tot
    return w
