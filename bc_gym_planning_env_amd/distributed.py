"""Sharding the env batch over the GPUs of one node: one process per GPU, envs split by contiguous index blocks,
nothing crosses GPUs on the data path except ONE all-gather of the uint8 done mask per step (RCCL over xGMI when the
backend is "nccl"; "gloo" on CPU for tests).  The reference has no counterpart: its envs are independent objects.
"""
import os

import torch
import torch.distributed as dist


def env_block(n_total, rank, world_size):
    """Contiguous block of env indices owned by `rank`: (first, count).  Remainder goes to the lowest ranks."""
    base, rem = divmod(int(n_total), int(world_size))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* when launched by torch.distributed.run.
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


class DoneGather(object):
    """All-gather of the per-rank done mask into the global mask [world_size * n_local] (uint8).

    Equal shard sizes are required (all_gather_into_tensor); pad the last shard if the batch does not divide."""

    def __init__(self, n_local, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n_local = int(n_local)
        self.out = torch.zeros(self.world * self.n_local, dtype=torch.uint8, device=device)

    def __call__(self, done_local, async_op=False):
        if self.world == 1:
            self.out.copy_(done_local)
            return None if async_op else self.out
        work = dist.all_gather_into_tensor(self.out, done_local, group=self.group, async_op=async_op)
        return work if async_op else self.out
