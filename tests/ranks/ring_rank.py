"""One rank of the CPU rehearsal of bench.py's launcher at the world size the driver uses (8): started by
bench.spawn_ranks (RANK / WORLD_SIZE / MASTER_* in the environment), rendezvous over gloo, then the done-mask ring of
bench.py -- a mask per step into row k % R, the ring all-gathered every R steps (R = 8 here, 128 in bench.py), results one launch late -- with masks
derived from the GLOBAL env index so that every rank can check what it gathered.  No GPU, no env: plumbing only.
Rank 0 prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def mask_of(first, count, step, torch):
    g = torch.arange(first, first + count, dtype=torch.int64)
    return (((g * 2654435761 + step * 40503) >> 3) % 5 == 0).to(torch.uint8)


def main():
    n_local, steps = int(sys.argv[1]), int(sys.argv[2])
    import torch
    import torch.distributed as dist
    from bc_gym_planning_env_amd import distributed as bdist
    rank, world, local_rank = bdist.init_from_env()
    assert dist.get_backend() == "gloo"
    first, count = bdist.env_block(n_local * world, rank, world)
    assert (first, count) == (rank * n_local, n_local)
    every = 8
    ring = torch.zeros(every, n_local, dtype=torch.uint8)
    gather = bdist.DoneGather(every * n_local, torch.device("cpu"), packed=True)   # (as bench.py: one bit per env on the wire)
    checked = 0
    for k in range(steps):
        ring[k % every].copy_(mask_of(first, count, k, torch))
        if k % every == every - 1:
            gather.launch(ring.view(-1))
            ring.zero_()   # the next steps overwrite the ring right away
            got = gather.result().view(world, every, n_local)
            for j in range(every):
                step = k - every + 1 + j
                for w in range(world):
                    assert torch.equal(got[w, j], mask_of(w * n_local, n_local, step, torch)), (rank, step, w)
                    checked += 1
    gather.flush()
    t = torch.tensor([float(checked)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"world": world, "backend": dist.get_backend(), "masks_checked": int(t.item())}))
        sys.stdout.flush()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
