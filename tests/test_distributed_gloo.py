"""world_size-2 gloo test (CPU) of the multi-GPU plumbing: env-block sharding and the done-mask all-gather.
The data path itself has no collective; the gather is the only exchange step."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from bc_gym_planning_env_amd import distributed as bdist
    r, w, _ = bdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    first, count = bdist.env_block(n_total, rank, world)
    # every rank derives its local done mask from the GLOBAL env index, so the gathered result is checkable
    gidx = torch.arange(first, first + count)
    done_local = ((gidx * 2654435761) % 7 == 0).to(torch.uint8)
    gather = bdist.DoneGather(count, torch.device("cpu"))
    for step in range(3):
        out = gather(done_local ^ (step & 1))
        exp = ((torch.arange(n_total) * 2654435761) % 7 == 0).to(torch.uint8) ^ (step & 1)
        assert torch.equal(out, exp), (rank, step)
    work = gather(done_local, async_op=True)
    work.wait()
    assert torch.equal(gather.out, ((torch.arange(n_total) * 2654435761) % 7 == 0).to(torch.uint8))
    # pipelined form: results come back one launch late, in order
    base = ((torch.arange(n_total) * 2654435761) % 7 == 0).to(torch.uint8)
    for step in range(5):
        buf = done_local ^ (step & 1)
        gather.launch(buf)
        buf.zero_()  # the caller may overwrite its mask right away (the step kernel does)
        assert torch.equal(gather.result(), base ^ (step & 1)), (rank, step)
    gather.flush()
    np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([first, count]))
    dist.barrier()
    dist.destroy_process_group()


def test_done_gather_world2(tmp_path):
    world, n_total = 2, 4096
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    blocks = [np.load(os.path.join(str(tmp_path), "ok_%d.npy" % r)) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[0][0] + blocks[0][1] == blocks[1][0] and blocks[1][0] + blocks[1][1] == n_total
