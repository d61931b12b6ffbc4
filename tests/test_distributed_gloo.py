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
    # ... and with one BIT per env on the wire (what bench.py's ring sends): the same masks come back, as zeros and ones,
    # or as the gathered bits themselves
    packed = bdist.DoneGather(count, torch.device("cpu"), packed=True)
    for step in range(4):
        buf = (done_local ^ (step & 1)) * (3 if step == 2 else 1)   # (any non-zero byte is "done")
        packed.launch(buf)
        buf.zero_()
        assert torch.equal(packed.result(), base ^ (step & 1)), (rank, step)
        bits = packed.result(unpack=False)
        assert bits.numel() == n_total // 8 and torch.equal(bits, bdist.pack_mask_bits(base ^ (step & 1)))
    packed.flush()
    np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([first, count]))
    dist.barrier()
    dist.destroy_process_group()


def test_done_gather_world2(tmp_path):
    world, n_total = 2, 4096
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    blocks = [np.load(os.path.join(str(tmp_path), "ok_%d.npy" % r)) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[0][0] + blocks[0][1] == blocks[1][0] and blocks[1][0] + blocks[1][1] == n_total


def test_bench_launcher_world8_ring(capfd):
    """bench.py's own launcher (spawn_ranks: fresh rank processes, free port, exit-code relay) at the world size the
    driver's scaling run uses, with the rendezvous and the done-mask ring of bench.py over gloo on the CPU: 8 ranks,
    64 steps, every rank checks every mask it gathered (tests/ranks/ring_rank.py)."""
    import argparse
    import json
    sys.path.insert(0, ROOT)
    import bench
    env_before = dict(os.environ)
    os.environ["BCP_DIST_BACKEND"] = "gloo"
    os.environ["BCP_BENCH_TIMEOUT"] = "240"
    try:
        rc = bench.spawn_ranks(argparse.Namespace(gpus=8), script=os.path.join(ROOT, "tests", "ranks", "ring_rank.py"),
                               argv=["512", "64"])
    finally:
        os.environ.clear()
        os.environ.update(env_before)
    assert rc == 0
    line = [x for x in capfd.readouterr().out.splitlines() if x.startswith("{")][-1]
    got = json.loads(line)
    assert got == {"world": 8, "backend": "gloo", "masks_checked": 8 * 8 * 64}


def test_bench_launcher_relays_a_failing_rank(tmp_path):
    """A rank that dies stops the job: the launcher returns its exit code instead of waiting for the others."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    code = "import os, sys, time\nif os.environ['RANK'] == '2':\n    sys.exit(7)\ntime.sleep(60)\n"
    path = str(tmp_path / "failing_rank.py")
    with open(path, "w") as f:
        f.write(code)
    import time
    t0 = time.time()
    rc = bench.spawn_ranks(argparse.Namespace(gpus=4), script=path, argv=[])
    assert rc == 7 and time.time() - t0 < 30


def _geometry_worker(rank, world, port, out_dir, tamper):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from bc_gym_planning_env_amd import distributed as bdist
    bdist.init_from_env(backend="gloo")
    g = np.load(os.path.join(ROOT, "tests", "golden", "g8_traj_mini_00.npz"))
    arrays = [g["costmap"].copy(), g["origin"].copy(), np.asarray(g["resolution"]), g["path"].copy()]
    if rank == 1:
        arrays[0][40, 41] ^= 254          # this rank read a different map ...
        arrays[3][5, 0] += 1e-9           # ... and a path that differs in one bit pattern
    caught = None
    try:
        bdist.check_same_geometry(bdist.geometry_digest(*arrays))
    except RuntimeError as exc:
        caught = str(exc)
    assert caught is not None and "[1]" in caught, (rank, caught)     # raised on EVERY rank, naming the odd one
    # rank 0's copy for everybody (one broadcast at set-up): afterwards the digests agree
    if not tamper:
        arrays = bdist.broadcast_geometry(arrays)
    else:                                  # (the broadcast really comes from `src`: rank 1's tampered copy wins here)
        arrays = bdist.broadcast_geometry(arrays, src=1)
    bdist.check_same_geometry(bdist.geometry_digest(*arrays))
    ref = [g["costmap"], g["origin"], np.asarray(g["resolution"]), g["path"]]
    same = all(np.array_equal(np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)) for a, b in zip(arrays, ref))
    assert same == (not tamper), rank
    np.save(os.path.join(out_dir, "geom_ok_%d_%d.npy" % (rank, int(tamper))), np.array([1]))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_with_different_geometry_are_caught(tmp_path):
    """A sharded job replicates the shared costmap / path on every rank: check_same_geometry stops the job when a rank
    holds a different copy, broadcast_geometry hands out rank 0's (world size 2, gloo)."""
    for tamper in (False, True):
        mp.spawn(_geometry_worker, args=(2, _free_port(), str(tmp_path), tamper), nprocs=2, join=True)
        assert all(os.path.exists(os.path.join(str(tmp_path), "geom_ok_%d_%d.npy" % (r, int(tamper)))) for r in range(2))
