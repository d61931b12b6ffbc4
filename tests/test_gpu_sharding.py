"""The property the multi-GPU design rests on (include/bcplan.h: bcp_create's env_id_base): a batch sharded into
contiguous env blocks IS the unsharded batch.  In one process (two handles against one) and across two real rank
processes that share this box's GPU (rehearsal transport gloo, chosen explicitly; RCCL needs a GPU per rank)."""
import os
import socket
import sys

import numpy as np
import pytest

from util import GOLDEN

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "ranks"))


def _assert_same(whole, parts, what):
    joined = np.concatenate(parts, axis=-1)
    assert np.array_equal(whole, joined, equal_nan=True), what


def test_two_half_handles_equal_one_handle_shared_map(torch_cuda):
    """65 536 replicas of a RandomMiniEnv world, on-device noise, auto-reset, 100 steps: handles of 32 768 envs with
    env_id_base 0 and 32 768 against one handle of 65 536 -- every state value, reward, done flag and the normals the
    steps drew, bit for bit."""
    torch = torch_cuda
    from sharded_rank import global_actions, make_shard
    n, steps = 65536, 100
    whole = make_shard(n, 0, 0, timeout=90)
    halves = [make_shard(n // 2, 0, 0, timeout=90), make_shard(n // 2, n // 2, 0, timeout=90)]
    acts = global_actions(n, 8)
    zw = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    zh = [torch.zeros(n // 2, 3, dtype=torch.float64, device="cuda") for _ in range(2)]
    dones = 0
    for k in range(steps):
        a = torch.from_numpy(acts[k % 8]).cuda()
        whole.step(a, noise_z_out=zw)
        for h, env in enumerate(halves):
            env.step(a[h * (n // 2):(h + 1) * (n // 2)].contiguous(), noise_z_out=zh[h])
        _assert_same(whole.state.robot.cpu().numpy(), [e.state.robot.cpu().numpy() for e in halves], "robot state, step %d" % k)
        _assert_same(whole.reward.cpu().numpy(), [e.reward.cpu().numpy() for e in halves], "reward")
        _assert_same(whole.done.cpu().numpy(), [e.done.cpu().numpy() for e in halves], "done")
        _assert_same(whole.state.target_idx.cpu().numpy(), [e.state.target_idx.cpu().numpy() for e in halves], "target")
        _assert_same(whole.state.current_iter.cpu().numpy(), [e.state.current_iter.cpu().numpy() for e in halves], "iter")
        assert np.array_equal(zw.cpu().numpy(), np.concatenate([z.cpu().numpy() for z in zh]), equal_nan=True)
        dones += int(whole.done.sum())
    assert dones >= n   # every env finished an episode (wall or the 90-step timeout) and was reset inside the kernel
    # ... and the stream itself: the normals of global env e at step t do not depend on the handle that draws them
    from bc_gym_planning_env_amd import _lib
    out_w = torch.empty(3, 64, 3, dtype=torch.float64, device="cuda")
    out_h = torch.empty(3, 64, 3, dtype=torch.float64, device="cuda")
    _lib.check(whole._lib.bcp_device_normals(whole._h, n // 2 + 100, 64, 7, 3, out_w.data_ptr(), None))
    _lib.check(whole._lib.bcp_device_normals(halves[1]._h, 100, 64, 7, 3, out_h.data_ptr(), None))
    assert torch.equal(out_w, out_h)


def test_two_half_handles_equal_one_handle_endless_pool(torch_cuda):
    """The same with a world of its own per env: RandomMiniEnv(seed = GLOBAL env index) streams sampled on the device,
    every reset moves an env on along its stream (pool seeds keyed by env_id_base too)."""
    torch = torch_cuda
    from bc_gym_planning_env_amd import mini_env
    from sharded_rank import global_actions
    n, steps, per = 4096, 50, 4
    p = mini_env.default_random_mini_env_params()
    import attr
    p = attr.evolve(p, env_params=attr.evolve(p.env_params, iteration_timeout=20))

    def make(count, base):
        return mini_env.BatchedRandomMiniEnv(count, params=p, episodes=per, endless=True, auto_reset=True, seed=5,
                                             env_id_base=base)
    whole = make(n, 0)
    halves = [make(n // 2, 0), make(n // 2, n // 2)]
    acts = global_actions(n, 8, seed=3)
    resets = 0
    for k in range(steps):
        a = torch.from_numpy(acts[k % 8]).cuda()
        whole.step(a)
        for h, env in enumerate(halves):
            env.step(a[h * (n // 2):(h + 1) * (n // 2)].contiguous())
        if k % 10 == 9:
            for e in [whole] + halves:
                e.refresh()
        _assert_same(whole.state.robot.cpu().numpy(), [e.state.robot.cpu().numpy() for e in halves], "robot state, step %d" % k)
        _assert_same(whole.reward.cpu().numpy(), [e.reward.cpu().numpy() for e in halves], "reward")
        _assert_same(whole.done.cpu().numpy(), [e.done.cpu().numpy() for e in halves], "done")
        # pool entries are handle-local: entry = local env * per + ring slot
        gw = whole.geom_of_env.cpu().numpy()
        gh = np.concatenate([halves[0].geom_of_env.cpu().numpy(), halves[1].geom_of_env.cpu().numpy() + (n // 2) * per])
        assert np.array_equal(gw, gh)
        resets += int(whole.done.sum())
    assert resets > n


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_processes_step_real_shards(torch_cuda, launcher, tmp_path):
    """World size 2, one process per rank, both on this box's GPU: each rank steps a real BatchedPlanEnv shard and the
    done masks are all-gathered (per step, and as the ring bench.py uses).  What rank 0 gathered must be the done
    masks -- and the final state the state -- of the same batch stepped in ONE process."""
    torch = torch_cuda
    from sharded_rank import global_actions, make_shard
    n, steps = 8192, 48
    out = str(tmp_path / "rank0.npz")
    port = _free_port()
    script = os.path.join(ROOT, "tests", "ranks", "sharded_rank.py")
    group = [{"argv": [sys.executable, script, out, str(n), str(steps)],
              "env": {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": "2", "LOCAL_WORLD_SIZE": "2",
                      "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "BCP_DIST_BACKEND": "gloo",
                      "HSA_ENABLE_IPC_MODE_LEGACY": "0"}} for r in range(2)]
    results = launcher.run_group(group, timeout=420, cwd=ROOT)
    for r in results:
        assert r["rc"] == 0, r["err"][-3000:]
    got = np.load(out)
    assert str(got["backend"]) == "gloo"
    env = make_shard(n, 0, 0)
    acts = global_actions(n, steps)
    expect = []
    for k in range(steps):
        env.step(torch.from_numpy(acts[k]).cuda())
        expect.append(env.done.cpu().numpy().copy())
    expect = np.stack(expect)
    assert expect.sum() > n // 2
    np.testing.assert_array_equal(got["done_per_step"], expect)
    np.testing.assert_array_equal(got["done_ring"], expect)
    np.testing.assert_array_equal(got["robot"], env.state.robot.cpu().numpy())


def test_rccl_refuses_more_ranks_than_gpus(launcher):
    """Without the explicit rehearsal transport a rank of a 2-rank job on a 1-GPU box fails at once and says why --
    it neither hangs nor changes transport on its own."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has a GPU per rank")
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from bc_gym_planning_env_amd import distributed as d\n"
            "d.init_from_env()\n" % ROOT)
    r = launcher.run([sys.executable, "-c", code],
                     env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1",
                          "MASTER_PORT": str(_free_port()), "BCP_DIST_BACKEND": ""}, timeout=120)
    assert r["rc"] != 0 and "BCP_DIST_BACKEND=gloo" in r["err"]


def _single_process_masks(torch, n, steps):
    from sharded_rank import global_actions, make_shard
    env = make_shard(n, 0, 0)
    acts = global_actions(n, steps)
    expect = []
    for k in range(steps):
        env.step(torch.from_numpy(acts[k]).cuda())
        expect.append(env.done.cpu().numpy().copy())
    return np.stack(expect), env.state.robot.cpu().numpy()


def test_mask_bits_native_vs_numpy(torch_cuda):
    """bcp_pack_mask_bits / bcp_unpack_mask_bits (what DoneGather(packed=True) sends): numpy's little-endian packbits, for
    whole words, a ragged tail and a source that is not 16-byte aligned"""
    torch = torch_cuda
    import ctypes as C
    from bc_gym_planning_env_amd import _lib, distributed as bdist
    lib = _lib.load()
    rng = np.random.RandomState(3)
    for n, offset in ((65536 * 8, 0), (4096 + 19, 0), (1024, 5), (31, 1)):
        host = (rng.rand(n + offset) < 0.3).astype(np.uint8) * rng.randint(1, 255, n + offset).astype(np.uint8)
        src = torch.from_numpy(host).cuda()[offset:]
        words = torch.zeros((n + 31) // 32, dtype=torch.int32, device="cuda")
        _lib.check(lib.bcp_pack_mask_bits(src.data_ptr(), n, words.data_ptr(), None))
        want = np.packbits(np.pad(host[offset:] != 0, (0, (-n) % 32)), bitorder="little").view(np.uint32)
        assert (words.cpu().numpy().view(np.uint32) == want).all(), (n, offset)
        back = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        _lib.check(lib.bcp_unpack_mask_bits(words.data_ptr(), n, back.data_ptr(), None))
        assert (back.cpu().numpy() == (host[offset:] != 0)).all(), (n, offset)
    m = torch.from_numpy((rng.rand(4096) < 0.5).astype(np.uint8) * 3).cuda()
    assert torch.equal(bdist.pack_mask_bits(m).cpu(), bdist.pack_mask_bits(m.cpu()))            # native == tensor arithmetic
    assert torch.equal(bdist.unpack_mask_bits(bdist.pack_mask_bits(m)).cpu(), (m != 0).to(torch.uint8).cpu())


def test_rccl_branch_at_world_size_one(torch_cuda, launcher, tmp_path):
    """The RCCL path for real, on the one GPU of this box: BCP_DIST_FORCE=1 makes init_from_env build a `nccl` process
    group at world size 1 (device_id path, communicator set-up) and DoneGather take its collective branch --
    all_gather_into_tensor per step, and the pipelined form bench.py uses (staging copy on the step's stream,
    asynchronous work handles, wait two launches later) -- while the rank steps a real shard with done_out = ring row.
    What it gathered must be the single-process masks."""
    torch = torch_cuda
    n, steps = 8192, 48
    out = str(tmp_path / "rank0.npz")
    script = os.path.join(ROOT, "tests", "ranks", "sharded_rank.py")
    r = launcher.run([sys.executable, script, out, str(n), str(steps)],
                     env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "LOCAL_WORLD_SIZE": "1",
                          "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "BCP_DIST_BACKEND": "nccl",
                          "BCP_DIST_FORCE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=420, cwd=ROOT)
    assert r["rc"] == 0, r["err"][-3000:]
    got = np.load(out)
    assert str(got["backend"]) == "nccl"
    expect, robot = _single_process_masks(torch, n, steps)
    assert expect.sum() > n // 2
    np.testing.assert_array_equal(got["done_per_step"], expect)
    np.testing.assert_array_equal(got["done_ring"], expect)
    np.testing.assert_array_equal(got["robot"], robot)


def _bench_line(result):
    assert result["rc"] == 0, result["err"][-3000:]
    lines = [x for x in result["out"].splitlines() if x.startswith("{")]
    assert len(lines) == 1, result["out"][-2000:]
    import json
    return json.loads(lines[0])


def test_bench_rccl_at_world_size_one(launcher):
    """bench.py itself through the RCCL ring (forced group at N = 1): the line says so, and the timed region obeys its
    rules -- reps x steps steps, wall time within a few percent of the HIP-event time of the same region."""
    r = launcher.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                      "--envs-per-gpu", "16384", "--no-aux", "--no-cpu-baseline"],
                     env={"BCP_DIST_FORCE": "1", "BCP_DIST_BACKEND": "nccl", "MASTER_ADDR": "127.0.0.1",
                          "MASTER_PORT": str(_free_port()), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=600, cwd=ROOT)
    line = _bench_line(r)
    assert line["n_gpus"] == 1 and line["steps"] == 20 and "RCCL" in line["config"]["sharding"]
    tr = line["timed_region"]
    assert line["reps"] >= 1 and tr["steps_timed"] == line["reps"] * 20 and tr["wall_ms"] >= 45.0
    assert abs(line["ms_per_step"] - tr["wall_ms"] / tr["steps_timed"]) < 1e-9
    assert line["value"] == pytest.approx(16384 * tr["steps_timed"] / (tr["wall_ms"] * 1e-3), rel=1e-9)
    assert tr["device_ms"] <= tr["wall_ms"] * 1.001


def test_bench_four_rank_rehearsal_on_one_gpu(launcher):
    """`BCP_DIST_BACKEND=gloo python bench.py --gpus 4 --envs-per-gpu 8192` on this box's one GPU: bench.py's own
    launcher, the rendezvous, four rank processes stepping real shards, the ring and the MAX-over-ranks timing, end to
    end.  (Four, not eight: a GPU box of this pool admits at most 6 processes on its card and this session is one of
    them; the launcher and the ring at world size 8 run on the CPU in tests/test_distributed_gloo.py.)"""
    r = launcher.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "20", "--warmup", "5",
                      "--envs-per-gpu", "8192", "--no-aux", "--no-cpu-baseline", "--reps", "4"],
                     env={"BCP_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "BCP_BENCH_TIMEOUT": "500"},
                     timeout=600, cwd=ROOT)
    line = _bench_line(r)
    assert line["n_gpus"] == 4 and line["config"]["envs_total"] == 4 * 8192 and line["scaling"] == "weak"
    assert "REHEARSAL" in line["config"]["sharding"] and line["reps"] == 4
    assert line["timed_region"]["steps_timed"] == 80 and line["value"] > 0
