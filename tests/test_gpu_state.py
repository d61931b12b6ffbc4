"""Batched get_state / set_state / serialize and the Monte-Carlo fan-out."""
import os
import pickle

import numpy as np
import pytest

from util import GOLDEN

pytestmark = pytest.mark.gpu


def _env(torch, n, **kw):
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    g = np.load(os.path.join(GOLDEN, "g8_traj_mini_03.npz"))
    res = float(g["resolution"])
    params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=res, refine_path=False, **kw)
    return BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], params, n_envs=n, seed=5)


@pytest.mark.parametrize("kw", [dict(), dict(control_delay=2, pose_delay=1, state_delay=3)], ids=["plain", "delays"])
def test_snapshot_restore_replays_bit_for_bit(torch_cuda, kw):
    """set_state(get_state()) + the same actions and normals => the same trajectory, bit for bit; the snapshot survives
    pickling through serialize()"""
    torch = torch_cuda
    from bc_gym_planning_env_amd.batched_env import BatchedState
    n = 512
    env = _env(torch, n, **kw)
    rng = np.random.RandomState(0)
    acts = [env.action_space.sample_batch(n, rng) for _ in range(40)]
    zs = [rng.normal(size=(n, 3)) for _ in range(40)]
    for t in range(15):
        env.step(acts[t], noise_z=zs[t])
    snap = env.get_state()
    blob = pickle.dumps(snap.serialize())
    first = []
    for t in range(15, 40):
        env.step(acts[t], noise_z=zs[t])
        first.append((env.state.robot.cpu().numpy().copy(), env.reward.cpu().numpy().copy(), env.done.cpu().numpy().copy()))
    for source in (snap, BatchedState.deserialize(pickle.loads(blob), device="cuda")):
        env.set_state(source)
        for k, t in enumerate(range(15, 40)):
            env.step(acts[t], noise_z=zs[t])
            assert (env.state.robot.cpu().numpy() == first[k][0]).all()
            assert (env.reward.cpu().numpy() == first[k][1]).all() and (env.done.cpu().numpy() == first[k][2]).all()


@pytest.mark.parametrize("kw", [dict(), dict(control_delay=1, pose_delay=2, state_delay=1)], ids=["plain", "delays"])
def test_fan_out_from_one_state(torch_cuda, kw):
    """many rollouts from one state: after fan_out(src) every env equals env src, and identical inputs keep them equal;
    a mask leaves the other envs untouched"""
    torch = torch_cuda
    n = 300
    env = _env(torch, n, **kw)
    rng = np.random.RandomState(1)
    for t in range(12):
        env.step(env.action_space.sample_batch(n, rng))
    before = env.get_state()
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
    mask[::2] = 1
    env.fan_out(7, mask)
    m = mask.cpu().numpy().astype(bool)
    for name in before.FIELDS:
        a, b = getattr(env.state, name), getattr(before, name)
        if a is None:
            continue
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert (a[..., ~m] == b[..., ~m]).all(), name                       # untouched
        assert (a[..., m] == b[..., 7:8]).all(), name                       # copies of env 7
    env.fan_out(7)
    one = env.action_space.sample_batch(1, rng)
    z = rng.normal(size=(1, 3))
    for t in range(10):
        env.step(np.repeat(one, n, axis=0), noise_z=np.repeat(z, n, axis=0))
        st = env.state.robot.cpu().numpy()
        assert (st == st[:, :1]).all()
    # the per-env view round-trips through the reference-shaped State, too
    from bc_gym_planning_env_amd.api import State
    s = env.envs[3].get_state()
    assert State.deserialize(pickle.loads(pickle.dumps(s.serialize()))) == s


def test_fan_out_carries_the_geometry_entry(torch_cuda):
    torch = torch_cuda
    from bc_gym_planning_env_amd import mini_env
    pool = mini_env.sample_pool(None, [1, 2, 3, 4], 2)
    env = mini_env.BatchedRandomMiniEnv(64, pool=pool, seed=1)
    g = env.geom_of_env.cpu().numpy()
    assert len(np.unique(g)) > 1
    env.fan_out(5)
    assert (env.geom_of_env.cpu().numpy() == g[5]).all()
    snap = env.get_state()
    env.reset()
    assert (env.geom_of_env.cpu().numpy() == pool.next_geom[g[5]]).all()
    env.set_state(snap)
    assert (env.geom_of_env.cpu().numpy() == g[5]).all()


def test_done_out_ring(torch_cuda):
    """step(..., done_out=row) stores the done mask straight into a caller-owned buffer (the multi-GPU ring)"""
    torch = torch_cuda
    n = 256
    a, b = _env(torch, n, iteration_timeout=6), _env(torch, n, iteration_timeout=6)
    ring = torch.full((4, n), 7, dtype=torch.uint8, device="cuda")
    rng = np.random.RandomState(3)
    for t in range(8):
        act = a.action_space.sample_batch(n, rng)
        z = rng.normal(size=(n, 3))
        _o, _r, d_a, _ = a.step(act, noise_z=z)
        _o, _r, d_b, _ = b.step(act, noise_z=z, done_out=ring[t % 4])
        assert d_b.data_ptr() == ring[t % 4].data_ptr()
        assert (d_a == ring[t % 4]).all()
    assert int(ring[1].sum()) == n      # step 6 (t = 5) timed every env out


def test_per_env_state_round_trip_with_delays(torch_cuda):
    """envs[i].set_state(envs[j].get_state()) transplants one env's full State (queues included): both envs then
    evolve identically under identical inputs"""
    torch = torch_cuda
    n = 8
    env = _env(torch, n, control_delay=2, pose_delay=3, state_delay=1)
    rng = np.random.RandomState(4)
    for t in range(9):
        env.step(env.action_space.sample_batch(n, rng), noise_z=rng.normal(size=(n, 3)))
    s5 = env.envs[5].get_state()
    true5 = env.state.robot[:, 5].clone()
    env.envs[2].set_state(s5)
    env.state.robot[:, 2] = true5        # (the robot's undelayed state is not part of State; see EnvView.set_state)
    assert env.envs[2].get_state() == s5
    for t in range(12):
        a = env.action_space.sample_batch(n, rng)
        z = rng.normal(size=(n, 3))
        a[2], z[2] = a[5], z[5]
        obs, rew, done, _ = env.step(a, noise_z=z)
        assert (env.state.robot[:, 2] == env.state.robot[:, 5]).all()
        assert (obs.pose[:, 2] == obs.pose[:, 5]).all() and rew[2] == rew[5] and done[2] == done[5]


def test_whole_env_serialize_roundtrip(torch_cuda):
    """PlanEnv.serialize / deserialize (env.py:251-276) per env of a batch: records of basic types (picklable) rebuild a
    batch -- private costmaps and paths, parametrisation and mid-episode state included -- that steps identically"""
    import pickle
    torch = torch_cuda
    from bc_gym_planning_env_amd import BatchedPlanEnv, EnvParams, mini_env
    params = mini_env.RandomMiniEnvParams(env_params=EnvParams(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2,
                                                               iteration_timeout=40, pose_delay=1, control_delay=1))
    pool = mini_env.sample_pool(params, [3, 4, 5], 2)
    n = 6
    env = BatchedPlanEnv(pool.costmaps, pool.paths, params.env_params, n_envs=n, seed=5, auto_reset=True)
    rng = np.random.RandomState(1)
    for _ in range(7):
        env.step(env.action_space.sample_batch(n, rng))
    records = pickle.loads(pickle.dumps([env.envs[i].serialize() for i in range(n)]))
    assert set(records[0]) == {'version', 'state', 'params', 'path', 'costmap'}
    assert EnvParams.deserialize(records[0]['params']) == params.env_params
    twin = BatchedPlanEnv.deserialize(records, seed=5, auto_reset=True)
    for i in range(n):
        assert twin.envs[i].get_state() == env.envs[i].get_state()
    z = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    for _ in range(60):
        a = env.action_space.sample_batch(n, rng)
        env.step(a, noise_z_out=z)
        twin.step(a, noise_z=z)
        assert torch.equal(env.reward, twin.reward) and torch.equal(env.done, twin.done)
        assert torch.equal(env.state.robot, twin.state.robot)
    assert twin.envs[2].get_state() == env.envs[2].get_state()


# records whose State holds the robot's true state (no state delay): the rebuilt env must reproduce the live env's
# next 100 steps.  With a state delay State.robot_state is the DELAYED state, and set_state hands exactly that to the
# robot (env.py:278-285) -- the reference's own deserialize would do the same -- so only the record itself is compared.
G13_EXACT = ["plain", "delay_c2p3", "delay_fresh", "pp", "pp_delay"]
G13_ALL = G13_EXACT + ["delay_p1s1", "delay_c2p3s1", "delay_filling"]


@pytest.mark.parametrize("tag", G13_ALL)
def test_reference_serialize_record_rebuilds_the_env(torch_cuda, tag):
    """PlanEnv.serialize() records made by the GENUINE reference mid-episode (g13) -> BatchedPlanEnv.deserialize:
    EnvView.serialize() of the rebuilt env is the reference's record key for key, and the env carries on as the
    reference's live env did (true state, what State exposes, reward, done, provider state) for 100 steps."""
    torch = torch_cuda
    from oracle import records
    from bc_gym_planning_env_amd import BatchedPlanEnv
    from util import ATOL, GOLDEN, z_in
    g = np.load(os.path.join(GOLDEN, "g13_serialized_%s.npz" % tag))
    rec = records.unpack(g["record"], g)
    env = BatchedPlanEnv.deserialize([records.unpack(g["record"], g), records.unpack(g["record"], g)])
    for i in range(2):
        assert records.same(rec, env.envs[i].serialize()) == []
    if tag not in G13_EXACT:
        return
    sd = env.params.state_delay
    for t in range(len(g["cont_actions"])):
        a = np.repeat(g["cont_actions"][t][None], 2, axis=0)
        z = np.repeat(z_in(g["cont_z"][t])[None], 2, axis=0)
        obs, rew, done, _ = env.step(a, noise_z=z)
        st = env.state.robot.cpu().numpy()
        np.testing.assert_allclose(st[:, 0], g["cont_true_states"][t], rtol=0, atol=ATOL, err_msg="%s step %d" % (tag, t))
        assert (st[:, 0] == st[:, 1]).all()
        np.testing.assert_allclose(obs.pose.cpu().numpy()[:, 0], g["cont_seen_pose"][t], rtol=0, atol=ATOL)
        seen = (env.state.robot_state_seen if sd else env.state.robot).cpu().numpy()[:, 0]
        np.testing.assert_allclose(seen, g["cont_seen_states"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(rew.cpu().numpy(), g["cont_reward"][t], rtol=0, atol=ATOL)
        assert (done.cpu().numpy() == g["cont_done"][t]).all()
        assert (env.state.robot_collided.cpu().numpy() == g["cont_collided"][t]).all()
        assert (env.state.target_idx.cpu().numpy() == g["cont_target_idx"][t]).all()
        np.testing.assert_allclose(env.state.min_spat_dist_so_far.cpu().numpy(), g["cont_min_dist"][t], rtol=0, atol=ATOL)
        if t in (0, 30, 99):
            assert len(env.envs[0].observation().path) == int(g["cont_obs_path_len"][t])


@pytest.mark.parametrize("kind", ["two-kernel", "single-kernel"])
def test_steps_replayed_from_a_captured_graph(torch_cuda, kind):
    """The step counter (noise stream, parity of the parking counters) lives on the device: steps captured into a HIP
    graph and replayed are the very steps an ordinary launch sequence performs."""
    torch = torch_cuda
    n = 3000
    env, twin = _env(torch, n), _env(torch, n)
    if kind == "single-kernel":
        env.set_tuning(defer=0)
        twin.set_tuning(defer=0)
    rng = np.random.RandomState(4)
    acts = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(12)])).cuda()
    a_static = acts[0].clone()
    for env_ in (env, twin):                       # ordinary steps first: uploads the parameter block
        env_.step(a_static)
        env_.step(a_static)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            env.step(a_static)
            env.step(a_static)                     # two steps per replay
    torch.cuda.current_stream().wait_stream(side)
    # capture does not execute: both envs are still in step; now 6 replays (12 steps) against 12 launched steps
    for k in range(6):
        a_static.copy_(acts[2 * k])                # (both steps of a replay read the same action tensor)
        graph.replay()
        twin.step(acts[2 * k])
        twin.step(acts[2 * k])
        torch.cuda.synchronize()
        assert torch.equal(env.state.robot, twin.state.robot), k
        assert torch.equal(env.reward, twin.reward) and torch.equal(env.done, twin.done), k
        assert torch.equal(env.state.current_iter, twin.state.current_iter), k
    # and back to ordinary launches
    env.step(acts[3])
    twin.step(acts[3])
    assert torch.equal(env.state.robot, twin.state.robot)


def test_reseed_restarts_the_noise_stream_whatever_the_step_parity(torch_cuda):
    """seed(s) restarts the step counter; the parity-keyed parking counters are re-armed with it, so an env reseeded
    after an odd number of steps carries on exactly like one reseeded after an even number"""
    torch = torch_cuda
    n = 4096
    odd, even = _env(torch, n), _env(torch, n)
    rng = np.random.RandomState(2)
    acts = torch.from_numpy(np.stack([odd.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
    acts[:, :, 0] *= 3.0
    for k in range(3):
        odd.step(acts[k])
    for k in range(4):
        even.step(acts[k])
    even.set_state(odd.get_state())
    odd.seed(77)
    even.seed(77)
    for k in range(40):
        odd.step(acts[k % 8])
        even.step(acts[k % 8])
        assert torch.equal(odd.state.robot, even.state.robot), k
        assert torch.equal(odd.reward, even.reward) and torch.equal(odd.done, even.done), k
        assert torch.equal(odd.state.robot_collided, even.state.robot_collided), k
