"""bcp_rollout (K steps per call; with the single-launch step form ONE launch of step_local_kernel<.., ROLL = true>) against
K calls of bcp_step on a twin batch: every state value, reward, done / collision flag, drawn normal and pool entry bit for
bit -- shared map (the metric configuration), private maps and paths (BASELINE configs[3]'s shape), a geometry pool with
in-kernel resets onto new worlds, delay queues, and the step forms that have no rollout kernel (stepped launch by launch)."""
import os

import numpy as np
import pytest

from util import GOLDEN

pytestmark = pytest.mark.gpu


def _same(torch, a, b, what):
    assert torch.equal(a, b), what


def _compare_envs(torch, env, twin, tag):
    _same(torch, env.state.robot, twin.state.robot, tag + " robot state")
    _same(torch, env.state.min_spat_dist_so_far, twin.state.min_spat_dist_so_far, tag + " min_dist")
    _same(torch, env.state.target_idx, twin.state.target_idx, tag + " target_idx")
    _same(torch, env.state.current_iter, twin.state.current_iter, tag + " current_iter")
    _same(torch, env.state.robot_collided, twin.state.robot_collided, tag + " robot_collided")
    if env.geom_of_env is not None:
        _same(torch, env.geom_of_env, twin.geom_of_env, tag + " pool entry")
    for name in ("pose_seen", "robot_state_seen", "control_queue", "poses_queue", "robot_state_queue"):
        a, b = getattr(env.state, name, None), getattr(twin.state, name, None)
        if a is not None:
            _same(torch, a, b, tag + " " + name)


def _roll_and_compare(torch, make, k_steps, rounds, tuning=None, scale=1.0, f64=False, scatter=0.0):
    env, twin = make(), make()
    if tuning:
        env.set_tuning(**tuning)
        twin.set_tuning(**tuning)
    n = env.n_envs
    rng = np.random.RandomState(5)
    if scatter:   # knock the robots off their start poses (the same way in both batches): walls get hit within a few steps
        kick = torch.from_numpy(np.concatenate([rng.normal(0, scatter, (2, n)), rng.normal(0, 0.6, (1, n))])).cuda()
        for e in (env, twin):
            e.state.robot[0:3] += kick
            if e.state.pose_seen is not None:
                e.state.pose_seen.copy_(e.state.robot[0:3])
            if e.state.robot_state_seen is not None:
                e.state.robot_state_seen.copy_(e.state.robot)
    hits = dones = 0
    for r in range(rounds):
        acts = np.stack([env.action_space.sample_batch(n, rng) for _ in range(k_steps)])
        acts[..., 0] *= scale
        a = torch.from_numpy(acts.astype(np.float64 if f64 else np.float32)).cuda()
        zout = torch.zeros(k_steps, n, 3, dtype=torch.float64, device="cuda")
        coll = torch.zeros(k_steps, n, dtype=torch.uint8, device="cuda")
        err = torch.zeros(k_steps, n, dtype=torch.int32, device="cuda")
        rew, done = env.rollout(a, noise_z_out=zout, collided_out=coll, err_out=err)
        z1 = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
        for k in range(k_steps):
            twin.step(a[k], noise_z_out=z1)
            _same(torch, rew[k], twin.reward, "reward, round %d step %d" % (r, k))
            _same(torch, done[k], twin.done, "done, round %d step %d" % (r, k))
            _same(torch, coll[k], twin.collided_now, "collided_now, round %d step %d" % (r, k))
            _same(torch, err[k], twin.err, "err, round %d step %d" % (r, k))
            assert torch.equal(torch.nan_to_num(zout[k], nan=123.0), torch.nan_to_num(z1, nan=123.0)), (r, k)
        _compare_envs(torch, env, twin, "round %d" % r)
        hits += int(coll.sum())
        dones += int(done.sum())
    env.check_errors()
    return hits, dones


def _mini(n, **kw):
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    g = np.load(os.path.join(GOLDEN, "g8_traj_mini_00.npz"))
    res = float(g["resolution"])
    params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=res, refine_path=False, iteration_timeout=40, **kw)
    return lambda: BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], params, n_envs=n, auto_reset=True, seed=77)


@pytest.mark.parametrize("k_steps", [1, 2, 7, 33])
def test_rollout_shared_map_vs_steps(torch_cuda, k_steps):
    """the metric configuration's shape: shared map and path in LDS, tricycle + on-device noise, in-kernel resets"""
    torch = torch_cuda
    hits, dones = _roll_and_compare(torch, _mini(4096 + 37), k_steps, rounds=max(3, 70 // k_steps), scale=2.0, scatter=0.5)
    assert dones > 100 and hits > 20
    # replayed normals (noise_z given): the same thing once more, float64 actions
    make = _mini(1000)
    env, twin = make(), make()
    rng = np.random.RandomState(1)
    a = torch.from_numpy(np.stack([env.action_space.sample_batch(1000, rng) for _ in range(5)]).astype(np.float64)).cuda()
    z = torch.from_numpy(rng.standard_normal((5, 1000, 3))).cuda()
    rew, done = env.rollout(a, noise_z=z)
    for k in range(5):
        twin.step(a[k], noise_z=z[k])
        assert torch.equal(rew[k], twin.reward) and torch.equal(done[k], twin.done)
    _compare_envs(torch, env, twin, "replayed normals")


def test_rollout_delays_and_pure_pursuit(torch_cuda):
    torch = torch_cuda
    hits, _ = _roll_and_compare(torch, _mini(2048, control_delay=2, pose_delay=1, state_delay=3), 9, rounds=3, scale=2.0, scatter=0.5)
    hits2, _ = _roll_and_compare(torch, _mini(2048, pose_delay=1, reward_provider_name='continuous_reward_pure_pursuit'), 6, rounds=3, scale=2.0,
                                 scatter=0.5)
    assert hits > 5 and hits2 > 5


def test_rollout_private_maps_and_paths(torch_cuda):
    """BASELINE configs[3]'s shape: private maps and private paths read from global memory"""
    torch = torch_cuda
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    names = ["g8_traj_aisle_c4_00.npz", "g8_traj_aisle_c4_10.npz", "g8_traj_aisle_c4_01.npz", "g8_traj_aisle_c4_11.npz"]
    gs = [np.load(os.path.join(GOLDEN, nm)) for nm in names]
    res = float(gs[0]["resolution"])
    n = 3000
    make = lambda: BatchedPlanEnv([CostMap2D(x["costmap"], res, x["origin"]) for x in gs], [x["path"] for x in gs],
                                  EnvParams(resolution=res, refine_path=False, iteration_timeout=60), n_envs=n, auto_reset=True,
                                  template_of_env=np.arange(n) % 4, map_storage=(256, 256), seed=3)
    hits, dones = _roll_and_compare(torch, make, 24, rounds=4, scatter=0.35)
    assert hits > 20 and dones > 50


def test_rollout_geometry_pool(torch_cuda):
    """in-kernel resets move an env to its next world between the steps of one launch"""
    torch = torch_cuda
    from bc_gym_planning_env_amd import EnvParams, mini_env
    params = mini_env.RandomMiniEnvParams(env_params=EnvParams(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2, iteration_timeout=15))
    pool = mini_env.sample_pool(params, [1, 2, 3, 4, 5], 3)
    make = lambda: mini_env.BatchedRandomMiniEnv(640, params, pool=pool, auto_reset=True, seed=2)
    hits, dones = _roll_and_compare(torch, make, 20, rounds=3, scale=3.0, scatter=0.3)
    assert dones > 1000 and hits > 10


@pytest.mark.parametrize("tuning", [dict(fused=0), dict(defer=0), dict(local_pairs=2)], ids=["two-launch", "single-kernel", "8-wave-workgroups"])
def test_rollout_other_step_forms(torch_cuda, tuning):
    """forms without a rollout kernel are stepped launch by launch (8-wave workgroups: the rollout kernel is the 16-wave one)"""
    torch = torch_cuda
    _roll_and_compare(torch, _mini(2048), 5, rounds=2, tuning=tuning, scale=2.0)
