"""EnvParams delays > 0 and the pure-pursuit reward provider on the GPU: the reference's
recorded trajectories (g11, from the genuine PlanEnv) and a batch against the oracle."""
import os

import numpy as np
import pytest

from util import ATOL, GOLDEN, z_in

pytestmark = pytest.mark.gpu

G11 = ["g11_traj_delay_p1s1.npz", "g11_traj_delay_c2p3s1.npz", "g11_traj_delay_c1_wall.npz", "g11_traj_pp.npz",
       "g11_traj_pp_delay.npz"]


def _env_params(g, **kw):
    from bc_gym_planning_env_amd import EnvParams
    mini = int(g["pure_pursuit"]) == 0
    return EnvParams(goal_spat_dist=0.2 if mini else 1.0, goal_ang_dist=np.pi / 8 if mini else np.pi / 2,
                     resolution=float(g["resolution"]), refine_path=False, control_delay=int(g["control_delay"]),
                     pose_delay=int(g["pose_delay"]), state_delay=int(g["state_delay"]),
                     reward_provider_name='continuous_reward_pure_pursuit' if int(g["pure_pursuit"]) else 'continuous_reward',
                     **kw)


@pytest.mark.parametrize("name", G11)
def test_g11_trajectories(torch_cuda, name):
    """replay of the reference's own steps: true robot state, what State exposes, reward, done, provider state"""
    torch = torch_cuda
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D
    g = np.load(os.path.join(GOLDEN, name))
    n = 3   # replicas must agree with each other, too
    env = BatchedPlanEnv(CostMap2D(g["costmap"], float(g["resolution"]), g["origin"]), g["path"], _env_params(g), n_envs=n)
    assert int(env.state.target_idx[0]) == int(g["init_target_idx"])
    assert float(env.state.min_spat_dist_so_far[0]) == float(g["init_min_dist"])
    pd, sd = int(g["pose_delay"]), int(g["state_delay"])
    for t in range(len(g["actions"])):
        a = np.repeat(g["actions"][t][None], n, axis=0)
        z = np.repeat(z_in(g["z"][t])[None], n, axis=0)
        obs, rew, done, _ = env.step(a, noise_z=z)
        st = env.state.robot.cpu().numpy()
        np.testing.assert_allclose(st[:, 0], g["true_states"][t], rtol=0, atol=ATOL, err_msg="%s step %d" % (name, t))
        assert (st == st[:, :1]).all()
        seen_pose = obs.pose.cpu().numpy()[:, 0]
        np.testing.assert_allclose(seen_pose, g["seen_pose"][t], rtol=0, atol=ATOL)
        seen = (env.state.robot_state_seen if sd else env.state.robot).cpu().numpy()[:, 0]
        np.testing.assert_allclose(seen, g["seen_states"][t], rtol=0, atol=ATOL)
        np.testing.assert_allclose(rew.cpu().numpy(), g["reward"][t], rtol=0, atol=ATOL)
        assert (done.cpu().numpy() == g["done"][t]).all() and (env.state.robot_collided.cpu().numpy() == g["collided"][t]).all()
        assert (env.state.target_idx.cpu().numpy() == g["target_idx"][t]).all()
        np.testing.assert_allclose(env.state.min_spat_dist_so_far.cpu().numpy(), g["min_dist"][t], rtol=0, atol=ATOL)
        if t in (0, 5, 40):
            # per-env view: reference-shaped State / Observation
            o = env.envs[1].observation()
            assert len(o.path) == int(g["obs_path_len"][t])
            np.testing.assert_allclose(o.pose, g["seen_pose"][t], rtol=0, atol=ATOL)
            s = env.envs[1].get_state()
            assert len(s.poses_queue) == min(t + 1, pd) and len(s.robot_state_queue) == min(t + 1, sd)
            assert len(s.control_queue) == min(t + 1, int(g["control_delay"]))


@pytest.mark.parametrize("cfg", [dict(control_delay=1, pose_delay=2, state_delay=3), dict(pose_delay=1, state_delay=1),
                                 dict(pure_pursuit=1), dict(pure_pursuit=1, control_delay=2, pose_delay=1)],
                         ids=lambda c: "-".join("%s%d" % (k[:4], v) for k, v in sorted(c.items())))
def test_batch_with_delays_vs_oracle(torch_cuda, oracle, cfg):
    """2048 envs, short episodes with in-kernel resets (queues restart), collisions, every step against the oracle"""
    torch = torch_cuda
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    g = np.load(os.path.join(GOLDEN, "g8_traj_mini_03.npz"))
    res = float(g["resolution"])
    pp = cfg.get("pure_pursuit", 0)
    delays = {k: v for k, v in cfg.items() if k != "pure_pursuit"}
    params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=res, refine_path=False, iteration_timeout=30,
                       reward_provider_name='continuous_reward_pure_pursuit' if pp else 'continuous_reward', **delays)
    n, steps = 2048, 80
    env = BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], params, n_envs=n, auto_reset=True, seed=3)
    p = oracle.make_params("tricycle", noise=oracle.PLANENV_NOISE, spatial_precision=0.2, angular_precision=np.pi / 8,
                           iteration_timeout=30, reward_provider=pp, **delays)
    ref = oracle.OracleBatch(p, n, g["costmap"], g["origin"], res, g["path"])
    ref.reset_from_paths()
    rng = np.random.RandomState(12)
    # spread the robots: a third of them next to the wall, so that collisions (and their rollbacks) enter the queues
    lethal = np.argwhere(g["costmap"] == 254)
    pick = lethal[rng.randint(0, len(lethal), n)]
    xy = g["origin"][None] + (pick[:, ::-1] + rng.uniform(-12, 12, (n, 2))) * res
    start = np.stack([xy[:, 0], xy[:, 1], rng.uniform(-np.pi, np.pi, n)])
    third = np.arange(n) % 3 == 0
    env.state.robot[0:3, torch.from_numpy(third).cuda()] = torch.from_numpy(start[:, third]).cuda()
    for f in range(3):
        ref.st[f][third] = start[f][third]
    if env.state.pose_seen is not None:
        env.state.pose_seen.copy_(env.state.robot[0:3])
    if env.state.robot_state_seen is not None:
        env.state.robot_state_seen.copy_(env.state.robot)
    ref.obs_pose[:] = np.stack(ref.st[:3], axis=1)
    ref.obs_state[:] = np.stack(ref.st, axis=1)
    zout = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    hits = resets = 0
    for t in range(steps):
        a = env.action_space.sample_batch(n, rng)
        a[:, 0] *= 2.0
        obs, rew, done, _ = env.step(a, noise_z_out=zout)
        ref.step(a.astype(np.float64), z_in(zout.cpu().numpy()), auto_reset=True, threads=8)
        np.testing.assert_array_equal(done.cpu().numpy(), ref.done, err_msg="done, step %d" % t)
        np.testing.assert_array_equal(env.collided_now.cpu().numpy(), ref.collided_now)
        np.testing.assert_array_equal(env.state.target_idx.cpu().numpy(), ref.target_idx)
        np.testing.assert_array_equal(env.state.current_iter.cpu().numpy(), ref.cur_iter)
        np.testing.assert_allclose(env.state.robot.cpu().numpy(), np.stack(ref.st), rtol=0, atol=ATOL)
        np.testing.assert_allclose(rew.cpu().numpy(), ref.reward, rtol=0, atol=ATOL)
        np.testing.assert_allclose(obs.pose.cpu().numpy(), ref.obs_pose.T, rtol=0, atol=ATOL)
        if env.state.robot_state_seen is not None:
            np.testing.assert_allclose(env.state.robot_state_seen.cpu().numpy(), ref.obs_state.T, rtol=0, atol=ATOL)
        hits += int(ref.collided_now.sum())
        resets += int(ref.done.sum())
    assert hits > 50 and resets > n
