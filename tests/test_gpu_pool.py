"""Geometry pool on the GPU: the batched sampler with its GPU acceptance test against the
reference's worlds, and stepping / resetting along pool chains against the oracle."""
import os

import numpy as np
import pytest

from util import ATOL, GOLDEN, z_in

pytestmark = pytest.mark.gpu

MODES = [dict(), dict(defer=0), dict(cull=0, exact_mode=1), dict(cull=0, exact_mode=2), dict(dense_threshold=0),
         dict(dense_threshold=64), dict(local_pairs=2), dict(local_pairs=1)]


def _ids(m):
    return "-".join("%s%d" % (k[:4], v) for k, v in sorted(m.items())) or "default"


def test_gpu_sampler_reproduces_reference_worlds(torch_cuda):
    from bc_gym_planning_env_amd import mini_env
    g = np.load(os.path.join(GOLDEN, "g9_mini_geometry.npz"))
    seeds, episodes = [int(s) for s in g["seeds"]], g["worlds"].shape[1]
    pool = mini_env.sample_pool(None, seeds, episodes)
    cols = int(g["map_shape"][1])
    for k, w in enumerate(pool.worlds):
        ref = g["worlds"][k // episodes, k % episodes]
        mine = np.concatenate([w.start_pos, w.end_pos, w.obstacle_a, w.obstacle_o, w.obstacle_b, [w.h, w.w]])
        assert (mine == ref).all(), k
        want = np.unpackbits(g["maps"][k], axis=1)[:, :cols].astype(bool)
        assert ((pool.costmaps[k].get_data() == 254) == want).all()


def _pool_and_oracle(oracle, mini_env, n, n_chains, episodes, timeout, next_geom=True, pure_pursuit=0, **delays):
    from bc_gym_planning_env_amd import EnvParams
    params = mini_env.RandomMiniEnvParams(
        env_params=EnvParams(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2, iteration_timeout=timeout,
                             reward_provider_name='continuous_reward_pure_pursuit' if pure_pursuit else 'continuous_reward',
                             **delays))
    pool = mini_env.sample_pool(params, list(range(100, 100 + n_chains)), episodes)
    env = mini_env.BatchedRandomMiniEnv(n, params, pool=pool, auto_reset=True, seed=11,
                                        draw_new_turn_on_reset=next_geom)
    maps = np.stack([c.get_data() for c in pool.costmaps])
    origins = np.stack([c.get_origin() for c in pool.costmaps])
    paths = env._paths
    max_len = max(len(p) for p in paths)
    pbuf = np.zeros((len(paths), max_len, 3))
    for k, p in enumerate(paths):
        pbuf[k, :len(p)] = p
    i = np.arange(n)
    geom0 = (i % n_chains) * episodes + (i // n_chains) % episodes
    p = oracle.make_params("tricycle", noise=oracle.PLANENV_NOISE, spatial_precision=0.2, angular_precision=np.pi / 8,
                           iteration_timeout=timeout, reward_provider=pure_pursuit, **delays)
    ref = oracle.OracleBatch(p, n, maps, origins, params.env_params.resolution, pbuf, lens=[len(q) for q in paths],
                             geom=geom0, next_geom=pool.next_geom if next_geom else None)
    ref.reset_from_paths()
    ref.reset_all_to_geom(advance=True)    # the constructor's reset() moves every env to world 1 of its chain
    return env, ref, pool


def _compare(env, ref, t=None):
    np.testing.assert_array_equal(env.geom_of_env.cpu().numpy(), ref.geom, err_msg="geom step %s" % t)
    np.testing.assert_array_equal(env.state.target_idx.cpu().numpy(), ref.target_idx)
    np.testing.assert_array_equal(env.state.current_iter.cpu().numpy(), ref.cur_iter)
    np.testing.assert_array_equal(env.state.robot_collided.cpu().numpy(), ref.collided)
    np.testing.assert_allclose(env.state.robot.cpu().numpy(), np.stack(ref.st), rtol=0, atol=ATOL)
    np.testing.assert_allclose(env.state.min_spat_dist_so_far.cpu().numpy(), ref.min_dist, rtol=0, atol=ATOL)


@pytest.mark.parametrize("mode", MODES, ids=_ids)
def test_pool_steps_and_resets_vs_oracle(torch_cuda, oracle, mode):
    """Short episodes (timeout 25) over a small pool: hundreds of resets, each moving an env to its next world."""
    torch = torch_cuda
    from bc_gym_planning_env_amd import mini_env
    n, steps = 768, 90
    env, ref, pool = _pool_and_oracle(oracle, mini_env, n, n_chains=7, episodes=3, timeout=25)
    env.set_tuning(**mode)
    _compare(env, ref, "init")
    rng = np.random.RandomState(4)
    zout = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    resets = hits = 0
    for t in range(steps):
        a = env.action_space.sample_batch(n, rng)
        a[:, 0] *= 3.0
        env.step(a, noise_z_out=zout)
        ref.step(a.astype(np.float64), z_in(zout.cpu().numpy()), auto_reset=True, threads=8)
        np.testing.assert_array_equal(env.done.cpu().numpy(), ref.done, err_msg="done step %d" % t)
        np.testing.assert_array_equal(env.collided_now.cpu().numpy(), ref.collided_now)
        np.testing.assert_allclose(env.reward.cpu().numpy(), ref.reward, rtol=0, atol=ATOL)
        _compare(env, ref, t)
        resets += int(ref.done.sum())
        hits += int(ref.collided_now.sum())
    assert resets > 2 * n and hits > 20
    assert len(np.unique(ref.geom)) == len(pool)        # every world of the pool was in use at the end


@pytest.mark.parametrize("cfg", [dict(pose_delay=1, state_delay=1, control_delay=2), dict(pure_pursuit=1, pose_delay=1)],
                         ids=["delays", "pure-pursuit"])
def test_pool_with_delays_and_pure_pursuit_vs_oracle(torch_cuda, oracle, cfg):
    """the geometry pool together with the delay queues / the second reward provider (general step kernel): resets move
    an env to its next world AND restart its FIFOs"""
    torch = torch_cuda
    from bc_gym_planning_env_amd import mini_env
    n, steps = 512, 70
    env, ref, pool = _pool_and_oracle(oracle, mini_env, n, n_chains=5, episodes=3, timeout=20, **cfg)
    _compare(env, ref, "init")
    rng = np.random.RandomState(8)
    zout = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    resets = 0
    for t in range(steps):
        a = env.action_space.sample_batch(n, rng)
        a[:, 0] *= 3.0
        obs, rew, done, _ = env.step(a, noise_z_out=zout)
        ref.step(a.astype(np.float64), z_in(zout.cpu().numpy()), auto_reset=True, threads=8)
        np.testing.assert_array_equal(done.cpu().numpy(), ref.done, err_msg="done step %d" % t)
        np.testing.assert_allclose(rew.cpu().numpy(), ref.reward, rtol=0, atol=ATOL)
        np.testing.assert_allclose(obs.pose.cpu().numpy(), ref.obs_pose.T, rtol=0, atol=ATOL)
        _compare(env, ref, t)
        resets += int(ref.done.sum())
    assert resets > 2 * n


def test_pool_without_successor_table_keeps_worlds(torch_cuda, oracle):
    torch = torch_cuda
    from bc_gym_planning_env_amd import mini_env
    n = 256
    env, ref, pool = _pool_and_oracle(oracle, mini_env, n, n_chains=5, episodes=2, timeout=10, next_geom=False)
    g0 = env.geom_of_env.cpu().numpy().copy()
    rng = np.random.RandomState(1)
    zout = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    for t in range(25):
        a = env.action_space.sample_batch(n, rng)
        env.step(a, noise_z_out=zout)
        ref.step(a.astype(np.float64), z_in(zout.cpu().numpy()), auto_reset=True, threads=4)
        _compare(env, ref, t)
    np.testing.assert_array_equal(env.geom_of_env.cpu().numpy(), g0)


def test_pool_masked_reset_and_views(torch_cuda, oracle):
    """reset(mask) advances only the masked envs; envs[i] hands back the world the env is currently on."""
    torch = torch_cuda
    from bc_gym_planning_env_amd import mini_env
    n = 128
    env, ref, pool = _pool_and_oracle(oracle, mini_env, n, n_chains=4, episodes=4, timeout=1200)
    before = env.geom_of_env.cpu().numpy().copy()
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
    mask[::3] = 1
    env.reset(mask)
    after = env.geom_of_env.cpu().numpy()
    m = mask.cpu().numpy().astype(bool)
    np.testing.assert_array_equal(after[~m], before[~m])
    np.testing.assert_array_equal(after[m], pool.next_geom[before[m]])
    for i in (0, 1, 3, 127):
        st = env.envs[i].get_state()
        k = int(after[i])
        assert st.costmap is pool.costmaps[k]
        np.testing.assert_array_equal(st.original_path, env._paths[k])
        np.testing.assert_array_equal(st.pose, env._paths[k][0])


def test_pool_full_size_two_kernel_path(torch_cuda, oracle):
    """65 536 envs over a 64 x 4 pool: the fast + pending kernel pair with per-env geometry entries; every env is
    checked against the oracle."""
    torch = torch_cuda
    from bc_gym_planning_env_amd import mini_env
    n, steps = 65536, 12
    env, ref, pool = _pool_and_oracle(oracle, mini_env, n, n_chains=64, episodes=4, timeout=8)
    rng = np.random.RandomState(2)
    zout = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    resets = 0
    for t in range(steps):
        a = env.action_space.sample_batch(n, rng)
        a[:, 0] *= 3.0
        env.step(a, noise_z_out=zout)
        ref.step(a.astype(np.float64), z_in(zout.cpu().numpy()), auto_reset=True, threads=16)
        np.testing.assert_array_equal(env.done.cpu().numpy(), ref.done, err_msg="done step %d" % t)
        np.testing.assert_array_equal(env.collided_now.cpu().numpy(), ref.collided_now)
        _compare(env, ref, t)
        resets += int(ref.done.sum())
    assert resets >= n


def test_device_sampler_matches_host_and_reference(torch_cuda):
    """bcp_sample_mini_worlds (one wavefront per RandomState stream: MT19937, rejection sampler, walls, acceptance test
    on the device) against the reference's own worlds (g9) and against the host sampler on more streams"""
    from bc_gym_planning_env_amd import mini_env
    g = np.load(os.path.join(GOLDEN, "g9_mini_geometry.npz"))
    seeds, episodes = [int(s) for s in g["seeds"]], g["worlds"].shape[1]
    pool = mini_env.sample_pool_device(None, seeds, episodes)
    cols = int(g["map_shape"][1])
    exact = 0
    for k, w in enumerate(pool.worlds):
        ref = g["worlds"][k // episodes, k % episodes]
        mine = np.concatenate([w.start_pos, w.end_pos, w.obstacle_a, w.obstacle_o, w.obstacle_b, [w.h, w.w]])
        np.testing.assert_allclose(mine, ref, rtol=0, atol=1e-12, err_msg="world %d" % k)
        exact += int((mine == ref).all())
        want = np.unpackbits(g["maps"][k], axis=1)[:, :cols].astype(bool)
        assert ((pool.costmaps[k].get_data() == 254) == want).all(), k
        assert set(np.unique(pool.costmaps[k].get_data())) <= {0, 254}
    assert exact >= len(pool.worlds) // 2       # (device sin / cos / atan2 differ from numpy's in the last bit now and then)
    # more streams, longer chains: host sampler (bit-exact to the reference) vs device sampler
    seeds = list(range(500, 564))
    a = mini_env.sample_pool(None, seeds, 6)
    b = mini_env.sample_pool_device(None, seeds, 6)
    same_maps = 0
    for wa, wb, ca, cb in zip(a.worlds, b.worlds, a.costmaps, b.costmaps):
        for fa, fb in ((wa.start_pos, wb.start_pos), (wa.end_pos, wb.end_pos), (wa.obstacle_o, wb.obstacle_o),
                       (wa.obstacle_a, wb.obstacle_a), (wa.obstacle_b, wb.obstacle_b)):
            np.testing.assert_allclose(fa, fb, rtol=0, atol=1e-12)
        same_maps += int((ca.get_data() == cb.get_data()).all())
    assert same_maps >= len(a.worlds) - 1       # a last-bit difference can move a wall end by a pixel, very rarely
    # and the pool drives an env like any other
    env = mini_env.BatchedRandomMiniEnv(256, pool=b, auto_reset=True)
    env.step(env.action_space.sample_batch(256))
    env.check_errors()


def test_device_resident_pool(torch_cuda, oracle):
    """sample_pool_device(keep_on_device=True): refined paths and initial reward states from the device agree with the
    host's numpy ones, and an env built straight from the device tensors steps like the oracle"""
    torch = torch_cuda
    from bc_gym_planning_env_amd import EnvParams, host_init, mini_env
    params = mini_env.RandomMiniEnvParams(
        env_params=EnvParams(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2, iteration_timeout=20))
    seeds, episodes = list(range(40, 72)), 3
    dp = mini_env.sample_pool_device(params, seeds, episodes, keep_on_device=True)
    hp = mini_env.sample_pool(params, seeds, episodes)
    assert len(dp) == len(hp) == 96 and (dp.next_geom == hp.next_geom).all()
    rp = params.env_params.reward_provider_params
    flips = 0
    for k in range(len(dp)):
        # the device's paths against numpy's refine_path of the DEVICE's worlds: a start / end coordinate that differs
        # from the host sampler's in its last bit can move int(d / path_delta) across an integer (the circle method puts
        # start and end exactly 3.5 m = 70 path_delta apart), i.e. give a path with one way point more or less
        w = dp.worlds[k]
        flips += int(len(host_init.refine_path(hp.paths[k], params.env_params.path_delta)) != int(dp.lens[k]))
        np.testing.assert_allclose(np.concatenate([w.start_pos, w.end_pos]),
                                   np.concatenate([hp.worlds[k].start_pos, hp.worlds[k].end_pos]), rtol=0, atol=1e-12)
        want = host_init.refine_path(np.array([w.start_pos, w.end_pos]), params.env_params.path_delta)
        got = dp.paths[k]
        assert got.shape == want.shape, k
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
        md, ti = host_init.initial_reward_state(want, rp)
        assert int(dp.init[k, 1]) == ti and abs(float(dp.init[k, 0]) - md) < 1e-12
        assert (dp.costmaps[k].get_data() == hp.costmaps[k].get_data()).all()
    assert flips <= 3
    n = 384
    env = mini_env.BatchedRandomMiniEnv(n, params, pool=dp, auto_reset=True, seed=4)
    maps = dp.maps.cpu().numpy()
    origins = np.tile(dp.origin, (len(dp), 1))
    lens = dp.lens.cpu().numpy()
    pbuf = dp.path_points.cpu().numpy()
    prev = np.empty_like(dp.next_geom)
    prev[dp.next_geom] = np.arange(len(dp.next_geom), dtype=np.int32)
    p = oracle.make_params("tricycle", noise=oracle.PLANENV_NOISE, spatial_precision=0.2, angular_precision=np.pi / 8,
                           iteration_timeout=20)
    ref = oracle.OracleBatch(p, n, maps, origins, params.env_params.resolution, pbuf, lens=lens,
                             geom=prev[env.geom_of_env.cpu().numpy()], next_geom=dp.next_geom)
    ref.reset_from_paths()
    ref.reset_all_to_geom(advance=True)
    _compare(env, ref, "init")
    rng = np.random.RandomState(6)
    zout = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    for t in range(50):
        a = env.action_space.sample_batch(n, rng)
        a[:, 0] *= 3.0
        env.step(a, noise_z_out=zout)
        ref.step(a.astype(np.float64), z_in(zout.cpu().numpy()), auto_reset=True, threads=8)
        np.testing.assert_array_equal(env.done.cpu().numpy(), ref.done)
        _compare(env, ref, t)
    st = env.envs[5].get_state()      # host views are fetched on demand
    k = int(env.geom_of_env[5])
    assert (st.costmap.get_data() == maps[k]).all() and (st.original_path == pbuf[k, :lens[k]]).all()


def _endless_setup(mini_env, n, episodes, long_episodes, timeout):
    from bc_gym_planning_env_amd import EnvParams
    params = mini_env.RandomMiniEnvParams(
        env_params=EnvParams(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2, iteration_timeout=timeout))
    seeds = list(range(300, 300 + n))
    env = mini_env.BatchedRandomMiniEnv(n, params, seeds=seeds, episodes=episodes, endless=True, auto_reset=True, seed=9)
    # the same streams sampled `long_episodes` worlds deep in one go: world j of stream c is entry c * long_episodes + j
    deep = mini_env.sample_pool_device(params, seeds, long_episodes, keep_on_device=True)
    return params, env, deep


def _check_ring_entries(env, deep, world_of_env, generated):
    """every ring entry holds the world of its stream that the bookkeeping says, bit for bit, with its path and reward
    state; the env sits on the entry of its world"""
    E, K = env.pool.episodes, deep.episodes
    pw, dw = env.pool.world_params.cpu().numpy(), deep.world_params.cpu().numpy()
    pm, dm = env.pool.maps.cpu().numpy(), deep.maps.cpu().numpy()
    pp, dpth = env.pool.path_points.cpu().numpy(), deep.path_points.cpu().numpy()
    pl, dl = env.pool.lens.cpu().numpy(), deep.lens.cpu().numpy()
    pi, di = env.pool.init.cpu().numpy(), deep.init.cpu().numpy()
    init = env._initial_state
    ix, imd, iti = init.robot[0:3].cpu().numpy(), init.min_spat_dist_so_far.cpu().numpy(), init.target_idx.cpu().numpy()
    geom = env.geom_of_env.cpu().numpy()
    nxt = env._keep["next_geom"].cpu().numpy()
    for c in range(env.n_envs):
        assert geom[c] == c * E + world_of_env[c] % E, c
        for j in range(generated[c] - E, generated[c]):
            g, d = c * E + j % E, c * K + j
            assert (pw[g] == dw[d]).all() and (pm[g] == dm[d]).all(), (c, j)
            assert pl[g] == dl[d] and (pp[g, :pl[g]] == dpth[d, :dl[d]]).all() and (pi[g] == di[d]).all(), (c, j)
            assert (ix[:, g] == dpth[d, 0]).all() and imd[g] == di[d, 0] and iti[g] == int(di[d, 1]), (c, j)
            assert nxt[g] == (g if j == generated[c] - 1 else c * E + (j + 1) % E), (c, j)
    # the 1-bit tiles of the step's outer test follow the distance fields through every re-sampling
    import torch
    near, t_out = env.near_field(0, env.n_envs * E)
    assert torch.equal(near, env.distance_field(0, env.n_envs * E)[0] < t_out)


@pytest.mark.parametrize("overlap,E", [(False, 3), (True, 5), ("masked", 5)], ids=["in-order", "side-stream", "cu-masked-side-stream"])
def test_endless_pool_follows_the_streams(torch_cuda, overlap, E):
    """BatchedRandomMiniEnv(endless=True) + refresh(): every env walks through the worlds of its own RandomState
    stream in order, never an old one again; re-sampled entries (costmap, path, initial state, and the handle's
    distance fields / path index behind them) make the env step exactly like one built on a pool that holds the
    whole stream up front.  Both with the sampling in stream order and with it on a side stream under the steps."""
    torch = torch_cuda
    from bc_gym_planning_env_amd import mini_env
    n, K = 96, 40
    params, env, deep = _endless_setup(mini_env, n, E, K, timeout=12)
    if overlap == "masked":   # the refresh on a stream that may only use half of the compute units (bcp_side_stream)
        env.side_cu_percent = 50
        overlap = True
    ref = mini_env.BatchedRandomMiniEnv(n, params, pool=deep, auto_reset=True, seed=9)
    world = np.ones(n, dtype=np.int64)          # both constructors end with the reset() that moves on to world 1
    generated = np.full(n, E, dtype=np.int64)   # worlds planned so far ...
    reach = generated.copy()                    # ... and released: an env can be on worlds < reach
    _check_ring_entries(env, deep, world, generated)
    rng = np.random.RandomState(2)
    z = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    resampled, planned = 0, None
    for t in range(150):
        a = env.action_space.sample_batch(n, rng)
        a[:, 0] *= 3.0
        if t in (90, 120) and E == 5:
            # the refreshes so far only made the 1-bit tiles of the re-sampled worlds (near_dilate_kernel: nothing else is read
            # under the single-launch step); the two-launch form reads the uint8 fields -- they are brought up to date on demand
            env.set_tuning(fused=0 if t == 90 else 1)
        env.step(a, noise_z_out=z)
        ref.step(a, noise_z=z)
        for name in ("reward", "done", "collided_now"):
            assert torch.equal(getattr(env, name), getattr(ref, name)), (name, t)
        assert torch.equal(env.state.robot, ref.state.robot) and torch.equal(env.state.target_idx, ref.state.target_idx), t
        assert torch.equal(env.state.min_spat_dist_so_far, ref.state.min_spat_dist_so_far), t
        world += env.done.cpu().numpy().astype(np.int64)
        assert (world <= reach - 1).all() and world.max() < K     # nobody had to wait at a guard
        if t % 2 == 1:
            got = env.refresh(check=True, overlap=overlap)
            now = (int((world - (generated - E)).sum()), int((world == generated - 1).sum()))
            if overlap:           # this call released the previous round and planned the next one
                assert got == planned
                reach = generated.copy()
                generated, planned = world + E, now
            else:
                assert got == now
                generated = world + E
                reach = generated.copy()
            resampled += now[0]
            if t % 20 == 19:
                env.finish_refresh(check=True)
                reach, planned = generated.copy(), None
                _check_ring_entries(env, deep, world, generated)
    env.finish_refresh(check=True)
    _check_ring_entries(env, deep, world, generated)
    assert resampled > 5 * n and world.min() >= 5


def test_endless_pool_guard_holds_envs_back(torch_cuda):
    """without refresh() an env never wraps onto an old world: it repeats its newest one; the next refresh() reports it
    and the stream goes on"""
    torch = torch_cuda
    from bc_gym_planning_env_amd import mini_env
    n, E, K = 8, 2, 6
    params, env, deep = _endless_setup(mini_env, n, E, K, timeout=3)
    a = np.zeros((n, 2))
    for _ in range(10):                         # three timeouts, no refresh: everybody stays on world 1
        env.step(a)
    assert (env.geom_of_env.cpu().numpy() == np.arange(n) * E + 1).all()
    assert env.refresh(check=True) == (n, n)    # world 2 replaces world 0 behind every env
    world, generated = np.ones(n, dtype=np.int64), np.full(n, E + 1, dtype=np.int64)
    _check_ring_entries(env, deep, world, generated)
    for _ in range(3):
        env.step(a)
    _check_ring_entries(env, deep, world + 1, generated)
    with pytest.raises(RuntimeError):
        mini_env.BatchedRandomMiniEnv(4, params, n_chains=4, episodes=2).refresh()


def test_near_field_of_a_shared_map(torch_cuda):
    """shared 183 x 183 map: the padded field is 379 cells wide -- not a multiple of the 32-cell tiles"""
    torch = torch_cuda
    import bench
    env, _ = bench.make_env(256, 0, 0, 5)
    field, pad, clamp = env.distance_field(0, 1)
    near, t_out = env.near_field(0, 1)
    assert field.shape[2] % 32 != 0 and 0 < t_out <= clamp
    assert torch.equal(near, field < t_out) and bool(near.any()) and not bool(near.all())


def test_distance_fields_lds_kernel_vs_two_pass_vs_scipy(torch_cuda):
    """the distance fields behind the pose pre-classification: the LDS-resident transform (pool / private maps), the
    two-pass global-memory kernels and scipy's exact EDT agree cell for cell"""
    torch = torch_cuda
    from scipy import ndimage
    from bc_gym_planning_env_amd import mini_env
    pool = mini_env.sample_pool_device(None, list(range(7, 19)), 2, keep_on_device=True)
    env = mini_env.BatchedRandomMiniEnv(24, pool=pool)
    g_n = len(pool)
    fast, pad, clamp = env.distance_field(0, g_n)
    env.set_tuning(edt_lds=0)
    env.set_costmap_tensors(env._keep["map"], env._keep["origins"], env.resolution)
    slow, pad2, clamp2 = env.distance_field(0, g_n)
    assert (pad, clamp) == (pad2, clamp2) and torch.equal(fast, slow)
    near, t_out = env.near_field(0, g_n)          # the 1-bit tiles the step's outer test reads
    assert 0 < t_out <= clamp and torch.equal(near, slow < t_out)
    maps = pool.maps.cpu().numpy()
    for k in (0, 5, g_n - 1):
        free = np.pad(maps[k] != 254, pad, constant_values=True)
        want = np.minimum(np.floor(ndimage.distance_transform_edt(free) + 1e-9), clamp).astype(np.uint8)
        assert (fast[k].cpu().numpy() == want).all(), k
    # and a map with lethal cells right at its border and in every row (all window edge cases of the bit scans)
    rng = np.random.RandomState(0)
    dense = np.where(rng.rand(g_n, *maps.shape[1:]) < 0.02, 254, 0).astype(np.uint8)
    dense[:, 0, 0] = dense[:, -1, -1] = dense[:, 0, -1] = 254
    env.set_tuning(edt_lds=1)
    dmap = torch.from_numpy(dense).cuda()
    env.set_costmap_tensors(dmap, env._keep["origins"], env.resolution)
    fast = env.distance_field(0, g_n)[0].cpu().numpy()
    env.set_tuning(edt_lds=0)
    env.set_costmap_tensors(dmap, env._keep["origins"], env.resolution)
    assert (env.distance_field(0, g_n)[0].cpu().numpy() == fast).all()
    near, t_out = env.near_field(0, g_n)
    assert (near.cpu().numpy() == (fast < t_out)).all()
    for k in (1, g_n - 2):
        free = np.pad(dense[k] != 254, pad, constant_values=True)
        want = np.minimum(np.floor(ndimage.distance_transform_edt(free) + 1e-9), clamp).astype(np.uint8)
        assert (fast[k] == want).all(), k


@pytest.mark.parametrize("density", [0.0005, 0.02, 0.5])
def test_near_tiles_by_dilation_vs_thresholded_field(torch_cuda, density):
    """near_dilate_kernel (what a pool refresh runs under the single-launch step: the lethal mask dilated by the sample
    disc, no uint8 field) writes the very words near_tiles_kernel gets by thresholding the distance field -- padding bits
    and rows included --, on sampled worlds and on random maps with lethal cells in the corners and along the edges"""
    torch = torch_cuda
    from bc_gym_planning_env_amd import mini_env
    pool = mini_env.sample_pool_device(None, list(range(3, 11)), 2, keep_on_device=True)
    env = mini_env.BatchedRandomMiniEnv(16, pool=pool)
    g_n = len(pool)
    rng = np.random.RandomState(int(density * 1e4))
    maps = pool.maps.cpu().numpy().copy()
    half = g_n // 2
    maps[half:] = np.where(rng.rand(g_n - half, *maps.shape[1:]) < density, 254, 0).astype(np.uint8)
    maps[half:, 0, 0] = maps[half:, -1, -1] = maps[half, 0, -1] = maps[half, -1, 0] = 254
    maps[half + 1, :, 0] = maps[half + 1, 0, :] = 254
    maps[half + 2, :, -1] = maps[half + 2, -1, :] = 254
    dmap = torch.from_numpy(maps).cuda()
    radii, base_res = set(), env.resolution
    for scale in (1.0, 4.0, 0.6, 0.45):   # the same cells read at other resolutions: other radii of the sample disc (t_out)
        res = base_res * scale
        env.set_tuning(near_dilate=0)
        env.set_costmap_tensors(dmap, env._keep["origins"], res)
        want, t_out = env.near_field(0, g_n, raw=True)
        cells = env.near_field(0, g_n)[0]
        assert torch.equal(cells, env.distance_field(0, g_n)[0] < t_out) and bool(cells.any())
        assert density > 0.1 or not bool(cells.all())
        env.set_tuning(near_dilate=2)
        env.set_costmap_tensors(dmap, env._keep["origins"], res)
        got, t2 = env.near_field(0, g_n, raw=True)
        assert t2 == t_out and torch.equal(got, want), (scale, t_out)
        radii.add(t_out)
    assert len(radii) >= 3 and min(radii) <= 10 and max(radii) >= 24, radii
