"""Long randomised parity run: the HIP step against the oracle, every step, every env, several configurations.
Not collected by pytest (minutes of CPU oracle time); it lives under tests/ because it drives the oracle.
Usage: python tests/soak.py [steps] [n_envs] [seed]   (from the repository root)"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, '.')
import oracle as O
from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams, mini_env

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
G = os.path.join('tests', 'golden')
ATOL = 1e-9
O.build()


def z_in(z):
    return np.where(np.isnan(z), 1e300, z)


def run(tag, env, ref, scale=(2.0, 1.0), geom=False, seen=False):
    rng = np.random.RandomState(seed + 17)
    zout = torch.zeros(env.n_envs, 3, dtype=torch.float64, device="cuda")
    t0 = time.time()
    hits = resets = goals = 0
    worst = 0.0
    for t in range(steps):
        a = env.action_space.sample_batch(env.n_envs, rng) * np.array(scale, dtype=np.float32)
        obs, rew, done, _ = env.step(a, noise_z_out=zout)
        ref.step(a.astype(np.float64), z_in(zout.cpu().numpy()), auto_reset=True, threads=16)
        st = env.state.robot.cpu().numpy()
        assert (done.cpu().numpy() == ref.done).all(), (tag, t, "done")
        assert (env.collided_now.cpu().numpy() == ref.collided_now).all(), (tag, t, "collided_now")
        assert (env.state.target_idx.cpu().numpy() == ref.target_idx).all(), (tag, t, "target_idx")
        assert (env.state.current_iter.cpu().numpy() == ref.cur_iter).all(), (tag, t, "iter")
        d = np.abs(st - np.stack(ref.st)).max()
        dr = np.abs(rew.cpu().numpy() - ref.reward).max()
        worst = max(worst, d, dr)
        assert d <= ATOL and dr <= ATOL, (tag, t, d, dr)
        if geom:
            assert (env.geom_of_env.cpu().numpy() == ref.geom).all(), (tag, t, "geom")
        if seen:
            assert np.abs(obs.pose.cpu().numpy() - ref.obs_pose.T).max() <= ATOL, (tag, t, "seen pose")
        hits += int(ref.collided_now.sum())
        resets += int(ref.done.sum())
        goals += int((ref.reward == 1.0).sum())
    print("%-28s %d steps x %d envs ok  (collisions %d, resets %d, way points reached %d, max |diff| %.2e)  %.0f s" % (
        tag, steps, env.n_envs, hits, resets, goals, worst, time.time() - t0), flush=True)


# 1. shared map, tricycle + noise (the metric configuration), a different geometry per soak seed
names = sorted(f for f in os.listdir(G) if f.startswith("g8_traj_mini_") and "nonoise" not in f)
g = np.load(os.path.join(G, names[seed % len(names)]))
res = float(g["resolution"])
params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=res, refine_path=False, iteration_timeout=300)
env = BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], params, n_envs=n, auto_reset=True, seed=seed)
p = O.make_params("tricycle", noise=O.PLANENV_NOISE, spatial_precision=0.2, angular_precision=np.pi / 8, iteration_timeout=300)
ref = O.OracleBatch(p, n, g["costmap"], g["origin"], res, g["path"])
ref.reset_from_paths()
run("shared map " + names[seed % len(names)][8:-4], env, ref)
del env

# 2. geometry pool with short episodes
mp = mini_env.RandomMiniEnvParams(env_params=EnvParams(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2, iteration_timeout=150))
pool = mini_env.sample_pool(mp, list(range(1000 * seed, 1000 * seed + 64)), 4)
env = mini_env.BatchedRandomMiniEnv(n, mp, pool=pool, auto_reset=True, seed=seed)
maps = np.stack([c.get_data() for c in pool.costmaps])
origins = np.stack([c.get_origin() for c in pool.costmaps])
paths = env._paths
pbuf = np.zeros((len(paths), max(len(q) for q in paths), 3))
for k, q in enumerate(paths):
    pbuf[k, :len(q)] = q
prev = np.empty_like(pool.next_geom)
prev[pool.next_geom] = np.arange(len(pool.next_geom), dtype=np.int32)
p = O.make_params("tricycle", noise=O.PLANENV_NOISE, spatial_precision=0.2, angular_precision=np.pi / 8, iteration_timeout=150)
ref = O.OracleBatch(p, n, maps, origins, mp.env_params.resolution, pbuf, lens=[len(q) for q in paths],
                    geom=prev[env.geom_of_env.cpu().numpy()], next_geom=pool.next_geom)
ref.reset_from_paths()
ref.reset_all_to_geom(advance=True)
run("geometry pool 64 x 4", env, ref, scale=(3.0, 1.0), geom=True)
del env

# 3. delays + pure pursuit on the aisle map
ga = np.load(os.path.join(G, "g8_traj_aisle_default.npz"))
res = float(ga["resolution"])
for tag, kw, pp in (("delays c2 p1 s3", dict(control_delay=2, pose_delay=1, state_delay=3), 0),
                    ("pure pursuit + delays", dict(control_delay=1, pose_delay=2), 1)):
    params = EnvParams(resolution=res, refine_path=False, iteration_timeout=200,
                       reward_provider_name='continuous_reward_pure_pursuit' if pp else 'continuous_reward', **kw)
    env = BatchedPlanEnv(CostMap2D(ga["costmap"], res, ga["origin"]), ga["path"], params, n_envs=n, auto_reset=True, seed=seed)
    p = O.make_params("tricycle", noise=O.PLANENV_NOISE, iteration_timeout=200, reward_provider=pp, **kw)
    ref = O.OracleBatch(p, n, ga["costmap"], ga["origin"], res, ga["path"])
    ref.reset_from_paths()
    run(tag, env, ref, scale=(3.0, 1.0), seen=True)
    del env

# 4. diff-drive, no noise, small map
gd = np.load(os.path.join(G, "g8dd_traj_mini64_00.npz"))
res = float(gd["resolution"])
params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=res, refine_path=False, iteration_timeout=250,
                   robot_name='industrial_diffdrive_v1')
env = BatchedPlanEnv(CostMap2D(gd["costmap"], res, gd["origin"]), gd["path"], params, n_envs=n, noise_parameters=None,
                     auto_reset=True, seed=seed)
p = O.make_params("diffdrive", noise=None, spatial_precision=0.2, angular_precision=np.pi / 8, iteration_timeout=250)
ref = O.OracleBatch(p, n, gd["costmap"], gd["origin"], res, gd["path"])
ref.reset_from_paths()
run("diff-drive 64x64", env, ref, scale=(1.0, 1.0))
print("soak ok")

# 5. egocentric views: thousands of random poses on random-byte costmaps, shared and private, against the oracle
from bc_gym_planning_env_amd.ops import NativeOps
rng = np.random.RandomState(seed + 99)
for trial in range(3):
    rows, cols = rng.randint(40, 200), rng.randint(40, 200)
    res = float(rng.choice([0.03, 0.05, 10. / 256]))
    data = rng.randint(0, 256, (rows, cols)).astype(np.uint8)
    org = rng.uniform(-3, 0, 2)
    ops = NativeOps()
    ops.set_costmap(data, org, res)
    m = 3000
    poses = np.stack([rng.uniform(org[0] - 2, org[0] + cols * res + 2, m), rng.uniform(org[1] - 2, org[1] + rows * res + 2, m),
                      rng.uniform(-10, 10, m)], axis=1)
    for worg, wsize, border in (((-0.5, -2.0), (3.5, 4.0), 0), ((-1.3, -0.4), (2.21, 1.07), 200), (None, None, 9)):
        got = ops.extract_egocentric_costmap(poses, worg, wsize, border).cpu().numpy()
        bad = 0
        for k in range(m):
            want = O.extract_egocentric(data, org, res, poses[k], worg, wsize, border)
            bad += int((want != got[k]).sum())
        assert bad == 0, (trial, worg, bad)
    print("egocentric views  map %dx%d @%.4f: %d poses x 3 windows identical to the oracle" % (rows, cols, res, m), flush=True)
    ops.close()
print("soak ok (egocentric)")
