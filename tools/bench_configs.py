"""Informational timings of the other BASELINE.json configs (they are parity-test cases, not bench lines):
C2 4096 envs diff-drive shared 64x64;  C4 65536 envs AisleTurn, private 256x256 costmaps + private paths."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, '.')
from bc_gym_planning_env_amd import _lib
if os.environ.get('BCP_LIB'):   # A/B of kernel variants: another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ['BCP_LIB'])
from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
G = os.path.join('tests', 'golden')

def run(env, n, steps, lo, hi, tag, bytes_per_env):
    rng = np.random.RandomState(0)
    pool = torch.from_numpy(np.stack([rng.uniform(lo, hi, (n, 2)).astype(np.float32) for _ in range(8)])).cuda()
    env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, n).astype(np.int32)).cuda())
    for k in range(1200):
        env.step(pool[k % 8])
    torch.cuda.synchronize()
    ms = env.time_steps(pool[0], steps)
    coll = float(env.collided_now.float().mean())
    print("%s: %d envs  %.3f ms/step  %.3e env-steps/s  algorithmic %.1f GB/s  (collisions/step %.4f, done %.4f)" % (
        tag, n, ms, n / ms * 1e3, bytes_per_env * n / ms / 1e6, coll, float(env.done.float().mean())), flush=True)

# ---- C2
g = np.load(os.path.join(G, 'g8dd_traj_mini64_00.npz'))
res = float(g['resolution'])
params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=res, refine_path=False,
                   robot_name='industrial_diffdrive_v1')
env = BatchedPlanEnv(CostMap2D(g['costmap'], res, g['origin']), g['path'], params, n_envs=4096, noise_parameters=None,
                     auto_reset=True)
run(env, 4096, 200, [0.105, -np.pi / 2], [0.524, np.pi / 2], 'C2 diff-drive shared 64x64', 57 + 57 + 8 + 9)
del env
# ---- C4
names = ['g8_traj_aisle_c4_00.npz', 'g8_traj_aisle_c4_10.npz', 'g8_traj_aisle_c4_01.npz', 'g8_traj_aisle_c4_11.npz']
gs = [np.load(os.path.join(G, nm)) for nm in names]
res = float(gs[0]['resolution'])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
idx = np.arange(n) % 4
params = EnvParams(resolution=res, refine_path=False)
cms = [CostMap2D(np.pad(x['costmap'], ((0, 256 - x['costmap'].shape[0]), (0, 256 - x['costmap'].shape[1]))) * 0 + 0, res, x['origin']) for x in gs]
# pad to 256x256 storage but keep the true shape through valid_rows/valid_cols: hand the unpadded maps over
cms = [CostMap2D(x['costmap'], res, x['origin']) for x in gs]
t0 = time.time()
env = BatchedPlanEnv(cms, [x['path'] for x in gs], params, n_envs=n, auto_reset=True, template_of_env=idx)
torch.cuda.synchronize()
print('C4 set-up %.1f s, map storage %s' % (time.time() - t0, tuple(env._keep['map'].shape)), flush=True)
for mode in (dict(exact_mode=0), dict(exact_mode=2), dict(exact_mode=1), dict(exact_mode=0, dense_threshold=64),
             dict(exact_mode=0, dense_threshold=16)):
    env.set_tuning(**mode)
    env.reset()
    run(env, n, 50, [np.pi / 30, -np.pi / 2], [np.pi / 6, np.pi / 2], 'C4 aisle private maps %s' % mode, 163 + 900 + 3120)
