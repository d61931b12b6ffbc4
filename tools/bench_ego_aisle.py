import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from bc_gym_planning_env_amd import _lib
if os.environ.get("BCP_LIB"):   # an experimental build of the library
    _lib.LIB_PATH = os.environ["BCP_LIB"]
from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
from bc_gym_planning_env_amd.egocentric import BatchedColoredEgoCostmap, BatchedEgocentricCostmap
g = np.load('tests/golden/' + (sys.argv[1] if len(sys.argv) > 1 else 'g12_colored_ego.npz'))
n = 65536
res = float(g['resolution'])
env = BatchedPlanEnv(CostMap2D(g['costmap'], res, g['origin']), g['path'], EnvParams(resolution=res, refine_path=False), n_envs=n, auto_reset=True)
wrap = (BatchedEgocentricCostmap if len(sys.argv) > 1 else BatchedColoredEgoCostmap)(env)
rng = np.random.RandomState(0)
acts = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
import bench
bench.steady_state(env, acts, rng)   # (random episode phases + one full timeout of pre-roll, as bench.py's legs)
for k in range(3):
    wrap.observation()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(20):
    wrap.observation()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("aisle map %s, window %s: %.4f ms -> %.1f GB/s written, lethal fraction %.4f, %s" % (g['costmap'].shape, wrap.image_shape, ms, wrap.images.numel() / ms / 1e6, float((wrap.images == 254).float().mean()), wrap.route()))
lit = (wrap.images != 0).flatten(1).sum(1).float()
print("lit pixels per image: median %d  p90 %d  max %d" % (lit.median(), lit.quantile(0.9), lit.max()))
ms2 = env.time_steps(acts[0], 50)
print("step alone %.4f ms" % ms2)
