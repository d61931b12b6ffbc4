"""Profiling workload for the egocentric observation on the AisleTurn maps the reference's consumers use: the fixture's
geometry (g10_ego_aisle.npz: EgocentricCostmap on the default AisleTurn map, 333 x 183 -> 133 x 117; default
g12_colored_ego.npz: ColoredEgoCostmapRandomAisleTurnEnv's 350 x 512 -> 133 x 133), 65 536 envs at steady state, 10 observations.
   rocprofv3 --kernel-trace --stats -- python3 tools/ego_prof_aisle.py [fixture]      (or --pmc WRITE_SIZE)"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
from bc_gym_planning_env_amd.egocentric import BatchedColoredEgoCostmap, BatchedEgocentricCostmap
fixture = sys.argv[1] if len(sys.argv) > 1 else 'g12_colored_ego.npz'
g = np.load('tests/golden/' + fixture)
n = 65536
res = float(g['resolution'])
env = BatchedPlanEnv(CostMap2D(g['costmap'], res, g['origin']), g['path'], EnvParams(resolution=res, refine_path=False), n_envs=n,
                     auto_reset=True, seed=17)
wrap = (BatchedColoredEgoCostmap if fixture.startswith('g12') else BatchedEgocentricCostmap)(env)
rng = np.random.RandomState(5)
acts = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
bench.steady_state(env, acts, rng)
for k in range(10):
    wrap.observation()
torch.cuda.synchronize()
print("done: %s, %d bytes per call" % (wrap.route(), wrap.images.numel()))
