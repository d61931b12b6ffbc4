"""Profiling workload for the egocentric kernel: steady-state C3 batch, then 10 observations.
   rocprofv3 --kernel-trace --stats -- python3 tools/ego_prof.py      (or --pmc ...)"""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
from bc_gym_planning_env_amd.egocentric import BatchedEgocentricCostmap
n = 65536
g = np.load(os.path.join('tests', 'golden', 'g8_traj_mini_00.npz'))
res = float(g['resolution'])
params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=res, refine_path=False)
env = BatchedPlanEnv(CostMap2D(g['costmap'], res, g['origin']), g['path'], params, n_envs=n, auto_reset=True, seed=1)
wrap = BatchedEgocentricCostmap(env)
rng = np.random.RandomState(0)
acts = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, n).astype(np.int32)).cuda())
for k in range(600):
    env.step(acts[k % 8])
torch.cuda.synchronize()
for k in range(10):
    wrap.observation()
torch.cuda.synchronize()
print("done")
