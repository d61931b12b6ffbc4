"""Workload for the PMC passes over the geometry-pool step: one RandomState stream (and so one private world) per env,
steady-state pre-roll, then 40 steps.   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/pmc_pool_run.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bc_gym_planning_env_amd import mini_env  # noqa: E402

n = 65536
env = mini_env.BatchedRandomMiniEnv(n, n_chains=n, episodes=2, auto_reset=True, seed=3, sampler="device_resident")
rng = np.random.RandomState(0)
acts = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, n).astype(np.int32)).cuda())
for k in range(1200):
    env.step(acts[k % 8])
torch.cuda.synchronize()
for k in range(40):
    env.step(acts[k % 8])
torch.cuda.synchronize()
print("done")
