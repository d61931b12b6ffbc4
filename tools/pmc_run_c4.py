"""Workload for the kernel-trace / PMC passes over BASELINE configs[3] (C4): 65 536 AisleTurn envs with private costmaps
stored [N, 256, 256] and private 130-point paths -- steady-state pre-roll, then 40 steps.
rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/pmc_run_c4.py   (and a second pass with WRITE_SIZE)"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
n = 65536
env = bench.make_c4_env(n, 0)
rng = np.random.RandomState(0)
pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
bench.steady_state(env, pool, rng)
for k in range(40):
    env.step(pool[k % 8])
torch.cuda.synchronize()
print("done: %s, collisions/step %.4f" % (env.step_kernels(), float(env.collided_now.float().mean())))
