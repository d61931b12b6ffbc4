"""bcp_rollout against bcp_step on the metric workload (C3, 65 536 envs, steady state): K steps per call as ONE launch of
step_local_kernel<.., ROLL = true> (workgroups advance independently, launch + staging once) against K launches.
python tools/bench_rollout.py [n_envs]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.ENVS_PER_GPU
env, g = bench.make_env(n, 0, 0, 2024)
rng = np.random.RandomState(1234)
pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(16)])).cuda()
bench.steady_state(env, pool, rng)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for K in (1, 8, 32, 128, 512):
    acts = pool[torch.arange(K) % 16].contiguous()          # [K, N, 2] float32
    for _ in range(2):
        env.rollout(acts)
    reps = max(2, 2048 // K)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        env.rollout(acts)
    e1.record()
    torch.cuda.synchronize()
    ms_roll = e0.elapsed_time(e1) / (reps * K)
    e0.record()
    for r in range(reps):
        for k in range(K):
            env.step(acts[k])
    e1.record()
    torch.cuda.synchronize()
    ms_step = e0.elapsed_time(e1) / (reps * K)
    print("K = %4d: rollout %.5f ms per step (%.3e env-steps/s)   |   %d x step() %.5f ms per step (%.3e env-steps/s)   ratio %.2f" % (
        K, ms_roll, n / ms_roll * 1e3, K, ms_step, n / ms_step * 1e3, ms_step / ms_roll), flush=True)
env.check_errors()
