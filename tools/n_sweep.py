"""Step time of the C3 geometry against the batch size, 16 k ... 1 M envs (one workgroup of step_local_kernel = 256 envs,
one workgroup per CU: 65 536 envs fill the 256 CUs exactly once): where does one-workgroup-per-CU stop being bound by
the latency of a single workgroup?  HIP-event timing of 4 x 100 back-to-back steps after a steady-state pre-roll.
python tools/n_sweep.py [lib] > gpurun_out/n_sweep.txt"""
import os, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
from bc_gym_planning_env_amd import _lib
from _variant import use_lib
use_lib(sys.argv[1] if len(sys.argv) > 1 else '')
import numpy as np, torch
import bench
pairs = os.environ.get("BCP_PAIRS")
for n in (16384, 32768, 65536, 65536 + 256, 98304, 131072, 196608, 262144, 524288, 1048576):
    env, g = bench.make_env(n, 0, 0, 2024)
    if pairs is not None:
        env.set_tuning(local_pairs=int(pairs))
    rng = np.random.RandomState(1234)
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(4)])).cuda()
    bench.steady_state(env, pool, rng)
    ms = min(env.time_steps(pool[i % 4], 100) for i in range(4))
    print("%8d envs (%5.2f workgroups per CU): %.4f ms/step  %.3e env-steps/s  %.1f GB/s algorithmic"
          % (n, n / 65536.0, ms, n / ms * 1e3, 163.0 * n / ms * 1e-6), flush=True)
    env.close()
    del env, pool
    torch.cuda.empty_cache()
