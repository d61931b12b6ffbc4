"""Step time of the metric workload (C3, 65 536 envs, steady state) with a given build of the library:
python tools/step_time.py [path/to/libbcplan_variant.so] [n_envs] -- HIP-event timing of 5 x 200 back-to-back steps."""
import os, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
from bc_gym_planning_env_amd import _lib
from _variant import use_lib
use_lib(sys.argv[1] if len(sys.argv) > 1 else '')
import numpy as np, torch
import bench
n = int(sys.argv[2]) if len(sys.argv) > 2 else bench.ENVS_PER_GPU
env, g = bench.make_env(n, 0, 0, 2024)
rng = np.random.RandomState(1234)
pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(16)])).cuda()
bench.steady_state(env, pool, rng)
if os.environ.get("BCP_PAIRS") is not None:
    env.set_tuning(local_pairs=int(os.environ["BCP_PAIRS"]))
if os.environ.get("BCP_FUSED") is not None:
    env.set_tuning(fused=int(os.environ["BCP_FUSED"]))
ms = [env.time_steps(pool[0], 200) for _ in range(5)]
k = env.time_step_kernels(pool[0], 200)
print("%s n=%d: ms/step %s  min %.5f  (bcp_time_step_kernels: %.5f + %.5f)  form: %s" % (
    os.path.basename(_lib.LIB_PATH), n, " ".join("%.5f" % m for m in ms), min(ms), k[0], k[1], env.step_kernels()), flush=True)
