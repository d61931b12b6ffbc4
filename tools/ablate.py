"""The C3 step (`c4` as an argument: BASELINE configs[3]) with stages switched off (ablation flags of bcp_step.h: results are wrong by construction, only the
timing is of interest).  HIP-event timing of 20 back-to-back steps, 5 repetitions each.  Needs the diagnostic build:
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DBCP_DIAG bc_gym_planning_env_amd/csrc/bcplan.hip
      -o tools/libbcplan_diag.so      (BCP_FUSED=0 in the environment: the two-launch step)"""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from bc_gym_planning_env_amd import _lib
_lib.LIB_PATH = os.path.join('tools', 'libbcplan_diag.so')   # -DBCP_DIAG build: the shipping library rejects these flags
import bench
sys.path.insert(0, 'tools')
from diag_flags import time_steps_with_flags
rng = np.random.RandomState(0)
if "c4" in sys.argv[1:]:   # BASELINE configs[3] instead of the metric workload
    env = bench.make_c4_env(65536, 0)
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(65536, rng) for _ in range(16)])).cuda()
    bench.steady_state(env, pool, rng)
else:
    env, g = bench.make_env(65536, 0, 0, 1)
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(65536, rng) for _ in range(16)])).cuda()
    env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, 65536).astype(np.int32)).cuda())
    for k in range(1200):
        env.step(pool[k % 16])
torch.cuda.synchronize()
if os.environ.get("BCP_FUSED") is not None:
    env.set_tuning(fused=int(os.environ["BCP_FUSED"]))
print(env.step_kernels(), flush=True)
st = env.get_state()
for name, fl in [('full', 0), ('no_park', 1 << 21), ('no_classify', 1 << 22), ('no_collision', 1 << 16),
                 ('no_reward', 1 << 17), ('neither', 3 << 16), ('no_exact_test', 1 << 19)]:
    env.set_state(st)
    ms = [time_steps_with_flags(env, pool[i % 16], 20, fl) for i in range(5)]
    print(name, ['%.4f' % m for m in ms], flush=True)
