"""Where the step's fetched bytes come from: the metric workload under the ablation flags of the diagnostic build
(tools/ablate.py), 40 steps per setting, for the FETCH_SIZE / WRITE_SIZE passes of rocprofv3:
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/abl_FETCH_SIZE -o p -- python3 tools/pmc_ablate_run.py
then   python3 tools/pmc_ablate_run.py --read gpurun_out/abl_   (last 6 x 40 step launches, in order).
`pool` as an argument: the geometry-pool workload (one private world per env) instead."""
import sys, os, glob, csv
PHASES = [('full', 0), ('no_park', 1 << 21), ('no_classify', 1 << 22), ('no_collision', 1 << 16), ('no_reward', 1 << 17),
          ('neither', 3 << 16)]
if len(sys.argv) > 2 and sys.argv[1] == '--read':
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = []
        for fn in glob.glob(sys.argv[2] + c + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                if "step_" in r["Kernel_Name"]:
                    rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        rows.sort()
        vals = [v for _, v in rows][-40 * len(PHASES):]
        for k, (name, _) in enumerate(PHASES):
            part = vals[40 * k + 10:40 * (k + 1)]
            print(c, name, "%.1f KiB per launch (counter as reported)" % (sum(part) / max(len(part), 1)))
    sys.exit(0)
import numpy as np, torch
sys.path.insert(0, '.')
from bc_gym_planning_env_amd import _lib
_lib.LIB_PATH = os.path.join('tools', 'libbcplan_diag.so')
import bench
sys.path.insert(0, 'tools')
from diag_flags import time_steps_with_flags
n = 65536
rng = np.random.RandomState(1234)
if "pool" in sys.argv[1:]:   # one private world per env (tools/pmc_pool_run.py)
    from bc_gym_planning_env_amd import mini_env
    env = mini_env.BatchedRandomMiniEnv(n, n_chains=n, episodes=2, auto_reset=True, seed=3, sampler="device_resident")
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(16)])).cuda()
    env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, n).astype(np.int32)).cuda())
    for k in range(1200):
        env.step(pool[k % 16])
else:
    env, g = bench.make_env(n, 0, 0, 2024)
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(16)])).cuda()
    bench.steady_state(env, pool, rng)
st = env.get_state()
for name, fl in PHASES:
    env.set_state(st)
    for k in range(40):
        time_steps_with_flags(env, pool[k % 16], 1, fl)   # (the library's timing loop: it takes the ablation flags)
    torch.cuda.synchronize()
print("done")
