"""Workload for the PMC passes: steady-state pre-roll, then 40 fused steps, then a calibration copy of known size.
rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/pmc_run.py   (and a second pass with WRITE_SIZE)"""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
if os.environ.get("BCP_LIB"):   # (another build of the library: A/B of counters)
    from _variant import use_lib
    use_lib(os.environ["BCP_LIB"])
import bench
n = 65536
env, g = bench.make_env(n, 0, 0, 2024)
rng = np.random.RandomState(1234)
pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(16)])).cuda()
bench.steady_state(env, pool, rng)
for k in range(40):
    env.step(pool[k % 16])
torch.cuda.synchronize()
# calibration: a float64 copy of 8 Mi elements = 64 MiB read + 64 MiB written
a = torch.rand(8 * 1024 * 1024, dtype=torch.float64, device='cuda')
b = torch.empty_like(a)
for _ in range(5):
    b.copy_(a)
torch.cuda.synchronize()
print("done")
# calibration in the library's own access pattern (8 B/lane SoA): robot_step_kernel on 4 Mi robots reads
# 7 state arrays + [n,2] actions (9 * 8 B) and writes 7 arrays + err (7 * 8 + 4 B) per robot
from bc_gym_planning_env_amd import NativeOps
ops = NativeOps()
m = 4 * 1024 * 1024
st = torch.zeros(m, 7, dtype=torch.float64, device='cuda')
ac = torch.rand(m, 2, dtype=torch.float64, device='cuda')
for _ in range(3):
    ops.robot_step(st, ac)
torch.cuda.synchronize()
print("calibration robot_step_kernel: read %d KiB, write %d KiB per launch" % (m * 72 // 1024, m * 60 // 1024))
