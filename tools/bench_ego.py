"""Timing of the egocentric observation on the metric workload: 65 536 envs, shared 183x183 map,
133 x 117 window.  Usage: python tools/bench_ego.py [n] [pool]"""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from bc_gym_planning_env_amd import _lib
if os.environ.get("BCP_LIB"):   # an experimental build of the library
    _lib.LIB_PATH = os.environ["BCP_LIB"]
from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams, mini_env
from bc_gym_planning_env_amd.egocentric import BatchedEgocentricCostmap

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
use_pool = len(sys.argv) > 2 and sys.argv[2] == "pool"
if use_pool:
    env = mini_env.BatchedRandomMiniEnv(n, n_chains=1024, episodes=4, auto_reset=True, seed=3)
else:
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
for k in range(5):
    wrap.step(acts[k % 8])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 30
e0.record()
for k in range(reps):
    wrap.observation()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
bytes_out = wrap.images.numel()
print("observation only: %.4f ms  -> %.1f GB/s written (%d B per env)" % (ms, bytes_out / ms / 1e6, bytes_out // n), flush=True)
# the same with the sampling kernels (BCP_TUNE_EGO_SPARSE = 0: what round 2 ran), and a plain device fill of the same bytes
env.set_tuning(ego_sparse=0)
for k in range(3):
    wrap.observation()
e0.record()
for k in range(reps):
    wrap.observation()
e1.record()
torch.cuda.synchronize()
ms_dense = e0.elapsed_time(e1) / reps
env.set_tuning(ego_sparse=1)
for k in range(3):
    wrap.images.zero_()
e0.record()
for k in range(reps):
    wrap.images.zero_()
e1.record()
torch.cuda.synchronize()
ms_fill = e0.elapsed_time(e1) / reps
print("  sampling kernels (ego_sparse = 0): %.4f ms -> %.1f GB/s;  torch zero_() of the same buffer: %.4f ms -> %.1f GB/s" % (
    ms_dense, bytes_out / ms_dense / 1e6, ms_fill, bytes_out / ms_fill / 1e6), flush=True)
e0.record()
for k in range(reps):
    wrap.step(acts[k % 8])
e1.record()
torch.cuda.synchronize()
ms2 = e0.elapsed_time(e1) / reps
print("step + observation: %.4f ms/step  %.3e env-steps/s  (lethal fraction of the images %.4f)" % (
    ms2, n / ms2 * 1e3, float((wrap.images == 254).float().mean())), flush=True)

# worst case for the row culling: every robot well inside the map (all rows are sampled)
if not use_pool:
    c = torch.tensor([0.0, 0.0], dtype=torch.float64, device='cuda')
    env.state.robot[0:2] = (torch.rand(2, n, dtype=torch.float64, device='cuda') - 0.5) * 1.0
    env.state.robot[2] = (torch.rand(n, dtype=torch.float64, device='cuda') - 0.5) * 6.28
    wrap.observation()
    torch.cuda.synchronize()
    e0.record()
    for k in range(reps):
        wrap.observation()
    e1.record()
    torch.cuda.synchronize()
    ms3 = e0.elapsed_time(e1) / reps
    print("observation only, all robots near the map centre: %.4f ms -> %.1f GB/s written (lethal fraction %.4f)" % (
        ms3, bytes_out / ms3 / 1e6, float((wrap.images == 254).float().mean())), flush=True)
