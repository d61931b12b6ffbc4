import sys, numpy as np, torch
sys.path.insert(0, '.')
from bc_gym_planning_env_amd import mini_env
n = 65536
pool = mini_env.sample_pool(None, list(range(1024)), 4)
env = mini_env.BatchedRandomMiniEnv(n, pool=pool, auto_reset=True, seed=3)
rng = np.random.RandomState(0)
acts = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, n).astype(np.int32)).cuda())
for k in range(1200):
    env.step(acts[k % 8])
torch.cuda.synchronize()
st = env.get_state()
for thr in (6, 2, 12, 24, 64):
    env.set_state(st)
    env.set_tuning(dense_threshold=thr)
    ms = [env.time_steps(acts[i % 8], 20) for i in range(4)]
    k = env.time_step_kernels(acts[0], 20)
    print("dense_threshold", thr, ['%.4f' % m for m in ms], "kernels", ['%.4f' % v for v in k], flush=True)
