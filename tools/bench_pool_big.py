"""One chain per env: 65 536 RandomMiniEnv streams x 4 worlds = 262 144 worlds, sampled, turned into paths / initial
states / distance fields and kept on the GPU (sampler="device_resident").  Prints the set-up time and the step rate."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from bc_gym_planning_env_amd import mini_env
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.cuda.synchronize()
t0 = time.time()
env = mini_env.BatchedRandomMiniEnv(n, n_chains=n, episodes=episodes, auto_reset=True, seed=3, sampler="device_resident")
torch.cuda.synchronize()
t1 = time.time()
print("%d worlds (one chain per env) ready in %.1f s; GPU memory in use %.1f GB" % (
    len(env.pool), t1 - t0, torch.cuda.memory_allocated() / 1e9), flush=True)
rng = np.random.RandomState(0)
acts = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, n).astype(np.int32)).cuda())
for k in range(1200):
    env.step(acts[k % 8])
torch.cuda.synchronize()
ms = env.time_steps(acts[0], 100)
print("%d envs  %.4f ms/step  %.3e env-steps/s  (distinct worlds in use %d)" % (
    n, ms, n / ms * 1e3, len(torch.unique(env.geom_of_env))), flush=True)
