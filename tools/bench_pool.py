"""Informational timing of the geometry-pool configuration: 65 536 RandomMiniEnv instances, every reset moving the
env to the next pre-sampled world of its chain.  Usage: python tools/bench_pool.py [n] [chains] [episodes]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, '.')
from bc_gym_planning_env_amd import _lib
if os.environ.get('BCP_LIB'):   # A/B of kernel variants: another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ['BCP_LIB'])
from bc_gym_planning_env_amd import mini_env

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
episodes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
t0 = time.time()
pool = mini_env.sample_pool(None, list(range(chains)), episodes)
t1 = time.time()
print("pool: %d worlds sampled in %.1f s (%.1f worlds/s, host numpy + batched GPU acceptance test)" % (
    len(pool), t1 - t0, len(pool) / (t1 - t0)), flush=True)
for big in (4096, 65536):
    torch.cuda.synchronize()
    t2 = time.time()
    dpool = mini_env.sample_pool_device(None, list(range(big)), episodes)
    t3 = time.time()
    print("device sampler: %d worlds in %.2f s (%.0f worlds/s, incl. download and host objects)" % (
        len(dpool), t3 - t2, len(dpool) / (t3 - t2)), flush=True)
    del dpool
t1 = time.time()
env = mini_env.BatchedRandomMiniEnv(n, pool=pool, auto_reset=True, seed=3)
torch.cuda.synchronize()
print("env set-up %.1f s" % (time.time() - t1), flush=True)
rng = np.random.RandomState(0)
acts = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, n).astype(np.int32)).cuda())
for mode in (dict(exact_mode=0), dict(exact_mode=1)):
    env.set_tuning(**mode)
    done = 0
    for k in range(1200):
        env.step(acts[k % 8])
        if k >= 1000:
            done += int(env.done.sum())
    torch.cuda.synchronize()
    ms = env.time_steps(acts[0], 100)
    print("pool %s: %d envs  %.4f ms/step  %.3e env-steps/s  (episodes ending per step %.1f, distinct worlds in use %d)" % (
        mode, n, ms, n / ms * 1e3, done / 200.0, len(torch.unique(env.geom_of_env))), flush=True)
