"""Soak of the endless geometry pool with the refresh on a side stream, at scale: N envs on E-entry rings against the
same envs on a pool that holds their whole streams K worlds deep; every step's reward / done / collision / state must be
identical, and in the end every ring entry must hold the stream's world of its number.  The steps are enqueued without
host synchronisation, so step kernels and refresh kernels really overlap.
Usage: python tools/soak_endless.py [n_envs] [ring] [deep] [steps] [refresh_every]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bc_gym_planning_env_amd import EnvParams, mini_env  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
E = int(sys.argv[2]) if len(sys.argv) > 2 else 6
K = int(sys.argv[3]) if len(sys.argv) > 3 else 48
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 600
every = int(sys.argv[5]) if len(sys.argv) > 5 else 16
timeout = 25

params = mini_env.RandomMiniEnvParams(env_params=EnvParams(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2,
                                                           iteration_timeout=timeout))
seeds = list(range(5000, 5000 + n))
env = mini_env.BatchedRandomMiniEnv(n, params, seeds=seeds, episodes=E, endless=True, auto_reset=True, seed=21)
deep = mini_env.sample_pool_device(params, seeds, K, keep_on_device=True)
ref = mini_env.BatchedRandomMiniEnv(n, params, pool=deep, auto_reset=True, seed=21)
g = torch.Generator(device="cuda").manual_seed(3)
lo = torch.tensor([0.0, -0.6], device="cuda", dtype=torch.float64)
span = torch.tensor([1.5, 1.2], device="cuda", dtype=torch.float64)
acts = [torch.rand(n, 2, device="cuda", generator=g, dtype=torch.float64) * span + lo for _ in range(32)]
bad = torch.zeros((), dtype=torch.int64, device="cuda")
resets = torch.zeros(n, dtype=torch.int64, device="cuda")
z = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
waiting = torch.zeros((), dtype=torch.int64, device="cuda")
resampled = torch.zeros((), dtype=torch.int64, device="cuda")
for t in range(steps):
    a = acts[t % 32]
    env.step(a, noise_z_out=z)
    ref.step(a, noise_z=z)
    for name in ("reward", "done", "collided_now"):
        bad += (getattr(env, name) != getattr(ref, name)).sum()
    bad += (env.state.robot != ref.state.robot).sum() + (env.state.target_idx != ref.state.target_idx).sum()
    bad += (env.state.min_spat_dist_so_far != ref.state.min_spat_dist_so_far).sum()
    resets += env.done
    if t % every == every - 1:
        info = env.refresh(overlap=True)
        if info is not None:
            resampled += info[0]
            waiting += info[1]
    if t % 100 == 99:
        print("step %d enqueued" % (t + 1), flush=True)
env.finish_refresh(check=True)
torch.cuda.synchronize()
world = 1 + resets.cpu().numpy()
assert world.max() < K, "the deep pool (%d worlds per stream) did not outlast the run (%d): raise `deep`" % (K, world.max())
print("mismatching values over %d steps x %d envs: %d" % (steps, n, int(bad)))
print("episodes per env: min %d max %d; worlds re-sampled %d; envs seen waiting at a guard %d"
      % (world.min() - 1, world.max() - 1, int(resampled), int(waiting)))
# ring contents against the deep pool
generated = env._generated.cpu().numpy()
pw, dw = env.pool.world_params.cpu().numpy(), deep.world_params.cpu().numpy()
geom = env.geom_of_env.cpu().numpy()
wrong = 0
if int(waiting) == 0:
    wrong += int((geom != np.arange(n) * E + world % E).sum())
for c in range(0, n, max(1, n // 2048)):
    for j in range(generated[c] - E, generated[c]):
        wrong += int((pw[c * E + j % E] != dw[c * K + j]).any())
print("ring entries / positions that differ from the stream: %d" % wrong)
ok = int(bad) == 0 and wrong == 0
print("SOAK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
