"""Step rate of the C3 geometry (65 536 envs, shared map) with delay queues and with the pure-pursuit reward provider
(step_local_kernel<*, PLAIN = false>), against the plain configuration and the general single-kernel step.
The time of a step is measured as bench.py does it: a pool of 8 action batches, python loop, HIP events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams, _lib  # noqa: E402
if os.environ.get("BCP_LIB"):   # A/B of kernel variants: another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["BCP_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "g8_traj_mini_00.npz"))
res = float(g["resolution"])
for tag, kw in (("no delays, continuous reward", dict()),
                ("pose_delay 1, state_delay 1 (the runner scripts)", dict(pose_delay=1, state_delay=1)),
                ("control 2, pose 1, state 3", dict(control_delay=2, pose_delay=1, state_delay=3)),
                ("pure pursuit", dict(reward_provider_name="continuous_reward_pure_pursuit")),
                ("pure pursuit + pose_delay 1", dict(reward_provider_name="continuous_reward_pure_pursuit", pose_delay=1)),
                ("no delays, forced single-kernel step", dict())):
    params = EnvParams(goal_spat_dist=0.2, goal_ang_dist=np.pi / 8, resolution=res, refine_path=False, **kw)
    env = BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], params, n_envs=n, auto_reset=True, seed=1)
    if tag.endswith("single-kernel step"):
        env.set_tuning(defer=0)
    rng = np.random.RandomState(0)
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
    env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, n).astype(np.int32)).cuda())
    for k in range(1200):
        env.step(pool[k % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(3):
        e0.record()
        for k in range(400):
            env.step(pool[k % 8])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 400)
    env.check_errors()
    print("%-52s %.4f ms/step  %.3e env-steps/s  (%s)" % (tag, best, n / best * 1e3, env.step_kernels()), flush=True)
    del env
