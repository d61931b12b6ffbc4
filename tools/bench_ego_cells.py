"""Calibration of the sparse / sampling routing of bcp_egocentric_costmaps (csrc/bcplan.hip: ego_sparse_limit): time per call of
the fill-and-patch kernel against the number of non-zero cells of the map, beside the sampling kernel the same call would fall
to, for a map that fits LDS (183 x 183) and one that does not (350 x 512).  65 536 images of 133 x 117 px, random poses
inside the map (every window is full of map: the worst case for the patches)."""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams   # noqa: E402
from bc_gym_planning_env_amd.egocentric import BatchedEgocentricCostmap   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
res = 0.03
rng = np.random.RandomState(0)
path = np.array([[0., 0., 0.], [1., 0., 0.], [2., 0., 0.]])
for shape in ((183, 183), (350, 512)):
    for cells in (128, 512, 1024, 2048, 4096, 8192, 16384):
        m = np.zeros(shape, dtype=np.uint8)
        m.reshape(-1)[rng.choice(m.size, cells, replace=False)] = 254
        env = BatchedPlanEnv(CostMap2D(m, res, np.zeros(2)), path, EnvParams(resolution=res, refine_path=False), n_envs=n)
        st = env.state.robot
        st[0].copy_(torch.from_numpy(rng.uniform(0.5, shape[1] * res - 0.5, n)).cuda())
        st[1].copy_(torch.from_numpy(rng.uniform(0.5, shape[0] * res - 0.5, n)).cuda())
        st[2].copy_(torch.from_numpy(rng.uniform(-3.1, 3.1, n)).cuda())
        wrap = BatchedEgocentricCostmap(env)
        row = []
        for tuning in (1 << 20, 0):
            env.set_tuning(ego_sparse=tuning)
            for _ in range(2):
                wrap.observation()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                wrap.observation()
            e1.record()
            torch.cuda.synchronize()
            row.append((wrap.route()["kernel"], e0.elapsed_time(e1) / 5))
        print("map %3d x %3d  %5d cells (%.1f per window):  %s %.4f ms   |   %s %.4f ms" % (
            shape[0], shape[1], cells, cells * 133.0 * 117.0 / m.size, row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)
        env.close()
        del wrap, env
        torch.cuda.empty_cache()
