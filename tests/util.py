"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAJ = sorted(glob.glob(os.path.join(GOLDEN, "g8_traj_*.npz")))

# tolerance of the north star: done/collision bit-exact, pose/reward within 1e-5.  The HIP path differs from the
# oracle only through the last-ulp behaviour of device sin/cos/hypot, so the tests hold it to a far tighter bound.
ATOL = 1e-9


def traj_config(name):
    """(noise_parameters, spatial_precision, angular_precision) used when a g8 trajectory was recorded."""
    mini = "mini" in name
    noise = None if "nonoise" in name else 'planenv'
    sp, ap = (0.2, np.pi / 8) if mini else (1.0, np.pi / 2)
    return noise, sp, ap


def env_from_traj(g, name, n_envs=1, **kw):
    """BatchedPlanEnv replicating the recorded env n_envs times (shared costmap / path)."""
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    noise, sp, ap = traj_config(name)
    params = EnvParams(goal_spat_dist=sp, goal_ang_dist=ap, resolution=float(g["resolution"]), refine_path=False)
    costmap = CostMap2D(g["costmap"], float(g["resolution"]), g["origin"])
    return BatchedPlanEnv(costmap, g["path"], params, n_envs=n_envs, noise_parameters=noise, **kw)


def oracle_params_for(oracle, name, **kw):
    noise, sp, ap = traj_config(name)
    return oracle.make_params("tricycle", noise=oracle.PLANENV_NOISE if noise else None, spatial_precision=sp,
                              angular_precision=ap, **kw)


def z_in(z):
    """NaN (= slot not drawn in the reference) -> a poison value that must never be consumed."""
    return np.where(np.isnan(z), 1e300, z)
