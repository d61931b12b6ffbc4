"""Robot constants flattened into the library's POD parameter block.

Values of robot_models/robot_dimensions_examples.py: tricycle :108-188 (footprint 16 vertices in mm / 1000,
front wheel 0.964 m ahead of the rear axle, +-85 deg steering, 60 deg/s steering speed, 0.4 m/s^2 and 0.5 rad/s^2
acceleration limits, front-column P gain 0.16), diff-drive :53-82 (24-vertex mock footprint).
"""
import numpy as np

from . import _lib
from .api import (CONTINUOUS_REWARD, CONTINUOUS_REWARD_PURE_PURSUIT, INDUSTRIAL_DIFFDRIVE_V1,
                  INDUSTRIAL_TRICYCLE_V1)

_TRICYCLE_MM = [
    (1348.35, 0.), (1338.56, 139.75), (1306.71, 280.12), (1224.36, 338.62), (1093.81, 374.64), (-214.37, 374.64),
    (-313.62, 308.56), (-366.36, 117.44), (-374.01, -135.75), (-227.96, -459.13), (-156.72, -458.78),
    (759.8, -442.96), (849.69, -426.4), (1171.05, -353.74), (1303.15, -286.54), (1341.34, -118.37)]
_DIFFDRIVE_MM = [
    (644.5, 0), (634.86, 61), (571.935, 130.54), (553.38, 161), (360.36, 186), (250, 186), (250, 186), (100, 186),
    (100, 186), (0, 196), (-119.21, 190.5), (-173.4, 146), (-193, 0), (-173.4, -143), (-111.65, -246), (-71.57, -246),
    (100, -246), (100, -246), (250, -246), (250, -246), (413.085, -223), (491.5, -204.5), (553, -161), (634.86, -62)]

FOOTPRINTS = {
    INDUSTRIAL_TRICYCLE_V1: np.array(_TRICYCLE_MM, dtype=np.float64) / 1000.,
    INDUSTRIAL_DIFFDRIVE_V1: np.array(_DIFFDRIVE_MM, dtype=np.float64) / 1000.,
}
MODELS = {INDUSTRIAL_TRICYCLE_V1: _lib.MODEL_TRICYCLE, INDUSTRIAL_DIFFDRIVE_V1: _lib.MODEL_DIFFDRIVE}

FRONT_WHEEL_FROM_AXIS = 0.964
MAX_FRONT_WHEEL_ANGLE = 0.5 * 170 * np.pi / 180.
MAX_FRONT_WHEEL_SPEED = 60. * np.pi / 180.
MAX_LINEAR_ACCELERATION = 1. / 2.5
MAX_ANGULAR_ACCELERATION = 1. / 2.
FRONT_COLUMN_P_GAIN = 0.16

# PlanEnv hard-codes this odometry noise on its TricycleRobot (envs/base/env.py:226-232)
PLANENV_NOISE = dict(alpha1=0.0, alpha2=0.0, alpha3=1.e-2, alpha4=1.e-2, alpha5=1.e-3, alpha6=1.e-3)


def get_footprint(robot_name, footprint_scale=1.0):
    if robot_name not in FOOTPRINTS:
        raise AssertionError("Unknown footprint {}. Should be one of {}".format(robot_name, list(FOOTPRINTS)))
    return FOOTPRINTS[robot_name] * footprint_scale


def make_bcp_params(env_params, robot_name, noise_parameters, footprint_scale=1.0, dynamic_model=True,
                    model_front_column_pid=True, unpinned_diffdrive_noise=False):
    """EnvParams + robot -> bcp_params (include/bcplan.h)."""
    p = _lib.BcpParams()
    p.abi_version = _lib.ABI_VERSION
    p.model = MODELS[robot_name]
    if p.model == _lib.MODEL_DIFFDRIVE and noise_parameters is not None:
        if not unpinned_diffdrive_noise:
            # what the reference's own DiffDriveRobot.step does with noise_parameters (differential_drive.py:73)
            raise IndexError("too many indices for array: DiffDriveRobot with noise_parameters fails in the reference "
                             "(new_pose[:, 2] on a 1-D pose); pass unpinned_diffdrive_noise=True for the unpinned analogue")
        p.options |= _lib.OPT_DIFFDRIVE_NOISE
    fp = get_footprint(robot_name, footprint_scale)
    p.n_verts = len(fp)
    for k, (x, y) in enumerate(fp):
        p.verts[k][0], p.verts[k][1] = float(x), float(y)
    p.dynamic_model = int(dynamic_model)
    p.model_front_column_pid = int(model_front_column_pid)
    p.noise_on = int(noise_parameters is not None)
    for k in range(6):
        p.alpha[k] = 0.0 if noise_parameters is None else float(noise_parameters['alpha%d' % (k + 1)])
    p.iteration_timeout = int(env_params.iteration_timeout)
    p.dt = float(env_params.dt)
    p.front_wheel_from_axis = FRONT_WHEEL_FROM_AXIS
    p.max_front_wheel_angle = MAX_FRONT_WHEEL_ANGLE
    p.max_front_wheel_speed = MAX_FRONT_WHEEL_SPEED
    p.max_linear_acceleration = MAX_LINEAR_ACCELERATION
    p.max_angular_acceleration = MAX_ANGULAR_ACCELERATION
    p.front_column_p_gain = FRONT_COLUMN_P_GAIN
    rp = env_params.reward_provider_params
    p.spatial_precision = float(rp.spatial_precision)
    p.angular_precision = float(rp.angular_precision)
    p.spatial_progress_multiplier = float(rp.spatial_progress_multiplier)
    providers = {CONTINUOUS_REWARD: _lib.REWARD_CONTINUOUS, CONTINUOUS_REWARD_PURE_PURSUIT: _lib.REWARD_PURE_PURSUIT}
    if env_params.reward_provider_name not in providers:
        raise AssertionError("Unknown reward provider: {}. Should be one of {}".format(
            env_params.reward_provider_name, list(providers)))
    p.reward_provider = providers[env_params.reward_provider_name]
    p.control_delay = int(env_params.control_delay)
    p.pose_delay = int(env_params.pose_delay)
    p.state_delay = int(env_params.state_delay)
    return p
