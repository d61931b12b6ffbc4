"""Egocentric observations for the whole batch: what the reference's EgocentricCostmap wrapper
(envs/egocentric.py:102-160) computes per env on the host -- the costmap rotated and cut around the robot
(extract_egocentric_costmap, utilities/costmap_utils.py:25-75) and the goal_n_state vector -- as two launches over
all envs (bcp_egocentric_costmaps, bcp_goal_n_state; include/bcplan.h)."""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib


def _f64x2(v):
    a = np.ascontiguousarray(v, dtype=np.float64)
    assert a.shape == (2,)
    return a


class BatchedEgocentricCostmap(object):
    """Observation wrapper around a BatchedPlanEnv (or BatchedRandomMiniEnv): step() / reset() return
    OrderedDict(env=uint8 [N, H, W, 1], goal_n_state=float32 [N, 9, 1]) device tensors (8 rows for a diff-drive robot).
    The window defaults are the reference's: 0.5 m behind to 3 m ahead of the robot, 2 m to each side."""

    def __init__(self, env, x_bounds=(-0.5, 3.), y_bounds=(-2., 2.), border_value=0):
        self.env = env
        self.action_space = env.action_space
        self._origin = _f64x2([x_bounds[0], y_bounds[0]])
        self._size = _f64x2([x_bounds[1] - x_bounds[0], y_bounds[1] - y_bounds[0]])
        self._border = int(border_value)
        self._lib = env._lib
        shape = (C.c_int32 * 2)()
        _lib.check(self._lib.bcp_egocentric_shape(env._h, self._size.ctypes.data_as(_lib._f64p), shape))
        self.image_shape = (int(shape[0]), int(shape[1]))
        res = env.resolution
        # CostMap2D.world_size() of the extracted map (utilities/costmap_2d.py:107-121)
        self._world = _f64x2([(self._origin[0] + res * shape[1]) - self._origin[0],
                              (self._origin[1] + res * shape[0]) - self._origin[1]])
        n, dev = env.n_envs, env.device
        self.n_state = 6 if env.is_tricycle else 5
        self.images = torch.zeros((n,) + self.image_shape + (1,), dtype=torch.uint8, device=dev)
        self.goal_n_state = torch.zeros((n, 3 + self.n_state, 1), dtype=torch.float32, device=dev)
        self._obs = OrderedDict((('env', self.images), ('goal_n_state', self.goal_n_state)))

    def unwrapped(self):
        return self.env

    def observation(self, _observation=None):
        """Refresh and return the observation of the envs' current state (device tensors, no sync)."""
        e = self.env
        stream = C.c_void_p(torch.cuda.current_stream(e.device).cuda_stream)
        _lib.check(self._lib.bcp_egocentric_costmaps(
            e._h, None, e.n_envs, self._origin.ctypes.data_as(_lib._f64p), self._size.ctypes.data_as(_lib._f64p),
            self._border, self.images.data_ptr(), stream))
        _lib.check(self._lib.bcp_goal_n_state(e._h, self._world.ctypes.data_as(_lib._f64p),
                                              self.goal_n_state.data_ptr(), stream))
        return self._obs

    def _refresh_images(self):
        e = self.env
        stream = C.c_void_p(torch.cuda.current_stream(e.device).cuda_stream)
        _lib.check(self._lib.bcp_egocentric_costmaps(
            e._h, None, e.n_envs, self._origin.ctypes.data_as(_lib._f64p), self._size.ctypes.data_as(_lib._f64p),
            self._border, self.images.data_ptr(), stream))
        return stream

    def route(self):
        """Which kernel drew the last observation (bcp_egocentric_route): dict(kernel=name, max_cells, list_stride, limit)."""
        info = (C.c_int32 * 4)()
        _lib.check(self._lib.bcp_egocentric_route(self.env._h, info))
        return {"kernel": _lib.EGO_KERNELS.get(int(info[0]), "?"), "max_cells": int(info[1]), "list_stride": int(info[2]),
                "limit": int(info[3])}

    def step(self, actions, **kw):
        _o, reward, done, info = self.env.step(actions, **kw)
        return self.observation(), reward, done, info

    def reset(self, mask=None):
        self.env.reset(mask)
        return self.observation()

    def seed(self, seed=None):
        self.env.seed(seed)

    def get_state(self):
        return self.env.get_state()

    def set_state(self, state):
        self.env.set_state(state)

    def render(self, mode='human'):
        return self.env.render(mode)

    def close(self):
        self.env.close()


class BatchedColoredEgoCostmap(BatchedEgocentricCostmap):
    """The observation of ColoredEgoCostmapRandomAisleTurnEnv (envs/synth_turn_env.py:376-451) for a whole batch:
    OrderedDict(environment=uint8 [N, 133, 133, 1], goal=float64 [N, 5, 1]) -- the egocentric costmap 0.5 m behind to
    3.5 m ahead of the robot, and (unit direction to the final way point, v, w, wheel_angle)."""

    def __init__(self, env, x_bounds=(-0.5, 3.5), y_bounds=(-2., 2.), border_value=0):
        super(BatchedColoredEgoCostmap, self).__init__(env, x_bounds, y_bounds, border_value)
        self.goal = torch.zeros((env.n_envs, 5, 1), dtype=torch.float64, device=env.device)
        self._obs = OrderedDict((('environment', self.images), ('goal', self.goal)))

    def observation(self, _observation=None):
        stream = self._refresh_images()
        _lib.check(self._lib.bcp_goal_direction_state(self.env._h, self._world.ctypes.data_as(_lib._f64p),
                                                      self.goal.data_ptr(), stream))
        return self._obs
