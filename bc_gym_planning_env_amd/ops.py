"""Operator-level seams: GPU stand-ins for the reference's optional native hooks (`*_impl` imports guarded by
try/except ImportError), batched over many inputs.

  get_pixel_footprint   utilities/path_tools.py:101-162       (get_pixel_footprint_impl)
  pose_collides         envs/base/env.py:464-489, utilities/costmap_utils.py:178-203
  normalize_angle       utilities/coordinate_transformations.py:17-36   (normalize_angle_impl)
  world_to_pixel        utilities/coordinate_transformations.py:169-205 (world_to_pixel_impl)
  extract_egocentric_costmap   utilities/costmap_utils.py:25-75 (cv2.getRotationMatrix2D + cv2.warpAffine, nearest)
  robot_step            IRobot.step: tricycle_model.py:478-538 / differential_drive.py:236-265
  is_robot_colliding    utilities/costmap_utils.py:106-164; is_footprint_colliding (is_footprint_colliding_impl, :106-136)
  reward / find_last_reached / path_velocity   envs/base/reward.py:184-259, utilities/path_tools.py:432-448, :298-323
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, robots
from .api import EnvParams, INDUSTRIAL_TRICYCLE_V1


class NativeOps(object):
    """A libbcplan handle used only for the stand-alone operators (no env state bound)."""

    def __init__(self, robot_name=INDUSTRIAL_TRICYCLE_V1, device=0, noise_parameters=None, params=None,
                 footprint_scale=1.0, dynamic_model=True, model_front_column_pid=True):
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("NativeOps needs a GPU (libbcplan has no CPU path)")
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.params = EnvParams() if params is None else params
        self._p = robots.make_bcp_params(self.params, robot_name, noise_parameters, footprint_scale, dynamic_model,
                                         model_front_column_pid)
        self._h = C.c_void_p()
        self._n_maps = 1
        _lib.check(self._lib.bcp_create(C.byref(self._p), 1, self.device.index or 0, 0, C.byref(self._h)))
        self._keep = {}

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_tuning(self, exact_mode=None, dense_threshold=None, cull=None, defer=None):
        """Execution knobs of libbcplan (bcp_set_tuning); results never depend on them.  `cull` must be chosen
        before set_costmap()."""
        for key, val in ((_lib.TUNE_EXACT_MODE, exact_mode), (_lib.TUNE_DENSE_THRESHOLD, dense_threshold),
                         (_lib.TUNE_CULL, cull), (_lib.TUNE_DEFER, defer)):
            if val is not None:
                _lib.check(self._lib.bcp_set_tuning(self._h, key, int(val)))

    def _dev(self, a, dtype):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(self.device, dtype).contiguous()

    def normalize_angle(self, z):
        zin = self._dev(np.atleast_1d(z) if not isinstance(z, torch.Tensor) else z, torch.float64)
        out = torch.empty_like(zin)
        _lib.check(self._lib.bcp_normalize_angle(self._h, zin.data_ptr(), out.data_ptr(), zin.numel(), self._stream()))
        return out

    def world_to_pixel(self, world_coords, origin, resolution):
        xy = self._dev(world_coords, torch.float64)
        flat = xy.reshape(-1, 2)
        out = torch.empty(flat.shape, dtype=torch.int64, device=self.device)
        org = np.ascontiguousarray(origin, dtype=np.float64)
        _lib.check(self._lib.bcp_world_to_pixel(self._h, flat.data_ptr(), flat.shape[0],
                                                org.ctypes.data_as(C.POINTER(C.c_double)), float(resolution),
                                                out.data_ptr(), self._stream()))
        return out.reshape(xy.shape)

    def get_pixel_footprint(self, angles, map_resolution, side=None):
        """-> (masks uint8 [n, side, side], shape_hw int32 [n, 2]); image i is masks[i, :h, :w]."""
        ang = self._dev(np.atleast_1d(angles) if not isinstance(angles, torch.Tensor) else angles, torch.float64)
        n = ang.numel()
        if side is None:
            fp = np.array([[self._p.verts[k][0], self._p.verts[k][1]] for k in range(self._p.n_verts)])
            side = 2 * int(np.ceil(np.linalg.norm(fp, axis=1).max() / map_resolution)) + 3
        masks = torch.empty((n, side, side), dtype=torch.uint8, device=self.device)
        shape = torch.zeros((n, 2), dtype=torch.int32, device=self.device)
        _lib.check(self._lib.bcp_pixel_footprint(self._h, ang.data_ptr(), n, float(map_resolution), masks.data_ptr(),
                                                 side, shape.data_ptr(), self._stream()))
        return masks, shape

    def set_costmap(self, data, origin, resolution):
        d = self._dev(data, torch.uint8)
        org = np.ascontiguousarray(origin, dtype=np.float64)
        self._keep["map"] = d
        _lib.check(self._lib.bcp_set_costmaps(self._h, d.data_ptr(), d.shape[0], d.shape[1], 1, None, None,
                                              org.ctypes.data, 0, float(resolution), self._stream()))

    def pose_collides(self, poses):
        """poses [n,3] against the costmap given to set_costmap -> uint8 [n]."""
        p = self._dev(poses, torch.float64)
        out = torch.empty(p.shape[0], dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.bcp_pose_collides(self._h, p.data_ptr(), p.shape[0], out.data_ptr(), self._stream()))
        return out

    def is_robot_colliding(self, poses):
        """is_robot_colliding (costmap_utils.py:106-164) for poses [n,3]: pose_collides, but never when the robot's own
        pixel is off the map -> uint8 [n]."""
        p = self._dev(poses, torch.float64)
        out = torch.empty(p.shape[0], dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.bcp_is_robot_colliding(self._h, p.data_ptr(), p.shape[0], out.data_ptr(), self._stream()))
        return out

    def is_footprint_colliding(self, image_slices, blit_masks, lethal=254):
        """is_footprint_colliding_impl(image_slice, blit_mask, lethal) for n pairs [n, rows, cols] -> uint8 [n]."""
        sl = self._dev(image_slices, torch.uint8)
        mk = self._dev(blit_masks, torch.uint8)
        if sl.dim() == 2:
            sl, mk = sl[None], mk[None]
        assert sl.shape == mk.shape and sl.dim() == 3
        out = torch.empty(sl.shape[0], dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.bcp_is_footprint_colliding(self._h, sl.data_ptr(), mk.data_ptr(), sl.shape[0], sl.shape[1],
                                                        sl.shape[2], int(lethal), out.data_ptr(), self._stream()))
        return out

    def set_path(self, path):
        """The (already refined) path [m,3] the reward operators below score against."""
        p = self._dev(path, torch.float64)
        assert p.dim() == 2 and p.shape[1] == 3
        self._keep["path"] = p
        _lib.check(self._lib.bcp_set_paths(self._h, p.data_ptr(), None, p.shape[0], 1, self._stream()))
        torch.cuda.current_stream(self.device).synchronize()

    def reward(self, poses, min_spat_dist_so_far, target_idx, robot_collided=None):
        """reward_provider.reward(state) / .done(state) (reward.py:184-259) for n (pose, provider state) pairs ->
        (reward float64 [n], new min_spat_dist_so_far [n], new target_idx int32 [n], goal_reached uint8 [n])."""
        p = self._dev(poses, torch.float64)
        n = p.shape[0]
        md = self._dev(min_spat_dist_so_far, torch.float64).clone()
        ti = self._dev(target_idx, torch.int32).clone()
        col = self._dev(robot_collided, torch.uint8) if robot_collided is not None else None
        rew = torch.empty(n, dtype=torch.float64, device=self.device)
        goal = torch.empty(n, dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.bcp_reward(self._h, p.data_ptr(), n, md.data_ptr(), ti.data_ptr(),
                                        col.data_ptr() if col is not None else None, rew.data_ptr(), goal.data_ptr(),
                                        self._stream()))
        return rew, md, ti, goal

    def find_last_reached(self, poses):
        """find_last_reached (path_tools.py:432-448) for poses [n,3] against the path of set_path -> int32 [n], -1 = None."""
        p = self._dev(poses, torch.float64)
        out = torch.empty(p.shape[0], dtype=torch.int32, device=self.device)
        _lib.check(self._lib.bcp_find_last_reached(self._h, p.data_ptr(), p.shape[0], out.data_ptr(), self._stream()))
        return out

    def path_velocity(self, path_txyth):
        """path_velocity (path_tools.py:298-323): rows of (t, x, y, angle) -> (v, w) of the n - 1 segments; raises like
        the reference on corrupted angle data / non-increasing time stamps."""
        p = self._dev(path_txyth, torch.float64)
        n = p.shape[0]
        v = torch.empty(n - 1, dtype=torch.float64, device=self.device)
        w = torch.empty(n - 1, dtype=torch.float64, device=self.device)
        err = torch.zeros(n - 1, dtype=torch.int32, device=self.device)
        _lib.check(self._lib.bcp_path_velocity(self._h, p.data_ptr(), n, v.data_ptr(), w.data_ptr(), err.data_ptr(),
                                               self._stream()))
        e = err.cpu().numpy()
        assert not (e & _lib.ERR_TIME_ORDER).any()
        if (e & _lib.ERR_ANGLE_JUMP).any():
            raise Exception("Path has missing/corrupted angle data at indices: %s." % ((e & _lib.ERR_ANGLE_JUMP).nonzero(),))
        return v, w

    def device_normals(self, n_envs, first_step=0, n_steps=1, first_env=0):
        """The standard normals the step kernels would draw (bcp_device_normals) -> float64 [n_steps, n_envs, 3]."""
        out = torch.empty((int(n_steps), int(n_envs), 3), dtype=torch.float64, device=self.device)
        _lib.check(self._lib.bcp_device_normals(self._h, int(first_env), int(n_envs), int(first_step), int(n_steps),
                                                out.data_ptr(), self._stream()))
        return out

    def seed(self, seed):
        _lib.check(self._lib.bcp_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def extract_egocentric_costmap(self, poses, resulting_origin=None, resulting_size=None, border_value=0):
        """The costmap given to set_costmap seen from each of poses [n,3] -> uint8 [n, rows, cols] (robot at (0, 0)
        heading +x; resulting_origin / resulting_size in metres, both or neither)."""
        p = self._dev(np.atleast_2d(poses) if not isinstance(poses, torch.Tensor) else poses, torch.float64)
        f64p = C.POINTER(C.c_double)
        org = sz = None
        if resulting_origin is not None:
            org = np.ascontiguousarray(resulting_origin, dtype=np.float64)
            sz = np.ascontiguousarray(resulting_size, dtype=np.float64)
        shape = (C.c_int32 * 2)()
        _lib.check(self._lib.bcp_egocentric_shape(self._h, sz.ctypes.data_as(f64p) if sz is not None else None, shape))
        out = torch.empty((p.shape[0], shape[0], shape[1]), dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.bcp_egocentric_costmaps(
            self._h, p.data_ptr(), p.shape[0], org.ctypes.data_as(f64p) if org is not None else None,
            sz.ctypes.data_as(f64p) if sz is not None else None, int(border_value), out.data_ptr(), self._stream()))
        return out

    def robot_step(self, state7, actions, noise_z=None):
        """state7 [n,7] rows {x,y,angle,v,w,steering_motor_command,wheel_angle}, actions [n,2] -> (new [n,7], err)."""
        st = self._dev(state7, torch.float64).t().contiguous()  # SoA [7, n]
        a = self._dev(actions, torch.float64)
        n = a.shape[0]
        z = self._dev(noise_z, torch.float64) if noise_z is not None else None
        err = torch.zeros(n, dtype=torch.int32, device=self.device)
        _lib.check(self._lib.bcp_robot_step(self._h, st.data_ptr(), n, a.data_ptr(),
                                            z.data_ptr() if z is not None else None, err.data_ptr(), self._stream()))
        return st.t().contiguous(), err

    def close(self):
        if self._h:
            self._lib.bcp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
