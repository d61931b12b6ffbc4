"""BatchedPlanEnv: N PlanEnv instances advanced by one fused HIP kernel per step.

Mirrors the reference's object API (envs/base/env.py:217-439): reset() / step() / get_state() / set_state() /
seed() / action_space, batched over N envs, plus `envs[i]` views that hand back reference-shaped
Observation / State objects for one env.  Device tensors are torch tensors only because they cross the C ABI as
raw pointers (tensor.data_ptr()); all arithmetic happens in libbcplan.so.
"""
import ctypes as C

import attr

import numpy as np
import torch

from . import _lib, host_init, robots
from .api import (Action, Box, CONTINUOUS_REWARD_PURE_PURSUIT, ContinuousRewardProviderState,
                  ContinuousRewardPurePursuitProviderState, CostMap2D, DiffdriveRobotState, EnvParams,
                  INDUSTRIAL_TRICYCLE_V1, Observation, State, TricycleRobotState)

_STATE_FIELDS = ("x", "y", "angle", "v", "w", "steering_motor_command", "wheel_angle")


def _as_device_actions(actions, n, device):
    """list[Action] | ndarray | tensor -> contiguous [n,2] float32/float64 tensor on the device."""
    if isinstance(actions, torch.Tensor):
        t = actions
    else:
        if isinstance(actions, Action):
            actions = [actions]
        if isinstance(actions, (list, tuple)) and len(actions) and isinstance(actions[0], Action):
            actions = np.stack([np.asarray(a.command) for a in actions])
        t = torch.from_numpy(np.ascontiguousarray(actions))
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    t = t.to(device).contiguous()
    if tuple(t.shape) != (n, 2):
        raise ValueError("actions must have shape (%d, 2), got %s" % (n, tuple(t.shape)))
    return t


class DeviceGeometryPool(object):
    """G geometries that already live on the GPU (e.g. from mini_env.sample_device_pool): what BatchedPlanEnv's
    geometry-pool mode needs, as device tensors.  `costmaps` / `paths` hand out host copies on demand, for the per-env
    views (envs[i].get_state())."""

    def __init__(self, maps, origin, resolution, paths, lens, init):
        self.maps = maps                  # uint8 [G, rows, cols]
        self.origin = np.asarray(origin, dtype=np.float64)   # one origin for all entries
        self.resolution = float(resolution)
        self.path_points = paths          # float64 [G, max_len, 3], already refined
        self.lens = lens                  # int32 [G]
        self.init = init                  # float64 [G, 2] = (min_spat_dist_so_far, target_idx)

    def __len__(self):
        return int(self.maps.shape[0])

    class _Lazy(object):
        def __init__(self, n, fetch):
            self._n, self._fetch = n, fetch

        def __len__(self):
            return self._n

        def __getitem__(self, k):
            if not -self._n <= k < self._n:
                raise IndexError(k)
            return self._fetch(int(k) % self._n)

    @property
    def costmaps(self):
        return self._Lazy(len(self), lambda k: CostMap2D(self.maps[k].cpu().numpy(), self.resolution, self.origin))

    @property
    def paths(self):
        return self._Lazy(len(self), lambda k: self.path_points[k, :int(self.lens[k])].cpu().numpy())


class BatchedState(object):
    """Snapshot of every env's mutable state (what PlanEnv.get_state() deep-copies, env.py:287-291)."""

    # with delays > 0 (EnvParams.pose_delay / state_delay / control_delay, env.py:27-49, 363-398): what State exposes
    # and the FIFO contents; `robot` is always the robot's TRUE state.  Element k pushed since the last reset lives
    # in slot (k - 1) % delay.
    DELAY_FIELDS = ("pose_seen", "robot_state_seen", "control_queue", "poses_queue", "robot_state_queue")

    def __init__(self, robot, min_spat_dist_so_far, target_idx, current_iter, robot_collided, pose_seen=None,
                 robot_state_seen=None, control_queue=None, poses_queue=None, robot_state_queue=None):
        self.robot = robot                      # float64 [7, N]: x, y, angle, v, w, steering_motor_command, wheel_angle
        self.min_spat_dist_so_far = min_spat_dist_so_far
        self.target_idx = target_idx
        self.current_iter = current_iter
        self.robot_collided = robot_collided
        self.pose_seen = pose_seen                      # [3, N] State.pose when pose_delay > 0
        self.robot_state_seen = robot_state_seen        # [7, N] State.robot_state when state_delay > 0
        self.control_queue = control_queue              # [control_delay, 2, N]
        self.poses_queue = poses_queue                  # [pose_delay, 3, N]
        self.robot_state_queue = robot_state_queue      # [state_delay, 7, N]

    FIELDS = ("robot", "min_spat_dist_so_far", "target_idx", "current_iter", "robot_collided") + DELAY_FIELDS
    VERSION = 1

    def copy(self):
        extra = {k: (getattr(self, k).clone() if getattr(self, k) is not None else None) for k in self.DELAY_FIELDS}
        return BatchedState(self.robot.clone(), self.min_spat_dist_so_far.clone(), self.target_idx.clone(),
                            self.current_iter.clone(), self.robot_collided.clone(), **extra)

    def serialize(self):
        """Basic python types only (dict of numpy arrays + version), as the reference's Serializable objects
        (utilities/serialize.py): picklable, device independent."""
        out = {k: (getattr(self, k).cpu().numpy() if getattr(self, k) is not None else None) for k in self.FIELDS}
        out['version'] = self.VERSION
        return out

    @classmethod
    def deserialize(cls, state, device="cpu"):
        state = dict(state)
        assert state.pop('version') == cls.VERSION
        return cls(**{k: (torch.from_numpy(np.ascontiguousarray(v)).to(device) if v is not None else None)
                      for k, v in state.items()})


class BatchedObservation(object):
    """Observation of all envs after a step: references to the live device tensors (as the reference's Observation
    holds references, obs.py:14-23).  `obs[i]` builds the reference-shaped Observation of env i."""

    def __init__(self, env):
        self._env = env
        st = env.state
        # (with delays the observation shows the delayed pose / robot state, env.py:377-394)
        self.pose = st.pose_seen if st.pose_seen is not None else st.robot[0:3]          # [3, N]
        seen = st.robot_state_seen if st.robot_state_seen is not None else st.robot
        self.robot_state = seen[3:7]              # [4, N] view: v, w, steering_motor_command, wheel_angle
        self.target_idx = env.state.target_idx
        self.current_iter = env.state.current_iter
        self.dt = env.params.dt

    @property
    def time(self):
        return self._env.time_of(self.current_iter)

    def __len__(self):
        return self._env.n_envs

    def __getitem__(self, i):
        return self._env.envs[i].observation()


class EnvView(object):
    """One env of the batch behind the reference's per-env API (copies a few scalars from the device on demand)."""

    def __init__(self, env, i):
        self._env, self._i = env, i

    def _robot_state(self, col):
        if self._env.is_tricycle:
            return TricycleRobotState(*[float(v) for v in col])
        return DiffdriveRobotState(*[float(v) for v in col[:5]])

    def get_state(self):
        e, i = self._env, self._i
        s = e.state
        col = s.robot[:, i].cpu().numpy()
        path = e.path_of(i)
        tidx = int(s.target_idx[i])
        it = int(s.current_iter[i])
        cls = (ContinuousRewardPurePursuitProviderState if e.params.reward_provider_name == CONTINUOUS_REWARD_PURE_PURSUIT
               else ContinuousRewardProviderState)
        rps = cls(min_spat_dist_so_far=float(s.min_spat_dist_so_far[i]), path=path, target_idx=tidx)
        pose = s.pose_seen[:, i].cpu().numpy() if s.pose_seen is not None else col[:3].copy()
        seen = s.robot_state_seen[:, i].cpu().numpy() if s.robot_state_seen is not None else col

        def fifo(q):   # the queue as the reference's list: oldest element first
            if q is None:
                return []
            d = q.shape[0]
            rows = q[:, :, i].cpu().numpy()
            return [rows[(k - 1) % d].copy() for k in range(max(1, it - d + 1), it + 1)]

        return State(reward_provider_state=rps, path=rps.current_path(), original_path=np.copy(path),
                     costmap=e.costmap_of(i), iter_timeout=e.params.iteration_timeout,
                     current_time=float(e.time_of(s.current_iter[i:i + 1])[0]), current_iter=it,
                     robot_collided=bool(s.robot_collided[i]), poses_queue=fifo(s.poses_queue),
                     robot_state_queue=[self._robot_state(v) for v in fifo(s.robot_state_queue)],
                     control_queue=[Action(command=v) for v in fifo(s.control_queue)], pose=pose,
                     robot_state=self._robot_state(seen))

    VERSION = 1

    def serialize(self):
        """PlanEnv.serialize (env.py:251-261): this env, its parametrisation included, as basic python types;
        BatchedPlanEnv.deserialize builds a batch from such records."""
        st = self.get_state()
        return {'version': self.VERSION, 'state': st.serialize(), 'params': self._env.params.serialize(),
                'path': st.original_path, 'costmap': st.costmap.get_state()}

    def set_state(self, state):
        """PlanEnv.set_state (env.py:278-285) for this env: like the reference, the robot takes over
        `state.robot_state` (with a state delay that is the delayed state -- the reference does the same)."""
        e, i = self._env, self._i
        s = e.state

        def vec(rs):
            return [rs.x, rs.y, rs.angle, rs.v, rs.w, getattr(rs, "steering_motor_command", 0.0),
                    getattr(rs, "wheel_angle", 0.0)]

        s.robot[:, i] = torch.tensor(vec(state.robot_state), dtype=torch.float64)
        s.min_spat_dist_so_far[i] = state.reward_provider_state.min_spat_dist_so_far
        s.target_idx[i] = state.reward_provider_state.target_idx
        s.current_iter[i] = state.current_iter
        s.robot_collided[i] = int(state.robot_collided)
        if s.pose_seen is not None:
            s.pose_seen[:, i] = torch.tensor(np.asarray(state.pose, dtype=np.float64))
        if s.robot_state_seen is not None:
            s.robot_state_seen[:, i] = torch.tensor(vec(state.robot_state), dtype=torch.float64)
        it = int(state.current_iter)
        for q, items in ((s.poses_queue, [np.asarray(p, dtype=np.float64) for p in state.poses_queue]),
                         (s.robot_state_queue, [np.array(vec(r)) for r in state.robot_state_queue]),
                         (s.control_queue, [np.asarray(a.command, dtype=np.float64) for a in state.control_queue])):
            if q is None:
                continue
            d = q.shape[0]
            # the list holds pushes it - len + 1 .. it (oldest first); push k lives in slot (k - 1) % d
            for k, item in zip(range(it - len(items) + 1, it + 1), items):
                q[(k - 1) % d, :, i] = torch.tensor(item)

    def observation(self):
        s = self.get_state()
        return Observation(pose=s.pose, path=s.path, costmap=s.costmap, robot_state=s.robot_state,
                           time=s.current_time, dt=self._env.params.dt)


class _EnvViews(object):
    def __init__(self, env):
        self._env = env

    def __len__(self):
        return self._env.n_envs

    def __getitem__(self, i):
        if not -self._env.n_envs <= i < self._env.n_envs:
            raise IndexError(i)
        return EnvView(self._env, i % self._env.n_envs)


class BatchedPlanEnv(object):
    """N planning envs on one MI355X.

    :param costmap: CostMap2D shared by all envs, or a list of N CostMap2D of equal resolution (private maps)
    :param path: array(M, 3) shared oriented path, or a list of N such arrays (private paths)
    :param params EnvParams: as the reference (delays must be 0; continuous reward provider)
    :param n_envs int: number of envs on this device
    :param device: torch device / index of the GPU
    :param robot_name: robot model + footprint; default params.robot_name (PlanEnv itself always drives a tricycle)
    :param noise_parameters: 'planenv' = the odometry noise PlanEnv hard-codes on its tricycle (env.py:226-232; for a
        diff-drive robot, which PlanEnv never builds, that default means None), None = off, or a dict alpha1..alpha6.
        A diff-drive robot WITH noise raises IndexError like the reference (differential_drive.py:73) unless
        unpinned_diffdrive_noise=True opts in to the library's unpinned analogue
    :param auto_reset bool: restore an env's initial state right after the step that finished it
    :param env_id_base int: global index of env 0 (rank * n_envs when sharded over GPUs); keys the noise stream
    :param template_of_env: optional int array [n_envs]; `costmap` and `path` are then lists of T templates and env i
        gets a PRIVATE copy of costmap[template_of_env[i]] / path[template_of_env[i]] (built on the device)
    :param geom_of_env: optional int array [n_envs] -> GEOMETRY POOL mode: `costmap` and `path` are lists of G pool
        entries, env i runs on entry geom_of_env[i], and every reset (reset(), auto-reset) moves an env to
        next_geom[entry] (RandomMiniEnv.reset with draw_new_turn_on_reset, envs/mini_env.py:469-481).  Note that the
        constructor ends with reset(), like the reference's usage `env = RandomMiniEnv(); env.reset()`.
    :param next_geom: optional int array [G], successor of every pool entry; None = stay on the same entry
    :param map_storage: optional (rows, cols): private costmaps are stored with at least this (padded) shape, e.g.
        (256, 256) for BASELINE's "per-env 256x256 costmap"; the true shapes still bound the collision test
    """

    def __init__(self, costmap, path, params=None, n_envs=1, device=0, robot_name=None, noise_parameters='planenv',
                 auto_reset=False, env_id_base=0, seed=0, footprint_scale=1.0, dynamic_model=True,
                 model_front_column_pid=True, template_of_env=None, geom_of_env=None, next_geom=None, map_storage=None,
                 unpinned_diffdrive_noise=False):
        params = EnvParams() if params is None else params
        self.params = params
        self._pure_pursuit = params.reward_provider_name == CONTINUOUS_REWARD_PURE_PURSUIT
        self.n_envs = int(n_envs)
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.robot_name = params.robot_name if robot_name is None else robot_name
        self.is_tricycle = self.robot_name == INDUSTRIAL_TRICYCLE_V1
        if noise_parameters == 'planenv':
            noise_parameters = dict(robots.PLANENV_NOISE) if self.is_tricycle else None
        self.noise_parameters = noise_parameters
        self.auto_reset = bool(auto_reset)
        self._lib = _lib.load()  # raises when libbcplan.so is missing: no fallback
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedPlanEnv needs a GPU (libbcplan has no CPU path)")
        self._bcp_params = robots.make_bcp_params(params, self.robot_name, noise_parameters, footprint_scale,
                                                  dynamic_model, model_front_column_pid, unpinned_diffdrive_noise)
        self._h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self._lib.bcp_create(C.byref(self._bcp_params), self.n_envs, dev_index, int(env_id_base),
                                        C.byref(self._h)))
        self.action_space = Box(low=np.array([robots.MAX_FRONT_WHEEL_SPEED / 10, -np.pi / 2]),
                                high=np.array([robots.MAX_FRONT_WHEEL_SPEED / 2, np.pi / 2]), dtype=np.float32)
        self.reward_range = (0.0, 1.0)
        self.time_table = host_init.time_table(params.dt, params.iteration_timeout + 1)
        self._time_table_dev = torch.from_numpy(self.time_table).to(self.device)

        n, dev = self.n_envs, self.device
        def f64(*shape):
            return torch.zeros(*shape, dtype=torch.float64, device=dev)

        cd, pd, sd = int(params.control_delay), int(params.pose_delay), int(params.state_delay)
        self.state = BatchedState(f64(7, n), f64(n), torch.zeros(n, dtype=torch.int32, device=dev),
                                  torch.zeros(n, dtype=torch.int32, device=dev),
                                  torch.zeros(n, dtype=torch.uint8, device=dev),
                                  pose_seen=f64(3, n) if pd else None, robot_state_seen=f64(7, n) if sd else None,
                                  control_queue=f64(cd, 2, n) if cd else None, poses_queue=f64(pd, 3, n) if pd else None,
                                  robot_state_queue=f64(sd, 7, n) if sd else None)
        self.reward = torch.zeros(n, dtype=torch.float64, device=dev)
        self.done = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.collided_now = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.err = torch.zeros(n, dtype=torch.int32, device=dev)
        self.envs = _EnvViews(self)
        self._keep = {}  # device buffers the library holds pointers to

        self._map_storage = (0, 0) if map_storage is None else (int(map_storage[0]), int(map_storage[1]))
        self._template_of_env = None if template_of_env is None else np.asarray(template_of_env, dtype=np.int64)
        self.geom_of_env = None
        if geom_of_env is not None:
            assert template_of_env is None
            self._device_pool = costmap if isinstance(costmap, DeviceGeometryPool) else None
            if self._device_pool is not None:
                costmap, path = self._device_pool.costmaps, self._device_pool.paths
            else:
                costmap, path = list(costmap), list(path)
            g0 = np.asarray(geom_of_env, dtype=np.int32)
            assert g0.shape == (n,) and len(costmap) == len(path) and 0 <= g0.min() and g0.max() < len(costmap)
            self.geom_of_env = torch.from_numpy(g0.copy()).to(dev)
            nxt = None
            if next_geom is not None:
                nx = np.asarray(next_geom, dtype=np.int32)
                assert nx.shape == (len(costmap),) and 0 <= nx.min() and nx.max() < len(costmap)
                nxt = torch.from_numpy(nx.copy()).to(dev)
            self._keep.update(next_geom=nxt)
            _lib.check(self._lib.bcp_set_geometry_pool(self._h, len(costmap), self.geom_of_env.data_ptr(),
                                                       nxt.data_ptr() if nxt is not None else None))
            self._set_from_templates(costmap, path)
        elif self._template_of_env is not None:
            self._set_from_templates(list(costmap), list(path))
        else:
            self._set_costmaps(costmap)
            self._set_paths(path)
        self._bind(self.state, self._lib.bcp_bind_state)
        self._initial_state = self._make_initial_state()
        self._bind(self._initial_state, self._lib.bcp_bind_initial_state)
        # per-step call state, built once (the step path itself should cost microseconds of host time)
        self._io = _lib.BcpStepIO()
        self._io.reward = self.reward.data_ptr()
        self._io.done = self.done.data_ptr()
        self._io.collided_now = self.collided_now.data_ptr()
        self._io.err = self.err.data_ptr()
        self._io_ref = C.byref(self._io)
        self._bcp_step = self._lib.bcp_step
        self._flags_f64 = _lib.STEP_AUTO_RESET if self.auto_reset else 0
        self._flags_f32 = self._flags_f64 | _lib.STEP_ACTIONS_F32
        self._obs = BatchedObservation(self)
        self._info = {}
        self.seed(seed)
        self.reset()

    # ------------------------------------------------------------------ construction helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _bind(self, s, fn):
        st = _lib.BcpState()
        for k, name in enumerate(_STATE_FIELDS):
            setattr(st, name, s.robot[k].data_ptr())
        st.min_spat_dist_so_far = s.min_spat_dist_so_far.data_ptr()
        st.target_idx = s.target_idx.data_ptr()
        st.current_iter = s.current_iter.data_ptr()
        st.robot_collided = s.robot_collided.data_ptr()
        for name in BatchedState.DELAY_FIELDS:
            t = getattr(s, name)
            setattr(st, name, t.data_ptr() if t is not None else None)
        _lib.check(fn(self._h, C.byref(st)))

    def _set_costmaps(self, costmap):
        n = self.n_envs
        if isinstance(costmap, CostMap2D) or hasattr(costmap, "get_data") and not isinstance(costmap, (list, tuple)):
            self._costmaps, self._shared_map = [costmap], True
            data = np.ascontiguousarray(costmap.get_data(), dtype=np.uint8)
            rows, cols = data.shape
            origins = np.ascontiguousarray(costmap.get_origin(), dtype=np.float64)
            res = float(costmap.get_resolution())
            data_dev = torch.from_numpy(data).to(self.device)
            self._keep["map"] = data_dev
            self._origin_host = origins
            _lib.check(self._lib.bcp_set_costmaps(self._h, data_dev.data_ptr(), rows, cols, 1, None, None,
                                                  origins.ctypes.data, 0, res, self._stream()))
        else:
            costmaps = list(costmap)
            if len(costmaps) != n:
                raise ValueError("need one costmap per env (%d), got %d" % (n, len(costmaps)))
            res = float(costmaps[0].get_resolution())
            if any(float(c.get_resolution()) != res for c in costmaps):
                raise ValueError("all costmaps must share one resolution")
            self._costmaps, self._shared_map = costmaps, False
            rows = max(self._map_storage[0], max(c.get_data().shape[0] for c in costmaps))
            cols = max(self._map_storage[1], max(c.get_data().shape[1] for c in costmaps))
            data = np.zeros((n, rows, cols), dtype=np.uint8)
            vr = np.zeros(n, dtype=np.int32)
            vc = np.zeros(n, dtype=np.int32)
            origins = np.zeros((n, 2), dtype=np.float64)
            for i, c in enumerate(costmaps):
                d = c.get_data()
                data[i, :d.shape[0], :d.shape[1]] = d
                vr[i], vc[i] = d.shape
                origins[i] = c.get_origin()
            self.set_costmap_tensors(torch.from_numpy(data).to(self.device), torch.from_numpy(origins).to(self.device),
                                     res, torch.from_numpy(vr).to(self.device), torch.from_numpy(vc).to(self.device))
        self.resolution = res

    def _set_from_templates(self, costmaps, paths):
        """Private per-env costmaps / paths expanded on the device from a few templates."""
        n, dev = self.n_envs, self.device
        dp = getattr(self, "_device_pool", None)
        if dp is not None:   # everything is on the device already (refined paths included)
            g_n = len(dp)
            self._costmaps, self._paths, self._shared_path = costmaps, paths, False
            self.set_costmap_tensors(dp.maps, torch.from_numpy(np.tile(dp.origin, (g_n, 1))).to(dev), dp.resolution)
            self._keep.update(path=dp.path_points, lens=dp.lens)
            _lib.check(self._lib.bcp_set_paths(self._h, dp.path_points.data_ptr(), dp.lens.data_ptr(),
                                               int(dp.path_points.shape[1]), 0, self._stream()))
            torch.cuda.current_stream(dev).synchronize()
            return
        if self.geom_of_env is not None:   # geometry pool: the library indexes the entries itself
            idx = torch.arange(len(costmaps), device=dev)
        else:
            idx = torch.from_numpy(self._template_of_env).to(dev)
            assert idx.numel() == n and int(idx.max()) < len(costmaps) == len(paths)
        res = float(costmaps[0].get_resolution())
        rows = max(self._map_storage[0], max(c.get_data().shape[0] for c in costmaps))
        cols = max(self._map_storage[1], max(c.get_data().shape[1] for c in costmaps))
        t_data = np.zeros((len(costmaps), rows, cols), dtype=np.uint8)
        t_shape = np.zeros((len(costmaps), 2), dtype=np.int32)
        t_org = np.zeros((len(costmaps), 2), dtype=np.float64)
        for t, c in enumerate(costmaps):
            d = c.get_data()
            t_data[t, :d.shape[0], :d.shape[1]] = d
            t_shape[t] = d.shape
            t_org[t] = c.get_origin()
        data = torch.from_numpy(t_data).to(dev)[idx].contiguous()
        shape = torch.from_numpy(t_shape).to(dev)[idx]
        self._costmaps, self._shared_map = costmaps, False
        self.set_costmap_tensors(data, torch.from_numpy(t_org).to(dev)[idx].contiguous(), res,
                                 shape[:, 0].contiguous(), shape[:, 1].contiguous())
        refine = (lambda p: host_init.refine_path(p, self.params.path_delta)) if self.params.refine_path else (lambda p: p)
        tp = [np.ascontiguousarray(refine(np.asarray(p)), dtype=np.float64) for p in paths]
        max_len = max(len(p) for p in tp)
        buf = np.zeros((len(tp), max_len, 3), dtype=np.float64)
        for t, p in enumerate(tp):
            buf[t, :len(p)] = p
        self._paths, self._shared_path = tp, False
        pdev = torch.from_numpy(buf).to(dev)[idx].contiguous()
        lens = torch.from_numpy(np.array([len(p) for p in tp], dtype=np.int32)).to(dev)[idx].contiguous()
        self._keep.update(path=pdev, lens=lens)
        _lib.check(self._lib.bcp_set_paths(self._h, pdev.data_ptr(), lens.data_ptr(), max_len, 0, self._stream()))
        torch.cuda.current_stream(dev).synchronize()

    def set_costmap_tensors(self, data, origins, resolution, valid_rows=None, valid_cols=None):
        """Private costmaps straight from device tensors: data uint8 [N, rows, cols], origins float64 [N, 2]
        (N = pool entries in geometry-pool mode)."""
        n = self.n_envs if self.geom_of_env is None else data.shape[0]
        assert data.dtype == torch.uint8 and data.dim() == 3 and data.shape[0] == n and data.is_contiguous()
        assert origins.dtype == torch.float64 and tuple(origins.shape) == (n, 2) and origins.is_contiguous()
        self._keep.update(map=data, origins=origins, vr=valid_rows, vc=valid_cols)
        self._shared_map = False
        self.resolution = float(resolution)
        _lib.check(self._lib.bcp_set_costmaps(
            self._h, data.data_ptr(), data.shape[1], data.shape[2], 0,
            valid_rows.data_ptr() if valid_rows is not None else None,
            valid_cols.data_ptr() if valid_cols is not None else None, origins.data_ptr(), 1, float(resolution),
            self._stream()))

    def _set_paths(self, path):
        refine = (lambda p: host_init.refine_path(p, self.params.path_delta)) if self.params.refine_path else (lambda p: p)
        if isinstance(path, np.ndarray) and path.ndim == 2:
            p = np.ascontiguousarray(refine(path), dtype=np.float64)
            assert p.shape[1] == 3
            self._paths, self._shared_path = [p], True
            dev = torch.from_numpy(p).to(self.device)
            self._keep["path"] = dev
            _lib.check(self._lib.bcp_set_paths(self._h, dev.data_ptr(), None, p.shape[0], 1, self._stream()))
        else:
            paths = [np.ascontiguousarray(refine(np.asarray(p)), dtype=np.float64) for p in path]
            if len(paths) != self.n_envs:
                raise ValueError("need one path per env (%d), got %d" % (self.n_envs, len(paths)))
            self._paths, self._shared_path = paths, False
            max_len = max(len(p) for p in paths)
            buf = np.zeros((self.n_envs, max_len, 3), dtype=np.float64)
            lens = np.zeros(self.n_envs, dtype=np.int32)
            for i, p in enumerate(paths):
                buf[i, :len(p)] = p
                lens[i] = len(p)
            dev, lens_dev = torch.from_numpy(buf).to(self.device), torch.from_numpy(lens).to(self.device)
            self._keep.update(path=dev, lens=lens_dev)
            _lib.check(self._lib.bcp_set_paths(self._h, dev.data_ptr(), lens_dev.data_ptr(), max_len, 0,
                                               self._stream()))
        torch.cuda.current_stream(self.device).synchronize()  # the [.,3] staging tensor may now be released

    def _make_initial_state(self):
        """make_initial_state (env.py:179-214): pose = path[0], v = w = 0, wheel at initial_wheel_angle... the
        reference's TricycleRobotState() default wheel angle is 0.0 and PlanEnv never applies
        params.initial_wheel_angle to it, so neither do we."""
        n = self.n_envs if self.geom_of_env is None else len(self._paths)
        dp = getattr(self, "_device_pool", None)
        if dp is not None:   # initial states computed on the device with the paths (bcp_mini_world_paths)
            dev = self.device
            robot_t = torch.zeros(7, n, dtype=torch.float64, device=dev)
            robot_t[0:3] = dp.path_points[:, 0, :].t()
            return BatchedState(robot_t, dp.init[:, 0].contiguous(), dp.init[:, 1].to(torch.int32).contiguous(),
                                torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev))
        robot = np.zeros((7, n), dtype=np.float64)
        md = np.zeros(n, dtype=np.float64)
        ti = np.zeros(n, dtype=np.int32)
        rp = self.params.reward_provider_params
        if self.geom_of_env is not None:   # one initial state per pool entry
            for g, p in enumerate(self._paths):
                md[g], ti[g] = self._first_reward_state(p, rp)
                robot[0:3, g] = p[0]
        elif self._shared_path:
            p = self._paths[0]
            m0, t0 = self._first_reward_state(p, rp)
            robot[0:3, :] = p[0][:, None]
            md[:], ti[:] = m0, t0
        elif self._template_of_env is not None:
            per = [self._first_reward_state(p, rp) for p in self._paths]
            tix = self._template_of_env
            md[:] = np.array([m for m, _ in per])[tix]
            ti[:] = np.array([t for _, t in per])[tix]
            robot[0:3, :] = np.stack([p[0] for p in self._paths])[tix].T
        else:
            for i, p in enumerate(self._paths):
                md[i], ti[i] = self._first_reward_state(p, rp)
                robot[0:3, i] = p[0]
        dev = self.device
        return BatchedState(torch.from_numpy(robot).to(dev), torch.from_numpy(md).to(dev), torch.from_numpy(ti).to(dev),
                            torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev))

    def _first_reward_state(self, path, reward_params):
        if self._pure_pursuit:
            return host_init.initial_pure_pursuit_state(path)
        return host_init.initial_reward_state(path, reward_params)

    @classmethod
    def deserialize(cls, records, **kw):
        """PlanEnv.deserialize (env.py:263-276), batched: one env per record of `EnvView.serialize()` (or of the
        reference's `PlanEnv.serialize()`), each with its own costmap and path; the parametrisation is the first
        record's (they must agree).  Extra keyword arguments go to the constructor (device, seed, auto_reset, ...)."""
        records = [dict(r) for r in records]
        for r in records:
            assert r.pop('version') == EnvView.VERSION
        params = EnvParams.deserialize(records[0]['params'])
        if any(EnvParams.deserialize(r['params']) != params for r in records[1:]):
            raise ValueError("all envs of a batch share one EnvParams")
        costmaps = [CostMap2D.from_state(r['costmap']) for r in records]
        paths = [np.asarray(r['path'], dtype=np.float64) for r in records]
        # `path` is State.original_path, i.e. already refined: it is taken as it is.  (The reference's deserialize goes
        # through the constructor's refine_path once more -- a no-op except for segments within rounding of path_delta --
        # but then set_state puts the recorded path back anyway.)
        env = cls(costmaps, paths, attr.evolve(params, refine_path=False), n_envs=len(records), **kw)
        env.params = params
        for i, r in enumerate(records):
            env.envs[i].set_state(State.deserialize(r['state']))
        return env

    def set_tuning(self, exact_mode=None, dense_threshold=None, cull=None, defer=None, edt_lds=None, fused=None,
                   ego_sparse=None, near_dilate=None, local_pairs=None, ego_list_stride=None, near_shift=None):
        """Execution knobs of libbcplan (bcp_set_tuning); results never depend on them.
        (near_shift takes effect when the costmaps are bound the next time.)"""
        for key, val in ((_lib.TUNE_EXACT_MODE, exact_mode), (_lib.TUNE_DENSE_THRESHOLD, dense_threshold),
                         (_lib.TUNE_CULL, cull), (_lib.TUNE_DEFER, defer), (_lib.TUNE_EDT_LDS, edt_lds),
                         (_lib.TUNE_FUSED, fused), (_lib.TUNE_EGO_SPARSE, ego_sparse), (_lib.TUNE_LOCAL_PAIRS, local_pairs), (_lib.TUNE_EGO_LIST_STRIDE, ego_list_stride),
                         (_lib.TUNE_NEAR_DILATE, near_dilate), (_lib.TUNE_NEAR_SHIFT, near_shift)):
            if val is not None:
                _lib.check(self._lib.bcp_set_tuning(self._h, key, int(val)))

    def distance_field(self, first=0, count=1):
        """The distance fields libbcplan pre-classifies poses with (bcp_get_distance_field), for `count` map entries
        from `first`: (uint8 device tensor [count, rows + 2 pad, cols + 2 pad], pad, clamp)."""
        shape = (C.c_int32 * 4)()
        _lib.check(self._lib.bcp_get_distance_field(self._h, 0, 0, None, shape, None))
        out = torch.empty((int(count), shape[0], shape[1]), dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.bcp_get_distance_field(self._h, int(first), int(count), out.data_ptr(), shape, self._stream()))
        return out, int(shape[2]), int(shape[3])

    def near_field(self, first=0, count=1, raw=False):
        """The 1-bit form of the distance fields the step's outer test reads (bcp_get_near_field), unpacked:
        (bool device tensor [count, rows + 2 pad, cols + 2 pad] -- True where the field is < t_out --, t_out);
        raw: the tile words themselves, int32 [count, tile rows, tile columns, 32]."""
        shape = (C.c_int32 * 3)()
        _lib.check(self._lib.bcp_get_near_field(self._h, 0, 0, None, shape, None))
        ty, tx, t_out = int(shape[0]), int(shape[1]), int(shape[2])
        words = torch.empty((int(count), ty, tx, 32), dtype=torch.int32, device=self.device)
        _lib.check(self._lib.bcp_get_near_field(self._h, int(first), int(count), words.data_ptr(), shape, self._stream()))
        dshape = (C.c_int32 * 4)()
        _lib.check(self._lib.bcp_get_distance_field(self._h, 0, 0, None, dshape, None))
        bits = (words.unsqueeze(-1) >> torch.arange(32, device=self.device, dtype=torch.int32)) & 1   # [count, ty, tx, 32 rows, 32 bits]
        cells = bits.permute(0, 1, 3, 2, 4).reshape(int(count), ty * 32, tx * 32)
        if raw:
            return words, t_out
        return cells[:, :dshape[0], :dshape[1]].bool(), t_out

    # ------------------------------------------------------------------ per-env lookups
    def path_of(self, i):
        if self.geom_of_env is not None:
            return self._paths[int(self.geom_of_env[i])]
        if self._template_of_env is not None:
            return self._paths[self._template_of_env[i]]
        return self._paths[0] if self._shared_path else self._paths[i]

    def costmap_of(self, i):
        if self.geom_of_env is not None:
            return self._costmaps[int(self.geom_of_env[i])]
        if self._template_of_env is not None:
            return self._costmaps[self._template_of_env[i]]
        if self._shared_map or len(self._costmaps) == 1:
            return self._costmaps[0]
        return self._costmaps[i]

    def time_of(self, current_iter):
        """Observation.time for iteration counters (device tensor): dt accumulated current_iter times."""
        idx = current_iter.to(torch.int64)
        top = int(idx.max()) if idx.numel() else 0
        if top >= len(self.time_table):  # envs stepped past the timeout without a reset: grow the table
            self.time_table = host_init.time_table(self.params.dt, 2 * top)
            self._time_table_dev = torch.from_numpy(self.time_table).to(self.device)
        return self._time_table_dev[idx]

    # ------------------------------------------------------------------ reference API
    def seed(self, seed=None):
        """Seeds the on-device odometry-noise stream (the reference draws from numpy's global RNG)."""
        if seed is not None:
            _lib.check(self._lib.bcp_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def reset(self, mask=None):
        """PlanEnv.reset for all envs, or for those with mask[i] != 0 (uint8/bool device tensor)."""
        ptr = None
        if mask is not None:
            mask = mask.to(self.device).to(torch.uint8).contiguous()
            ptr = mask.data_ptr()
        _lib.check(self._lib.bcp_reset_masked(self._h, ptr, self._stream()))
        return self._obs

    def get_state(self):
        """Snapshot of every env's state (device tensors); with a geometry pool the entries the envs are on ride
        along as `.geom_of_env`."""
        snap = self.state.copy()
        snap.geom_of_env = self.geom_of_env.clone() if self.geom_of_env is not None else None
        return snap

    def set_state(self, state):
        s = self.state
        s.robot.copy_(state.robot)
        s.min_spat_dist_so_far.copy_(state.min_spat_dist_so_far)
        s.target_idx.copy_(state.target_idx)
        s.current_iter.copy_(state.current_iter)
        s.robot_collided.copy_(state.robot_collided)
        for name in BatchedState.DELAY_FIELDS:
            if getattr(s, name) is not None:
                getattr(s, name).copy_(getattr(state, name))
        if self.geom_of_env is not None and getattr(state, "geom_of_env", None) is not None:
            self.geom_of_env.copy_(state.geom_of_env)

    def fan_out(self, src, mask=None):
        """Monte-Carlo fan-out (reference README, 'Statefullness of the env'): every env (or those with mask[i] != 0)
        takes over env `src`'s complete state -- the batched `s = env.get_state(); others.set_state(s)`."""
        ptr = None
        if mask is not None:
            mask = mask.to(self.device).to(torch.uint8).contiguous()
            ptr = mask.data_ptr()
        _lib.check(self._lib.bcp_broadcast_state(self._h, int(src), ptr, self._stream()))
        self._last_inputs = (mask,)   # keep the mask alive until the stream has consumed it

    def step(self, actions, noise_z=None, noise_z_out=None, done_out=None):
        """One tick for every env.  actions: [N,2] (float32 or float64) tensor / array, or a list of Action.
        noise_z: optional [N,3] float64 standard normals (slot order) replacing the on-device RNG.
        done_out: optional uint8 [N] device tensor that receives the done mask instead of `self.done` (e.g. a row of
        a ring buffer that is all-gathered every few steps).
        Returns (BatchedObservation, reward float64[N], done uint8[N], {}) -- device tensors, no sync."""
        # fast path: a device tensor of the right shape and dtype goes straight to the library
        if not (isinstance(actions, torch.Tensor) and actions.device == self.device and actions.is_contiguous()
                and actions.dtype in (torch.float32, torch.float64) and tuple(actions.shape) == (self.n_envs, 2)):
            actions = _as_device_actions(actions, self.n_envs, self.device)
        io = self._io
        io.actions = actions.data_ptr()
        flags = self._flags_f32 if actions.dtype == torch.float32 else self._flags_f64
        z = None
        if noise_z is not None:
            z = noise_z if isinstance(noise_z, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(noise_z))
            z = z.to(self.device, torch.float64).contiguous()
            assert tuple(z.shape) == (self.n_envs, 3)
            io.noise_z = z.data_ptr()
        else:
            io.noise_z = None
        if noise_z_out is not None:
            assert noise_z_out.dtype == torch.float64 and tuple(noise_z_out.shape) == (self.n_envs, 3)
            io.noise_z_out = noise_z_out.data_ptr()
        else:
            io.noise_z_out = None
        done = self.done
        if done_out is not None:
            assert done_out.dtype == torch.uint8 and done_out.numel() == self.n_envs and done_out.is_contiguous()
            done = done_out
        io.done = done.data_ptr()
        rc = self._bcp_step(self._h, self._io_ref, flags, torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            _lib.check(rc)
        self._last_inputs = (actions, z, done)  # keep inputs alive until the stream has consumed them
        return self._obs, self.reward, done, self._info

    def rollout(self, actions, noise_z=None, noise_z_out=None, collided_out=None, err_out=None):
        """K ticks for every env in one library call (bcp_rollout): actions [K, N, 2] float32 / float64 on the device (or
        anything torch can put there).  Returns (reward float64 [K, N], done uint8 [K, N]) device tensors, no sync; the
        state ends where K calls of step() with actions[k] would leave it, bit for bit (auto-reset and the on-device
        noise stream included).  noise_z / noise_z_out: optional [K, N, 3]; collided_out uint8 / err_out int32: optional
        [K, N] (without them only the last step's collided_now / err are kept).  With the single-launch step form the K
        steps are ONE kernel launch -- open-loop Monte-Carlo rollouts from one state (the reference's README) pay launch,
        argument fetch and staging once, and no workgroup waits for the chip's slowest one between steps."""
        if not isinstance(actions, torch.Tensor):
            actions = torch.from_numpy(np.ascontiguousarray(actions))
        if actions.dtype not in (torch.float32, torch.float64):
            actions = actions.to(torch.float64)
        actions = actions.to(self.device).contiguous()
        k, n = int(actions.shape[0]), self.n_envs
        assert tuple(actions.shape) == (k, n, 2) and k >= 1
        io = _lib.BcpStepIO()
        io.actions = actions.data_ptr()
        flags = self._flags_f32 if actions.dtype == torch.float32 else self._flags_f64
        keep = [actions]
        if noise_z is not None:
            z = noise_z if isinstance(noise_z, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(noise_z))
            z = z.to(self.device, torch.float64).contiguous()
            assert tuple(z.shape) == (k, n, 3)
            io.noise_z = z.data_ptr()
            keep.append(z)
        if noise_z_out is not None:
            assert noise_z_out.dtype == torch.float64 and tuple(noise_z_out.shape) == (k, n, 3) and noise_z_out.is_contiguous()
            io.noise_z_out = noise_z_out.data_ptr()
        reward = torch.empty((k, n), dtype=torch.float64, device=self.device)
        done = torch.empty((k, n), dtype=torch.uint8, device=self.device)
        io.reward, io.done = reward.data_ptr(), done.data_ptr()
        if collided_out is not None:
            assert collided_out.dtype == torch.uint8 and tuple(collided_out.shape) == (k, n) and collided_out.is_contiguous()
            io.collided_now = collided_out.data_ptr()
        if err_out is not None:
            assert err_out.dtype == torch.int32 and tuple(err_out.shape) == (k, n) and err_out.is_contiguous()
            io.err = err_out.data_ptr()
        _lib.check(self._lib.bcp_rollout(self._h, C.byref(io), k, flags, self._stream()))
        # the per-step views of step() show the last row
        self.reward.copy_(reward[-1])
        self.done.copy_(done[-1])
        if collided_out is not None:
            self.collided_now.copy_(collided_out[-1])
        if err_out is not None:
            self.err.copy_(err_out[-1])
        self._last_inputs = tuple(keep)
        return reward, done

    def check_errors(self):
        """Raise what the reference would have raised during the last step (synchronises); and a RuntimeError if a
        wait inside the step kernel ever gave up (bcp_expired_waits: a defect of the library, not of the data)."""
        gave_up = C.c_int64()
        _lib.check(self._lib.bcp_expired_waits(self._h, C.byref(gave_up), self._stream()))
        if gave_up.value:
            raise RuntimeError("libbcplan: %d bounded wait(s) of the step kernel ran into their limit" % gave_up.value)
        bad = torch.nonzero(self.err & _lib.ERR_ANGLE_JUMP).flatten()
        if len(bad):
            raise Exception("Path has missing/corrupted angle data at env indices: %s" % bad.cpu().numpy())

    def geometry_digest(self):
        """Digest of the geometry this rank's envs share (distributed.geometry_digest): the costmap(s) and path(s) as they
        were given -- for distributed.check_same_geometry at set-up of a sharded job."""
        import numpy as np
        from . import distributed
        parts = []
        dp = getattr(self, "_device_pool", None)
        if dp is not None:   # a pool that lives on the GPU: its first entries stand for it
            return distributed.geometry_digest(dp.maps[:16].cpu().numpy(), dp.origin, np.float64(dp.resolution),
                                               dp.path_points[:16].cpu().numpy(), dp.lens[:16].cpu().numpy())
        for cm in self._costmaps[:16]:
            parts += [cm.get_data(), np.asarray(cm.get_origin(), dtype=np.float64), np.float64(cm.get_resolution())]
        parts += [np.asarray(p, dtype=np.float64) for p in self._paths[:16]]
        return distributed.geometry_digest(*parts)

    def parked_poses(self):
        """Poses the single-launch step handed to the exact footprint test since the env was created (bcp_parked_poses;
        synchronises).  Measurement only."""
        count = C.c_int64()
        _lib.check(self._lib.bcp_parked_poses(self._h, C.byref(count), self._stream()))
        return int(count.value)

    def _timing_io(self, actions, noise_z):
        a = _as_device_actions(actions, self.n_envs, self.device)
        io = _lib.BcpStepIO()
        io.actions = a.data_ptr()
        flags = (_lib.STEP_ACTIONS_F32 if a.dtype == torch.float32 else 0) | (_lib.STEP_AUTO_RESET if self.auto_reset else 0)
        if noise_z is not None:
            io.noise_z = noise_z.data_ptr()
        io.reward, io.done = self.reward.data_ptr(), self.done.data_ptr()
        io.collided_now, io.err = self.collided_now.data_ptr(), self.err.data_ptr()
        return a, io, flags

    STEP_FORMS = {0: "step_kernel", 1: "step_fast_pair_kernel",
                  2: "step_fast_pair_kernel + step_pending_kernel (one step = these two launches)",
                  3: "step_local_kernel"}

    def step_kernels(self):
        """The kernels one step() launches as the handle is configured now (bcp_step_form)."""
        form = self._lib.bcp_step_form(self._h)
        if form < 0:
            _lib.check(form)
        return self.STEP_FORMS[form]

    def time_steps(self, actions, steps, noise_z=None):
        """Average device time (ms) of one step over `steps` back-to-back steps, measured with HIP events on the
        launch stream."""
        a, io, flags = self._timing_io(actions, noise_z)
        ms = C.c_float()
        _lib.check(self._lib.bcp_time_steps(self._h, C.byref(io), flags, int(steps), self._stream(), C.byref(ms)))
        return ms.value

    def time_step_kernels(self, actions, steps, noise_z=None):
        """(step_kernel ms, step_pending_kernel ms): average launch durations, HIP events around each launch."""
        a, io, flags = self._timing_io(actions, noise_z)
        ms = (C.c_float * 2)()
        _lib.check(self._lib.bcp_time_step_kernels(self._h, C.byref(io), flags, int(steps), self._stream(), ms))
        return ms[0], ms[1]

    def render(self, mode='human'):
        raise NotImplementedError("rendering is out of scope of the batched step path")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.bcp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
