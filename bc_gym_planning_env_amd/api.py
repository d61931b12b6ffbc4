"""Host-side mirror of the reference's boundary types for the PlanEnv.step() path.

Same names, field names, defaults and meaning as the reference, so code written against
`bc_gym_planning_env` keeps working on per-env views of the batched env:
  Action                 envs/base/action.py:11-19
  Observation            envs/base/obs.py:14-23
  EnvParams              envs/base/params.py:14-42
  RewardParams           envs/base/reward.py:162-171
  ContinuousRewardProviderState   envs/base/reward.py:12-75
  State                  envs/base/env.py:52-68
  TricycleRobotState     robot_models/tricycle_model.py:234-290
  DiffdriveRobotState    robot_models/differential_drive.py:77-124
  CostMap2D              utilities/costmap_2d.py:13-174
  Box (action space)     envs/base/spaces.py:80-145
Only what the step path touches is mirrored (no GUI, no drawing, no (de)serialisation plumbing).
"""
import attr
import numpy as np

INDUSTRIAL_TRICYCLE_V1 = 'industrial_tricycle_v1'
INDUSTRIAL_DIFFDRIVE_V1 = 'industrial_diffdrive_v1'
class Serializable(object):
    """utilities/serialize.py:8-33: render to / rebuild from a dict of basic types (int, float, np.ndarray, ...)"""
    VERSION = 1

    @classmethod
    def deserialize(cls, state):
        state = dict(state)
        ver = state.pop('version')
        assert ver == cls.VERSION
        return cls(**state)

    def serialize(self):
        state = attr.asdict(self)
        state['version'] = self.VERSION
        return state


CONTINUOUS_REWARD = 'continuous_reward'
CONTINUOUS_REWARD_STATE = 'continuous_reward_state'
CONTINUOUS_REWARD_PURE_PURSUIT_STATE = 'continuous_reward_pure_pursuit_state'
CONTINUOUS_REWARD_PURE_PURSUIT = 'continuous_reward_pure_pursuit'


@attr.s(eq=False)
class Action(object):
    """A motion primitive: command = (wheel_v, wheel_angle) for the tricycle, (v, w) for diff-drive."""
    command = attr.ib(type=np.ndarray)

    def __eq__(self, other):
        return isinstance(other, Action) and not (self.command != other.command).any()

    def __ne__(self, other):
        return not self.__eq__(other)


@attr.s
class RewardParams(Serializable):
    """envs/base/reward.py:39-53"""
    spatial_precision = attr.ib(type=float)
    angular_precision = attr.ib(type=float)
    spatial_progress_multiplier = attr.ib(type=float, default=0.0)


@attr.s(frozen=True)
class EnvParams(Serializable):
    """envs/base/params.py:14-59 (serialize / deserialize nest the reward provider's parameters)"""
    dt = attr.ib(type=float, default=0.05)
    goal_ang_dist = attr.ib(type=float, default=np.pi / 2)
    goal_spat_dist = attr.ib(type=float, default=1.0)
    initial_wheel_angle = attr.ib(default=0.0, type=float)
    iteration_timeout = attr.ib(type=int, default=1200)
    path_limiter_max_dist = attr.ib(type=float, default=5.0)
    robot_name = attr.ib(default=INDUSTRIAL_TRICYCLE_V1)
    resolution = attr.ib(default=0.03, type=float)
    refine_path = attr.ib(default=True, type=bool)
    path_delta = attr.ib(default=0.05, type=float)
    pose_delay = attr.ib(default=0, type=int)
    control_delay = attr.ib(default=0, type=int)
    state_delay = attr.ib(default=0, type=int)
    reward_provider_name = attr.ib(default=CONTINUOUS_REWARD)
    reward_provider_params = attr.ib(
        default=attr.Factory(lambda self: RewardParams(spatial_precision=self.goal_spat_dist,
                                                       angular_precision=self.goal_ang_dist), takes_self=True))

    def serialize(self):
        out = attr.asdict(self)
        out['version'] = self.VERSION
        out['reward_provider_params'] = self.reward_provider_params.serialize()
        return out

    @classmethod
    def deserialize(cls, state):
        state = dict(state)
        ver = state.pop('version')
        assert ver == cls.VERSION
        state['reward_provider_params'] = RewardParams.deserialize(state['reward_provider_params'])
        return cls(**state)


class CostMap2D(object):
    """uint8 occupancy grid with a world origin (position of data[0, 0]) and a resolution in m/px."""
    FREE_SPACE = 0
    LETHAL_OBSTACLE = 254
    NO_INFORMATION = 255

    def __init__(self, data, resolution, origin):
        origin = np.array(origin, dtype=np.float64)
        origin.setflags(write=False)
        self._data, self._resolution, self._origin = data, resolution, origin

    def get_data(self):
        return self._data

    def get_resolution(self):
        return self._resolution

    def get_origin(self):
        return self._origin

    def copy(self):
        return CostMap2D(self._data.copy(), self._resolution, self._origin.copy())

    def get_state(self):
        """utilities/costmap_2d.py:150-160"""
        return dict(version=1, data=self._data, resolution=self._resolution, origin=self._origin)

    @classmethod
    def from_state(cls, state):
        assert state['version'] == 1
        return cls(state['data'], state['resolution'], state['origin'])

    def __eq__(self, other):
        return (isinstance(other, CostMap2D) and self._resolution == other.get_resolution()
                and (self._origin == other.get_origin()).all() and self._data.shape == other.get_data().shape
                and (self._data == other.get_data()).all())

    def __ne__(self, other):
        return not self.__eq__(other)


@attr.s
class TricycleRobotState(Serializable):
    x = attr.ib(default=0.0, type=float)
    y = attr.ib(default=0.0, type=float)
    angle = attr.ib(default=0.0, type=float)
    v = attr.ib(default=0.0, type=float)
    w = attr.ib(default=0.0, type=float)
    steering_motor_command = attr.ib(default=0.0, type=float)
    wheel_angle = attr.ib(default=0.0, type=float)
    robot_type_name = INDUSTRIAL_TRICYCLE_V1

    def copy(self):
        return attr.evolve(self)

    def get_pose(self):
        return self.x, self.y, self.angle

    def set_pose(self, pose):
        self.x, self.y, self.angle = pose

    def to_numpy_array(self):
        return np.array([self.x, self.y, self.angle, self.v, self.w, self.wheel_angle], dtype=np.float64)

    def egocentric_state_numpy_array(self):
        return np.array([self.v, self.w, self.wheel_angle], dtype=np.float64)

    def get_robot_type_name(self):
        return self.robot_type_name


@attr.s
class DiffdriveRobotState(Serializable):
    x = attr.ib(default=0.0, type=float)
    y = attr.ib(default=0.0, type=float)
    angle = attr.ib(default=0.0, type=float)
    v = attr.ib(default=0.0, type=float)
    w = attr.ib(default=0.0, type=float)
    robot_type_name = INDUSTRIAL_DIFFDRIVE_V1

    def copy(self):
        return attr.evolve(self)

    def get_pose(self):
        return np.array([self.x, self.y, self.angle])

    def set_pose(self, pose):
        self.x, self.y, self.angle = pose

    def to_numpy_array(self):
        return np.array([self.x, self.y, self.angle, self.v, self.w], dtype=np.float64)

    def egocentric_state_numpy_array(self):
        return np.array([self.v, self.w, 0.0], dtype=np.float64)

    def get_robot_type_name(self):
        return self.robot_type_name


@attr.s(eq=False)
class ContinuousRewardProviderState(Serializable):
    reward_provider_state_type_name = CONTINUOUS_REWARD_STATE

    min_spat_dist_so_far = attr.ib(type=float)
    path = attr.ib(type=np.ndarray)
    target_idx = attr.ib(type=int)

    def copy(self):
        return attr.evolve(self, path=np.copy(self.path))

    def current_goal_pose(self):
        if self.target_idx < len(self.path):
            return self.path[self.target_idx]
        raise ValueError("No path left to follow.")

    def current_path(self):
        return self.path[self.target_idx:]

    def done(self):
        return self.target_idx > len(self.path) - 1

    def __eq__(self, other):
        return (isinstance(other, ContinuousRewardProviderState) and not (self.path != other.path).any()
                and self.min_spat_dist_so_far == other.min_spat_dist_so_far and self.target_idx == other.target_idx)

    def __ne__(self, other):
        return not self.__eq__(other)


@attr.s(eq=False)
class ContinuousRewardPurePursuitProviderState(Serializable):
    reward_provider_state_type_name = CONTINUOUS_REWARD_PURE_PURSUIT_STATE

    """envs/base/reward.py:76-159: the goal is always the last way point, target_idx the look-ahead way point"""
    min_spat_dist_so_far = attr.ib(type=float)
    path = attr.ib(type=np.ndarray)
    target_idx = attr.ib(type=int, default=0)

    def copy(self):
        return attr.evolve(self, path=np.copy(self.path))

    def current_goal_pose(self):
        return self.path[-1]

    def current_path(self):
        return self.path[:self.target_idx + 1]

    def __eq__(self, other):
        return (isinstance(other, ContinuousRewardPurePursuitProviderState) and not (self.path != other.path).any()
                and self.min_spat_dist_so_far == other.min_spat_dist_so_far and self.target_idx == other.target_idx)

    def __ne__(self, other):
        return not self.__eq__(other)


@attr.s(frozen=True, eq=False)
class Observation(object):
    pose = attr.ib(type=np.ndarray)
    path = attr.ib(type=np.ndarray, repr=False)
    costmap = attr.ib(type=CostMap2D)
    time = attr.ib(type=float)
    dt = attr.ib(type=float)
    robot_state = attr.ib(type=object)


@attr.s(eq=False)
class State(Serializable):
    """Full per-env state, field for field the reference's State (envs/base/env.py:52-176)."""
    reward_provider_state = attr.ib(type=object)
    path = attr.ib(type=np.ndarray)
    original_path = attr.ib(type=np.ndarray)
    costmap = attr.ib(type=CostMap2D)
    iter_timeout = attr.ib(type=int)
    current_time = attr.ib(type=float)
    current_iter = attr.ib(type=int)
    robot_collided = attr.ib(type=bool)
    poses_queue = attr.ib(type=list)
    robot_state_queue = attr.ib(type=list)
    control_queue = attr.ib(type=list)
    pose = attr.ib(type=np.ndarray)
    robot_state = attr.ib(type=object)

    def copy(self):
        return attr.evolve(self, reward_provider_state=self.reward_provider_state.copy(), path=np.copy(self.path),
                           pose=np.copy(self.pose), original_path=np.copy(self.original_path),
                           costmap=self.costmap.copy(), poses_queue=[np.copy(p) for p in self.poses_queue],
                           robot_state_queue=[r.copy() for r in self.robot_state_queue],
                           control_queue=list(self.control_queue), robot_state=self.robot_state.copy())

    def serialize(self):
        """envs/base/env.py:163-176, key for key.  The reference renders the state with a recursive attr.asdict first,
        so the entries of robot_state_queue are plain field dicts (no 'version') and those of control_queue are
        {'command': array} dicts; robot_state and reward_provider_state are then replaced by their own serialize()."""
        resu = attr.asdict(self, recurse=False)
        resu['version'] = self.VERSION
        resu['costmap'] = self.costmap.get_state()
        resu['reward_provider_state_type_name'] = self.reward_provider_state.reward_provider_state_type_name
        resu['reward_provider_state'] = self.reward_provider_state.serialize()
        resu['robot_type_name'] = self.robot_state.get_robot_type_name()
        resu['robot_state'] = self.robot_state.serialize()
        resu['robot_state_queue'] = [attr.asdict(r) for r in self.robot_state_queue]
        resu['control_queue'] = [{'command': np.asarray(a.command)} for a in self.control_queue]
        resu['poses_queue'] = list(self.poses_queue)
        return resu

    @classmethod
    def deserialize(cls, state):
        """envs/base/env.py:135-161.  Reads what State.serialize writes -- ours and the reference's own records.  (The
        reference cannot read its own: it writes 'reward_provider_state_type_name' and pops 'reward_provider_state_name',
        pops a 'version' the queue entries do not have, and leaves control_queue entries as dicts; all of these forms are
        accepted here, as are round 1's records with bare command arrays.)"""
        state = dict(state)
        assert state.pop('version') == cls.VERSION
        state['costmap'] = CostMap2D.from_state(state['costmap'])
        kind = state.pop('reward_provider_state_type_name', None) or state.pop('reward_provider_state_name')
        rp_cls = {CONTINUOUS_REWARD_STATE: ContinuousRewardProviderState,
                  CONTINUOUS_REWARD_PURE_PURSUIT_STATE: ContinuousRewardPurePursuitProviderState}[kind]
        state['reward_provider_state'] = rp_cls.deserialize(state['reward_provider_state'])
        rs_cls = {INDUSTRIAL_TRICYCLE_V1: TricycleRobotState, INDUSTRIAL_DIFFDRIVE_V1: DiffdriveRobotState}[
            state.pop('robot_type_name')]
        state['robot_state'] = rs_cls.deserialize(state['robot_state'])
        state['robot_state_queue'] = [rs_cls.deserialize(dict(r, version=r.get('version', rs_cls.VERSION)))
                                      for r in state['robot_state_queue']]
        state['control_queue'] = [Action(command=np.asarray(c['command'] if isinstance(c, dict) else c))
                                  for c in state['control_queue']]
        state['poses_queue'] = [np.asarray(p) for p in state['poses_queue']]
        return cls(**state)

    def __eq__(self, other):
        if not isinstance(other, State):
            return False
        same_arrays = all(np.shape(a) == np.shape(b) and (np.asarray(a) == np.asarray(b)).all() for a, b in (
            (self.path, other.path), (self.original_path, other.original_path), (self.pose, other.pose)))
        queues = (len(self.poses_queue) == len(other.poses_queue)
                  and all((np.asarray(a) == np.asarray(b)).all() for a, b in zip(self.poses_queue, other.poses_queue))
                  and self.robot_state_queue == other.robot_state_queue
                  and len(self.control_queue) == len(other.control_queue)
                  and all((np.asarray(a.command) == np.asarray(b.command)).all()
                          for a, b in zip(self.control_queue, other.control_queue)))
        return (same_arrays and queues and self.reward_provider_state == other.reward_provider_state
                and self.costmap == other.costmap and self.iter_timeout == other.iter_timeout
                and self.current_time == other.current_time and self.current_iter == other.current_iter
                and self.robot_collided == other.robot_collided and self.robot_state == other.robot_state)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


_SPACE_RNG = np.random.RandomState(0)  # module-level stream seeded 0, as envs/base/spaces.py:9-10


class Box(object):
    """Action space: Box(low, high) of float32 with sample() -> Action (envs/base/spaces.py:80-145)."""

    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low).astype(dtype)
        self.high = np.asarray(high).astype(dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def sample(self):
        v, w = _SPACE_RNG.uniform(low=self.low, high=self.high, size=self.low.shape).astype(self.dtype)
        return Action(command=np.array([v, w]))

    def sample_batch(self, n, rng=None):
        """[n, 2] float32 commands drawn uniformly from the box."""
        rng = _SPACE_RNG if rng is None else rng
        return rng.uniform(low=self.low, high=self.high, size=(n,) + self.low.shape).astype(self.dtype)

    def contains(self, x):
        return x.shape == self.shape and (x >= self.low).all() and (x <= self.high).all()


def seed_action_space(seed):
    _SPACE_RNG.seed(seed)
