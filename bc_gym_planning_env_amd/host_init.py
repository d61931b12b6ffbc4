"""Reset-time host logic (numpy): what PlanEnv.__init__ / make_initial_state do once per episode, before the
first step.  Not on the hot path; kept on the host so its float64 results are the reference's own numpy results.

  refine_path              utilities/path_tools.py:178-240 (angle_delta=None, as make_initial_state calls it)
  initial_reward_state     envs/base/reward.py:261-288 (ContinuousRewardProvider.generate_initial_state)
  initial_pure_pursuit_state   envs/base/reward.py:355-371
  time_table               envs/base/env.py:382 (current_time accumulates `+= dt`)
"""
import numpy as np


def normalize_angle(z):
    return (np.array(z) + np.pi) % (2 * np.pi) - np.pi


def refine_path(data, delta):
    """Insert evenly spaced points wherever consecutive way points are more than `delta` apart; inserted points
    carry the heading of the segment's first point."""
    data = np.array(data, dtype=float) if isinstance(data, (list, tuple)) else data
    if data.shape[1] not in (2, 3):
        raise Exception("This function takes n x (x, y) or n x (x, y, angle) arrays")
    seg = np.linalg.norm(np.diff(data[:, :2], axis=0), axis=1)
    rows = []
    for i, d in enumerate(seg):
        if d > delta:
            npoints = int(d / delta) + 2
            cols = [np.linspace(data[i, j], data[i + 1, j], num=npoints) for j in range(2)]
            if data.shape[1] == 3:
                cols.append(np.ones((npoints,), dtype=float) * data[i, 2])
            rows.append(np.vstack(cols).T[:-1])
        else:
            rows.append(data[i])
    rows.append(data[-1])
    return np.vstack(rows)


def find_last_reached(pose, segment, spatial_precision, angular_precision):
    """Last way point index that is within reach of `pose` (utilities/path_tools.py:408-448), or None."""
    dist = np.hypot(segment[:, 0] - pose[0], segment[:, 1] - pose[1])
    angle = np.abs(normalize_angle(pose[2] - segment[:, 2]))
    par = np.cos(segment[:, 2]) * (pose[0] - segment[:, 0]) + np.sin(segment[:, 2]) * (pose[1] - segment[:, 1])
    idx = np.where((dist < spatial_precision) & (angle < angular_precision) & (par >= -spatial_precision / 9))[0]
    return idx[-1] if len(idx) else None


def initial_reward_state(path, reward_params):
    """-> (min_spat_dist_so_far, target_idx) for a fresh episode on `path`."""
    last = find_last_reached(path[0], path, reward_params.spatial_precision, reward_params.angular_precision)
    if last == len(path) - 1:
        raise ValueError("Goal pose too close to initial pose")
    target_idx = last + 1
    goal = path[target_idx]
    return float(np.hypot(goal[0] - path[0][0], goal[1] - path[0][1])), int(target_idx)


def initial_pure_pursuit_state(path):
    """-> (min_spat_dist_so_far, target_idx) of ContinuousRewardPurePursuitProvider.generate_initial_state
    (envs/base/reward.py:355-371): distance from the first to the LAST way point, look-ahead index 1."""
    return float(np.hypot(path[-1][0] - path[0][0], path[-1][1] - path[0][1])), 1


def time_table(dt, n):
    """t[k] = dt added k times in float64 (k = 0..n): Observation.time after k steps."""
    t = np.zeros(n + 1, dtype=np.float64)
    acc = 0.0
    for k in range(1, n + 1):
        acc = acc + dt
        t[k] = acc
    return t
