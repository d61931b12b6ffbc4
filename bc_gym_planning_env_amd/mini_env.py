"""RandomMiniEnv geometry: the reset-time side of the batched env.

The reference's RandomMiniEnv (envs/mini_env.py:408-494) draws a fresh one-corner world on every reset(): rejection
sampling of an obstacle wedge plus a start and an end pose (:269-361), two 1-px walls rendered with cv2.line
(envs/base/maps.py:28-44, utilities/map_drawing_utils.py:140-156), and a collision test of the two path ends with the
tricycle footprint (:324-343).  At 65 536 envs about a hundred episodes end per step, so a reset cannot be a host
round trip: geometries are sampled ahead of time into a POOL (this module, host numpy + one batched GPU
pose_collides per round), uploaded once, and the step / reset kernels move an env along its chain of pool entries
(bcp_set_geometry_pool, include/bcplan.h).

The sampler consumes its numpy RandomState exactly like the reference, draw for draw, so chain c of a pool is the
sequence of worlds `RandomMiniEnv(seed=seeds[c])` goes through on successive resets.
"""
import ctypes as C

import attr
import numpy as np
import torch

from . import _lib, robots
from .api import CostMap2D, EnvParams
from .batched_env import BatchedPlanEnv, DeviceGeometryPool

_TWO_PI = 2 * np.pi
_MAX_TRIES = 1000


@attr.s
class RandomMiniEnvParams(object):
    """Space the mini worlds are drawn from (same fields and defaults as envs/mini_env.py:30-47)."""
    inner_h = attr.ib(default=3, type=float)
    inner_w = attr.ib(default=3, type=float)
    mid_margin = attr.ib(default=0.25, type=float)
    out_margin = attr.ib(default=1, type=float)
    min_obstacle_angle = attr.ib(default=np.pi / 8., type=float)
    max_obstacle_angle = attr.ib(default=np.pi, type=float)
    lim_euc_dist = attr.ib(default=1000, type=float)
    lim_ang_dist = attr.ib(default=np.pi, type=float)
    angular_pose_noise_scale = attr.ib(default=np.pi / 2.0, type=float)
    env_params = attr.ib(factory=EnvParams)


def default_random_mini_env_params():
    """What RandomMiniEnv() uses when no params are given (envs/mini_env.py:424-430)."""
    return RandomMiniEnvParams(env_params=EnvParams(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2))


@attr.s
class MiniEnvParams(object):
    """One sampled world (envs/mini_env.py:79-92); poses are (x, y, theta) arrays, points (x, y) arrays."""
    h = attr.ib()
    w = attr.ib()
    start_pos = attr.ib()
    end_pos = attr.ib()
    obstacle_a = attr.ib()
    obstacle_o = attr.ib()
    obstacle_b = attr.ib()
    env_params = attr.ib(factory=EnvParams)


class SpaceSeemsEmptyError(Exception):
    """No admissible start/end pair on the circle after 1000 draws (the caller redraws the obstacle)."""


def _wrap(z):
    # normalize_angle, utilities/coordinate_transformations.py:28-36 (float64 scalars: the same three IEEE operations
    # as on the reference's 0-d arrays, without building arrays)
    return (np.asarray(z, dtype=np.float64)[()] + np.pi) % _TWO_PI - np.pi


def _pose(x, y, theta):
    # OrientedPoint normalises its heading on construction (envs/mini_env.py:50-55)
    return np.array([x, y, _wrap(theta)], dtype=float)


class _Wedge(object):
    """The angular sector behind the corner obstacle, seen from its apex (not_inside_obstacle, mini_env.py:199-208)."""

    def __init__(self, apex, first, width):
        self.ax, self.ay = np.float64(apex[0]), np.float64(apex[1])
        self.first, self.last = first, first + width

    def clear_of(self, x, y):
        phi = _wrap(np.arctan2(y - self.ay, x - self.ax))  # cart2pol, coordinate_transformations.py:124-135
        if self.first <= phi <= self.last:
            return False
        return not (self.first <= phi + _TWO_PI <= self.last)

    def clear_of_many(self, x, y):
        phi = _wrap(np.arctan2(y - self.ay, x - self.ax))
        inside = ((self.first <= phi) & (phi <= self.last)) | ((self.first <= phi + _TWO_PI) & (phi + _TWO_PI <= self.last))
        return ~inside


def _ends_on_circle(rng, p, wedge):
    # _sample_pose_circ (mini_env.py:146-180): two antipodal points of a circle, both heading start -> end.
    # The reference tries up to 1000 (phi, unused heading) pairs one by one; here they are drawn and tested a chunk at a
    # time -- RandomState.uniform(size=n) yields the same stream as n scalar calls -- and the generator is then wound
    # back to just after the pair that was accepted, so the stream continues exactly where the reference's would.
    radius = min((p.inner_w + p.inner_h) / 4. + p.mid_margin, p.lim_euc_dist)
    done = 0
    while done < _MAX_TRIES:
        chunk = min(8 if done == 0 else 128, _MAX_TRIES - done)
        before = rng.get_state()
        phi = rng.uniform(0, _TWO_PI, size=2 * chunk)[0::2]
        x, y = radius * np.cos(phi), radius * np.sin(phi)
        ok = wedge.clear_of_many(x, y) & wedge.clear_of_many(-x, -y)
        if ok.any():
            k = int(np.argmax(ok))
            rng.set_state(before)
            rng.uniform(0, _TWO_PI, size=2 * (k + 1))
            xk, yk = x[k], y[k]
            heading = np.arctan2(-yk - yk, -xk - xk)
            return _pose(xk, yk, heading), _pose(-xk, -yk, heading)
        done += chunk
    raise SpaceSeemsEmptyError()


def _pose_in_square(rng, p, accept):
    # _sample_pose (mini_env.py:120-143)
    half_w, half_h = p.inner_w / 2 + p.mid_margin, p.inner_h / 2 + p.mid_margin
    for _ in range(_MAX_TRIES):
        x = rng.uniform(-half_w, half_w)
        y = rng.uniform(-half_h, half_h)
        cand = _pose(x, y, rng.uniform(0, _TWO_PI))
        if accept(cand):
            return cand
    raise ValueError("Something went wrong, the sampling space looks empty.")


def _ends_in_square(rng, p, wedge):
    # _pick_pts_square_method (mini_env.py:183-236)
    start = _pose_in_square(rng, p, lambda q: wedge.clear_of(q[0], q[1]))

    def admissible(q):
        if not wedge.clear_of(q[0], q[1]):
            return False
        if not np.mod(start[2] - q[2], _TWO_PI) < p.lim_ang_dist:       # angle_diff, coordinate_transformations.py:115
            return False
        return np.linalg.norm(np.array([start[0] - q[0], start[1] - q[1]])) < p.lim_euc_dist

    end = _pose_in_square(rng, p, admissible)
    heading = np.arctan2(end[1] - start[1], end[0] - start[0])
    return _pose(start[0], start[1], heading), _pose(end[0], end[1], heading)


def draw_candidate(params, rng):
    """One unchecked world (_sample_mini_env_params_no_final_check, mini_env.py:269-325)."""
    p = params
    apex = np.array([rng.uniform(-p.inner_w / 2, p.inner_w / 2), rng.uniform(-p.inner_h / 2, p.inner_h / 2)], dtype=float)
    first = rng.uniform(0, _TWO_PI)
    width = rng.uniform(p.min_obstacle_angle, p.max_obstacle_angle)
    reach = 3 * (p.inner_h + p.inner_w + p.mid_margin + p.out_margin)   # far outside the map: walls run off its edge

    def ray_end(phi):
        return np.array([reach * np.cos(phi) + apex[0], reach * np.sin(phi) + apex[1]], dtype=float)

    side_h = p.inner_h + 2 * p.mid_margin + 2 * p.out_margin
    side_w = p.inner_w + 2 * p.mid_margin + 2 * p.out_margin
    wedge = _Wedge(apex, first, width)
    start, end = (_ends_on_circle if rng.rand() < 0.7 else _ends_in_square)(rng, p, wedge)
    half = p.angular_pose_noise_scale / 2.0
    start = _pose(start[0], start[1], start[2] + rng.uniform(-half, half))
    end = _pose(end[0], end[1], end[2] + rng.uniform(-half, half))
    return MiniEnvParams(side_h, side_w, start, end, ray_end(first), apex, ray_end(first + width), p.env_params)


# ---- map construction ---------------------------------------------------------------------------------------
def _to_pixel(xy, origin, resolution):
    # world_to_pixel, coordinate_transformations.py:185-205 (multiplies by the reciprocal, rounds half to even)
    return np.round((np.asarray(xy, dtype=np.float64) - origin) * (1.0 / resolution)).astype(int)


def _clip_segment(cols, rows, x1, y1, x2, y2):
    """cv::clipLine for integer end points: the part of the segment inside [0, cols) x [0, rows), or None."""
    right, bottom = cols - 1, rows - 1

    def code(x, y):
        return (x < 0) + (x > right) * 2 + (y < 0) * 4 + (y > bottom) * 8

    c1, c2 = code(x1, y1), code(x2, y2)
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * float(x2 - x1) / float(y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * float(x2 - x1) / float(y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * float(y2 - y1) / float(x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * float(y2 - y1) / float(x2 - x1))
                x2 = a
                c2 = 0
    return (x1, y1, x2, y2) if (c1 | c2) == 0 else None


def draw_line(data, p0, p1, value):
    """cv2.line(data, p0, p1, value, thickness=1): the segment is clipped to the image, then traced left to right by
    the 8-connected Bresenham iterator; the minor coordinate after i major steps is floor((2*d*i + D - 1) / (2*D))."""
    rows, cols = data.shape
    seg = _clip_segment(cols, rows, int(p0[0]), int(p0[1]), int(p1[0]), int(p1[1]))
    if seg is None:
        return
    x1, y1, x2, y2 = seg
    if x2 < x1:
        x1, y1, x2, y2 = x2, y2, x1, y1
    dx, dy = x2 - x1, abs(y2 - y1)
    sy = 1 if y2 >= y1 else -1
    major, minor = (dy, dx) if dy > dx else (dx, dy)
    i = np.arange(major + 1, dtype=np.int64)
    k = (2 * minor * i + major - 1) // (2 * major) if major else np.zeros(1, dtype=np.int64)
    if dy > dx:
        data[y1 + sy * i, x1 + k] = value
    else:
        data[y1 + sy * k, x1 + i] = value


def add_wall(costmap, p0, p1, width=0.05, cost=CostMap2D.LETHAL_OBSTACLE):
    """Wall.render -> _mark_wall_on_static_map (map_drawing_utils.py:140-156)."""
    thickness = max(1, int(width / costmap.get_resolution()))
    if thickness != 1:
        raise NotImplementedError("walls thicker than one pixel (resolution < %g m) are not supported" % (width / 2))
    org, res = costmap.get_origin(), costmap.get_resolution()
    draw_line(costmap.get_data(), _to_pixel(p0, org, res), _to_pixel(p1, org, res), cost)


def map_shape(h, w, resolution):
    """(rows, cols) of CostMap2D.create_empty(world_size=(h, w), resolution) (utilities/costmap_2d.py:58-69)"""
    size = _to_pixel(np.array([h, w], dtype=np.float64), np.zeros(2), resolution)
    return int(size[1]), int(size[0])


def prepare_map_and_path(mp, out=None):
    """-> (CostMap2D with the two walls, coarse path [2, 3]) for one sampled world (mini_env.py:362-388).
    `out`: optional zeroed uint8 array of the map's shape to draw into (a row of a batch buffer)."""
    res = mp.env_params.resolution
    data = np.zeros(map_shape(mp.h, mp.w, res), dtype=np.uint8) if out is None else out
    costmap = CostMap2D(data, res, np.array([-mp.h / 2., -mp.w / 2.]))
    add_wall(costmap, mp.obstacle_o, mp.obstacle_a)
    add_wall(costmap, mp.obstacle_o, mp.obstacle_b)
    return costmap, np.array([mp.start_pos, mp.end_pos])


# ---- batched acceptance test on the GPU -----------------------------------------------------------------------
class PoseCollider(object):
    """pose_collides of the two path ends for up to `capacity` candidate worlds per call (one launch)."""

    def __init__(self, env_params, capacity, device=0):
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("sampling mini-env geometries needs a GPU (libbcplan has no CPU path)")
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.capacity = int(capacity)
        # the reference tests with a TricycleRobot carrying the configured robot's footprint (mini_env.py:336-337)
        self._p = robots.make_bcp_params(env_params, env_params.robot_name, None)
        self._h = C.c_void_p()
        _lib.check(self._lib.bcp_create(C.byref(self._p), self.capacity, self.device.index or 0, 0, C.byref(self._h)))
        self.resolution = float(env_params.resolution)

    def __call__(self, costmaps, paths, batch=None):
        """costmaps: list of K <= capacity CostMap2D of one shape; paths: list of K [2,3] -> bool [K, 2].
        batch: optionally the K maps as one contiguous uint8 [K, rows, cols] array (saves the stacking)."""
        k, cap = len(costmaps), self.capacity
        rows, cols = costmaps[0].get_data().shape
        if batch is None:
            batch = np.stack([c.get_data() for c in costmaps])
        origins = np.zeros((cap, 2), dtype=np.float64)
        poses = np.zeros((2, cap, 3), dtype=np.float64)
        for j in range(k):
            origins[j] = costmaps[j].get_origin()
            poses[:, j] = paths[j]
        dev = self.device
        if getattr(self, "_maps", None) is None or tuple(self._maps.shape[1:]) != (rows, cols):
            self._maps = torch.zeros((cap, rows, cols), dtype=torch.uint8, device=dev)   # rows >= k keep stale maps:
        d = self._maps                                                                   # their poses are ignored
        d[:k].copy_(torch.from_numpy(np.ascontiguousarray(batch)))
        o, p = torch.from_numpy(origins).to(dev), torch.from_numpy(poses).to(dev)
        out = torch.empty(2 * cap, dtype=torch.uint8, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(self._lib.bcp_set_costmaps(self._h, d.data_ptr(), rows, cols, 0, None, None, o.data_ptr(), 1,
                                              self.resolution, stream))
        _lib.check(self._lib.bcp_pose_collides(self._h, p.data_ptr(), 2 * cap, out.data_ptr(), stream))
        hit = out.cpu().numpy().reshape(2, cap)[:, :k].T.astype(bool)   # (synchronises: d, o, p may go)
        return hit

    def close(self):
        if self._h:
            self._lib.bcp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Chain(object):
    """The worlds one RandomMiniEnv goes through: a private RandomState and the 1000-try budget of the current draw."""

    def __init__(self, seed):
        self.rng = np.random.RandomState(seed=0)   # RandomMiniEnv.__init__: RandomState(0), then seed(seed)
        if seed is not None:
            self.rng.seed(seed)
        self.tries = 0
        self.accepted = []   # MiniEnvParams of the accepted worlds ...
        self.built = []      # ... and their (costmap, coarse path)

    def propose(self, params):
        while self.tries < _MAX_TRIES:
            self.tries += 1
            try:
                return draw_candidate(params, self.rng)
            except SpaceSeemsEmptyError:
                continue
        raise ValueError("Something went wrong, the sampling space looks empty.")


def _too_close(path, env_params):
    # pose_distances of the two path ends against the goal tolerances (mini_env.py:347-351)
    cart = np.hypot(path[0][0] - path[1][0], path[0][1] - path[1][1])
    ang = np.abs(_wrap(path[0][2] - path[1][2]))
    return bool(cart < env_params.goal_spat_dist and ang < env_params.goal_ang_dist)


class MiniEnvPool(object):
    """n_chains x episodes pre-sampled worlds.  Entry c * episodes + k is the k-th world of chain c; `next_geom`
    walks a chain and wraps around at its end."""

    def __init__(self, params, seeds, episodes, worlds, built=None):
        self.params, self.seeds, self.episodes = params, list(seeds), int(episodes)
        self.worlds = worlds                                        # list of MiniEnvParams, chain-major
        if built is None:
            built = [prepare_map_and_path(w) for w in worlds]
        self.costmaps = [b[0] for b in built]
        self.paths = [b[1] for b in built]
        g = np.arange(len(worlds), dtype=np.int32)
        self.next_geom = (g // self.episodes) * self.episodes + (g % self.episodes + 1) % self.episodes
        self.next_geom = self.next_geom.astype(np.int32)

    def __len__(self):
        return len(self.worlds)

    def first_of_chain(self, c):
        return c * self.episodes


def sample_pool(params=None, seeds=(0,), episodes=1, device=0, collider=None):
    """Pre-sample `episodes` successive worlds for every seed (_sample_mini_env_params, mini_env.py:328-359, run for
    all chains at once: each round every unfinished chain proposes its next candidate and one batched GPU call tests
    the candidates' path ends)."""
    params = default_random_mini_env_params() if params is None else params
    chains = [_Chain(s) for s in seeds]
    own = collider is None
    if own:
        collider = PoseCollider(params.env_params, min(len(chains), 4096), device)
    try:
        todo = [c for c in chains if len(c.accepted) < episodes]
        while todo:
            batch = todo[:collider.capacity]
            cands = [c.propose(params) for c in batch]
            maps = np.zeros((len(cands),) + map_shape(cands[0].h, cands[0].w, params.env_params.resolution), dtype=np.uint8)
            built = [prepare_map_and_path(m, out=maps[j]) for j, m in enumerate(cands)]
            hits = collider([b[0] for b in built], [b[1] for b in built], batch=maps)
            for c, m, (costmap, path), hit in zip(batch, cands, built, hits):
                if not hit.any() and not _too_close(path, params.env_params):
                    c.accepted.append(m)
                    c.built.append((costmap.copy(), path))   # (own memory: `maps` is this round's scratch)
                    c.tries = 0
            todo = [c for c in chains if len(c.accepted) < episodes]
    finally:
        if own:
            collider.close()
    return MiniEnvPool(params, seeds, episodes, [m for c in chains for m in c.accepted],
                       built=[b for c in chains for b in c.built])


class DeviceMiniEnvPool(DeviceGeometryPool):
    """A MiniEnvPool that never leaves the GPU: worlds, costmaps, refined paths and initial reward states as device
    tensors (sample_pool_device(..., keep_on_device=True)).  `worlds` / `costmaps` / `paths` download on demand."""

    def __init__(self, params, seeds, episodes, worlds, maps, origin, paths, lens, init, mt_state=None, sampler_params=None):
        super(DeviceMiniEnvPool, self).__init__(maps, origin, params.env_params.resolution, paths, lens, init)
        self.params, self.seeds, self.episodes = params, list(seeds), int(episodes)
        self.world_params = worlds        # float64 [G, 14]
        self.mt_state = mt_state          # int32 [chains, 625]: where every MT19937 stream stands (after `generated` worlds)
        self.sampler_params = sampler_params   # the BcpMiniWorldParams the worlds were drawn with
        g = np.arange(len(self), dtype=np.int32)
        self.next_geom = ((g // self.episodes) * self.episodes + (g % self.episodes + 1) % self.episodes).astype(np.int32)

    @property
    def worlds(self):
        ep = self.params.env_params

        def fetch(k):
            v = self.world_params[k].cpu().numpy()
            return MiniEnvParams(v[12], v[13], v[0:3], v[3:6], v[6:8], v[8:10], v[10:12], ep)
        return self._Lazy(len(self), fetch)


def sample_pool_device(params=None, seeds=(0,), episodes=1, device=0, keep_on_device=False):
    """The same pool as sample_pool(), sampled entirely on the GPU (bcp_sample_mini_worlds, csrc/bcp_sample.h): one
    wavefront per seed runs numpy's MT19937 stream, the rejection sampler, the wall rasteriser and the acceptance test.
    Orders of magnitude faster than the host sampler; a coordinate can differ from the host's (numpy's) in its last bit
    because the transcendentals are the device's.
    keep_on_device=True returns a DeviceMiniEnvPool: nothing is downloaded, refined paths and initial reward states are
    computed on the GPU as well (bcp_mini_world_paths) -- the way to build pools of 10^5 .. 10^6 worlds."""
    params = default_random_mini_env_params() if params is None else params
    ep = params.env_params
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError("sampling mini-env geometries needs a GPU (libbcplan has no CPU path)")
    dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    h = C.c_void_p()
    bp = robots.make_bcp_params(ep, ep.robot_name, None)   # (the reference tests with the configured robot's footprint)
    _lib.check(lib.bcp_create(C.byref(bp), 1, dev.index or 0, 0, C.byref(h)))
    try:
        seeds = [int(s) for s in seeds]
        n = len(seeds)
        side_h = params.inner_h + 2 * params.mid_margin + 2 * params.out_margin
        side_w = params.inner_w + 2 * params.mid_margin + 2 * params.out_margin
        rows, cols = map_shape(side_h, side_w, ep.resolution)
        mp = _lib.BcpMiniWorldParams(
            params.inner_h, params.inner_w, params.mid_margin, params.out_margin, params.min_obstacle_angle,
            params.max_obstacle_angle, params.lim_euc_dist, params.lim_ang_dist, params.angular_pose_noise_scale,
            ep.resolution, ep.goal_spat_dist, ep.goal_ang_dist)
        seed_t = torch.tensor(seeds, dtype=torch.int64, device=dev)
        mt = torch.empty((n, 625), dtype=torch.int32, device=dev)          # MT19937 records (uint32 bit patterns)
        worlds = torch.empty((n * episodes, 14), dtype=torch.float64, device=dev)
        maps = torch.empty((n * episodes, rows, cols), dtype=torch.uint8, device=dev)
        status = torch.zeros(n, dtype=torch.int32, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.bcp_mini_world_seed(h, seed_t.data_ptr(), n, mt.data_ptr(), stream))
        _lib.check(lib.bcp_sample_mini_worlds(h, C.byref(mp), mt.data_ptr(), n, int(episodes), rows, cols, worlds.data_ptr(),
                                              maps.data_ptr(), status.data_ptr(), stream))
        if int(status.sum()):
            raise ValueError("Something went wrong, the sampling space looks empty.")
        if keep_on_device:
            if not ep.refine_path:
                raise NotImplementedError("device pools always carry refined paths")
            diag = float(np.hypot(2 * (params.inner_w / 2 + params.mid_margin), 2 * (params.inner_h / 2 + params.mid_margin)))
            max_len = int(diag / ep.path_delta) + 4            # the longest start -> end segment inside the square
            g_n = n * episodes
            paths = torch.zeros((g_n, max_len, 3), dtype=torch.float64, device=dev)
            lens = torch.zeros(g_n, dtype=torch.int32, device=dev)
            init = torch.zeros((g_n, 2), dtype=torch.float64, device=dev)
            pstat = torch.zeros(g_n, dtype=torch.int32, device=dev)
            _lib.check(lib.bcp_mini_world_paths(h, worlds.data_ptr(), g_n, float(ep.path_delta), max_len, paths.data_ptr(),
                                                lens.data_ptr(), init.data_ptr(), pstat.data_ptr(), stream))
            worst = int(pstat.max())
            if worst == 2:
                raise ValueError("Goal pose too close to initial pose")
            assert worst == 0, "refined path longer than expected"
            return DeviceMiniEnvPool(params, seeds, episodes, worlds, maps, np.array([-side_h / 2., -side_w / 2.]), paths,
                                     lens, init, mt_state=mt, sampler_params=mp)
        w_host, m_host = worlds.cpu().numpy(), maps.cpu().numpy()
    finally:
        lib.bcp_destroy(h)
    origin = np.array([-side_h / 2., -side_w / 2.])
    world_list, built = [], []
    for g in range(n * episodes):
        v = w_host[g]
        world_list.append(MiniEnvParams(v[12], v[13], v[0:3].copy(), v[3:6].copy(), v[6:8].copy(), v[8:10].copy(),
                                        v[10:12].copy(), ep))
        built.append((CostMap2D(m_host[g], ep.resolution, origin), np.array([v[0:3], v[3:6]])))
    return MiniEnvPool(params, seeds, episodes, world_list, built=built)


class BatchedRandomMiniEnv(BatchedPlanEnv):
    """N RandomMiniEnv instances (envs/mini_env.py:408-494) on one GPU: a BatchedPlanEnv in geometry-pool mode.

    Env i follows chain i % n_chains of the pool, starting (i // n_chains) % episodes entries into it, so replicas of a
    chain are out of phase (and the undecided poses of a world spread over many wavefronts).  With one chain per env and
    seeds[i] = s_i, env i sees exactly the worlds `RandomMiniEnv(seed=s_i)` sees for its first `episodes` resets (then
    the chain wraps around).  As in the reference, construction leaves the env on world 0 of its chain and the first
    reset() moves it to world 1 -- BatchedPlanEnv's constructor already ends with that reset().

    :param pool MiniEnvPool: pre-sampled worlds, or None to sample `n_chains` x `episodes` here
    :param sampler: "device" (sample_pool_device: on the GPU, ~10^5 worlds/s, coordinates within 1e-12 of the
        reference's), "device_resident" (the same, and the pool never leaves the GPU: for 10^5 .. 10^6 worlds) or
        "host" (sample_pool: numpy, bit-identical to the reference, ~3 x 10^3 worlds/s)
    :param draw_new_turn_on_reset bool: False keeps every env on its first world (RandomMiniEnv's flag of that name)
    :param endless bool: one stream per env (seeds[i], by default env_id_base + i), pool kept on the GPU, and the `episodes` entries of an env
        are a ring over its stream that refresh() tops up behind it: env i sees the worlds of RandomMiniEnv(seed=seeds[i])
        for as long as it runs, never an old one again.  Call refresh() every few steps (see there).
    Remaining keyword arguments go to BatchedPlanEnv (auto_reset, seed, noise_parameters, env_id_base, ...).
    """

    def __init__(self, n_envs, params=None, pool=None, seeds=None, n_chains=None, episodes=4, device=0,
                 draw_new_turn_on_reset=True, sampler="device", endless=False, **kw):
        params = default_random_mini_env_params() if params is None else params
        if endless:
            if pool is not None or n_chains not in (None, int(n_envs)) or not draw_new_turn_on_reset or episodes < 2:
                raise ValueError("endless=True samples its own pool: one stream per env, episodes >= 2")
            # default streams: seed = GLOBAL env index, so that the ranks of a sharded batch draw different worlds
            base = int(kw.get("env_id_base", 0))
            sampler, seeds = "device_resident", (range(base, base + int(n_envs)) if seeds is None else seeds)
            if len(list(seeds)) != int(n_envs):
                raise ValueError("endless=True needs one seed per env")
        if pool is None:
            if seeds is None:
                seeds = range(int(n_chains) if n_chains else min(int(n_envs), 1024))
            if sampler == "device_resident":
                pool = sample_pool_device(params, list(seeds), episodes, device, keep_on_device=True)
            else:
                pool = {"device": sample_pool_device, "host": sample_pool}[sampler](params, list(seeds), episodes, device)
        chains, per = len(pool.seeds), pool.episodes
        i = np.arange(int(n_envs))
        geom = (i % chains) * per + (i // chains) % per
        self.pool = pool
        on_device = isinstance(pool, DeviceGeometryPool)
        next_geom = pool.next_geom if draw_new_turn_on_reset else None
        self.endless = bool(endless)
        if endless:   # the newest world's entry is the ring's guard (bcp_refresh_mini_worlds)
            next_geom = next_geom.copy()
            newest = np.arange(chains) * per + per - 1
            next_geom[newest] = newest
        super(BatchedRandomMiniEnv, self).__init__(
            pool if on_device else pool.costmaps, None if on_device else pool.paths, params.env_params, n_envs=n_envs,
            device=device, geom_of_env=geom, next_geom=next_geom, **kw)
        if endless:
            dev = self.device
            self._generated = torch.full((chains,), per, dtype=torch.int64, device=dev)
            self._ring_status = torch.zeros(chains, dtype=torch.int32, device=dev)
            self._ring_path_status = torch.zeros(chains * per, dtype=torch.int32, device=dev)
            self._ring_info = torch.zeros(4, dtype=torch.int32, device=dev)
            self._ring_info_done = torch.zeros(4, dtype=torch.int32, device=dev)
            self._ring_side, self._ring_pending = None, False

    def refresh(self, check=False, overlap=False):
        """endless=True: re-sample, on the GPU and in stream order, the pool entries of the worlds every env has left
        (bcp_plan_mini_worlds, bcp_refresh_mini_worlds, bcp_release_mini_worlds) -- RandomMiniEnv.reset's `_sample_mini_env_params`
        (envs/mini_env.py:441-459), done ahead of time.  Call it between steps.  An env can go through at most
        `episodes - 1` resets between two refreshes before it has to repeat its newest world; info[1] counts the envs
        that were found waiting like that.

        overlap=False: everything on the current stream, in order (a refresh takes milliseconds: sampling a world is a
        serial job for one wavefront, however few worlds there are).
        overlap=True: the sampling runs on a side stream WHILE the steps go on: this call completes the previous refresh
        -- its worlds come into reach from the next step on -- and starts the next one.  Nothing blocks on the host; on
        the GPU the steps wait for a previous refresh that has not finished yet, so call it every few hundred steps
        (a refresh takes 2 .. 10 ms) and size `episodes` so that no env gets through `episodes - 1` worlds during two
        such periods.  Stepping on a high-priority stream (torch.cuda.Stream(priority=-1)) keeps the refresh kernels
        from delaying the steps (5 % in tools/bench_endless.py).

        Snapshots (get_state / set_state) hold pool ENTRIES, not worlds: one taken before a refresh can be restored only
        as long as the entries it points to have not been re-sampled.

        :param check bool: wait for the refresh to finish and raise what the reference's sampler would raise
        :return: device int32 [4] = (entries re-sampled, envs that were waiting on their newest world, 0, 0) of the
            refresh that was completed by this call; with check=True the first two as python ints"""
        if not self.endless:
            raise RuntimeError("refresh() needs BatchedRandomMiniEnv(endless=True)")
        main = torch.cuda.current_stream(self.device)
        if overlap:
            if self._ring_side is None:
                self._ring_side = self._make_side_stream()
                self._ring_done, self._ring_go = torch.cuda.Event(), torch.cuda.Event()
            done = self.finish_refresh(check)
            self._plan_refresh(main)
            self._ring_go.record(main)
            self._ring_side.wait_event(self._ring_go)
            self._launch_refresh(self._ring_side)
            self._ring_done.record(self._ring_side)
            return done
        self.finish_refresh()
        self._plan_refresh(main)
        self._launch_refresh(main)
        return self.finish_refresh(check)

    def _make_side_stream(self):
        """The stream the overlapped refresh runs on: an ordinary stream, or -- `side_cu_percent` < 100 -- one restricted
        to that share of the compute units (bcp_side_stream).  Measured on MI355X (tools/bench_endless.py, 65 536 envs x 8
        entries, refresh every 128 steps): 0.064 ms/step on an ordinary side stream, 0.078 with 50 % or 25 % of the
        compute units -- a refresh is throughput-bound (distance fields of ~7000 re-sampled maps), so confining it only
        makes it last longer; the default stays 100."""
        share = int(getattr(self, "side_cu_percent", 100))
        if 0 < share < 100:
            ptr = C.c_void_p()
            if self._lib.bcp_side_stream(self._h, share, C.byref(ptr)) == 0 and ptr.value:
                return torch.cuda.ExternalStream(ptr.value, device=self.device)
        return torch.cuda.Stream(self.device)

    def _plan_refresh(self, stream):
        _lib.check(self._lib.bcp_plan_mini_worlds(self._h, self.pool.episodes, self._generated.data_ptr(),
                                                  self._ring_info.data_ptr(), C.c_void_p(stream.cuda_stream)))
        self._ring_pending = True

    def _launch_refresh(self, stream):
        pool, ep = self.pool, self.params
        _lib.check(self._lib.bcp_refresh_mini_worlds(
            self._h, C.byref(pool.sampler_params), pool.mt_state.data_ptr(), pool.world_params.data_ptr(),
            pool.maps.data_ptr(), pool.path_points.data_ptr(), pool.lens.data_ptr(), pool.init.data_ptr(),
            float(ep.path_delta), self._ring_status.data_ptr(), self._ring_path_status.data_ptr(),
            C.c_void_p(stream.cuda_stream)))

    def finish_refresh(self, check=False):
        """Complete the refresh in flight, if any: wait (on the current stream) for its side stream and open the ring to
        the new worlds.  refresh() does this itself; call it before reading pool tensors or taking snapshots."""
        if not self._ring_pending:
            return None
        main = torch.cuda.current_stream(self.device)
        if self._ring_side is not None:
            main.wait_event(self._ring_done)
        _lib.check(self._lib.bcp_release_mini_worlds(self._h, C.c_void_p(main.cuda_stream)))
        self._ring_info_done.copy_(self._ring_info)   # (the plan's tally: the next plan overwrites _ring_info)
        self._ring_pending = False
        if not check:
            return self._ring_info_done
        if int(self._ring_status.sum()):
            raise ValueError("Something went wrong, the sampling space looks empty.")
        worst = int(self._ring_path_status.max())
        if worst == 2:
            raise ValueError("Goal pose too close to initial pose")
        assert worst == 0, "refined path longer than expected"
        info = self._ring_info_done.cpu().numpy()
        return int(info[0]), int(info[1])
