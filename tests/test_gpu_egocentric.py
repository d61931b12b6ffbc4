"""Egocentric observation on the GPU against the reference's fixtures (g10, recorded
from the genuine EgocentricCostmap wrapper), the reference's known answers, and the oracle on random inputs."""
import os

import numpy as np
import pytest

from util import GOLDEN

pytestmark = pytest.mark.gpu


def _marks(img):
    r, c = np.where(img == 254)
    return list(zip(c.tolist(), r.tolist()))


def test_kat_extract_egocentric_costmap_gpu(torch_cuda):
    """utilities/test_costmap_utils.py:38-195 through the HIP kernel"""
    from bc_gym_planning_env_amd.ops import NativeOps
    ops = NativeOps()
    data = np.zeros((100, 100), dtype=np.uint8)
    data[10, 20] = 254
    ops.set_costmap(data, np.array([0.0, 0.0]), 0.05)

    def ego(pose, org=None, size=None):
        return ops.extract_egocentric_costmap(np.array([pose]), org, size)[0].cpu().numpy()

    assert _marks(ego((0., 0., 0.))) == [(20, 10)]
    assert _marks(ego((0.2, 0.2, 0.0))) == [(20, 10)]
    assert ego((0.0, 0.0, np.pi / 6 - 0.05))[0, 22] == 254
    assert _marks(ego((2.5, 2.5, -np.pi / 2.))) == [(90, 20)]
    assert _marks(ego((2.5, 2.5, -np.pi / 2), [-2.5, -2.5], (5., 5.))) == [(90, 20)]
    img = ego((2.5, 2.5, -np.pi / 2), [-2.5, -2.5], (4.6, 4.9))
    assert img.shape == (98, 92) and _marks(img) == [(90, 20)]
    assert _marks(ego((2.5, 2.5, 0.0), [-5., -4.], (5, 5))) == [(70, 40)]
    img = ego((2.5, 2.5, 0.0), [-5., -4.], (4, 4))
    assert img.shape == (80, 80) and _marks(img) == [(70, 40)]
    assert _marks(ego((1.5, 1.5, -np.pi / 4), [-2., -2.], (4, 4))) == [(47, 19)]
    img = ego((1., 1.5, -np.pi / 4), [-3., -3.], (7, 6))
    assert img.shape == (120, 140) and _marks(img) == [(74, 46)]
    ops.set_costmap(data, np.array([1.0, 2.0]), 0.05)
    assert _marks(ego((3.5, 3.5, -np.pi / 4), [-2., -2.], (4, 4))) == [(33, 5)]
    block = np.zeros((100, 100), dtype=np.uint8)
    block[30:32, 30:32] = 254
    ops.set_costmap(block, np.array([0.0, 0.0]), 0.05)
    r, c = np.where(ego((1.5, 1.5, -np.pi / 4), [-2.0, -2.0], (4., 4)) != 0)
    assert r.tolist() == [40, 41, 41, 41, 42] and c.tolist() == [40, 39, 40, 41, 40]


@pytest.mark.parametrize("name", ["g10_ego_mini_00.npz", "g10_ego_mini_05.npz", "g10_ego_aisle.npz"])
def test_g10_observation_from_reference_states(torch_cuda, name):
    """every recorded state of the fixture becomes one env of a batch; one observation() must reproduce the
    reference wrapper's images bit for bit and its goal_n_state vectors"""
    torch = torch_cuda
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    from bc_gym_planning_env_amd.egocentric import BatchedEgocentricCostmap
    g = np.load(os.path.join(GOLDEN, name))
    n = len(g["states"])
    res = float(g["resolution"])
    env = BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], EnvParams(resolution=res, refine_path=False),
                         n_envs=n)
    env.state.robot.copy_(torch.from_numpy(np.ascontiguousarray(g["states"].T)).cuda())
    env.state.target_idx.copy_(torch.from_numpy(g["target_idx"]).cuda())
    wrap = BatchedEgocentricCostmap(env)
    assert wrap.image_shape == tuple(int(v) for v in g["image_shape"])
    obs = wrap.observation()
    cols = wrap.image_shape[1]
    want = np.unpackbits(g["images"], axis=2)[:, :, :cols].astype(bool)
    img = obs['env'].cpu().numpy()
    assert img.shape == (n,) + wrap.image_shape + (1,)
    assert ((img[..., 0] == 254) == want).all() and set(np.unique(img)) <= {0, 254}
    vec = obs['goal_n_state'].cpu().numpy()
    assert vec.shape == (n, 9, 1) and vec.dtype == np.float32
    np.testing.assert_allclose(vec[:, :, 0], g["goal_n_state"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("shared", [True, False, "large", "medium"],
                         ids=["shared-map", "private-maps", "large-shared-map", "medium-shared-map"])
def test_random_poses_vs_oracle(torch_cuda, oracle, shared):
    """random poses (inside and far outside the map), several windows and border values, a costmap with arbitrary
    byte values; private maps of different shapes go through the global-memory gather"""
    torch = torch_cuda
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    rng = np.random.RandomState(8)
    n = 96
    res = 0.05
    large = shared in ("large", "medium")   # 168 KB: too big for LDS, each workgroup stages the part of the map its
    medium = shared == "medium"             # window sees; 96 KB: staged whole in one big (> 64 KB) LDS allocation
    shared = bool(shared)
    shapes = [(330, 290)] if medium else [(420, 400)] if large else (
        [(90, 70)] if shared else [(90, 70), (64, 101), (300, 260)])     # (the private 300 x 260 map needs > 64 KB of LDS)
    maps = [rng.randint(0, 256, s).astype(np.uint8) for s in shapes]
    orgs = [rng.uniform(-2, 0, 2) for _ in shapes]
    path = np.array([[0., 0., 0.], [1., 0., 0.], [2., 0., 0.]])
    params = EnvParams(resolution=res, refine_path=False)
    if shared:
        env = BatchedPlanEnv(CostMap2D(maps[0], res, orgs[0]), path, params, n_envs=n)
    else:
        env = BatchedPlanEnv([CostMap2D(maps[i % 3], res, orgs[i % 3]) for i in range(n)], [path] * n, params, n_envs=n)
    hi = 23 if large else 6
    poses = np.stack([rng.uniform(-4, hi, n), rng.uniform(-4, hi, n), rng.uniform(-7, 7, n)], axis=1)
    poses[0] = (0., 0., 0.)
    poses[1] = (1.0, 1.0, np.pi)
    pt = torch.from_numpy(poses).cuda()
    import ctypes as C
    from bc_gym_planning_env_amd import _lib
    f64p = C.POINTER(C.c_double)
    for org, size, border in (((-0.5, -2.0), (3.5, 4.0), 0), ((-1.0, -1.0), (2.0, 2.0), 255), (None, None, 7),
                              ((-3.0, -0.7), (6.05, 1.45), 100),
                              ((-0.1, -0.15), (0.3, 0.25), 9)):   # (6 x 5 px: the 4-pixels-per-lane kernels)
        o = None if org is None else np.array(org, dtype=np.float64)
        s = None if size is None else np.array(size, dtype=np.float64)
        shape = (C.c_int32 * 2)()
        _lib.check(env._lib.bcp_egocentric_shape(env._h, s.ctypes.data_as(f64p) if s is not None else None, shape))
        if size is None and not shared:
            assert tuple(shape) == (300, 260)        # allocation shape of the padded private maps
        out = torch.full((n, shape[0], shape[1]), 99, dtype=torch.uint8, device="cuda")
        guard = torch.full((16,), 123, dtype=torch.uint8, device="cuda")  # (allocated right after `out`, not adjacent)
        _lib.check(env._lib.bcp_egocentric_costmaps(env._h, pt.data_ptr(), n, o.ctypes.data_as(f64p) if o is not None else None,
                                                    s.ctypes.data_as(f64p) if s is not None else None, border,
                                                    out.data_ptr(), None))
        got = out.cpu().numpy()
        assert (guard.cpu().numpy() == 123).all()
        for i in range(n):
            k = 0 if shared else i % 3
            if size is None and not shared:
                continue   # whole-map output of padded private maps has no reference counterpart of that shape
            ref = oracle.extract_egocentric(maps[k], orgs[k], res, poses[i], o, s, border)
            assert ref.shape == got[i].shape
            assert (ref == got[i]).all(), (org, size, i, int((ref != got[i]).sum()))


def test_wrapper_steps_with_pool_env(torch_cuda, oracle):
    """EgocentricCostmap(RandomMiniEnv) batched: after every step the observation of each env matches the oracle on
    that env's current world and state (including envs that were just reset onto their next world)"""
    torch = torch_cuda
    from bc_gym_planning_env_amd import EnvParams, mini_env
    from bc_gym_planning_env_amd.egocentric import BatchedEgocentricCostmap
    params = mini_env.RandomMiniEnvParams(
        env_params=EnvParams(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2, iteration_timeout=12))
    pool = mini_env.sample_pool(params, [1, 2, 3], 3)
    n = 48
    env = mini_env.BatchedRandomMiniEnv(n, params, pool=pool, auto_reset=True, seed=2)
    wrap = BatchedEgocentricCostmap(env)
    rng = np.random.RandomState(0)
    res = params.env_params.resolution
    rows, cols = wrap.image_shape
    world = np.array([(-0.5 + res * cols) - -0.5, (-2.0 + res * rows) - -2.0])
    for t in range(30):
        obs, _r, _d, _ = wrap.step(env.action_space.sample_batch(n, rng) * np.array([3.0, 1.0], dtype=np.float32))
        img = obs['env'].cpu().numpy()[..., 0]
        vec = obs['goal_n_state'].cpu().numpy()[..., 0]
        st = env.state.robot.cpu().numpy()
        geom = env.geom_of_env.cpu().numpy()
        tidx = env.state.target_idx.cpu().numpy()
        for i in range(n):
            cm = pool.costmaps[geom[i]]
            ref = oracle.extract_egocentric(cm.get_data(), cm.get_origin(), res, st[:3, i], (-0.5, -2.0), (3.5, 4.0))
            assert (ref == img[i]).all(), (t, i)
            rs = np.array([st[0, i], st[1, i], st[2, i], st[3, i], st[4, i], st[6, i]])
            want = oracle.goal_n_state(st[:3, i], env._paths[geom[i]][tidx[i]:], world, rs)
            np.testing.assert_allclose(vec[i], want, rtol=0, atol=1e-6)
    assert len(np.unique(geom)) > 3


def test_g12_colored_ego_observation(torch_cuda):
    """ColoredEgoCostmapRandomAisleTurnEnv's observation from the reference's recorded states: a 350 x 512 costmap (too
    large for LDS: sampled from global memory), 133 x 133 window, float64 goal vector"""
    torch = torch_cuda
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    from bc_gym_planning_env_amd.egocentric import BatchedColoredEgoCostmap
    g = np.load(os.path.join(GOLDEN, "g12_colored_ego.npz"))
    n = len(g["states"])
    res = float(g["resolution"])
    env = BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], EnvParams(resolution=res, refine_path=False),
                         n_envs=n)
    env.state.robot.copy_(torch.from_numpy(np.ascontiguousarray(g["states"].T)).cuda())
    wrap = BatchedColoredEgoCostmap(env)
    assert wrap.image_shape == tuple(int(v) for v in g["image_shape"]) == (133, 133)
    obs = wrap.observation()
    want = np.unpackbits(g["images"], axis=2)[:, :, :133].astype(bool)
    img = obs['environment'].cpu().numpy()
    assert ((img[..., 0] == 254) == want).all() and set(np.unique(img)) <= {0, 254}
    vec = obs['goal'].cpu().numpy()
    assert vec.shape == (n, 5, 1) and vec.dtype == np.float64
    np.testing.assert_allclose(vec[:, :, 0], g["goal"], rtol=0, atol=1e-9)


def _route(env):
    import ctypes as C
    from bc_gym_planning_env_amd import _lib
    info = (C.c_int32 * 4)()
    _lib.check(env._lib.bcp_egocentric_route(env._h, info))
    return _lib.EGO_KERNELS[int(info[0])], int(info[1]), int(info[2]), int(info[3])


@pytest.mark.parametrize("kind,cells", [("shared", 60), ("private", 60), ("shared", 700), ("private", 1500), ("pooled", 2000)],
                         ids=["shared", "private", "shared-700-cells", "private-1500-cells", "pooled-2000-cells"])
def test_sparse_maps_fill_and_patch_vs_oracle(torch_cuda, oracle, kind, cells):
    """Sparse costmaps with border value 0 take the fill-and-patch kernel (ego_sparse_kernel): a zero fill plus one patch
    per non-zero source cell.  Random poses far in and out of the map, several windows, maps with thin walls, isolated
    cells, cells on the map's edges and arbitrary non-zero values -- against the oracle, and against the sampling kernels
    (BCP_TUNE_EGO_SPARSE = 0) bit for bit.  Lists of 60 to ~2200 cells (round 3 stopped at 512), shared, private and pooled
    maps; a dense map routes the call to the sampling kernels (same images)."""
    torch = torch_cuda
    import ctypes as C
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams, _lib
    rng = np.random.RandomState(21 + cells)
    n = 192
    res = 0.05

    def sparse_map(shape, cells):
        m = np.zeros(shape, dtype=np.uint8)
        m[rng.randint(0, shape[0], cells), rng.randint(0, shape[1], cells)] = rng.randint(1, 256, cells)
        m[shape[0] // 3, 5:shape[1] - 5] = 254            # a wall one cell thick
        m[7:shape[0] - 9, shape[1] // 2] = 254
        m[0, 0] = m[-1, -1] = m[0, -1] = m[-1, 0] = 200    # the corners
        m[0, 3:9] = 17                                     # on the edges
        m[5:11, -1] = 18
        if cells > 512:                                    # a filled block: hundreds of cells inside one window
            m[20:20 + 18, 30:30 + 22] = 77
        return m

    big = cells > 512
    shapes = [(90, 70)] if kind == "shared" else [(90, 70), (64, 101), (183, 183)]
    if big and kind == "shared":
        shapes = [(200, 240)]
    maps = [sparse_map(s, cells if i == len(shapes) - 1 else min(cells, 300)) for i, s in enumerate(shapes)]
    orgs = [rng.uniform(-2, 0, 2) for _ in shapes]
    path = np.array([[0., 0., 0.], [1., 0., 0.], [2., 0., 0.]])
    params = EnvParams(resolution=res, refine_path=False)
    if kind == "shared":
        env = BatchedPlanEnv(CostMap2D(maps[0], res, orgs[0]), path, params, n_envs=n)
        map_of = lambda i: 0
    elif kind == "private":
        env = BatchedPlanEnv([CostMap2D(maps[i % 3], res, orgs[i % 3]) for i in range(n)], [path] * n, params, n_envs=n)
        map_of = lambda i: i % 3
    else:   # a geometry pool: 3 entries, envs spread over them
        geom = (np.arange(n) * 7 % 3).astype(np.int32)
        env = BatchedPlanEnv([CostMap2D(m, res, o) for m, o in zip(maps, orgs)], [path] * 3, params, n_envs=n, geom_of_env=geom)
        geom = env.geom_of_env.cpu().numpy()      # (the constructor's reset() moves nobody: no successor table)
        map_of = lambda i: int(geom[i])
    most = max(int((m != 0).sum()) for m in maps)
    assert (most > 512) == big
    poses = np.stack([rng.uniform(-3, 7, n), rng.uniform(-3, 7, n), rng.uniform(-7, 7, n)], axis=1)
    poses[0] = (0., 0., 0.)
    poses[1] = (1.0, 1.0, np.pi)
    poses[2] = (2.0, 1.5, np.pi / 2)
    poses[3] = (2.0, 1.5, np.pi / 4)
    poses[4] = (1e4, -1e4, 0.3)      # nowhere near the map
    pt = torch.from_numpy(poses).cuda()
    f64p = C.POINTER(C.c_double)

    def draw(e, o, s, border):
        shape = (C.c_int32 * 2)()
        _lib.check(e._lib.bcp_egocentric_shape(e._h, s.ctypes.data_as(f64p) if s is not None else None, shape))
        out = torch.full((n, shape[0], shape[1]), 99, dtype=torch.uint8, device="cuda")
        _lib.check(e._lib.bcp_egocentric_costmaps(e._h, pt.data_ptr(), n, o.ctypes.data_as(f64p) if o is not None else None,
                                                  s.ctypes.data_as(f64p) if s is not None else None, border,
                                                  out.data_ptr(), None))
        return out.cpu().numpy()

    nonzero = 0
    for org, size in (((-0.5, -2.0), (3.5, 4.0)), ((-1.0, -1.0), (2.0, 2.0)), ((-3.0, -0.7), (6.05, 1.45)), ((-0.1, -0.15), (0.3, 0.25))):
        o, s = np.array(org, dtype=np.float64), np.array(size, dtype=np.float64)
        env.set_tuning(ego_sparse=4096)      # an explicit limit: small windows would otherwise send 2000 cells to the samplers
        got = draw(env, o, s, 0)
        kernel, counted, stride, limit = _route(env)
        assert kernel == "ego_sparse_kernel" and counted == most and stride >= max(512, most) and limit == 4096
        env.set_tuning(ego_sparse=0)
        sampled = draw(env, o, s, 0)
        assert _route(env)[0] != "ego_sparse_kernel"
        assert (got == sampled).all()
        for i in range(n):
            k = map_of(i)
            ref = oracle.extract_egocentric(maps[k], orgs[k], res, poses[i], o, s, 0)
            assert ref.shape == got[i].shape and (ref == got[i]).all(), (org, size, i, int((ref != got[i]).sum()))
        nonzero += int((got != 0).sum())
    assert nonzero > 2000
    # the cost model (BCP_TUNE_EGO_SPARSE = 1) on the reference's own window: these lists qualify
    env.set_tuning(ego_sparse=1)
    o, s = np.array((-0.5, -2.0)), np.array((3.5, 4.0))
    again = draw(env, o, s, 0)
    assert _route(env)[0] == ("ego_sparse_kernel" if most <= _route(env)[3] else _route(env)[0])
    env.set_tuning(ego_sparse=0)
    assert (again == draw(env, o, s, 0)).all()
    # a non-zero border value is not this kernel's business
    env.set_tuning(ego_sparse=1)
    draw(env, o, s, 9)
    assert _route(env)[0] != "ego_sparse_kernel"
    # a dense map: more non-zero cells than the cost model's limit -> the sampling kernels, same answers
    dense = rng.randint(0, 256, shapes[0]).astype(np.uint8)
    if kind == "shared":
        env2 = BatchedPlanEnv(CostMap2D(dense, res, orgs[0]), path, params, n_envs=n)
        got = draw(env2, o, s, 0)
        kernel, counted, _stride, limit = _route(env2)
        assert kernel == "ego_costmap_kernel<staged>" and counted > limit
        for i in range(0, n, 7):
            assert (oracle.extract_egocentric(dense, orgs[0], res, poses[i], o, s, 0) == got[i]).all()


def test_entry_with_more_cells_than_its_list_is_drawn_pixel_by_pixel(torch_cuda, oracle):
    """a map entry whose non-zero cells do not fit its list (a pool entry re-sampled with more cells than the
    lists were sized for) is drawn inside the same launch, pixel by pixel (ego_image_slow), the other entries from their
    lists.  BCP_TUNE_EGO_LIST_STRIDE pins the lists at 128 cells; one of three private maps has ~700."""
    torch = torch_cuda
    import ctypes as C
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams, _lib
    rng = np.random.RandomState(11)
    res, n = 0.05, 96
    shapes = [(90, 70), (64, 101), (120, 120)]
    maps = []
    for k, shp in enumerate(shapes):
        m = np.zeros(shp, dtype=np.uint8)
        cells = 60 if k < 2 else 700
        m[rng.randint(0, shp[0], cells), rng.randint(0, shp[1], cells)] = rng.randint(1, 256, cells)
        maps.append(m)
    orgs = [rng.uniform(-2, 0, 2) for _ in shapes]
    path = np.array([[0., 0., 0.], [1., 0., 0.], [2., 0., 0.]])
    env = BatchedPlanEnv([CostMap2D(maps[i % 3], res, orgs[i % 3]) for i in range(n)], [path] * n,
                         EnvParams(resolution=res, refine_path=False), n_envs=n)
    env.set_tuning(ego_sparse=4096, ego_list_stride=128)
    poses = np.stack([rng.uniform(-1, 5, n), rng.uniform(-1, 5, n), rng.uniform(-7, 7, n)], axis=1)
    pt = torch.from_numpy(poses).cuda()
    f64p = C.POINTER(C.c_double)
    o, s = np.array((-0.5, -2.0)), np.array((3.5, 4.0))
    shape = (C.c_int32 * 2)()
    _lib.check(env._lib.bcp_egocentric_shape(env._h, s.ctypes.data_as(f64p), shape))
    out = torch.full((n, shape[0], shape[1]), 99, dtype=torch.uint8, device="cuda")
    _lib.check(env._lib.bcp_egocentric_costmaps(env._h, pt.data_ptr(), n, o.ctypes.data_as(f64p), s.ctypes.data_as(f64p), 0,
                                                out.data_ptr(), None))
    kernel, counted, stride, _limit = _route(env)
    assert kernel == "ego_sparse_kernel" and stride == 128 and counted > 500
    got = out.cpu().numpy()
    lit = 0
    for i in range(n):
        ref = oracle.extract_egocentric(maps[i % 3], orgs[i % 3], res, poses[i], o, s, 0)
        assert (ref == got[i]).all(), i
        lit += int((ref != 0).sum()) if i % 3 == 2 else 0
    assert lit > 300


def test_sparse_window_overflow_streams_the_list(torch_cuda, oracle):
    """More cells inside one window than a wave holds back in LDS (kEgoHeld = 256): the wave streams the list behind the
    fill instead.  A filled 40 x 40 block under the robot."""
    torch = torch_cuda
    import ctypes as C
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams, _lib
    rng = np.random.RandomState(5)
    res = 0.05
    m = np.zeros((150, 160), dtype=np.uint8)
    m[60:100, 50:90] = rng.randint(1, 256, (40, 40))
    m[10, 5:150] = 254
    org = np.array([-1.0, -0.5])
    path = np.array([[0., 0., 0.], [1., 0., 0.], [2., 0., 0.]])
    n = 64
    env = BatchedPlanEnv(CostMap2D(m, res, org), path, EnvParams(resolution=res, refine_path=False), n_envs=n)
    poses = np.stack([rng.uniform(0.5, 4.5, n), rng.uniform(1.5, 5.0, n), rng.uniform(-7, 7, n)], axis=1)
    pt = torch.from_numpy(poses).cuda()
    f64p = C.POINTER(C.c_double)
    o, s = np.array((-0.5, -2.0)), np.array((3.5, 4.0))
    shape = (C.c_int32 * 2)()
    _lib.check(env._lib.bcp_egocentric_shape(env._h, s.ctypes.data_as(f64p), shape))
    out = torch.full((n, shape[0], shape[1]), 99, dtype=torch.uint8, device="cuda")
    env.set_tuning(ego_sparse=4096)
    _lib.check(env._lib.bcp_egocentric_costmaps(env._h, pt.data_ptr(), n, o.ctypes.data_as(f64p), s.ctypes.data_as(f64p), 0,
                                                out.data_ptr(), None))
    assert _route(env)[0] == "ego_sparse_kernel"
    got = out.cpu().numpy()
    crowded = 0
    for i in range(n):
        ref = oracle.extract_egocentric(m, org, res, poses[i], o, s, 0)
        assert (ref == got[i]).all(), i
        crowded += int((ref != 0).sum() > 400)
    assert crowded > 10


def test_observation_after_a_masked_stream_refresh(torch_cuda, oracle):
    """bcp_egocentric_costmaps used to destroy the handle's CU-masked side stream when it (re)built its cell
    lists.  An endless pool refreshes on the masked stream, then the first observation is drawn (lists built), then the pool
    refreshes on the same stream again and the observation follows the re-sampled worlds."""
    torch = torch_cuda
    from bc_gym_planning_env_amd import EnvParams, mini_env
    from bc_gym_planning_env_amd.egocentric import BatchedEgocentricCostmap
    params = mini_env.RandomMiniEnvParams(env_params=EnvParams(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2, iteration_timeout=6))
    n = 32
    env = mini_env.BatchedRandomMiniEnv(n, params, episodes=4, endless=True, auto_reset=True, seed=4)
    env.side_cu_percent = 50
    wrap = BatchedEgocentricCostmap(env)
    rng = np.random.RandomState(3)
    res = params.env_params.resolution

    def check(t):
        img = wrap.observation()['env'].cpu().numpy()[..., 0]
        assert wrap.route()["kernel"] == "ego_sparse_kernel"
        st = env.state.robot.cpu().numpy()
        geom = env.geom_of_env.cpu().numpy()
        maps = env.pool.maps.cpu().numpy()
        for i in range(n):
            ref = oracle.extract_egocentric(maps[geom[i]], env.pool.origin, res, st[:3, i], (-0.5, -2.0), (3.5, 4.0))
            assert (ref == img[i]).all(), (t, i)

    for t in range(40):
        env.step(env.action_space.sample_batch(n, rng))
        if t % 8 == 7:
            env.refresh(overlap=True)
        if t in (9, 10, 25, 39):
            if t == 39:
                env.finish_refresh()
            torch.cuda.synchronize()
            check(t)
    assert (env.geom_of_env.cpu().numpy() != np.arange(n) * 4 + 1).any()
