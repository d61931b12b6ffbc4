"""CPU-only tests of the host side: the C-ABI library loads and exports every symbol include/bcplan.h declares,
the API mirror types behave like the reference's, reset-time host logic matches the oracle/golden data, and the
product refuses to run without a GPU (no silent fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from bc_gym_planning_env_amd import _lib, build
    build.build()  # hipcc cross-compiles gfx950 without a GPU
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "bcplan.h")).read()
    declared = set(re.findall(r"\b(bcp_[a-z_0-9]+)\s*\(", header))
    declared -= {"bcp_handle"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.bcp_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header():
    """ctypes mirrors must have the C layout (hipcc and ctypes agree on sizes through a tiny probe of offsets)."""
    import ctypes as C
    from bc_gym_planning_env_amd import _lib
    assert C.sizeof(_lib.BcpParams) == 8 * 4 + 32 * 2 * 8 + 7 * 8 + 6 * 8 + 3 * 8 + 4 * 4
    assert C.sizeof(_lib.BcpState) == 16 * 8
    assert C.sizeof(_lib.BcpStepIO) == 7 * 8
    assert _lib.BcpParams.verts.offset == 32 and _lib.BcpParams.dt.offset == 32 + 512


def test_no_gpu_means_loud_failure():
    """Without a GPU the product must refuse to run: no CPU fallback, no oracle behind the scenes."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import ctypes as C
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams, _lib, robots
    with pytest.raises(RuntimeError):
        BatchedPlanEnv(CostMap2D(np.zeros((16, 16), np.uint8), 0.03, np.zeros(2)), np.array([[0., 0, 0], [1., 0, 0]]))
    # and the C ABI itself reports BCP_E_NO_DEVICE
    lib = _lib.load()
    p = robots.make_bcp_params(EnvParams(), 'industrial_tricycle_v1', None)
    h = C.c_void_p()
    rc = lib.bcp_create(C.byref(p), 4, 0, 0, C.byref(h))
    assert rc == _lib.E_NO_DEVICE
    assert b"no CPU path" in lib.bcp_last_error()


def test_diffdrive_noise_raises_like_the_reference():
    """differential_drive.py:73: DiffDriveRobot.step with noise_parameters is an IndexError in the reference"""
    import pytest
    from bc_gym_planning_env_amd import EnvParams, _lib, robots
    with pytest.raises(IndexError):
        robots.make_bcp_params(EnvParams(), 'industrial_diffdrive_v1', dict(robots.PLANENV_NOISE))
    p = robots.make_bcp_params(EnvParams(), 'industrial_diffdrive_v1', dict(robots.PLANENV_NOISE), unpinned_diffdrive_noise=True)
    assert p.options == _lib.OPT_DIFFDRIVE_NOISE
    assert robots.make_bcp_params(EnvParams(), 'industrial_diffdrive_v1', None).options == 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bc_gym_planning_env_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "bcp_oracle" not in text, f


def test_refine_path_and_initial_state_match_reference_fixtures(oracle, golden_dir):
    """host_init (reset-time numpy logic) against what the reference produced for the recorded envs."""
    from bc_gym_planning_env_amd import host_init
    from bc_gym_planning_env_amd.api import RewardParams
    g = np.load(os.path.join(golden_dir, "g8_traj_mini_00.npz"))
    path = g["path"]
    # the recorded path is already refined: refining its two end points again must reproduce it
    coarse = np.array([path[0], path[-1]])
    np.testing.assert_array_equal(host_init.refine_path(coarse, 0.05), path)
    md, ti = host_init.initial_reward_state(path, RewardParams(0.2, np.pi / 8))
    assert (md, ti) == (float(g["init_min_dist"]), int(g["init_target_idx"]))
    assert (md, ti) == oracle.initial_reward_state(path, 0.2, np.pi / 8)
    ga = np.load(os.path.join(golden_dir, "g8_traj_aisle_default.npz"))
    md, ti = host_init.initial_reward_state(ga["path"], RewardParams(1.0, np.pi / 2))
    assert (md, ti) == (float(ga["init_min_dist"]), int(ga["init_target_idx"]))
    with pytest.raises(ValueError):
        host_init.initial_reward_state(path[:3], RewardParams(1.0, np.pi))


def test_time_table_is_running_sum(golden_dir):
    from bc_gym_planning_env_amd import host_init
    g = np.load(os.path.join(golden_dir, "g8_traj_mini_nonoise_40.npz"))
    t = host_init.time_table(0.05, 1300)
    np.testing.assert_array_equal(t[1:1251], g["time"])


def test_api_types_mirror_reference_semantics():
    from bc_gym_planning_env_amd import (Action, ContinuousRewardProviderState, CostMap2D, EnvParams, RewardParams,
                                         TricycleRobotState)
    from bc_gym_planning_env_amd.api import Box, seed_action_space
    p = EnvParams()
    assert (p.dt, p.resolution, p.iteration_timeout, p.path_delta) == (0.05, 0.03, 1200, 0.05)
    assert p.reward_provider_params == RewardParams(1.0, np.pi / 2, 0.0)
    with pytest.raises(Exception):
        p.dt = 1.0  # frozen, like the reference
    a = Action(command=np.array([0.3, -0.2]))
    assert a == Action(command=np.array([0.3, -0.2])) and a != Action(command=np.array([0.3, 0.2]))
    cm = CostMap2D(np.zeros((4, 5), np.uint8), 0.03, np.array([1., 2.]))
    assert not cm.get_origin().flags.writeable and cm == cm.copy()
    s = TricycleRobotState(x=1., y=2., angle=3., wheel_angle=0.5)
    assert s.get_pose() == (1., 2., 3.) and list(s.to_numpy_array()) == [1., 2., 3., 0., 0., 0.5]
    rps = ContinuousRewardProviderState(0.5, np.zeros((5, 3)), 5)
    assert rps.done() and len(rps.current_path()) == 0
    with pytest.raises(ValueError):
        rps.current_goal_pose()
    box = Box(np.array([np.pi / 30, -np.pi / 2]), np.array([np.pi / 6, np.pi / 2]))
    seed_action_space(0)
    act = box.sample()
    assert act.command.dtype == np.float32 and box.contains(act.command)
    # same stream as the reference's module RandomState(0): first uniform pair
    r = np.random.RandomState(0).uniform(low=box.low, high=box.high, size=(2,)).astype(np.float32)
    np.testing.assert_array_equal(act.command, r)


def test_env_block_partition():
    from bc_gym_planning_env_amd.distributed import env_block
    for n, w in ((65536 * 8, 8), (10, 3), (7, 8)):
        blocks = [env_block(n, r, w) for r in range(w)]
        assert sum(c for _, c in blocks) == n
        assert all(blocks[r][0] + blocks[r][1] == blocks[r + 1][0] for r in range(w - 1))


def test_state_serialization_round_trip():
    """State / robot state / reward-provider state / costmap render to basic types and back (envs/base/env.py:135-176,
    utilities/serialize.py), for both reward providers and with filled delay queues."""
    import pickle
    from bc_gym_planning_env_amd import api
    path = np.arange(12, dtype=np.float64).reshape(4, 3)
    cm = api.CostMap2D(np.arange(12, dtype=np.uint8).reshape(3, 4), 0.05, np.array([1.0, -2.0]))
    for rps in (api.ContinuousRewardProviderState(min_spat_dist_so_far=0.5, path=path, target_idx=2),
                api.ContinuousRewardPurePursuitProviderState(min_spat_dist_so_far=1.5, path=path, target_idx=1)):
        rs = api.TricycleRobotState(1., 2., 3., 0.1, 0.2, 0.3, 0.4)
        st = api.State(reward_provider_state=rps, path=rps.current_path(), original_path=path.copy(), costmap=cm,
                       iter_timeout=1200, current_time=0.35, current_iter=7, robot_collided=True,
                       poses_queue=[np.array([1., 2., 3.]), np.array([4., 5., 6.])],
                       robot_state_queue=[api.TricycleRobotState(9., 8., 7.)],
                       control_queue=[api.Action(command=np.array([0.2, 0.1]))], pose=np.array([1., 2., 3.]), robot_state=rs)
        blob = pickle.dumps(st.serialize())
        back = api.State.deserialize(pickle.loads(blob))
        assert back == st and back is not st
        assert type(back.reward_provider_state) is type(rps) and back.costmap == cm
        cp = st.copy()
        assert cp == st and cp.poses_queue[0] is not st.poses_queue[0]
        cp.poses_queue[0][0] = 99.0
        assert cp != st


G13 = ["plain", "delay_p1s1", "delay_c2p3s1", "delay_c2p3", "delay_filling", "delay_fresh", "pp", "pp_delay"]


@pytest.mark.parametrize("tag", G13)
def test_reference_serialize_records_are_read_and_written_key_for_key(tag):
    """g13: records the genuine reference's PlanEnv.serialize() produced mid-episode (delay queues filled, both reward
    providers).  State / EnvParams / CostMap2D read them and write them back key for key, value for value."""
    from oracle import records
    from bc_gym_planning_env_amd import api
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g13_serialized_%s.npz" % tag))
    rec = records.unpack(g["record"], g)
    assert set(rec) == {'version', 'state', 'params', 'path', 'costmap'}
    st = api.State.deserialize(rec['state'])
    params = api.EnvParams.deserialize(rec['params'])
    assert records.same(rec['state'], st.serialize()) == []
    assert records.same(rec['params'], params.serialize(), 'params') == []
    assert records.same(rec['costmap'], api.CostMap2D.from_state(rec['costmap']).get_state(), 'costmap') == []
    it = int(rec['state']['current_iter'])
    assert it == int(g["steps_before"])
    assert len(st.poses_queue) == min(it, params.pose_delay) and len(st.control_queue) == min(it, params.control_delay)
    assert len(st.robot_state_queue) == min(it, params.state_delay)
    assert all(isinstance(a, api.Action) for a in st.control_queue)
    assert all(isinstance(r, api.TricycleRobotState) for r in st.robot_state_queue)
    assert st.copy() == st


def test_env_params_serialize_roundtrip():
    """EnvParams / RewardParams as the reference's Serializable (envs/base/params.py:46-59): nested dict of basic types"""
    import pickle
    from bc_gym_planning_env_amd import EnvParams
    from bc_gym_planning_env_amd.api import RewardParams
    p = EnvParams(goal_spat_dist=0.3, pose_delay=2, reward_provider_name='continuous_reward_pure_pursuit',
                  reward_provider_params=RewardParams(spatial_precision=0.3, angular_precision=0.5,
                                                      spatial_progress_multiplier=2.0))
    rec = pickle.loads(pickle.dumps(p.serialize()))
    assert rec['version'] == 1 and rec['reward_provider_params'] == dict(
        spatial_precision=0.3, angular_precision=0.5, spatial_progress_multiplier=2.0, version=1)
    assert EnvParams.deserialize(rec) == p and EnvParams.deserialize(EnvParams().serialize()) == EnvParams()


def test_c_client_compiles_against_the_header(tmp_path):
    """include/bcplan.h is a C header: the plain C11 client of tests/c_abi builds and links against libbcplan.so with gcc
    (it only runs on a GPU box: tests/test_gpu_c_abi.py)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "bc_gym_planning_env_amd", "libbcplan.so")
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(root, "include"),
           "-I/opt/rocm/include", os.path.join(root, "tests", "c_abi", "step_from_c.c"), lib, "-L/opt/rocm/lib",
           "-lamdhip64", "-lm", "-o", str(tmp_path / "step_from_c")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_exact_division_identities_of_the_sparse_exact_test():
    """coop_collides_sparse (csrc/bcp_coop.h) replaces the two exact integer divisions of an edge's set-up --
    ((u - up) * 65536) / ddy, truncated, and floor(2^32 / (2 dy)) + 1 -- by a float32 reciprocal, two Newton steps in float64,
    a product and a small push before the truncation (coop_exact_rcp).  The identities, for every divisor and dividend a
    footprint image of up to 512 pixels can produce and a hardware reciprocal that is off by up to two units in the last place."""
    def newton(d, r0):
        d, r = d.astype(np.longdouble), r0.astype(np.longdouble)
        for _ in range(2):   # fma(fma(-d, r, 1), r, r): one rounding per fma
            e = (1.0 - d * r).astype(np.float64).astype(np.longdouble)
            r = (r + e * r).astype(np.float64).astype(np.longdouble)
        return r.astype(np.float64)

    def rcp32(d, ulps):
        r = (np.float32(1.0) / d.astype(np.float32))
        for _ in range(abs(ulps)):
            r = np.nextafter(r, np.float32(np.inf if ulps > 0 else -np.inf))
        return r.astype(np.float64)

    d = np.arange(1, 513, dtype=np.float64)
    k = np.arange(-512, 513, dtype=np.int64)
    N, D = np.meshgrid(k * 65536, np.arange(1, 513, dtype=np.int64))
    for ulps in (-2, -1, 0, 1, 2):
        inv = np.floor(4294967296.0 * newton(2 * d, rcp32(2 * d, ulps)) + 2.0 ** -17).astype(np.uint64) + 1
        assert (inv == np.uint64(4294967296) // (2 * d).astype(np.uint64) + 1).all()
        r = newton(d, rcp32(d, ulps))[:, None]
        for sign in (1, -1):   # (a negative divisor: the reciprocal and its Newton steps are odd in d)
            q0 = N.astype(np.float64) * (sign * r)
            q = np.trunc(q0 + np.copysign(2.0 ** -12, q0)).astype(np.int64)
            assert (q == (np.abs(N) // D) * np.sign(N) * sign).all()
