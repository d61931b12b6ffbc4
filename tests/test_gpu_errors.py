"""Error conventions of the C ABI on a real device: negative return code + bcp_last_error(), no exceptions, no
crashes (the reference raises Python exceptions for the same misuse)."""
import ctypes as C
import os

import numpy as np
import pytest

from util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

E_INVALID, E_STATE = -1, -4


def _params(**kw):
    from bc_gym_planning_env_amd import EnvParams, robots
    return robots.make_bcp_params(EnvParams(**kw), 'industrial_tricycle_v1', None)


def _create(L, p, n=8):
    h = C.c_void_p()
    rc = L.bcp_create(C.byref(p), n, 0, 0, C.byref(h))
    return rc, h


def test_create_rejects_bad_parameters(torch_cuda):
    from bc_gym_planning_env_amd import _lib
    L = _lib.load()
    for field, value in (("reward_provider", 7), ("pose_delay", -1), ("n_verts", 2), ("abi_version", 1), ("model", 9)):
        p = _params()
        setattr(p, field, value)
        rc, h = _create(L, p)
        assert rc == E_INVALID and not h.value, field
        assert len(L.bcp_last_error()) > 10
    p = _params()
    p.dt = 0.0
    assert _create(L, p)[0] == E_INVALID       # path_velocity asserts dt > 0 (utilities/path_tools.py:307)
    assert _create(L, _params(), n=0)[0] == E_INVALID


def test_diffdrive_with_noise_needs_the_opt_in(torch_cuda):
    """DiffDriveRobot with noise_parameters raises IndexError in the reference itself (differential_drive.py:73): the
    library refuses the combination unless BCP_OPT_DIFFDRIVE_NOISE opts in to its unpinned analogue, and the Python layer
    raises the reference's exception."""
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams, _lib, robots
    L = _lib.load()
    noise = dict(robots.PLANENV_NOISE)
    with pytest.raises(IndexError):
        robots.make_bcp_params(EnvParams(), 'industrial_diffdrive_v1', noise)
    p = robots.make_bcp_params(EnvParams(), 'industrial_diffdrive_v1', noise, unpinned_diffdrive_noise=True)
    assert p.options & _lib.OPT_DIFFDRIVE_NOISE and p.noise_on
    rc, h = _create(L, p)
    assert rc == 0
    assert L.bcp_destroy(h) == 0
    p.options = 0
    rc, h = _create(L, p)
    assert rc == E_INVALID and not h.value and b"IndexError" in L.bcp_last_error()
    g = np.load(os.path.join(GOLDEN, "g8dd_traj_mini64_00.npz"))
    res = float(g["resolution"])
    cm = CostMap2D(g["costmap"], res, g["origin"])
    params = EnvParams(resolution=res, refine_path=False, robot_name='industrial_diffdrive_v1')
    with pytest.raises(IndexError):
        BatchedPlanEnv(cm, g["path"], params, n_envs=2, noise_parameters=noise)
    assert BatchedPlanEnv(cm, g["path"], params, n_envs=2).noise_parameters is None   # ('planenv' is the tricycle's noise)
    env = BatchedPlanEnv(cm, g["path"], params, n_envs=2, noise_parameters=noise, unpinned_diffdrive_noise=True)
    env.step(np.array([[0.3, 0.2], [0.3, -0.2]], dtype=np.float32))
    assert np.isfinite(env.state.robot.cpu().numpy()).all()


def test_call_order_and_argument_checks(torch_cuda):
    torch = torch_cuda
    from bc_gym_planning_env_amd import _lib
    L = _lib.load()
    rc, h = _create(L, _params(pose_delay=1), n=16)
    assert rc == 0
    io = _lib.BcpStepIO()
    assert L.bcp_step(h, C.byref(io), 0, None) == E_STATE                      # nothing set yet
    assert L.bcp_reset_masked(h, None, None) == E_STATE
    assert L.bcp_broadcast_state(h, 0, None, None) == E_STATE
    assert L.bcp_egocentric_shape(h, None, (C.c_int32 * 2)()) == E_STATE
    # a state without the arrays the configured pose delay needs
    st = _lib.BcpState()
    bufs = [torch.zeros(16, dtype=torch.float64, device="cuda") for _ in range(8)]
    for name, b in zip(("x", "y", "angle", "v", "w", "steering_motor_command", "wheel_angle", "min_spat_dist_so_far"), bufs):
        setattr(st, name, b.data_ptr())
    ints = [torch.zeros(16, dtype=torch.int32, device="cuda") for _ in range(2)]
    st.target_idx, st.current_iter = ints[0].data_ptr(), ints[1].data_ptr()
    flag = torch.zeros(16, dtype=torch.uint8, device="cuda")
    st.robot_collided = flag.data_ptr()
    assert L.bcp_bind_state(h, C.byref(st)) == E_INVALID
    assert b"delay" in L.bcp_last_error()
    # geometry pool misuse
    assert L.bcp_set_geometry_pool(h, 4, None, None) == E_INVALID
    assert L.bcp_set_geometry_pool(h, -1, None, None) == E_INVALID
    assert L.bcp_set_tuning(h, 99, 0) == E_INVALID and L.bcp_set_tuning(h, _lib.TUNE_EXACT_MODE, 5) == E_INVALID
    assert L.bcp_destroy(h) == 0


def test_egocentric_argument_checks(torch_cuda):
    torch = torch_cuda
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams, _lib
    g = np.load(os.path.join(GOLDEN, "g8_traj_mini_03.npz"))
    res = float(g["resolution"])
    env = BatchedPlanEnv(CostMap2D(g["costmap"], res, g["origin"]), g["path"], EnvParams(resolution=res, refine_path=False),
                         n_envs=4)
    L, h = env._lib, env._h
    f64p = C.POINTER(C.c_double)
    out = torch.zeros(4 * 200 * 200, dtype=torch.uint8, device="cuda")
    org = (C.c_double * 2)(-0.5, -2.0)
    size = (C.c_double * 2)(3.5, 4.0)
    tiny = (C.c_double * 2)(0.05, 1.0)
    assert L.bcp_egocentric_costmaps(h, None, 4, org, None, 0, out.data_ptr(), None) == E_INVALID      # origin without size
    assert L.bcp_egocentric_costmaps(h, None, 3, org, size, 0, out.data_ptr(), None) == E_INVALID      # n != n_envs
    assert L.bcp_egocentric_costmaps(h, None, 4, org, tiny, 0, out.data_ptr(), None) == E_INVALID      # 2 px wide
    assert L.bcp_egocentric_costmaps(h, None, 4, org, size, 0, None, None) == E_INVALID
    assert L.bcp_egocentric_costmaps(h, None, 4, org, size, 0, out.data_ptr(), None) == 0
    assert L.bcp_broadcast_state(h, 4, None, None) == E_INVALID and L.bcp_broadcast_state(h, -1, None, None) == E_INVALID
    assert L.bcp_goal_n_state(h, None, out.data_ptr(), None) == E_INVALID
    torch.cuda.synchronize()
    # the env still works after all the refused calls
    env.step(env.action_space.sample_batch(4))
    env.check_errors()


def test_python_layer_raises_like_the_reference(torch_cuda):
    from bc_gym_planning_env_amd import BatchedPlanEnv, CostMap2D, EnvParams
    g = np.load(os.path.join(GOLDEN, "g8_traj_mini_03.npz"))
    res = float(g["resolution"])
    cm = CostMap2D(g["costmap"], res, g["origin"])
    with pytest.raises(AssertionError):       # get_reward_provider_example (reward_provider_examples_factory.py:44-46)
        BatchedPlanEnv(cm, g["path"], EnvParams(resolution=res, reward_provider_name='no_such_provider'), n_envs=2)
    with pytest.raises(ValueError):           # generate_initial_state: "Goal pose too close" (reward.py:275-277)
        BatchedPlanEnv(cm, g["path"][:1].repeat(2, axis=0), EnvParams(resolution=res, refine_path=False), n_envs=2)
    env = BatchedPlanEnv(cm, g["path"], EnvParams(resolution=res, refine_path=False), n_envs=2)
    with pytest.raises(ValueError):
        env.step(np.zeros((3, 2), dtype=np.float32))
    with pytest.raises(NotImplementedError):
        env.render()


def test_a_wait_that_never_ends_gives_up(torch_cuda, launcher, tmp_path):
    """The hand-offs inside step_local_kernel are waits on LDS words; every one of them is bounded.  A -DBCP_DIAG build
    (compiled here, on the box) can withhold the verdicts of the parked poses: the movers' waits then run into their
    limit, the step RETURNS (milliseconds, not a wedged GPU), the envs concerned are finished as free with
    BCP_ERR_INTERNAL, bcp_expired_waits counts the waits and check_errors() raises."""
    import json
    import sys
    from bc_gym_planning_env_amd import build
    lib = str(tmp_path / "libbcplan_diag.so")
    r = launcher.run([build.hipcc()] + build.FLAGS + ["-DBCP_DIAG", build.SRC, "-o", lib], timeout=400, cwd=ROOT)
    assert r["rc"] == 0, r["err"][-3000:]
    r = launcher.run([sys.executable, os.path.join(ROOT, "tests", "ranks", "withheld_verdicts.py"), lib], timeout=300, cwd=ROOT)
    assert r["rc"] == 0, r["err"][-3000:]
    got = json.loads([x for x in r["out"].splitlines() if x.startswith("{")][-1])
    assert got["rc"] == 0 and got["seconds"] < 5.0
    assert got["expired_before"] == 0 and got["expired_after"] > 0
    assert got["envs_flagged"] > 0 and got["flagged_collided_now"] == 0 and got["check_errors_raised"]
