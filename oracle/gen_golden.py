#!/usr/bin/env python
"""Generate tests/golden/*.npz from the GENUINE reference, imported in the build container through
oracle/ref_harness.py.  Run from the repo root:  python oracle/gen_golden.py

The fixtures are data only (inputs + expected outputs).  Nothing here travels as reference source.
Fixture list:
  g1_tricycle_step.npz      TricycleRobot.step, noise off          (tricycle_model.py:478-538)
  g2_tricycle_step_noise.npz  same with noise, normals in slot form  (differential_drive.py:55-74)
  g3_diffdrive_step.npz     DiffDriveRobot.step with/without noise (differential_drive.py:236-265)
  g4_scalar_utils.npz       normalize_angle / world_to_pixel / path_velocity incl. the reference's KAT values
  g5_footprint_vertices.npz pre-fill output of get_pixel_footprint (integer polygon + half size)
  g6_pose_collides.npz      pose_collides verdicts on random poses (fill = oracle restatement, see harness note)
  g7_reward.npz             ContinuousRewardProvider traces
  g8_traj_*.npz             full PlanEnv.step trajectories (RandomMiniEnv seeds, AisleTurnEnv variants)
  g9 .. g12                 RandomMiniEnv worlds, egocentric observations, delays / pure pursuit, coloured ego costmap
  g13_serialized_*.npz      PlanEnv.serialize() records taken mid-episode + the next 100 steps of the live env
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle as O  # noqa: E402
from oracle import ref_harness as H  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print("wrote %-32s %7.1f KiB" % (name, os.path.getsize(path) / 1024.))


class SlotTap(object):
    """Records the standard normals consumed by one noisy motion step in SLOT form (slot = index of the
    _gaussian_noise call inside kinematic_body_pose_motion_step_with_noise; NaN = not drawn)."""

    def __init__(self, dd_module, seed):
        self.dd = dd_module
        self.tap = H.NoiseTap(seed)
        self.slots = []
        self._orig = None

    def __enter__(self):
        self.tap.__enter__()
        self._orig = self.dd._gaussian_noise
        orig = self._orig
        tap = self.tap
        slots = self.slots

        def counting(variance):
            before = len(tap.calls)
            r = orig(variance)
            slots.append(tap.calls[-1] if len(tap.calls) > before else np.nan)
            return r
        self.dd._gaussian_noise = counting
        return self

    def __exit__(self, *exc):
        self.dd._gaussian_noise = self._orig
        self.tap.__exit__(*exc)
        return False

    def take(self):
        assert len(self.slots) in (0, 3), self.slots
        out = np.array(self.slots if self.slots else [np.nan] * 3, dtype=np.float64)
        del self.slots[:]
        del self.tap.calls[:]
        return out


def tri_state_vec(s):
    return np.array([s.x, s.y, s.angle, s.v, s.w, s.steering_motor_command, s.wheel_angle], dtype=np.float64)


def dd_state_vec(s):
    return np.array([s.x, s.y, s.angle, s.v, s.w, 0.0, 0.0], dtype=np.float64)


def random_robot_inputs(rng, n, tricycle):
    st = np.zeros((n, 7))
    st[:, 0:2] = rng.uniform(-3, 3, (n, 2))
    st[:, 2] = rng.uniform(-np.pi, np.pi, n)
    st[:, 3] = rng.uniform(-0.2, 1.2, n) * (rng.rand(n) > 0.1)
    st[:, 4] = rng.uniform(-1.0, 1.0, n) * (rng.rand(n) > 0.1)
    if tricycle:
        st[:, 6] = rng.uniform(-1.6, 1.6, n)
    cmd = np.stack([rng.uniform(-0.2, 1.5, n), rng.uniform(-2.0, 2.0, n)], axis=1)
    # a share of float32 commands, as action_space.sample() produces (envs/base/spaces.py:134-141)
    f32 = rng.rand(n) < 0.5
    cmd[f32] = cmd[f32].astype(np.float32).astype(np.float64)
    # angles right at the wrap-around
    st[:16, 2] = np.pi - 1e-9 * np.arange(16)
    st[16:32, 2] = -np.pi + 1e-9 * np.arange(16)
    cmd[32:40] = 0.0
    return st, cmd, f32


def gen_robot_steps():
    from bc_gym_planning_env.robot_models import differential_drive as dd
    from bc_gym_planning_env.robot_models.tricycle_model import TricycleRobot, TricycleRobotState
    from bc_gym_planning_env.robot_models.differential_drive import DiffDriveRobot, DiffdriveRobotState
    from bc_gym_planning_env.robot_models.robot_dimensions_examples import get_dimensions_example
    from bc_gym_planning_env.envs.base.action import Action
    tri_dims = get_dimensions_example('industrial_tricycle_v1')
    dd_dims = get_dimensions_example('industrial_diffdrive_v1')
    dt = 0.05

    def run_tricycle(st, cmd, f32, noise, seed, dynamic=True, pid=True):
        n = len(st)
        out = np.zeros((n, 7))
        z = np.full((n, 3), np.nan)
        robot = TricycleRobot(dimensions=tri_dims, noise_parameters=noise, dynamic_model=dynamic,
                              model_front_column_pid=pid)
        with SlotTap(dd, seed) as tap:
            for i in range(n):
                robot.set_state(TricycleRobotState(x=st[i, 0], y=st[i, 1], angle=st[i, 2], v=st[i, 3], w=st[i, 4],
                                                   steering_motor_command=st[i, 5], wheel_angle=st[i, 6]))
                c = cmd[i].astype(np.float32) if f32[i] else cmd[i]
                robot.step(dt, Action(command=c))
                out[i] = tri_state_vec(robot.get_state())
                z[i] = tap.take()
        return out, z

    rng = np.random.RandomState(101)
    st, cmd, f32 = random_robot_inputs(rng, 2048, True)
    out, _ = run_tricycle(st, cmd, f32, None, 1)
    save("g1_tricycle_step.npz", state=st, cmd=cmd, dt=dt, out=out)
    # kinematic-only / no front-column PID variants (tricycle_model.py:38-68, :61, :110)
    outs = {}
    for name, (dyn, pid) in dict(kin_pid=(False, True), kin_nopid=(False, False), dyn_nopid=(True, False)).items():
        # without the front-column PID a float32 command stays float32 through np.clip/np.cos (a dtype artefact of
        # a mode PlanEnv never enables, tricycle_model.py:61,110): those variants are pinned on float64 commands.
        outs[name], _ = run_tricycle(st[:512], cmd[:512], f32[:512] & pid, None, 1, dynamic=dyn, pid=pid)
    save("g1b_tricycle_variants.npz", state=st[:512], cmd=cmd[:512], dt=dt, **outs)

    rng = np.random.RandomState(102)
    st, cmd, f32 = random_robot_inputs(rng, 2048, True)
    alphas = np.array([O.PLANENV_NOISE, (1e-2, 1e-3, 1e-2, 1e-2, 1e-3, 1e-3), (0.05, 0.0, 0.0, 0.02, 0.0, 0.01),
                       (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)])
    outs, zs, aidx = [], [], []
    for a_i, a in enumerate(alphas):
        noise = dict(("alpha%d" % (k + 1), a[k]) for k in range(6))
        sl = slice(a_i * 512, (a_i + 1) * 512)
        o, z = run_tricycle(st[sl], cmd[sl], f32[sl], noise, 200 + a_i)
        outs.append(o)
        zs.append(z)
        aidx.append(np.full(512, a_i))
    save("g2_tricycle_step_noise.npz", state=st, cmd=cmd, dt=dt, alphas=alphas, alpha_idx=np.concatenate(aidx),
         z=np.concatenate(zs), out=np.concatenate(outs))

    # DiffDriveRobot.step: noise off only -- with noise the reference itself raises IndexError
    # (differential_drive.py:73 indexes a 1-D pose with [:, 2]), so there is nothing to pin.
    rng = np.random.RandomState(103)
    st, cmd, f32 = random_robot_inputs(rng, 2048, False)
    st[:, 5:] = 0.0
    robot = DiffDriveRobot(dimensions=dd_dims, noise_parameters=None)
    o = np.zeros((2048, 7))
    for i in range(2048):
        robot.set_state(DiffdriveRobotState(x=st[i, 0], y=st[i, 1], angle=st[i, 2], v=st[i, 3], w=st[i, 4]))
        # The command VALUES are float32-representable for half the rows but are handed over as float64: the
        # diff-drive model does scalar arithmetic on the command (differential_drive.py:34-35), which numpy 1.x
        # (the reference's era, value-based promotion) performs in float64 and numpy >= 2 (NEP 50) in float32.
        # The build follows the numpy 1.x semantics: commands are widened to float64 first.
        robot.step(dt, Action(command=cmd[i]))
        o[i] = dd_state_vec(robot.get_state())
    save("g3_diffdrive_step.npz", state=st, cmd=cmd, dt=dt, out=o)


def gen_scalar_utils():
    from bc_gym_planning_env.utilities.coordinate_transformations import normalize_angle, world_to_pixel, diff_angles
    from bc_gym_planning_env.utilities.path_tools import path_velocity, pose_distances
    rng = np.random.RandomState(104)
    # normalize_angle: KAT inputs of test_coordinate_transformations.py:32-83 plus random / boundary values
    kat = np.array([0., 1., -1., np.pi, -np.pi, 3 * np.pi, -3 * np.pi, 2 * np.pi, -2 * np.pi, 4 * np.pi, 0.5 * np.pi,
                    -0.5 * np.pi, 1.5 * np.pi, -1.5 * np.pi, 2.5 * np.pi, 100., -100., 1e-12, -1e-12,
                    np.pi + 1e-12, -np.pi - 1e-12, np.nextafter(np.pi, 4), np.nextafter(-np.pi, -4)])
    na_in = np.concatenate([kat, rng.uniform(-30, 30, 4096), rng.uniform(-np.pi, np.pi, 1024)])
    na_out = normalize_angle(na_in)
    da_a, da_b = rng.uniform(-7, 7, 2048), rng.uniform(-7, 7, 2048)
    da_out = diff_angles(da_a, da_b)
    # world_to_pixel: the KAT cases of test_coordinate_transformations.py:1544-1649 (half-to-even and
    # multiply-by-reciprocal) + random
    w2p_cases = []
    for xy, origin, res in [
        (np.array([[0., 0.], [1., 1.], [2.5, 3.5], [-1.5, -0.5], [0.025, 0.075]]), np.array([0., 0.]), 0.05),
        (np.array([[0.5, 1.5], [2.5, 3.5], [-0.5, -1.5]]), np.array([0., 0.]), 1.0),
        (np.array([[1.45, 2.85], [0.15, 0.25], [0.35, 0.45]]), np.array([-0.3, 0.1]), 0.1),
        (np.array([[4.35, 0.29], [0.57, 1.14]]), np.array([0., 0.]), 0.03),
        (rng.uniform(-5, 5, (2048, 2)), np.array([-2.75, -2.75]), 0.03),
        (rng.uniform(-5, 5, (1024, 2)), np.array([-3.1, 1.7]), 5.5 / 64),
        (np.round(rng.uniform(-100, 100, (1024, 2))) * 0.015, np.array([0., 0.]), 0.03),  # many exact halves
    ]:
        w2p_cases.append((xy, origin, res, world_to_pixel(xy, origin, res)))
    # path_velocity two-row cases (as called from the robot models)
    n = 2048
    p0 = np.stack([rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), rng.uniform(-np.pi, np.pi, n)], axis=1)
    d = np.stack([rng.uniform(-0.1, 0.1, n), rng.uniform(-0.1, 0.1, n), rng.uniform(-0.3, 0.3, n)], axis=1)
    d[:64, :2] = 0.0                    # no motion: sign() == 0 path
    d[64:128, 0] = 0.0
    p0[128:192, 2] = np.pi - 0.01       # wrap-around of the angle difference
    d[128:192, 2] = 0.05
    p1 = p0 + d
    p1[:, 2] = normalize_angle(p1[:, 2])
    pv = np.zeros((n, 2))
    for i in range(n):
        v, w = path_velocity(np.vstack([np.hstack([0., p0[i]]), np.hstack([0.05, p1[i]])]))
        pv[i] = v[0], w[0]
    a, b = p0.copy(), p1.copy()
    pd_lin, pd_ang = pose_distances(a, b)
    save("g4_scalar_utils.npz", na_in=na_in, na_out=na_out, da_a=da_a, da_b=da_b, da_out=da_out,
         pv_p0=p0, pv_p1=p1, pv_dt=0.05, pv_out=pv, pd_lin=pd_lin, pd_ang=pd_ang,
         **dict(("w2p%d_%s" % (i, k), v) for i, c in enumerate(w2p_cases)
                for k, v in zip(("xy", "origin", "res", "out"), c)))


def gen_footprints():
    import cv2  # the harness stub
    from bc_gym_planning_env.utilities.path_tools import get_pixel_footprint
    rng = np.random.RandomState(105)
    captured = {}
    orig_fill = cv2.fillPoly

    def capture(img, pts, color, *a, **k):
        captured["shape"] = img.shape
        captured["pts"] = np.array(pts[0], dtype=np.int32)
        return orig_fill(img, pts, color, *a, **k)
    cv2.fillPoly = capture
    try:
        out = {}
        for fname, fp in (("tri", O.TRICYCLE_FOOTPRINT), ("dd", O.DIFFDRIVE_FOOTPRINT)):
            for rname, res in (("r003", 0.03), ("r64", 5.5 / 64), ("r256", 10. / 256)):
                angles = np.concatenate([rng.uniform(-np.pi, np.pi, 2000),
                                         np.arange(-8, 9) * np.pi / 8, rng.uniform(-10, 10, 31)])
                pts = np.zeros((len(angles), len(fp), 2), dtype=np.int32)
                shapes = np.zeros((len(angles), 2), dtype=np.int32)
                area = np.zeros(len(angles), dtype=np.int32)
                for i, a in enumerate(angles):
                    kern = get_pixel_footprint(a, fp, res)
                    pts[i] = captured["pts"]
                    shapes[i] = captured["shape"]
                    area[i] = np.count_nonzero(kern)
                key = "%s_%s" % (fname, rname)
                out[key + "_angles"] = angles
                out[key + "_pts"] = pts
                out[key + "_shape"] = shapes
                out[key + "_area"] = area  # area depends on the (unpinned) fill; informational
                out[key + "_res"] = res
        save("g5_footprint_vertices.npz", **out)
    finally:
        cv2.fillPoly = orig_fill


def make_mini_env(seed, resolution=0.03, robot_name=None):
    from bc_gym_planning_env.envs.mini_env import RandomMiniEnv, RandomMiniEnvParams
    from bc_gym_planning_env.envs.base.params import EnvParams
    kw = dict(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2, resolution=resolution)
    if robot_name is not None:
        kw["robot_name"] = robot_name
    params = RandomMiniEnvParams(env_params=EnvParams(**kw))
    return RandomMiniEnv(params=params, rng=np.random.RandomState(seed), draw_new_turn_on_reset=False)


def gen_pose_collides():
    from bc_gym_planning_env.envs.base.env import pose_collides
    from bc_gym_planning_env.robot_models.tricycle_model import TricycleRobot
    from bc_gym_planning_env.robot_models.differential_drive import DiffDriveRobot
    from bc_gym_planning_env.robot_models.robot_dimensions_examples import get_dimensions_example
    rng = np.random.RandomState(106)
    out = {}
    for tag, seed, res, robot in (("mini0", 0, 0.03, "tri"), ("mini3", 3, 0.03, "tri"), ("mini64", 0, 5.5 / 64, "dd")):
        env = make_mini_env(seed, res)
        costmap = env._env._state.costmap
        robot_obj = (TricycleRobot(dimensions=get_dimensions_example('industrial_tricycle_v1')) if robot == "tri"
                     else DiffDriveRobot(dimensions=get_dimensions_example('industrial_diffdrive_v1')))
        n = 1500
        poses = np.stack([rng.uniform(-3.6, 3.6, n), rng.uniform(-3.6, 3.6, n), rng.uniform(-np.pi, np.pi, n)], axis=1)
        verdict = np.array([pose_collides(p[0], p[1], p[2], robot_obj, costmap) for p in poses], dtype=np.uint8)
        out[tag + "_map"] = costmap.get_data().copy()
        out[tag + "_origin"] = np.array(costmap.get_origin())
        out[tag + "_res"] = costmap.get_resolution()
        out[tag + "_poses"] = poses
        out[tag + "_collides"] = verdict
        out[tag + "_robot"] = np.array(0 if robot == "tri" else 1)
        print("   %s: %d/%d collide" % (tag, verdict.sum(), n))
    save("g6_pose_collides.npz", **out)


def gen_reward():
    from bc_gym_planning_env.envs.base.reward import ContinuousRewardProvider, RewardParams
    from bc_gym_planning_env.utilities.path_tools import refine_path
    rng = np.random.RandomState(107)
    out = {}
    for tag, (sp, ap, mult) in dict(a=(0.2, np.pi / 8, 0.0), b=(1.0, np.pi / 2, 0.0), c=(0.3, 0.5, 2.5)).items():
        coarse = np.array([[0., 0., 0.3], [1.5, 0.5, 0.3], [2.5, 2.0, 1.2], [2.5, 3.5, np.pi / 2]])
        path = refine_path(coarse, 0.05)
        params = RewardParams(spatial_precision=sp, angular_precision=ap, spatial_progress_multiplier=mult)
        prov = ContinuousRewardProvider(params=params)
        prov.set_state(ContinuousRewardProvider.generate_initial_state(path, params))
        init = prov.get_state()

        class S(object):
            pass
        # a noisy walk along the path (sometimes jumping back / far away)
        n = 600
        idx = np.clip((np.arange(n) * len(path) / 500.).astype(int) + rng.randint(-3, 4, n), 0, len(path) - 1)
        poses = path[idx] + np.stack([rng.normal(0, 0.05, n), rng.normal(0, 0.05, n), rng.normal(0, 0.15, n)], axis=1)
        poses[::37] += np.array([1.0, -1.0, 0.5])
        rew = np.zeros(n)
        md = np.zeros(n)
        ti = np.zeros(n, dtype=np.int32)
        for i in range(n):
            s = S()
            s.pose = poses[i]
            rew[i] = prov.reward(s)
            st = prov.get_state()
            md[i], ti[i] = st.min_spat_dist_so_far, st.target_idx
        out.update({tag + "_path": path, tag + "_params": np.array([sp, ap, mult]), tag + "_poses": poses,
                    tag + "_reward": rew, tag + "_min_dist": md, tag + "_target_idx": ti,
                    tag + "_init": np.array([init.min_spat_dist_so_far, init.target_idx])})
    save("g7_reward.npz", **out)


def record_trajectory(env, plan_env, n_steps, action_seed, noise_seed, robot_kind="tri"):
    """Drive the reference PlanEnv with action_space.sample() actions and record everything."""
    from bc_gym_planning_env.robot_models import differential_drive as dd
    from bc_gym_planning_env.envs.base import spaces
    spaces.SPACE_LOCAL_RANDOM_STATE.seed(action_seed)
    state0 = plan_env.get_state()
    rs = state0.robot_state
    rec = dict(
        costmap=state0.costmap.get_data().copy(), origin=np.array(state0.costmap.get_origin()),
        resolution=np.float64(state0.costmap.get_resolution()),
        path=np.array(state0.reward_provider_state.path), init_target_idx=np.int32(state0.reward_provider_state.target_idx),
        init_min_dist=np.float64(state0.reward_provider_state.min_spat_dist_so_far),
        init_state=tri_state_vec(rs) if robot_kind == "tri" else dd_state_vec(rs),
    )
    T = n_steps
    act = np.zeros((T, 2))
    z = np.full((T, 3), np.nan)
    st = np.zeros((T, 7))
    rew = np.zeros(T)
    done = np.zeros(T, dtype=np.uint8)
    coll = np.zeros(T, dtype=np.uint8)
    tidx = np.zeros(T, dtype=np.int32)
    mind = np.zeros(T)
    tm = np.zeros(T)
    obs_path_len = np.zeros(T, dtype=np.int32)
    with SlotTap(dd, noise_seed) as tap:
        for t in range(T):
            a = env.action_space.sample()
            act[t] = a.command.astype(np.float64)
            assert a.command.dtype == np.float32
            obs, r, d, _ = env.step(a)
            z[t] = tap.take()
            s = plan_env._state
            st[t] = tri_state_vec(s.robot_state) if robot_kind == "tri" else dd_state_vec(s.robot_state)
            assert (obs.pose == st[t, :3]).all()
            rew[t], done[t], coll[t] = r, d, s.robot_collided
            tidx[t] = s.reward_provider_state.target_idx
            mind[t] = s.reward_provider_state.min_spat_dist_so_far
            tm[t] = obs.time
            obs_path_len[t] = len(obs.path)
    rec.update(actions=act, z=z, states=st, reward=rew, done=done, collided=coll, target_idx=tidx, min_dist=mind,
               time=tm, obs_path_len=obs_path_len)
    return rec


def gen_trajectories():
    from bc_gym_planning_env.envs.synth_turn_env import AisleTurnEnv, AisleTurnEnvParams, TurnParams
    from bc_gym_planning_env.envs.base.params import EnvParams
    # RandomMiniEnv (metric config): 24 seeds x 400 steps, tricycle, PlanEnv's default noise
    for seed in range(24):
        env = make_mini_env(seed)
        rec = record_trajectory(env, env._env, 400, action_seed=1000 + seed, noise_seed=2000 + seed)
        save("g8_traj_mini_%02d.npz" % seed, **rec)
    # noise off (set_noise_parameters(None), tricycle_model.py:575-579), runs to the timeout
    for seed in (40, 41):
        env = make_mini_env(seed)
        env._env._robot.set_noise_parameters(None)
        rec = record_trajectory(env, env._env, 1250, action_seed=1000 + seed, noise_seed=2000 + seed)
        save("g8_traj_mini_nonoise_%02d.npz" % seed, **rec)
    # AisleTurnEnv variants: default res, and the C4 config (res 10/256, flips)
    variants = [("default", dict(), dict()),
                ("c4_00", dict(resolution=10. / 256), dict()),
                ("c4_10", dict(resolution=10. / 256), dict(flip_arnd_oy=True)),
                ("c4_01", dict(resolution=10. / 256), dict(flip_arnd_ox=True)),
                ("c4_11", dict(resolution=10. / 256), dict(flip_arnd_oy=True, flip_arnd_ox=True))]
    for i, (tag, ekw, tkw) in enumerate(variants):
        env = AisleTurnEnv(AisleTurnEnvParams(env_params=EnvParams(**ekw), turn_params=TurnParams(**tkw)))
        rec = record_trajectory(env, env, 500, action_seed=3000 + i, noise_seed=4000 + i)
        save("g8_traj_aisle_%s.npz" % tag, **rec)


def gen_diffdrive_trajectories():
    """C2 shape: PlanEnv hard-codes the tricycle, so the diff-drive robot is driven the way SURVEY 8(a9) says:
    _env_step(costmap, DiffDriveRobot, dt, action) (envs/base/env.py:442-461) + the reward provider, by hand, with
    the bookkeeping of PlanEnv._resolve_state_transition / step (env.py:334-398)."""
    from bc_gym_planning_env.envs.base.env import _env_step
    from bc_gym_planning_env.envs.base.action import Action
    from bc_gym_planning_env.envs.base.reward import ContinuousRewardProvider, RewardParams
    from bc_gym_planning_env.robot_models.differential_drive import DiffDriveRobot
    from bc_gym_planning_env.robot_models.robot_dimensions_examples import get_dimensions_example
    from bc_gym_planning_env.utilities.path_tools import refine_path
    for seed, toward_wall in ((0, False), (5, False), (3, True), (9, True)):
        env = make_mini_env(seed, resolution=5.5 / 64)
        costmap = env._env._state.costmap
        path = np.array(env._env._state.original_path)
        rng = np.random.RandomState(900 + seed)
        robot = DiffDriveRobot(dimensions=get_dimensions_example('industrial_diffdrive_v1'), noise_parameters=None)
        rp = RewardParams(spatial_precision=0.2, angular_precision=np.pi / 8)
        prov = ContinuousRewardProvider(params=rp)
        prov.set_state(ContinuousRewardProvider.generate_initial_state(path, rp))
        init = prov.get_state()
        robot.set_pose(*path[0])
        if toward_wall:  # start 1.4 m in front of a lethal cell, heading at it: collisions and rollbacks for sure
            ly, lx = np.nonzero(costmap.get_data() == 254)
            k = len(ly) // 2
            wx = costmap.get_origin()[0] + lx[k] * costmap.get_resolution()
            wy = costmap.get_origin()[1] + ly[k] * costmap.get_resolution()
            th0 = 0.3 + seed
            robot.set_pose(wx - 1.4 * np.cos(th0), wy - 1.4 * np.sin(th0), np.arctan2(np.sin(th0), np.cos(th0)))
        start_state = dd_state_vec(robot.get_state())

        class S(object):
            pass
        T = 500
        act = np.zeros((T, 2)); st = np.zeros((T, 7)); rew = np.zeros(T); done = np.zeros(T, np.uint8)
        coll = np.zeros(T, np.uint8); tidx = np.zeros(T, np.int32); mind = np.zeros(T)
        collided = False
        for t in range(T):
            # (v, w): mostly forward, float32-representable values handed over as float64 (see gen_robot_steps)
            a = np.array([rng.uniform(0.1, 0.6), rng.uniform(-1.2, 1.2) * (0.15 if toward_wall else 1.0)])
            a = a.astype(np.float32).astype(np.float64)
            act[t] = a
            hit = _env_step(costmap, robot, 0.05, Action(command=a))
            collided = collided or hit
            s = S(); s.pose = robot.get_pose()
            rew[t] = prov.reward(s)
            ps = prov.get_state()
            st[t] = dd_state_vec(robot.get_state())
            done[t] = ps.done() or (t + 1 >= 1200) or collided
            coll[t], tidx[t], mind[t] = collided, ps.target_idx, ps.min_spat_dist_so_far
        save("g8dd_traj_mini64_%02d.npz" % seed, costmap=costmap.get_data().copy(), origin=np.array(costmap.get_origin()),
             resolution=np.float64(costmap.get_resolution()), path=path, init_target_idx=np.int32(init.target_idx),
             init_min_dist=np.float64(init.min_spat_dist_so_far), start_state=start_state, actions=act, states=st, reward=rew, done=done,
             collided=coll, target_idx=tidx, min_dist=mind)
        print("   dd traj seed %d: collided at %s, final target %d/%d" % (seed, np.nonzero(coll)[0][:1], tidx[-1], len(path)))


def gen_kat_collision_table():
    """Inputs of the reference KAT test_costmap_utils.py:251-314 (its 20 expected verdicts are written in the
    test itself and restated in tests/test_oracle_kat.py); here we only store the costmap the reference's own
    map-building code produces through the harness' cv2.line stand-in."""
    from bc_gym_planning_env.utilities.costmap_2d import CostMap2D
    from bc_gym_planning_env.utilities.map_drawing_utils import add_wall_to_static_map
    costmap = CostMap2D.create_empty((10, 6), 0.05, (-1, -3))
    add_wall_to_static_map(costmap, (3.9, -4.), (3.9, -1 + 1.5))
    add_wall_to_static_map(costmap, (1.5, -4.), (1.5, -1 + 1.5))
    add_wall_to_static_map(costmap, (5., -4.), (5. + 1, -1 + 4.5))
    save("kat_collision_map.npz", costmap=costmap.get_data(), origin=np.array(costmap.get_origin()),
         resolution=costmap.get_resolution())


def gen_mini_geometry():
    """G9: the worlds RandomMiniEnv(seed=s) goes through on construction and on three successive reset() calls."""
    from bc_gym_planning_env.envs.mini_env import RandomMiniEnv
    seeds, episodes = list(range(12)), 4
    worlds = np.zeros((len(seeds), episodes, 14))
    maps, paths, lens, init = [], [], [], []
    for si, s in enumerate(seeds):
        env = RandomMiniEnv(seed=s)
        for e in range(episodes):
            if e:
                env.reset()
            cfg = env._env._config
            worlds[si, e] = np.concatenate([cfg.start_pos.as_np(), cfg.end_pos.as_np(), cfg.obstacle_a.as_np(),
                                            cfg.obstacle_o.as_np(), cfg.obstacle_b.as_np(), [cfg.h, cfg.w]])
            st = env._env.get_state()
            data = st.costmap.get_data()
            assert set(np.unique(data)) <= {0, 254}
            maps.append(np.packbits(data == 254, axis=1))
            paths.append(st.original_path)
            lens.append(len(st.original_path))
            init.append([st.pose[0], st.pose[1], st.pose[2], st.reward_provider_state.min_spat_dist_so_far,
                         st.reward_provider_state.target_idx])
    save("g9_mini_geometry.npz", seeds=np.array(seeds), worlds=worlds, maps=np.stack(maps),
         map_shape=np.array(data.shape), origin=np.array(st.costmap.get_origin()),
         resolution=st.costmap.get_resolution(), paths=np.concatenate(paths), lens=np.array(lens),
         init=np.array(init))


def record_delayed(env, plan_env, n_steps, action_seed, noise_seed, speed=1.0):
    """Like record_trajectory, for envs with delays / the pure-pursuit provider: the robot's true state and what the
    State / Observation expose are recorded separately."""
    from bc_gym_planning_env.robot_models import differential_drive as dd
    from bc_gym_planning_env.envs.base import spaces
    spaces.SPACE_LOCAL_RANDOM_STATE.seed(action_seed)
    state0 = plan_env.get_state()
    rec = dict(
        costmap=state0.costmap.get_data().copy(), origin=np.array(state0.costmap.get_origin()),
        resolution=np.float64(state0.costmap.get_resolution()),
        path=np.array(state0.reward_provider_state.path), init_target_idx=np.int32(state0.reward_provider_state.target_idx),
        init_min_dist=np.float64(state0.reward_provider_state.min_spat_dist_so_far),
        init_state=tri_state_vec(state0.robot_state),
    )
    act, z, true_st, seen_pose, seen_st = [], [], [], [], []
    rew, done, coll, tidx, mind, plen = [], [], [], [], [], []
    with SlotTap(dd, noise_seed) as tap:
        for t in range(n_steps):
            a = env.action_space.sample()
            a = type(a)(command=np.array([a.command[0] * speed, a.command[1]]))
            act.append(np.asarray(a.command, dtype=np.float64))
            obs, r, d, _ = env.step(a)
            z.append(tap.take())
            s = plan_env._state
            true_st.append(tri_state_vec(plan_env._robot.get_state()))
            seen_pose.append(np.array(obs.pose, dtype=np.float64))
            seen_st.append(tri_state_vec(obs.robot_state))
            assert (np.asarray(s.pose) == obs.pose).all()
            rew.append(r); done.append(d); coll.append(s.robot_collided)
            tidx.append(s.reward_provider_state.target_idx)
            mind.append(s.reward_provider_state.min_spat_dist_so_far)
            plen.append(len(obs.path))
    rec.update(actions=np.stack(act), z=np.stack(z), true_states=np.stack(true_st), seen_pose=np.stack(seen_pose),
               seen_states=np.stack(seen_st), reward=np.array(rew, dtype=np.float64), done=np.array(done, dtype=np.uint8),
               collided=np.array(coll, dtype=np.uint8), target_idx=np.array(tidx, dtype=np.int32),
               min_dist=np.array(mind, dtype=np.float64), obs_path_len=np.array(plen, dtype=np.int32))
    return rec


def gen_delays_and_pure_pursuit():
    """G11: EnvParams delays > 0 (env.py:27-49, 363-398) and the pure-pursuit reward provider (reward.py:78-159, 291-371)."""
    from bc_gym_planning_env.envs.mini_env import RandomMiniEnv, RandomMiniEnvParams
    from bc_gym_planning_env.envs.synth_turn_env import AisleTurnEnv, AisleTurnEnvParams
    from bc_gym_planning_env.envs.base.params import EnvParams
    cases = [
        ("delay_p1s1", dict(pose_delay=1, state_delay=1), 2, 1.0),          # the runner scripts' setting
        ("delay_c2p3s1", dict(control_delay=2, pose_delay=3, state_delay=1), 4, 1.0),
        ("delay_c1_wall", dict(control_delay=1, pose_delay=2, state_delay=2), 9, 3.0),   # drives into the wall
    ]
    for tag, kw, seed, speed in cases:
        params = RandomMiniEnvParams(env_params=EnvParams(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2, **kw))
        env = RandomMiniEnv(params=params, rng=np.random.RandomState(seed), draw_new_turn_on_reset=False)
        rec = record_delayed(env, env._env, 300, 500 + seed, 600 + seed, speed)
        rec.update(control_delay=np.int32(kw.get("control_delay", 0)), pose_delay=np.int32(kw.get("pose_delay", 0)),
                   state_delay=np.int32(kw.get("state_delay", 0)), pure_pursuit=np.int32(0))
        save("g11_traj_%s.npz" % tag, **rec)
    for tag, kw, speed in (("pp", dict(), 2.0), ("pp_delay", dict(pose_delay=1, control_delay=1), 3.0)):
        ep = EnvParams(reward_provider_name='continuous_reward_pure_pursuit', **kw)
        env = AisleTurnEnv(AisleTurnEnvParams(env_params=ep))
        rec = record_delayed(env, env, 400, 700, 701, speed)
        rec.update(control_delay=np.int32(kw.get("control_delay", 0)), pose_delay=np.int32(kw.get("pose_delay", 0)),
                   state_delay=np.int32(kw.get("state_delay", 0)), pure_pursuit=np.int32(1))
        save("g11_traj_%s.npz" % tag, **rec)
        print("   %s: reward range %.3f .. %.3f, collided %d, done at %s" % (
            tag, rec["reward"].min(), rec["reward"].max(), rec["collided"].sum(), np.argmax(rec["done"])))


def gen_serialized_records():
    """G13: genuine PlanEnv.serialize() records (env.py:251-261, State.serialize :163-176) taken mid-episode -- with delay
    queues filled and with both reward providers -- followed by the next 100 steps of the SAME live env.  The record is
    what a batch is rebuilt from (BatchedPlanEnv.deserialize); the continuation is what it must then reproduce."""
    import copy
    from oracle import records
    from bc_gym_planning_env.envs.mini_env import RandomMiniEnv, RandomMiniEnvParams
    from bc_gym_planning_env.envs.synth_turn_env import AisleTurnEnv, AisleTurnEnvParams
    from bc_gym_planning_env.envs.base.params import EnvParams
    from bc_gym_planning_env.envs.base import spaces

    def mini(seed, **kw):
        params = RandomMiniEnvParams(env_params=EnvParams(goal_ang_dist=np.pi / 8., goal_spat_dist=0.2, **kw))
        env = RandomMiniEnv(params=params, rng=np.random.RandomState(seed), draw_new_turn_on_reset=False)
        return env, env._env

    def aisle(**kw):
        env = AisleTurnEnv(AisleTurnEnvParams(env_params=EnvParams(reward_provider_name='continuous_reward_pure_pursuit', **kw)))
        return env, env

    cases = [("plain", lambda: mini(3), 37, 1.0),
             ("delay_p1s1", lambda: mini(2, pose_delay=1, state_delay=1), 23, 1.0),
             ("delay_c2p3s1", lambda: mini(4, control_delay=2, pose_delay=3, state_delay=1), 41, 1.0),
             ("delay_c2p3", lambda: mini(5, control_delay=2, pose_delay=3), 29, 1.0),   # no state delay: State holds the robot's true state
             ("delay_filling", lambda: mini(6, control_delay=3, pose_delay=2, state_delay=3), 2, 1.0),   # queues not full yet
             ("delay_fresh", lambda: mini(7, control_delay=1, pose_delay=1, state_delay=1), 0, 1.0),    # empty queues
             ("pp", lambda: aisle(), 60, 2.0),
             ("pp_delay", lambda: aisle(pose_delay=1, control_delay=1), 35, 2.0)]
    for tag, make, before, speed in cases:
        env, plan_env = make()
        spaces.SPACE_LOCAL_RANDOM_STATE.seed(900)
        np.random.seed(901)
        for _ in range(before):
            a = env.action_space.sample()
            env.step(type(a)(command=np.array([a.command[0] * speed, a.command[1]], dtype=np.float32)))
        skeleton, arrays = records.pack(copy.deepcopy(plan_env.serialize()))
        rec = record_delayed(env, plan_env, 100, 910, 911, speed)
        cont = dict(("cont_" + k, v) for k, v in rec.items() if k in (
            "actions", "z", "true_states", "seen_pose", "seen_states", "reward", "done", "collided", "target_idx",
            "min_dist", "obs_path_len"))
        save("g13_serialized_%s.npz" % tag, record=np.array(skeleton), steps_before=np.int32(before), **dict(arrays, **cont))


def gen_colored_ego():
    """G12: ColoredEgoCostmapRandomAisleTurnEnv observations (envs/synth_turn_env.py:376-451): 133 x 133 egocentric
    costmap + the 5-vector (unit goal direction, v, w, wheel angle), along a sampled-action trajectory."""
    from bc_gym_planning_env.envs.synth_turn_env import ColoredEgoCostmapRandomAisleTurnEnv
    from bc_gym_planning_env.envs.base import spaces
    from bc_gym_planning_env.robot_models import differential_drive as dd
    env = ColoredEgoCostmapRandomAisleTurnEnv()
    env.seed(3)
    env.reset()
    plan_env = env._env
    spaces.SPACE_LOCAL_RANDOM_STATE.seed(78)
    st0 = plan_env.get_state()
    imgs, vecs, states = [], [], []
    with SlotTap(dd, 6):
        for t in range(120):
            a = env.action_space.sample()
            a = type(a)(command=np.array([a.command[0] * 2.5, a.command[1]]))
            obs, _r, done, _ = env.step(a)
            img = obs['environment'][:, :, 0]
            assert set(np.unique(img)) <= {0, 254}
            imgs.append(np.packbits(img == 254, axis=1))
            vecs.append(obs['goal'][:, 0].copy())
            states.append(tri_state_vec(plan_env._robot.get_state()))
            if done:
                break
    save("g12_colored_ego.npz", costmap=st0.costmap.get_data().copy(), origin=np.array(st0.costmap.get_origin()),
         resolution=np.float64(st0.costmap.get_resolution()), path=np.array(st0.reward_provider_state.path),
         images=np.stack(imgs), image_shape=np.array(img.shape), goal=np.stack(vecs), states=np.stack(states),
         window_origin=np.array([-0.5, -2.0]), window_size=np.array([4.0, 4.0]))


def gen_egocentric():
    """G10: EgocentricCostmap(env).step observations (envs/egocentric.py:102-160) along sampled-action trajectories:
    the 133 x 117 egocentric costmap and the goal_n_state vector, with the state they were computed from."""
    from bc_gym_planning_env.envs.egocentric import EgocentricCostmap
    from bc_gym_planning_env.envs.synth_turn_env import AisleTurnEnv, AisleTurnEnvParams
    from bc_gym_planning_env.envs.base import spaces
    from bc_gym_planning_env.robot_models import differential_drive as dd
    cases = [("mini_00", lambda: make_mini_env(0), 120), ("mini_05", lambda: make_mini_env(5), 120),
             ("aisle", lambda: AisleTurnEnv(AisleTurnEnvParams()), 150)]
    for tag, make, steps in cases:
        base = make()
        plan_env = base._env if hasattr(base, "_env") else base
        env = EgocentricCostmap(base)
        spaces.SPACE_LOCAL_RANDOM_STATE.seed(77)
        st0 = plan_env.get_state()
        imgs, vecs, states, tidx = [], [], [], []
        with SlotTap(dd, 5):
            for t in range(steps):
                a = env.action_space.sample()
                a = type(a)(command=np.array([a.command[0] * 2.5, a.command[1]]))    # drive faster: see more of the map
                obs, _r, done, _ = env.step(a)
                s = plan_env._state
                img = obs['env'][:, :, 0]
                assert set(np.unique(img)) <= {0, 254}
                imgs.append(np.packbits(img == 254, axis=1))
                vecs.append(obs['goal_n_state'][:, 0].copy())
                states.append(tri_state_vec(s.robot_state))
                tidx.append(s.reward_provider_state.target_idx)
                if done:
                    break
        save("g10_ego_%s.npz" % tag, costmap=st0.costmap.get_data().copy(), origin=np.array(st0.costmap.get_origin()),
             resolution=np.float64(st0.costmap.get_resolution()), path=np.array(st0.reward_provider_state.path),
             images=np.stack(imgs), image_shape=np.array(img.shape), goal_n_state=np.stack(vecs),
             states=np.stack(states), target_idx=np.array(tidx, dtype=np.int32),
             window_origin=np.array([-0.5, -2.0]), window_size=np.array([3.5, 4.0]))


def main():
    os.makedirs(OUT, exist_ok=True)
    O.build()
    H.load()
    gen_robot_steps()
    gen_scalar_utils()
    gen_footprints()
    gen_pose_collides()
    gen_reward()
    gen_kat_collision_table()
    gen_mini_geometry()
    gen_egocentric()
    gen_colored_ego()
    gen_delays_and_pure_pursuit()
    gen_serialized_records()
    gen_diffdrive_trajectories()
    gen_trajectories()


if __name__ == "__main__":
    main()
