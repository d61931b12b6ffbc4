"""The CPU oracle (oracle/bcp_oracle.c) against golden vectors captured from the genuine reference
(oracle/gen_golden.py).  Bit-exact unless stated.  Runs without a GPU and without /root/reference."""
import glob
import os

import numpy as np
import pytest


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_g1_tricycle_step(oracle, golden_dir):
    g = load(golden_dir, "g1_tricycle_step.npz")
    p = oracle.make_params("tricycle", dt=float(g["dt"]))
    for st, cmd, exp in zip(g["state"], g["cmd"], g["out"]):
        out, err, drawn = oracle.robot_step(p, st, cmd)
        assert err == 0 and drawn == 0
        np.testing.assert_array_equal(out, exp)


@pytest.mark.parametrize("name,dyn,pid", [("kin_pid", 0, 1), ("kin_nopid", 0, 0), ("dyn_nopid", 1, 0)])
def test_g1b_tricycle_variants(oracle, golden_dir, name, dyn, pid):
    g = load(golden_dir, "g1b_tricycle_variants.npz")
    p = oracle.make_params("tricycle", dt=float(g["dt"]), dynamic_model=dyn, model_front_column_pid=pid)
    for st, cmd, exp in zip(g["state"], g["cmd"], g[name]):
        out, err, _ = oracle.robot_step(p, st, cmd)
        np.testing.assert_array_equal(out, exp)


def test_g2_tricycle_step_noise(oracle, golden_dir):
    g = load(golden_dir, "g2_tricycle_step_noise.npz")
    n_drawn = 0
    for st, cmd, exp, z, ai in zip(g["state"], g["cmd"], g["out"], g["z"], g["alpha_idx"]):
        p = oracle.make_params("tricycle", dt=float(g["dt"]), noise=g["alphas"][ai])
        zz = np.where(np.isnan(z), 1e300, z)  # an unconsumed slot must not be read
        out, err, drawn = oracle.robot_step(p, st, cmd, zz)
        assert drawn == sum(1 << k for k in range(3) if not np.isnan(z[k]))
        n_drawn += bin(drawn).count("1")
        np.testing.assert_array_equal(out, exp)
    assert n_drawn > 2000


def test_g3_diffdrive_step(oracle, golden_dir):
    g = load(golden_dir, "g3_diffdrive_step.npz")
    p = oracle.make_params("diffdrive", dt=float(g["dt"]))
    for st, cmd, exp in zip(g["state"], g["cmd"], g["out"]):
        out, err, _ = oracle.robot_step(p, st, cmd)
        np.testing.assert_array_equal(out, exp)


def test_g4_scalar_utils(oracle, golden_dir):
    g = load(golden_dir, "g4_scalar_utils.npz")
    np.testing.assert_array_equal(oracle.normalize_angle(g["na_in"]), g["na_out"])
    np.testing.assert_array_equal(oracle.normalize_angle(g["da_a"] - g["da_b"]), g["da_out"])
    i = 0
    while "w2p%d_xy" % i in g:
        out = oracle.world_to_pixel(g["w2p%d_xy" % i], g["w2p%d_origin" % i], float(g["w2p%d_res" % i]))
        np.testing.assert_array_equal(out, g["w2p%d_out" % i])
        i += 1
    assert i == 7
    for p0, p1, exp in zip(g["pv_p0"], g["pv_p1"], g["pv_out"]):
        v, w, err = oracle.path_velocity(p0, p1, float(g["pv_dt"]))
        assert err == 0
        assert (v, w) == (exp[0], exp[1])


def test_g5_footprint_vertices(oracle, golden_dir):
    g = load(golden_dir, "g5_footprint_vertices.npz")
    for fname, fp in (("tri", oracle.TRICYCLE_FOOTPRINT), ("dd", oracle.DIFFDRIVE_FOOTPRINT)):
        for rname in ("r003", "r64", "r256"):
            key = "%s_%s" % (fname, rname)
            res = float(g[key + "_res"])
            for a, pts, shape in zip(g[key + "_angles"], g[key + "_pts"], g[key + "_shape"]):
                v, half = oracle.footprint_vertices(a, fp, res)
                np.testing.assert_array_equal(v, pts)
                assert (2 * half[1] + 1, 2 * half[0] + 1) == tuple(shape)


def test_g6_pose_collides(oracle, golden_dir):
    g = load(golden_dir, "g6_pose_collides.npz")
    for tag in ("mini0", "mini3", "mini64"):
        fp = oracle.TRICYCLE_FOOTPRINT if int(g[tag + "_robot"]) == 0 else oracle.DIFFDRIVE_FOOTPRINT
        got = [oracle.pose_collides(p[0], p[1], p[2], fp, g[tag + "_map"], g[tag + "_origin"], float(g[tag + "_res"]))
               for p in g[tag + "_poses"]]
        np.testing.assert_array_equal(np.array(got, dtype=np.uint8), g[tag + "_collides"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g7_reward(oracle, golden_dir, tag):
    g = load(golden_dir, "g7_reward.npz")
    sp, ap, mult = g[tag + "_params"]
    p = oracle.make_params("tricycle", spatial_precision=sp, angular_precision=ap, spatial_progress_multiplier=mult)
    path = g[tag + "_path"]
    md, ti = oracle.initial_reward_state(path, sp, ap)
    assert (md, ti) == (g[tag + "_init"][0], int(g[tag + "_init"][1]))
    for pose, er, emd, eti in zip(g[tag + "_poses"], g[tag + "_reward"], g[tag + "_min_dist"], g[tag + "_target_idx"]):
        r, md, ti = oracle.reward(p, pose, path, md, ti)
        assert (r, md, ti) == (er, emd, eti)


def _replay(oracle, g, noise):
    p = oracle.make_params("tricycle", noise=noise, spatial_precision=float(g["sp"]) if "sp" in g else None)
    return p


TRAJ = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "g8_traj_*.npz")))


@pytest.mark.parametrize("path", TRAJ, ids=[os.path.basename(p)[8:-4] for p in TRAJ])
def test_g8_trajectories(oracle, path):
    g = np.load(path)
    name = os.path.basename(path)
    mini = "mini" in name
    noise = None if "nonoise" in name else oracle.PLANENV_NOISE
    sp, ap = (0.2, np.pi / 8) if mini else (1.0, np.pi / 2)
    p = oracle.make_params("tricycle", noise=noise, spatial_precision=sp, angular_precision=ap)
    env = oracle.OracleBatch(p, 1, g["costmap"], g["origin"], float(g["resolution"]), g["path"])
    env.reset_from_paths()
    assert env.target_idx[0] == int(g["init_target_idx"]) and env.min_dist[0] == float(g["init_min_dist"])
    np.testing.assert_array_equal([env.st[f][0] for f in range(7)], g["init_state"])
    T = len(g["actions"])
    n_coll = 0
    for t in range(T):
        z = np.where(np.isnan(g["z"][t]), 1e300, g["z"][t])[None]
        env.step(g["actions"][t][None], z if noise is not None else None)
        got = np.array([env.st[f][0] for f in range(7)])
        np.testing.assert_array_equal(got, g["states"][t], err_msg="step %d" % t)
        assert env.reward[0] == g["reward"][t], t
        assert env.done[0] == g["done"][t], t
        assert env.collided[0] == g["collided"][t], t
        assert env.target_idx[0] == g["target_idx"][t], t
        assert env.min_dist[0] == g["min_dist"][t], t
        assert env.cur_time[0] == g["time"][t], t
        assert len(g["path"]) - env.target_idx[0] == g["obs_path_len"][t] or env.target_idx[0] >= len(g["path"])
        n_coll += int(env.collided_now[0])
    assert env.err[0] == 0


DD_TRAJ = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "g8dd_traj_*.npz")))


@pytest.mark.parametrize("path", DD_TRAJ, ids=[os.path.basename(p)[5:-4] for p in DD_TRAJ])
def test_g8_diffdrive_trajectories(oracle, path):
    """_env_step with the DiffDriveRobot + reward provider (C2 shape: 64x64 costmap, res 5.5/64, noise off)."""
    g = np.load(path)
    p = oracle.make_params("diffdrive", spatial_precision=0.2, angular_precision=np.pi / 8)
    env = oracle.OracleBatch(p, 1, g["costmap"], g["origin"], float(g["resolution"]), g["path"])
    env.reset_from_paths()
    assert env.target_idx[0] == int(g["init_target_idx"]) and env.min_dist[0] == float(g["init_min_dist"])
    for f in range(7):
        env.st[f][0] = g["start_state"][f]
    for t in range(len(g["actions"])):
        env.step(g["actions"][t][None])
        np.testing.assert_array_equal([env.st[f][0] for f in range(7)], g["states"][t], err_msg="step %d" % t)
        assert (env.reward[0], env.done[0], env.collided[0], env.target_idx[0], env.min_dist[0]) == \
            (g["reward"][t], g["done"][t], g["collided"][t], g["target_idx"][t], g["min_dist"][t]), t


# ---- G10: egocentric observation (envs/egocentric.py:102-160 through the genuine reference) ---------------------
@pytest.mark.parametrize("name", ["g10_ego_mini_00.npz", "g10_ego_mini_05.npz", "g10_ego_aisle.npz"])
def test_g10_egocentric_observation(oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    res, org = float(g["resolution"]), g["origin"]
    wo, ws = g["window_origin"], g["window_size"]
    rows, cols = [int(v) for v in g["image_shape"]]
    # CostMap2D.world_size() of the extracted map (costmap_2d.py:107-121)
    world = np.array([(wo[0] + res * cols) - wo[0], (wo[1] + res * rows) - wo[1]])
    for t in range(len(g["states"])):
        st = g["states"][t]
        img = oracle.extract_egocentric(g["costmap"], org, res, st[:3], wo, ws)
        want = np.unpackbits(g["images"][t], axis=1)[:, :cols].astype(bool)
        assert img.shape == (rows, cols)
        assert ((img == 254) == want).all() and set(np.unique(img)) <= {0, 254}, (name, t)
        # robot_state.to_numpy_array() = x, y, angle, v, w, wheel_angle (tricycle_model.py:267-271)
        rs = np.array([st[0], st[1], st[2], st[3], st[4], st[6]])
        vec = oracle.goal_n_state(st[:3], g["path"][g["target_idx"][t]:], world, rs)
        np.testing.assert_allclose(vec, g["goal_n_state"][t], rtol=0, atol=1e-6)


# ---- G11: delays > 0 and the pure-pursuit reward provider (env.py:27-49, 363-398; reward.py:78-159, 291-371) ----
G11 = ["g11_traj_delay_p1s1.npz", "g11_traj_delay_c2p3s1.npz", "g11_traj_delay_c1_wall.npz", "g11_traj_pp.npz",
       "g11_traj_pp_delay.npz"]


def oracle_env_for_g11(oracle, g, n=1):
    mini = int(g["pure_pursuit"]) == 0
    p = oracle.make_params("tricycle", noise=oracle.PLANENV_NOISE,
                           spatial_precision=0.2 if mini else 1.0, angular_precision=np.pi / 8 if mini else np.pi / 2,
                           reward_provider=int(g["pure_pursuit"]), control_delay=int(g["control_delay"]),
                           pose_delay=int(g["pose_delay"]), state_delay=int(g["state_delay"]))
    ref = oracle.OracleBatch(p, n, g["costmap"], g["origin"], float(g["resolution"]), g["path"])
    ref.reset_from_paths()
    return ref


@pytest.mark.parametrize("name", G11)
def test_g11_delays_and_pure_pursuit(oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    ref = oracle_env_for_g11(oracle, g)
    assert ref.target_idx[0] == int(g["init_target_idx"]) and ref.min_dist[0] == float(g["init_min_dist"])
    assert (np.array([a[0] for a in ref.st]) == g["init_state"]).all()
    for t in range(len(g["actions"])):
        z = np.where(np.isnan(g["z"][t]), 1e300, g["z"][t])[None]
        ref.step(g["actions"][t][None], z)
        assert (np.array([a[0] for a in ref.st]) == g["true_states"][t]).all(), (name, t)
        assert (ref.obs_pose[0] == g["seen_pose"][t]).all(), (name, t)
        assert (ref.obs_state[0] == g["seen_states"][t]).all(), (name, t)
        assert ref.reward[0] == g["reward"][t], (name, t, ref.reward[0], g["reward"][t])
        assert ref.done[0] == g["done"][t] and ref.collided[0] == g["collided"][t], (name, t)
        assert ref.target_idx[0] == g["target_idx"][t] and ref.min_dist[0] == g["min_dist"][t], (name, t)


def test_g12_colored_ego_observation(oracle, golden_dir):
    """ColoredEgoCostmapRandomAisleTurnEnv (envs/synth_turn_env.py:376-451): 133 x 133 view + unit goal direction"""
    g = np.load(os.path.join(golden_dir, "g12_colored_ego.npz"))
    res, org = float(g["resolution"]), g["origin"]
    wo, ws = g["window_origin"], g["window_size"]
    rows, cols = [int(v) for v in g["image_shape"]]
    world = np.array([(wo[0] + res * cols) - wo[0], (wo[1] + res * rows) - wo[1]])
    for t in range(len(g["states"])):
        st = g["states"][t]
        img = oracle.extract_egocentric(g["costmap"], org, res, st[:3], wo, ws)
        want = np.unpackbits(g["images"][t], axis=1)[:, :cols].astype(bool)
        assert img.shape == (rows, cols) and ((img == 254) == want).all(), t
        vec = oracle.goal_direction_state(st[:3], g["path"][-1], world, [st[3], st[4], st[6]])
        np.testing.assert_allclose(vec, g["goal"][t], rtol=0, atol=1e-12)
