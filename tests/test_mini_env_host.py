"""Host side of the geometry pool (bc_gym_planning_env_amd/mini_env.py) against the reference's own RandomMiniEnv
worlds (tests/golden/g9_mini_geometry.npz, made by oracle/gen_golden.py from the genuine reference) and against the
oracle's cv2.line restatement.  CPU only: the batched GPU acceptance test is replaced by the oracle's pose_collides."""
import os

import numpy as np
import pytest

import oracle as O
from bc_gym_planning_env_amd import host_init, mini_env
from bc_gym_planning_env_amd.api import EnvParams

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


class OracleCollider(object):
    capacity = 64

    def __init__(self, env_params):
        self.verts = O.TRICYCLE_FOOTPRINT if env_params.robot_name == 'industrial_tricycle_v1' else O.DIFFDRIVE_FOOTPRINT

    def __call__(self, costmaps, paths, batch=None):
        return np.array([[O.pose_collides(p[0], p[1], p[2], self.verts, c.get_data(), c.get_origin(), c.get_resolution())
                          for p in path] for c, path in zip(costmaps, paths)], dtype=bool)


def test_draw_line_matches_oracle_line():
    rng = np.random.RandomState(5)
    for trial in range(3000):
        rows, cols = rng.randint(1, 60), rng.randint(1, 60)
        span = 40 if trial % 3 else 900   # every third segment starts / ends far outside the image
        p0 = rng.randint(-span, span + cols, 2)
        p1 = rng.randint(-span, span + cols, 2)
        if trial % 7 == 0:
            p1 = p0.copy()
        if trial % 11 == 0:
            p1[0] = p0[0]
        if trial % 13 == 0:
            p1[1] = p0[1]
        a = np.zeros((rows, cols), dtype=np.uint8)
        b = np.zeros((rows, cols), dtype=np.uint8)
        mini_env.draw_line(a, p0, p1, 254)
        O.line(b, p0, p1, 254)
        assert (a == b).all(), (trial, rows, cols, p0, p1)


def test_pool_reproduces_reference_worlds():
    g = np.load(os.path.join(GOLDEN, "g9_mini_geometry.npz"))
    seeds, episodes = [int(s) for s in g["seeds"]], g["worlds"].shape[1]
    params = mini_env.default_random_mini_env_params()
    pool = mini_env.sample_pool(params, seeds, episodes, collider=OracleCollider(params.env_params))
    assert len(pool) == len(seeds) * episodes
    rows, cols = [int(v) for v in g["map_shape"]]
    offs = np.concatenate([[0], np.cumsum(g["lens"])])
    rp = params.env_params.reward_provider_params
    for k, w in enumerate(pool.worlds):
        ref = g["worlds"][k // episodes, k % episodes]
        mine = np.concatenate([w.start_pos, w.end_pos, w.obstacle_a, w.obstacle_o, w.obstacle_b, [w.h, w.w]])
        assert (mine == ref).all(), (k, mine - ref)                      # bit-exact: same draws, same arithmetic
        cm = pool.costmaps[k]
        assert cm.get_data().shape == (rows, cols)
        assert (cm.get_origin() == g["origin"]).all() and cm.get_resolution() == float(g["resolution"])
        want = np.unpackbits(g["maps"][k], axis=1)[:, :cols].astype(bool)
        assert ((cm.get_data() == 254) == want).all() and set(np.unique(cm.get_data())) <= {0, 254}
        path = host_init.refine_path(pool.paths[k], params.env_params.path_delta)
        assert (path == g["paths"][offs[k]:offs[k + 1]]).all()
        md, ti = host_init.initial_reward_state(path, rp)
        assert (path[0] == g["init"][k, :3]).all() and md == g["init"][k, 3] and ti == int(g["init"][k, 4])
    # chains wrap around
    assert pool.next_geom.tolist()[:episodes] == [1, 2, 3, 0][:episodes]
    assert pool.next_geom[episodes] == episodes + 1


def test_chain_is_prefix_stable():
    """a longer chain starts with the shorter one (the pool is the RNG stream of RandomMiniEnv(seed))"""
    params = mini_env.default_random_mini_env_params()
    col = OracleCollider(params.env_params)
    a = mini_env.sample_pool(params, [3], 2, collider=col)
    b = mini_env.sample_pool(params, [3], 5, collider=col)
    for wa, wb in zip(a.worlds, b.worlds):
        assert (wa.start_pos == wb.start_pos).all() and (wa.obstacle_o == wb.obstacle_o).all()


def test_thick_walls_are_refused():
    params = mini_env.RandomMiniEnvParams(env_params=EnvParams(resolution=0.02))
    m = mini_env.draw_candidate(params, np.random.RandomState(0))
    with pytest.raises(NotImplementedError):
        mini_env.prepare_map_and_path(m)
