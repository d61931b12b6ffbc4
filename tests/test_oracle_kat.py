"""Known-answer tests the reference itself holds at the cv2 boundary, replayed against the oracle's OpenCV
restatement (fill + 1-px line).  These are the only pins of the polygon fill."""
import os

import numpy as np


def _rect():
    # _rectangular_footprint(), utilities/test_costmap_utils.py:242-248 (same polygon as test_path_tools.py:465-468)
    return np.array([[-0.77, -0.385], [-0.77, 0.385], [0.67, 0.385], [0.67, -0.385]])


def test_compute_robot_area_493(oracle):
    # utilities/test_path_tools.py:465-468: compute_robot_area(0.05, rect) == 493
    mask = oracle.pixel_footprint(0., _rect(), 0.05)
    assert np.count_nonzero(mask) == 493
    assert mask.shape == (17, 33)


def test_is_robot_colliding_table(oracle, golden_dir):
    # utilities/test_costmap_utils.py:251-314.  The costmap fixture was produced by the reference's own map code
    # (CostMap2D.create_empty + add_wall_to_static_map) drawing through the harness' cv2.line stand-in.
    g = np.load(os.path.join(golden_dir, "kat_collision_map.npz"))
    costmap, origin, res = g["costmap"], g["origin"], float(g["resolution"])
    assert costmap.shape == (120, 200)
    poses = [(0., 0., 0.2), (1., 0., 0.2), (2., 0., 0.2), (3., 0., 0.2), (4., 0., 0.2), (5., 0., 0.2), (6., 0., 0.2)]
    poses += [(x, 1.2, np.pi / 2 + 0.4) for x in (0., 1., 2., 3., 4., 5., 6.)]
    poses += [(0., -3, 0.2), (1., -3, 0.2), (2., -3, 0.2), (0., -3.2, 0.2), (1., -3.2, 0.2), (2., -3.2, 0.2)]
    expected = [False, True, True, False, True, True, True,
                False, False, False, False, True, True, True,
                False, True, True, False, False, False]
    for pose, exp in zip(poses, expected):
        # is_robot_colliding first rejects robot centres outside the map (costmap_utils.py:140-164)
        px, py = oracle.world_to_pixel(np.array(pose[:2]), origin, res)
        inside = 0 <= px < costmap.shape[1] and 0 <= py < costmap.shape[0]
        got = inside and oracle.pose_collides(pose[0], pose[1], pose[2], _rect(), costmap, origin, res)
        assert got == exp, pose


def test_wall_line_matches_oracle_line(oracle, golden_dir):
    """The oracle's line drawing re-creates the KAT map on its own (vertical walls x=98, x=50; slanted wall)."""
    g = np.load(os.path.join(golden_dir, "kat_collision_map.npz"))
    res, origin = float(g["resolution"]), g["origin"]
    img = np.zeros((120, 200), dtype=np.uint8)
    for p0, p1 in (((3.9, -4.), (3.9, .5)), ((1.5, -4.), (1.5, .5)), ((5., -4.), (6., 3.5))):
        a = oracle.world_to_pixel(np.array(p0), origin, res)
        b = oracle.world_to_pixel(np.array(p1), origin, res)
        oracle.line(img, a, b, 254)
    np.testing.assert_array_equal(img, g["costmap"])


def test_polyline_diagonal_pixels(oracle):
    # utilities/test_map_drawing_utils.py:94-112 pins a 1-px 8-connected diagonal: (i, i) for every step
    img = np.zeros((12, 12), dtype=np.uint8)
    oracle.line(img, (1, 1), (9, 9), 255)
    ys, xs = np.nonzero(img)
    assert list(zip(xs, ys)) == [(i, i) for i in range(1, 10)]


def test_fill_matches_scanline_definition(oracle):
    """Self-consistency: filled mask == outline pixels | interior spans, for both robot footprints."""
    rng = np.random.RandomState(5)
    for fp in (oracle.TRICYCLE_FOOTPRINT, oracle.DIFFDRIVE_FOOTPRINT):
        for res in (0.03, 5.5 / 64, 10. / 256):
            for a in rng.uniform(-np.pi, np.pi, 50):
                m = oracle.pixel_footprint(a, fp, res)
                assert m.any()
                # every row of a filled simple polygon is one contiguous run for these near-convex footprints
                for row in m:
                    nz = np.flatnonzero(row)
                    if len(nz):
                        assert nz[-1] - nz[0] + 1 == len(nz)


# ---- egocentric costmap: the reference's own known answers (utilities/test_costmap_utils.py:38-207) ------------
def _marks(img):
    r, c = np.where(img == 254)
    return list(zip(c.tolist(), r.tolist()))   # (x, y) like assert_mark_at


def test_kat_extract_egocentric_costmap(oracle):
    O = oracle
    data = np.zeros((100, 100), dtype=np.uint8)
    data[10, 20] = 254
    org, res = np.array([0.0, 0.0]), 0.05
    # dummy cut / pure shift of the robot: the data stay where they are (test_costmap_utils.py:45-63)
    assert _marks(O.extract_egocentric(data, org, res, (0., 0., 0.))) == [(20, 10)]
    assert _marks(O.extract_egocentric(data, org, res, (0.2, 0.2, 0.0))) == [(20, 10)]
    # rotated so that the mark is almost in front of the robot (:66-73)
    img = O.extract_egocentric(data, org, res, (0.0, 0.0, np.pi / 6 - 0.05))
    assert img.shape == (100, 100) and img[0, 22] == 254
    # robot in the centre, turned by -pi/2, with and without explicit window (:76-93)
    assert _marks(O.extract_egocentric(data, org, res, (2.5, 2.5, -np.pi / 2.))) == [(90, 20)]
    img = O.extract_egocentric(data, org, res, (2.5, 2.5, -np.pi / 2), np.array([-2.5, -2.5]), np.array((5., 5.)))
    assert img.shape == (100, 100) and _marks(img) == [(90, 20)]
    img = O.extract_egocentric(data, org, res, (2.5, 2.5, -np.pi / 2), np.array([-2.5, -2.5]), (4.6, 4.9))
    assert img.shape == (98, 92) and _marks(img) == [(90, 20)]
    # shift, shift and cut (:107-129)
    img = O.extract_egocentric(data, org, res, (2.5, 2.5, 0.0), np.array([-5., -4.]), (5, 5))
    assert img.shape == (100, 100) and _marks(img) == [(70, 40)]
    img = O.extract_egocentric(data, org, res, (2.5, 2.5, 0.0), np.array([-5., -4.]), (4, 4))
    assert img.shape == (80, 80) and _marks(img) == [(70, 40)]
    # rotate, shift and cut / expand (:132-155)
    img = O.extract_egocentric(data, org, res, (1.5, 1.5, -np.pi / 4), np.array([-2., -2.]), (4, 4))
    assert img.shape == (80, 80) and _marks(img) == [(47, 19)]
    img = O.extract_egocentric(data, org, res, (1., 1.5, -np.pi / 4), np.array([-3., -3.]), (7, 6))
    assert img.shape == (120, 140) and _marks(img) == [(74, 46)]
    # non-zero map origin (:161-176)
    img = O.extract_egocentric(data, np.array([1.0, 2.0]), res, (3.5, 3.5, -np.pi / 4), np.array([-2., -2.]), (4, 4))
    assert img.shape == (80, 80) and _marks(img) == [(33, 5)]


def test_kat_extract_egocentric_binary_block(oracle):
    """a 2x2 lethal block seen from a robot turned by -pi/4 is exactly these five pixels (:179-195)"""
    data = np.zeros((100, 100), dtype=np.uint8)
    data[30:32, 30:32] = 254
    img = oracle.extract_egocentric(data, np.array([0., 0.]), 0.05, (1.5, 1.5, -np.pi / 4), np.array([-2.0, -2.0]),
                                    (4., 4))
    r, c = np.where(img != 0)
    assert r.tolist() == [40, 41, 41, 41, 42] and c.tolist() == [40, 39, 40, 41, 40]
    assert (img[r, c] == 254).all()


def test_kat_rotate_costmap(oracle):
    """rotate_costmap by -pi/4 moves the mark at (30, 30) of a 80 x 100 map to (29, 47) (:198-207)"""
    data = np.zeros((80, 100), dtype=np.uint8)   # create_empty((5, 4), 0.05): 100 columns, 80 rows
    data[30, 30] = 254
    assert _marks(oracle.rotate_costmap(data, -np.pi / 4)) == [(29, 47)]
    assert (oracle.rotate_costmap(data, 0.0) == data).all()
