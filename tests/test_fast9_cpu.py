"""The FAST-9/16 oracle against a brute-force reading of the segment test (reference call site:
cv::ORB::create(..., fastThreshold 20) at src/imagefeatures.cpp:73-87; OpenCV absent: parity unpinned)."""
import numpy as np

from oracle import pyoracle as orc

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _is_corner(img, x, y, t):
    v = int(img[y, x])
    ring = [int(img[y + dy, x + dx]) for dx, dy in RING]
    for start in range(16):
        arc = [ring[(start + k) % 16] for k in range(9)]
        if all(q > v + t for q in arc) or all(q < v - t for q in arc):
            return True
    return False


def test_strength_is_the_largest_threshold_plus_one():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (30, 34), dtype=np.uint8)
    for y in range(3, 27):
        for x in range(3, 31):
            s = orc.fast9_strength(img, x, y)
            assert (s > 0) == _is_corner(img, x, y, 0)
            if s > 0:
                assert _is_corner(img, x, y, s - 1) and not _is_corner(img, x, y, s)
    assert orc.fast9_strength(img, 2, 10) == 0 and orc.fast9_strength(img, 10, 27) == 0       # border


def test_detector_without_suppression_is_the_corner_set():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (24, 40), dtype=np.uint8)
    for t in (0, 20, 70):
        k = orc.fast9_detect(img, t, nonmax=False)
        want = {(x, y) for y in range(3, 21) for x in range(3, 37) if _is_corner(img, x, y, t)}
        assert {(int(a), int(b)) for a, b in zip(k["x"], k["y"])} == want
        assert all(_is_corner(img, int(a), int(b), int(c)) for a, b, c in zip(k["x"], k["y"], k["score"]))
        assert np.all(np.diff(k["score"]) <= 0)                      # sorted by score


def test_suppression_keeps_strict_local_maxima_of_strength():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 40), dtype=np.uint8)
    t = 15
    st = np.zeros((40, 40), int)
    for y in range(3, 37):
        for x in range(3, 37):
            s = orc.fast9_strength(img, x, y)
            st[y, x] = s if s > t else 0
    want = {(x, y) for y in range(3, 37) for x in range(3, 37)
            if st[y, x] and all(st[y, x] > st[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dx or dy)}
    k = orc.fast9_detect(img, t, nonmax=True)
    assert {(int(a), int(b)) for a, b in zip(k["x"], k["y"])} == want


def test_hand_checkable_square():
    img = np.full((40, 40), 50, np.uint8)
    img[10:30, 10:30] = 200
    k = orc.fast9_detect(img, 20, nonmax=False)
    assert len(k) == 24 and set(k["score"].tolist()) == {149.0}          # six pixels around each corner of the square
    assert len(orc.fast9_detect(img, 20, nonmax=True)) == 0             # equal strengths suppress each other
    img[10, 10] = 210
    k = orc.fast9_detect(img, 20, nonmax=True)
    assert [(int(a), int(b), float(c)) for a, b, c in k] == [(10, 10, 159.0)]


def test_orb_level_plan_matches_published_series():
    """Level sizes and feature budgets of cv::ORB's pyramid for the reference's parameters (10 000 features, 1.2, 8 levels
    on 1920 x 1080): the oracle's restatement and the library's host function agree, budgets sum to nfeatures."""
    from visualslam_amd import api
    lw, lh, lf, ls = orc.orb_level_plan(1920, 1080)
    assert lw.tolist() == [1920, 1600, 1333, 1111, 926, 772, 643, 536] and lh.tolist() == [1080, 900, 750, 625, 521, 434, 362, 301]
    assert lf.sum() == 10000 and lf[0] == 2172
    glw, glh, glf, gls = api.orb_level_plan(api.orb_params(), 1920, 1080)
    assert np.array_equal(glw, lw) and np.array_equal(glh, lh) and np.array_equal(glf, lf) and np.array_equal(gls, ls)
