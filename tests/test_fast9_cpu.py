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
