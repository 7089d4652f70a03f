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


def test_orientation_follows_a_rotated_patch():
    """A bright half-plane through the keypoint: the intensity centroid points into the bright side, so the angle
    follows the direction the half-plane is turned to (to the accuracy of the discretised patch)."""
    from oracle import pyoracle as orc
    size = 96
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64) - size // 2
    for deg in (0, 45, 90, 200, 315):
        t = np.deg2rad(deg)
        img = np.where(xx * np.cos(t) + yy * np.sin(t) > 0, 220, 30).astype(np.uint8)
        kp = np.zeros(1, dtype=orc.FEATURE_DTYPE)
        kp["x"], kp["y"] = size // 2, size // 2
        ang, _ = orc.orb_describe_level(img, kp)
        d = abs(float(ang[0]) - deg)
        assert min(d, 360 - d) < 3.0


def test_polynomial_arctangent_stays_close_to_atan2():
    """cv::fastAtan2's polynomial: degrees in [0, 360), within 0.02 degrees of atan2 (OpenCV documents 0.3)."""
    import math
    from oracle import pyoracle as orc
    worst = 0.0
    for t in np.linspace(-math.pi, math.pi, 1441):
        y, x = 1000 * math.sin(t), 1000 * math.cos(t)
        a = orc.fast_atan2(y, x)
        assert 0.0 <= a <= 360.0
        ref = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - ref)
        worst = max(worst, min(d, 360.0 - d))
    assert worst < 0.02
    assert orc.fast_atan2(0.0, 0.0) == 0.0


def test_blur_is_a_unit_gain_gaussian():
    import scipy.ndimage as ndi
    from oracle import pyoracle as orc
    img = np.random.default_rng(12).integers(0, 256, (90, 140), dtype=np.uint8)
    got = orc.gaussian_blur7(img).astype(np.float64)
    want = ndi.gaussian_filter(img.astype(np.float64), 2.0, mode="mirror", truncate=1.5)     # 7 taps, reflect-101
    assert np.abs(got - want).max() <= 2.0 and abs((got - want).mean()) < 0.2
    flat = np.full((20, 30), 77, dtype=np.uint8)
    assert (orc.gaussian_blur7(flat) == 77).all()


def test_default_pattern_is_fixed_and_inside_the_patch():
    from oracle import pyoracle as orc
    from visualslam_amd import api
    pat = orc.orb_default_pattern()
    assert pat.shape == (1024,) and pat.min() >= -13 and pat.max() <= 13
    assert np.array_equal(pat, api.orb_default_pattern())                                   # library and oracle hold the same table
    t = pat.reshape(256, 4)
    assert not ((t[:, 0] == t[:, 2]) & (t[:, 1] == t[:, 3])).any()                         # no test compares a point with itself
    assert list(pat[:8]) == [2, -10, -9, 11, 13, -12, 12, -1]


def test_descriptor_of_a_rotated_image_is_stable():
    """rBRIEF steers the pattern by the patch orientation: a keypoint described on an image and on the same image
    turned by 90 degrees gets nearly the same bits (exactly, up to the arctangent's and the rounding's noise)."""
    from oracle import pyoracle as orc
    rng = np.random.default_rng(21)
    import scipy.ndimage as ndi
    img = np.clip(ndi.gaussian_filter(rng.normal(128, 60, (161, 161)), 3.0) * 4 - 384, 0, 255).astype(np.uint8)
    kp = np.zeros(1, dtype=orc.FEATURE_DTYPE)
    kp["x"], kp["y"] = 80, 80
    a0, d0 = orc.orb_describe_level(img, kp)
    a1, d1 = orc.orb_describe_level(np.ascontiguousarray(np.rot90(img)), kp)               # centre pixel stays in place
    turn = (float(a0[0]) - float(a1[0])) % 360.0
    assert min(abs(turn - 90.0), abs(turn - 270.0)) < 1.0
    differing = int(np.unpackbits(d0[0] ^ d1[0]).sum())
    assert differing <= 40                                                                  # 256 bits; unrelated patches differ in ~128

