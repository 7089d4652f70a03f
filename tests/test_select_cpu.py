"""pixelDistance and the new-landmark selection loop (reference src/SLAM.cpp:449-472, 746-760): the
reference's own known answers (test/src/pixelDistance.cpp) on the oracle and on the library's host
function, and the oracle's loop against a direct reading of the rule."""
import numpy as np
import pytest

from oracle import pyoracle as orc

KATS = [  # test/src/pixelDistance.cpp: landmarks at (0,0) and (100,0)
    ((0.0, 0.0), 10.0, True),       # :9-39
    ((0.0, 0.0), 200.0, True),      # :41-71
    ((500.0, 500.0), 200.0, False),  # :73-105
]


@pytest.mark.parametrize("pt,thresh,want", KATS)
def test_pixel_distance_reference_kats(pt, thresh, want):
    from visualslam_amd import api
    lm = [[0.0, 0.0], [100.0, 0.0]]
    assert orc.pixel_distance(lm, pt[0], pt[1], thresh) is want
    assert api.pixel_distance(lm, pt[0], pt[1], thresh) is want


def test_pixel_distance_strict_inequality_and_empty_map():
    from visualslam_amd import api
    lm = [[3.0, 4.0]]
    for f in (orc.pixel_distance, api.pixel_distance):
        assert f(lm, 0.0, 0.0, 5.0) is False            # dist == thresh is not "within"
        assert f(lm, 0.0, 0.0, 5.0000001) is True
        assert f(np.zeros((0, 2)), 1.0, 1.0, 100.0) is False


def _python_rule(feats, lm, fthr, wthr, hthr, dthr, iw, ih, max_new):
    lm = [tuple(map(np.float32, p)) for p in np.asarray(lm, dtype=np.float32).reshape(-1, 2)]
    out = []
    for i, f in enumerate(feats):
        x, y = np.float32(f["x"]), np.float32(f["y"])
        within = any(np.sqrt(np.float32(np.float32((a - x) * (a - x)) + np.float32((b - y) * (b - y)))) < dthr for a, b in lm)
        if len(out) < max_new and not within and wthr <= x <= iw - wthr and hthr <= y <= ih - hthr and f["score"] > fthr:
            out.append(i)
            lm.append((x, y))
    return out


@pytest.mark.parametrize("seed,L,max_new", [(0, 0, 50), (1, 20, 30), (2, 150, 5), (3, 40, 0), (4, 10, 1000)])
def test_selection_loop_matches_the_rule(seed, L, max_new):
    rng = np.random.default_rng(seed)
    K = 600
    feats = np.zeros(K, dtype=orc.FEATURE_DTYPE)
    feats["x"] = rng.integers(0, 1920, K)
    feats["y"] = rng.integers(0, 1080, K)
    feats["score"] = np.sort(rng.uniform(0, 1e-3, K).astype(np.float32))[::-1]
    lm = np.stack([rng.uniform(0, 1920, L), rng.uniform(0, 1080, L)], axis=1)
    got = orc.select_new_landmarks(feats, lm, 1e-4, 250, 100, 150, 1920, 1080, max_new)
    assert got.tolist() == _python_rule(feats, lm, 1e-4, 250, 100, 150, 1920, 1080, max_new)
