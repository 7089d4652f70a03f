"""FAST-9/16 detector (vslam_fast9_detect*, first stage of cv::ORB::detect at reference
src/imagefeatures.cpp:73-87) against the CPU oracle: integer work, keypoint lists identical bit for bit.
OpenCV is not part of the reference and absent here: parity with it is unpinned; the oracle states the
contract and tests/test_fast9_cpu.py checks the oracle against a brute-force reading of the definition."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    return a.shape == b.shape and np.array_equal(a["x"], b["x"]) and np.array_equal(a["y"], b["y"]) and np.array_equal(a["score"], b["score"])


def _images():
    from visualslam_amd import synthetic as syn
    rng = np.random.default_rng(9)
    yield "noise", rng.integers(0, 256, (97, 131), dtype=np.uint8)
    yield "smooth_noise", (128 + 40 * np.sin(np.arange(200)[None, :] / 9.0) + rng.normal(0, 12, (150, 200))).clip(0, 255).astype(np.uint8)
    yield "frame", syn.frame(5001)
    sq = np.full((64, 64), 50, np.uint8)
    sq[20:44, 20:44] = 200
    sq[20, 20] = 210
    yield "square", sq
    yield "flat", np.full((40, 70), 99, np.uint8)


@pytest.mark.parametrize("threshold,nonmax", [(20, True), (20, False), (0, True), (60, True), (254, True)])
def test_fast9_matches_oracle(ctx, threshold, nonmax):
    from oracle import pyoracle as orc
    from visualslam_amd import api
    for name, img in _images():
        if name == "frame" and threshold < 20:
            continue                     # more corners than a candidate list holds (65 536): see test_fast9_capacity
        want = orc.fast9_detect(img, threshold, nonmax)
        got = api.fast9_detect(ctx, img, threshold, nonmax)
        assert _same(got, want), (name, threshold, nonmax, len(got), len(want))


@pytest.mark.parametrize("w,h", [(7, 7), (8, 9), (63, 15), (64, 16), (65, 17), (128, 32), (129, 33), (191, 7)])
def test_fast9_tile_boundaries(ctx, w, h):
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(w * 100 + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    for nonmax in (True, False):
        assert _same(api.fast9_detect(ctx, img, 30, nonmax), orc.fast9_detect(img, 30, nonmax))


def test_fast9_cap_and_max_features(ctx):
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (120, 160), dtype=np.uint8)
    full = orc.fast9_detect(img, 25, True)
    assert len(full) > 50
    assert _same(api.fast9_detect(ctx, img, 25, True, max_features=17), full[:17])
    assert _same(api.fast9_detect(ctx, img, 25, True, max_features=100, cap=9), full[:9])


def test_fast9_dense_corners_per_tile(ctx):
    """More corners in one 64 x 32 tile than its staging buffer holds (512): the rest goes the slow way."""
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (70, 140), dtype=np.uint8)
    want = orc.fast9_detect(img, 3, False)
    assert len(want) > 3000
    assert _same(api.fast9_detect(ctx, img, 3, False), want)


def test_fast9_capacity(ctx):
    """A frame with more corners than a candidate list holds is an error, not a silent truncation."""
    from visualslam_amd import api, synthetic as syn
    with pytest.raises(api.VslamError):
        api.fast9_detect(ctx, syn.frame(5001), 0, False)


def test_fast9_errors(ctx):
    from visualslam_amd import api
    with pytest.raises(api.VslamError):
        api.fast9_detect(ctx, np.zeros((6, 20), np.uint8))
    with pytest.raises(api.VslamError):
        api.fast9_detect(ctx, np.zeros((20, 20), np.uint8), threshold=255)
    with pytest.raises(api.VslamError):
        api.fast9_detect(ctx, np.zeros((20, 20), np.uint8), max_features=0)


def test_fast9_batch_dev_with_row_stride(ctx):
    """Frames inside a larger buffer (row stride > width, frame stride > rows * stride)."""
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(12)
    F, h, w, stride = 5, 70, 150, 160
    buf = rng.integers(0, 256, (F, h + 3, stride), dtype=np.uint8)
    dev = torch.device("cuda:0")
    g = torch.tensor(buf, device=dev)
    cap = 4096
    out = torch.zeros((F, cap, 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros(F, dtype=torch.int32, device=dev)
    api.fast9_detect_batch_dev(ctx, g, w, h, stride, (h + 3) * stride, F, 35, True, cap, out, cap, cnt)
    torch.cuda.synchronize()
    res = out.cpu().numpy().view(api.FEATURE_DTYPE).reshape(F, cap)
    for f in range(F):
        want = orc.fast9_detect(np.ascontiguousarray(buf[f, :h, :w]), 35, True)
        n = int(cnt[f])
        assert _same(res[f, :n], want)
