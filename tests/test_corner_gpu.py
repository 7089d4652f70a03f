"""GPU parity of the corner front end: response maps and keypoint lists must be
bit-identical to the CPU oracle (integer structure tensor, float32 response
with one rounding per operation).  The reference delegates this arithmetic to
OpenCV and has no test for it: parity with the reference is unpinned; the
oracle defines the contract (see oracle/vslam_oracle.c)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _frame_1023():
    from PIL import Image
    return np.asarray(Image.open(os.path.join(GOLDEN, "frame_1023_gray.png")))


def _small_pattern():
    img = np.full((48, 64), 40, np.uint8)
    img[10:30, 12:40] = 200
    img[20:44, 30:60] = 90
    rng = np.random.default_rng(0)
    return np.clip(img.astype(np.int32) + rng.integers(-3, 4, img.shape), 0, 255).astype(np.uint8)


def _params(api_mod, **kw):
    return api_mod.corner_params(**kw)


def _both(gray, kw, ctx):
    from oracle import pyoracle as orc
    from visualslam_amd import api
    gp, op = api.corner_params(**kw), orc.corner_params(**kw)
    res = {}
    for det in (0, 1):
        res[det] = (api.corner_detect(ctx, gray, gp, det), orc.corner_detect(gray, op, det))
    return res


@pytest.mark.parametrize("name", ["small", "frame1023", "synthetic1080p"])
def test_response_maps_bit_exact(ctx, name):
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    gray = {"small": _small_pattern, "frame1023": _frame_1023, "synthetic1080p": lambda: syn.frame(2)}[name]()
    har, shi = api.corner_response(ctx, gray, api.corner_params())
    har0, shi0 = orc.corner_response(gray, orc.corner_params())
    assert np.array_equal(har.view(np.uint32), har0.view(np.uint32))
    assert np.array_equal(shi.view(np.uint32), shi0.view(np.uint32))


@pytest.mark.parametrize("name,kw", [
    ("small", dict(harris_thresh=1e-6, shi_thresh=1e-5)),
    ("frame1023", dict(harris_thresh=1e-4, shi_thresh=1e-3)),
    ("frame1023", dict(harris_thresh=1e-6, shi_thresh=1e-4, max_features=500)),
    ("synthetic1080p", dict(harris_thresh=1e-4, shi_thresh=1e-3)),
    ("synthetic1080p", dict(harris_thresh=1e-5, shi_thresh=2e-4, block_size=3)),
    ("synthetic1080p", dict(harris_thresh=1e-4, shi_thresh=1e-3, block_size=11, border=75)),
])
def test_keypoints_identical(ctx, name, kw):
    from visualslam_amd import synthetic as syn
    gray = {"small": _small_pattern, "frame1023": _frame_1023, "synthetic1080p": lambda: syn.frame(2)}[name]()
    for det, (got, want) in _both(gray, kw, ctx).items():
        assert len(got) == len(want), (det, len(got), len(want))
        assert np.array_equal(got["x"], want["x"]) and np.array_equal(got["y"], want["y"])
        assert np.array_equal(got["score"].view(np.uint32), want["score"].view(np.uint32))
        if len(got) > 1:
            assert np.all(np.diff(got["score"]) <= 0), "sorted by score, descending (imagefeatures.cpp:20-22)"


@pytest.mark.parametrize("harris_thresh,shi_thresh", [(0.0055, 0.068),      # 18436 / 25614 candidates: two chunks of 16384
                                                      (0.004, 0.058)])      # 49053 / 58067: three and four chunks
def test_many_candidates_merge_path(ctx, harris_thresh, shi_thresh):
    """> 16384 candidates per detector: exercises the chunk-merge kernel."""
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(5)
    gray = rng.integers(0, 256, (1080, 1920), dtype=np.uint8)       # pure noise: corners everywhere
    kw = dict(harris_thresh=harris_thresh, shi_thresh=shi_thresh, max_features=16384)
    gp, op = api.corner_params(**kw), orc.corner_params(**kw)
    for det in (0, 1):
        want = orc.corner_detect(gray, op, det, cap=16384)
        try:
            got = api.corner_detect(ctx, gray, gp, det, cap=16384)
        except api.VslamError as e:
            pytest.fail("capacity: %s" % e)
        assert len(want) == 16384, "test input must overflow one sort chunk (got %d)" % len(want)
        assert np.array_equal(got["x"], want["x"]) and np.array_equal(got["y"], want["y"])
        assert np.array_equal(got["score"].view(np.uint32), want["score"].view(np.uint32))


def test_strip_and_tile_kernels_agree_on_other_window_sizes(ctx):
    """block_size 7 takes the sliding-window kernel, every other size the tiled kernel; both follow the
    same contract, so each must equal the oracle on the same frame."""
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    gray = syn.frame(11, w=701, h=403, n_rect=80, n_disc=40)
    for bs, border in ((5, 8), (7, 5), (7, 31), (9, 9)):
        kw = dict(harris_thresh=1e-5, shi_thresh=3e-4, block_size=bs, border=border)
        gp, op = api.corner_params(**kw), orc.corner_params(**kw)
        for det in (0, 1):
            got, want = api.corner_detect(ctx, gray, gp, det), orc.corner_detect(gray, op, det)
            assert len(got) == len(want) and len(want) > 50
            assert np.array_equal(got["x"], want["x"]) and np.array_equal(got["y"], want["y"])
            assert np.array_equal(got["score"].view(np.uint32), want["score"].view(np.uint32))


def test_no_keypoints_and_errors(ctx):
    from visualslam_amd import api
    flat = np.full((120, 160), 77, np.uint8)
    assert len(api.corner_detect(ctx, flat, api.corner_params(), 0)) == 0
    with pytest.raises(api.VslamError):
        api.corner_detect(ctx, flat, api.corner_params(block_size=4), 0)
    with pytest.raises(api.VslamError):
        api.corner_detect(ctx, flat[:8, :8], api.corner_params(), 0)


def test_batch_dev_both_detectors(ctx):
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    frames = np.stack([syn.frame(100 + i, w=640, h=360, n_rect=60, n_disc=30) for i in range(3)])
    kw = dict(harris_thresh=1e-4, shi_thresh=1e-3, max_features=2000)
    gp, op = api.corner_params(**kw), orc.corner_params(**kw)
    dev = torch.device("cuda:0")
    t = torch.tensor(frames, device=dev)
    cap = 2000
    out = torch.zeros((3, 2, cap, 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros((3, 2), dtype=torch.int32, device=dev)
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    api.corner_detect_batch_dev(tctx, t, 640, 360, 640, 640 * 360, 3, gp, 2, out, cap, cnt)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    for f in range(3):
        for det in (0, 1):
            want = orc.corner_detect(frames[f], op, det)
            k = int(cnt[f, det])
            assert k == len(want)
            got = o[f, det, :k]
            assert np.array_equal(got[:, 0], want["x"]) and np.array_equal(got[:, 1], want["y"])
            assert np.array_equal(got[:, 2].view(np.uint32), want["score"].view(np.uint32))
    tctx.close()


@pytest.mark.gpu
def test_sqrtf_is_correctly_rounded_on_device(ctx):
    """The Shi-Tomasi response uses sqrtf; the bit-exact contract needs it correctly rounded for every
    non-negative finite float (checked against the fp64 square root rounded once)."""
    import ctypes as C
    from visualslam_amd import _lib
    L = _lib.lib()
    L.vslam_debug_sqrtf_mismatches.restype = C.c_int32
    L.vslam_debug_sqrtf_mismatches.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    bad = C.c_ulonglong(1)
    assert L.vslam_debug_sqrtf_mismatches(ctx.ptr, C.byref(bad)) == 0
    assert bad.value == 0
