"""ORB keypoint detection on the GPU (vslam_orb_detect*, cv::ORB::detect as configured at reference
src/imagefeatures.cpp:73-87) against the CPU oracle: keypoint lists identical, bit for bit.

Parity with the reference is UNPINNED: OpenCV is not in the reference tree and the reference's only test of this
code (test/src/matcher.cpp) is commented out; the oracle restates the published algorithm and is the contract."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _frame_1023():
    from PIL import Image
    return np.asarray(Image.open(os.path.join(GOLDEN, "frame_1023_gray.png")))


def _level(ctx, img, n, **kw):
    import torch
    from visualslam_amd import api
    h, w = img.shape
    dev = torch.device("cuda:0")
    t = torch.tensor(img, device=dev)
    cap = 16384
    out = torch.zeros((1, cap, 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    api.orb_detect_level_batch_dev(ctx, t, w, h, w, w * h, 1, n, out, cap, cnt, **kw)
    ctx.synchronize()
    k = int(cnt[0])
    assert k >= 0
    from oracle import pyoracle as orc
    return out[0, :k].cpu().numpy().view(orc.FEATURE_DTYPE).reshape(-1)


@pytest.mark.parametrize("name,n", [("frame1023", 2172), ("synthetic1080p", 2172), ("synthetic1080p", 300), ("noise", 500), ("frame1023", 0)])
def test_level0_keypoints_identical_to_oracle(ctx, name, n):
    from oracle import pyoracle as orc
    from visualslam_amd import synthetic as syn
    if name == "noise":
        img = np.random.default_rng(3).integers(0, 256, (600, 800), dtype=np.uint8)
    else:
        img = {"frame1023": _frame_1023, "synthetic1080p": lambda: syn.frame(2)}[name]()
    want = orc.orb_detect_level(img, n)
    got = _level(ctx, img, n)
    assert got.shape == want.shape
    assert np.array_equal(got["x"], want["x"]) and np.array_equal(got["y"], want["y"])
    assert np.array_equal(got["score"].view(np.uint32), want["score"].view(np.uint32))      # responses bit for bit
    if n > 0 and name == "synthetic1080p":
        assert want.size > 50                                                             # the case is not vacuous
        assert (want["x"] >= 75).all() and (want["x"] < img.shape[1] - 75).all()


def test_reference_frame_at_a_lower_fast_threshold(ctx):
    """data/1023.jpg is a low-contrast frame: at the reference's FAST threshold of 20 only a handful of corners lie
    inside the 75-pixel edge; at 5 there are hundreds, which makes the real-image comparison worth something."""
    from oracle import pyoracle as orc
    img = _frame_1023()
    want = orc.orb_detect_level(img, 2172, 5, 75, 7, 0.04)
    got = _level(ctx, img, 2172, fast_threshold=5)
    assert want.size > 100 and got.tobytes() == want.tobytes()


def test_other_thresholds_and_block(ctx):
    from oracle import pyoracle as orc
    from visualslam_amd import synthetic as syn
    img = syn.frame(7)[:700, :900].copy()
    for n, ft, edge, block in ((400, 10, 31, 7), (150, 40, 20, 5), (1000, 20, 75, 9)):
        want = orc.orb_detect_level(img, n, ft, edge, block, 0.04)
        got = _level(ctx, img, n, fast_threshold=ft, edge_threshold=edge, harris_block=block)
        assert got.tobytes() == want.tobytes()


def test_ties_at_the_cut_are_kept(ctx):
    """A lattice of identical isolated bright pixels: every FAST score and every Harris response is shared by all of them,
    so both retain-best steps keep more than they were asked for."""
    from oracle import pyoracle as orc
    img = np.full((400, 600), 40, dtype=np.uint8)
    img[100:300:20, 100:500:20] = 200
    want = orc.orb_detect_level(img, 10, 20, 31, 7, 0.04)
    got = _level(ctx, img, 10, edge_threshold=31)
    assert want.size > 10 and got.tobytes() == want.tobytes()


def test_resize_bit_exact(ctx):
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    img = syn.frame(9)
    dev = torch.device("cuda:0")
    src = torch.tensor(img, device=dev)
    for dw, dh in ((1600, 900), (1333, 750), (1920, 1080), (7, 5), (2500, 1300)):
        dst = torch.zeros((dh, dw), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        api.resize_bilinear_dev(ctx, src, dst)
        ctx.synchronize()
        assert np.array_equal(dst.cpu().numpy(), orc.resize_bilinear(img, dw, dh))


@pytest.mark.parametrize("name", ["frame1023", "synthetic1080p"])
def test_all_levels_identical_to_oracle(ctx, name):
    """The 8-level pyramid with the reference's parameters: per level the same keypoints, positions scaled to the base
    image by the same float multiplication."""
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    img = {"frame1023": _frame_1023, "synthetic1080p": lambda: syn.frame(2)}[name]()
    kps, lev = api.orb_detect(ctx, img)
    want, ls = orc.orb_detect(img)
    assert lev.tolist() == [len(w_) for w_ in want]
    assert sum(lev) == kps.size
    if name == "synthetic1080p":
        assert lev[0] > 50 and lev[3] > 10
    pos = 0
    for l, w_ in enumerate(want):
        g = kps[pos:pos + len(w_)]
        pos += len(w_)
        sx = w_["x"].astype(np.float32) * ls[l] if l else w_["x"].astype(np.float32)
        sy = w_["y"].astype(np.float32) * ls[l] if l else w_["y"].astype(np.float32)
        assert np.array_equal(g["x"], sx) and np.array_equal(g["y"], sy)
        assert np.array_equal(g["response"].view(np.uint32), w_["score"].view(np.uint32))
        assert (g["octave"] == l).all()


def test_orb_errors(ctx):
    from visualslam_amd import api
    img = np.zeros((200, 300), dtype=np.uint8)
    with pytest.raises(api.VslamError):
        api.orb_detect(ctx, img, api.orb_params(nfeatures=100000))          # level budget beyond the sorted list
    with pytest.raises(api.VslamError):
        api.orb_detect(ctx, img, api.orb_params(nlevels=40))
    kps, lev = api.orb_detect(ctx, img, api.orb_params(nfeatures=500))      # levels that fit inside the edge: none has corners
    assert kps.size == 0 and not lev.any()
