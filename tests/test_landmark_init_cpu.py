"""Initialisation of new landmarks (reference src/SLAM.cpp:876-941): the oracle's optical ray against the
camera model it inverts (a landmark placed on the ray projects back onto its keypoint), the state growth
against a direct reading of the rule, and the library's host function against the oracle.
Parity unpinned: the reference removes the distortion with cv::undistortPoints, which is not part of the
reference's sources and holds no fixture there; the oracle restates its documented iteration."""
import numpy as np
import pytest

from oracle import pyoracle as orc

PIXELS = [(960.0, 540.0), (300.0, 200.0), (1600.0, 900.0), (250.0, 100.0), (1670.0, 980.0), (1234.56, 321.25)]


@pytest.mark.parametrize("uv", PIXELS)
def test_ray_projects_back_onto_its_pixel(ocam, uv):
    eta = np.array([0.3, -0.2, 1.1, 0.05, -0.1, 0.4])
    ray = orc.optical_ray(ocam, uv[0], uv[1], 8.0)
    assert ray[2] == 8.0
    Rnc = orc.rpy2rot(eta[3:])
    code, px = orc.world_to_pixel(eta[:3] + Rnc @ ray, eta, ocam)
    assert code == 0
    assert np.max(np.abs(px - np.array(uv))) < 0.01        # five fixed-point steps + float pixel


def test_ray_without_distortion_is_the_pinhole_ray():
    cam = orc.make_camera(1000.0, 900.0, 640.0, 360.0, [0.0] * 12, 0, 1280, 720)
    r = orc.optical_ray(cam, 740.0, 270.0, 3.0)
    np.testing.assert_allclose(r, [0.3, -0.3, 3.0], rtol=0, atol=1e-12)


def test_ray_scales_with_its_length(ocam):
    a, b = orc.optical_ray(ocam, 400.0, 700.0, 1.0), orc.optical_ray(ocam, 400.0, 700.0, 8.0)
    np.testing.assert_allclose(8.0 * a, b, rtol=1e-15)


def test_state_growth(ocam):
    rng = np.random.default_rng(5)
    n = 12 + 3 * 4
    mu = rng.normal(size=n)
    mu[9:12] *= 0.2
    S = np.triu(rng.normal(size=(n, n)))
    mo, So = orc.initialize_landmark(mu, S, ocam, 700.0, 400.0, 8.0, 0.5)
    assert mo.shape == (n + 3,) and So.shape == (n + 3, n + 3)
    np.testing.assert_array_equal(mo[:n], mu)
    np.testing.assert_array_equal(So[:n, :n], S)
    np.testing.assert_array_equal(So[n:, :n], 0.0)
    np.testing.assert_array_equal(So[:n, n:], 0.0)
    np.testing.assert_array_equal(So[n:, n:], 0.5 * np.eye(3))
    want = mu[6:9] + orc.rpy2rot(mu[9:12]) @ orc.optical_ray(ocam, 700.0, 400.0, 8.0)
    np.testing.assert_allclose(mo[n:], want, rtol=1e-14)
    # the new landmark is seen where its keypoint was
    code, px = orc.world_to_pixel(mo[n:], mu[6:12], ocam)
    assert code == 0 and np.max(np.abs(px - [700.0, 400.0])) < 0.01


@pytest.mark.parametrize("uv", PIXELS)
def test_library_host_ray_equals_oracle(ocam, gcam, uv):
    from visualslam_amd import api
    for length in (8.0, 3.0):
        np.testing.assert_array_equal(api.optical_ray(gcam, uv[0], uv[1], length), orc.optical_ray(ocam, uv[0], uv[1], length))


def test_library_host_ray_rejects_unusable_constants(gcam):
    import copy
    from visualslam_amd import api
    bad = copy.copy(gcam)
    bad.fx = 0.0
    with pytest.raises(api.VslamError):
        api.optical_ray(bad, 10.0, 10.0)
    with pytest.raises(api.VslamError):
        api.optical_ray(gcam, 10.0, 10.0, float("nan"))
