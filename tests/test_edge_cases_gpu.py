"""Edge cases and size-independent properties of the hot path at the benchmark's full sizes."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h", [(65, 33), (127, 64), (333, 217), (1921, 17), (16, 16), (12, 40),
                                 (116, 70), (117, 129), (232, 65), (233, 200), (243, 131), (349, 64)])
def test_corner_ragged_image_sizes(ctx, w, h):
    """Widths that are not multiples of 4, of the tile or of the 116-column strips of the keypoint kernel,
    heights around its row bands, images of a single tile row, and images smaller than the borders (no
    valid pixel): maps bit-identical to the oracle, keypoints identical."""
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(w * 1000 + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[h // 4: h // 2, w // 4: w // 2] = 230
    gp, op = api.corner_params(harris_thresh=1e-6, shi_thresh=1e-5), orc.corner_params(harris_thresh=1e-6, shi_thresh=1e-5)
    har, shi = api.corner_response(ctx, img, gp)
    har0, shi0 = orc.corner_response(img, op)
    assert np.array_equal(har.view(np.uint32), har0.view(np.uint32))
    assert np.array_equal(shi.view(np.uint32), shi0.view(np.uint32))
    for det in (0, 1):
        got, want = api.corner_detect(ctx, img, gp, det), orc.corner_detect(img, op, det)
        assert len(got) == len(want)
        assert np.array_equal(got["x"], want["x"]) and np.array_equal(got["y"], want["y"])
        assert np.array_equal(got["score"].view(np.uint32), want["score"].view(np.uint32))


def test_corner_row_stride_and_unaligned_base(ctx):
    """The device-pointer entry point with a row stride larger than the width and a base address that is
    not 4-byte aligned (the tile loader's byte path)."""
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    full = syn.frame(7, 640, 360)
    view = full[3:303, 5:522]                                   # 300 x 517 window, stride 640, odd base offset
    dev = torch.device("cuda:0")
    t = torch.tensor(full, device=dev)
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    params = api.corner_params(harris_thresh=1e-4, shi_thresh=1e-3, max_features=2000)
    out = torch.zeros((1, 2, 2000, 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros((1, 2), dtype=torch.int32, device=dev)
    sub = t[3:, 5:]                                             # data_ptr at the window origin
    api.corner_detect_batch_dev(tctx, sub, 517, 300, 640, 640 * 360, 1, params, 2, out, 2000, cnt)
    torch.cuda.synchronize()
    op = orc.corner_params(harris_thresh=1e-4, shi_thresh=1e-3, max_features=2000)
    res = out.cpu().numpy()
    for det in (0, 1):
        want = orc.corner_detect(np.ascontiguousarray(view), op, det)
        k = int(cnt[0, det])
        assert k == len(want)
        assert np.array_equal(res[0, det, :k, 0], want["x"]) and np.array_equal(res[0, det, :k, 1], want["y"])
        assert np.array_equal(res[0, det, :k, 2].view(np.uint32), want["score"].view(np.uint32))
    tctx.close()


def test_corner_translation_property_full_frame(ctx):
    """1080p: moving the content by (dx, dy) moves every keypoint that stays clear of the borders by
    (dx, dy) and leaves its score bit-identical (no dependence on the tile grid)."""
    from visualslam_amd import api, synthetic as syn
    img = syn.frame(11)
    dx, dy = 37, 21
    moved = np.roll(np.roll(img, dy, axis=0), dx, axis=1)
    p = api.corner_params(harris_thresh=1e-4, shi_thresh=1e-3, max_features=16384, border=80)
    for det in (0, 1):
        a = api.corner_detect(ctx, img, p, det)
        b = api.corner_detect(ctx, moved, p, det)
        inner = lambda k, ox, oy: (k["x"] + ox >= 100) & (k["x"] + ox < 1820) & (k["y"] + oy >= 100) & (k["y"] + oy < 980) & \
                                  (k["x"] >= 100) & (k["x"] < 1820) & (k["y"] >= 100) & (k["y"] < 980)
        sa = {(int(k["x"]) + dx, int(k["y"]) + dy, k["score"].tobytes()) for k in a[inner(a, dx, dy)]}
        sb = {(int(k["x"]), int(k["y"]), k["score"].tobytes()) for k in b[inner(b, -dx, -dy)]}
        assert len(sa) > 100 and sa == sb


def test_point_loglik_additivity_at_benchmark_size(ctx, gcam):
    """512 x (L=150): the value, gradient and Hessian of all visible landmarks equal the sums over a
    partition of the visibility mask into three parts (the terms are independent per landmark), and every
    Hessian is exactly symmetric with zero velocity rows."""
    import torch
    from visualslam_amd import api, synthetic as syn
    B, L = 512, 150
    n = 12 + 3 * L
    scs = [syn.Scene(L, 700 + i, dense_S=False) for i in range(8)]
    rng = np.random.default_rng(1)
    x = np.tile(np.stack([s.x for s in scs]), (B // 8, 1)) + 1e-3 * rng.standard_normal((B, n))
    y = np.tile(np.stack([s.y for s in scs]), (B // 8, 1))
    vis = (rng.random((B, L)) < 0.9).astype(np.uint8)
    part = rng.integers(0, 3, (B, L))
    dev = torch.device("cuda:0")
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    xt, yt = torch.tensor(x, device=dev), torch.tensor(y, device=dev)

    def run(mask):
        val = torch.empty(B, dtype=torch.float64, device=dev)
        g = torch.empty((B, n), dtype=torch.float64, device=dev)
        H = torch.empty((B, n, n), dtype=torch.float64, device=dev)
        code = torch.empty((B, L), dtype=torch.int32, device=dev)
        api.point_loglik_batch_dev(tctx, gcam, n, xt, yt, torch.tensor(mask, device=dev), 7.0, val, g, H, code)
        torch.cuda.synchronize()
        return val, g, H

    v, g, H = run(vis)
    vs, gs, Hs = None, None, None
    for k in range(3):
        vk, gk, Hk = run((vis & (part == k)).astype(np.uint8))
        vs, gs, Hs = (vk, gk, Hk) if vs is None else (vs + vk, gs + gk, Hs + Hk)
    assert float((v - vs).abs().max() / v.abs().max()) <= 1e-12
    assert float((g - gs).abs().max() / g.abs().max()) <= 1e-12
    assert float((H - Hs).abs().max() / H.abs().max()) <= 1e-12
    assert bool((H == H.transpose(1, 2)).all())
    assert not bool(H[:, :6, :].any()) and not bool(g[:, :6].any())
    tctx.close()


def test_qr_idempotent_on_triangular_input_and_gram(ctx):
    """R of an upper-triangular matrix is the matrix itself (every reflector is the identity, signs kept),
    and R^T R = A^T A at the 2,000-landmark size of the square-root update."""
    from visualslam_amd import api
    rng = np.random.default_rng(3)
    U = np.triu(rng.standard_normal((1500, 1500)))
    assert np.array_equal(api.qr_upper(ctx, U), U)
    A = rng.standard_normal((6012, 1024))
    R = api.qr_upper(ctx, A)[:1024]
    assert rel_err(R.T @ R, A.T @ A) <= 1e-12


def test_bundle_empty_visibility_and_single_landmark(ctx, gcam, ocam):
    """n = 15 (one landmark) is the smallest legal state; everything culled gives a zero Jacobian block."""
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    sc = syn.Scene(1, 9)
    h, SR, Cm, code = api.bundle_measure(ctx, gcam, sc.x)
    h0, SR0, C0, code0 = orc.bundle_measure(sc.x, ocam)
    assert np.array_equal(code, code0) and rel_err(h, h0) <= 1e-12 and rel_err(Cm, C0) <= 1e-9
    x = sc.x.copy()
    x[12:15] = x[6:9] - np.array([0.0, 0.0, 5.0])             # behind the camera in every orientation used here
    v, g, H, code = api.point_loglik(ctx, gcam, x, sc.y, np.ones(1, np.uint8), 7.0)
    v0, g0, H0, c0 = orc.point_loglik_dual_sparse(x, sc.y, np.ones(1, np.uint8), ocam, 7.0)
    assert np.array_equal(code, c0)
    if c0[0] != 0:
        assert v == 0.0 and not g.any() and not H.any()


def test_empty_map_frame(ctx, gcam, ocam):
    """The frames before the first landmark: the state is the 12 pose / velocity entries alone.  The reference still runs
    the measurement update on every frame (src/SLAM.cpp:489-495); pointLogLikelihood loops over param.landmarks.size()
    landmarks (src/model.cpp:392), none, so the likelihood is 0 with zero gradient and Hessian -- the case
    test/src/measurementModel.cpp:462-519 sets up ("points" with an empty landmark list) -- and the update returns the
    prior.  Checked against the CPU oracle on the same inputs."""
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(12)
    n = 12
    mu = np.concatenate([0.1 * rng.standard_normal(6), [0.0, -5.0, -1.8, -1.57, 3.14, 0.0]])
    d = np.concatenate([np.full(6, 0.1), np.full(6, 0.001)])
    S = np.asfortranarray(np.triu(0.05 * np.sqrt(np.outer(d, d)) * rng.standard_normal((n, n)), 1) + np.diag(d))
    none, novis = np.zeros(0), np.zeros(0, np.uint8)
    # PointLogLikelihood x3 (src/model.cpp:408-437)
    v, g, H, code = api.point_loglik(ctx, gcam, mu, none, novis, 7.0)
    v0, g0, H0, _ = orc.point_loglik_dual_sparse(mu, none, novis, ocam, 7.0)
    assert v == 0.0 and v0 == 0.0 and code.size == 0
    assert g.shape == (n,) and H.shape == (n, n) and not g.any() and not H.any() and not g0.any() and not H0.any()
    v, _, _, _ = api.point_loglik(ctx, gcam, mu, none, novis, 7.0, want_g=False, want_H=False)
    assert v == 0.0
    # the strided-batch device entry, three empty maps
    dev = torch.device("cuda:0")
    xs = torch.tensor(np.stack([mu, mu + 1.0, mu - 1.0]), device=dev)
    empty_y = torch.zeros((3, 0), dtype=torch.float64, device=dev)
    empty_v = torch.zeros((3, 0), dtype=torch.uint8, device=dev)
    val = torch.full((3,), float("nan"), dtype=torch.float64, device=dev)
    gb = torch.full((3, n), float("nan"), dtype=torch.float64, device=dev)
    Hb = torch.full((3, n, n), float("nan"), dtype=torch.float64, device=dev)
    api.point_loglik_batch_dev(ctx, gcam, n, xs, empty_y, empty_v, 7.0, val, gb, Hb)
    torch.cuda.synchronize()
    assert not val.cpu().numpy().any() and not gb.cpu().numpy().any() and not Hb.cpu().numpy().any()
    # CostJointDensity (src/gaussian.hpp:455-466): the prior alone
    x = mu + S.T @ (0.3 * rng.standard_normal(n))
    V, gc, Hc = api.cost_joint_density(ctx, gcam, x, mu, S, none, novis, 7.0)
    lp, gp, Hp = orc.log_gaussian(x, mu, S)
    assert abs(V + lp) <= 1e-9 * abs(lp) and rel_err(gc, -gp) <= 1e-9 and rel_err(Hc, -Hp) <= 1e-9
    # measurementUpdateIEKF (src/gaussian.hpp:469-532): posterior = prior, on both trust-region routes and in a bank
    ret0, mup0, Sp0, it0, ev0 = orc.measurement_update_iekf(mu, S, none, novis, ocam, 7.0)
    assert ret0 == 0 and np.abs(mup0 - mu).max() <= 1e-12
    for solver in (api.TRS_EIG, api.TRS_CHOLESKY):
        ctx.set_trs_solver(solver)
        try:
            ret, mup, Sp, iters, evals = api.measurement_update_iekf(ctx, gcam, mu, S, none, novis, 7.0)
        finally:
            ctx.set_trs_solver(api.TRS_CHOLESKY)
        assert ctx.last_trs_route() == solver
        assert (ret, iters, evals) == (ret0, it0, ev0)
        assert np.abs(mup - mu).max() <= 1e-12
        assert not np.tril(Sp, -1).any()
        assert rel_err(Sp.T @ Sp, S.T @ S) <= 1e-9 and rel_err(Sp.T @ Sp, Sp0.T @ Sp0) <= 1e-9
    mub = torch.tensor(np.stack([mu, mu + 0.5]), device=dev)
    Sb = torch.tensor(np.stack([np.ascontiguousarray(S.T)] * 2), device=dev)
    mpb, Spb = torch.empty_like(mub), torch.empty_like(Sb)
    code, iters, status = api.measurement_update_iekf_batch_dev(ctx, gcam, mub, Sb, empty_y[:2], empty_v[:2], 7.0, mpb, Spb)
    torch.cuda.synchronize()
    assert not status.any() and not code.any()
    assert np.abs(mpb.cpu().numpy() - mub.cpu().numpy()).max() <= 1e-12
    for b in range(2):
        S1 = Spb[b].cpu().numpy().T
        assert rel_err(S1.T @ S1, S.T @ S) <= 1e-9
    # affineTransform through the bundle (src/gaussian.hpp:85-131): a 0-dimensional predicted measurement
    muy, Syy, codes = api.affine_bundle(ctx, gcam, mu, S)
    assert muy.size == 0 and Syy.size == 0 and codes.size == 0
    h, SR, Cm, codes = api.bundle_measure(ctx, gcam, mu)
    assert h.size == 0 and Cm.shape == (0, n)
