"""GPU parity of the data-association stage (individualCompatibility, jointCompatibility, snn;
src/dataAssociation.cpp) against the reference's known-answer tests and the CPU oracle."""
import numpy as np
import pytest

from test_oracle_cpu import KAT_LUT, KAT_MUY, KAT_SU, KAT_SYY, KAT_Y3, KAT_Y13

pytestmark = pytest.mark.gpu


def test_chi2inv_normcdf_host_functions():
    """test/src/normcdf.cpp, test/src/chi2inv.cpp and the LUT of test/src/jointCompatibility.cpp:89 (host code of the library)."""
    from visualslam_amd import api
    assert abs(api.normcdf(1) - 0.841344746068543) < 1e-14 and abs(api.normcdf(2) - 0.977249868051821) < 1e-14
    for (c, nu), want in (((0.5, 1), 0.454936423119573), ((0.5, 2), 1.386294361119890), ((0.5, 3), 2.365973884375338),
                          ((0.75, 3), 4.108344935632316)):
        assert abs(api.chi2inv(c, nu) - want) <= 1e-13 * want
    c = 2 * api.normcdf(3) - 1
    for k, want in enumerate(KAT_LUT):
        assert abs(api.chi2inv(c, 2 * (k + 1)) - want) <= 1e-12 * want


def test_individual_compatibility_reference_kats(ctx):
    """test/src/individualCompatibility.cpp:11-125, all pairs in one call."""
    from visualslam_amd import api
    got = api.individual_compat(ctx, KAT_Y3, KAT_MUY[:2], KAT_SYY[:2, :2], KAT_LUT[:1])
    assert got[:, 0].tolist() == [False, True, False]
    expected = [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1]
    got = api.individual_compat(ctx, KAT_Y13, KAT_MUY, KAT_SYY, KAT_LUT)
    assert got.astype(int).ravel().tolist() == expected


def test_joint_compatibility_reference_kats(ctx):
    """test/src/jointCompatibility.cpp:10-129 (inputs printed to 10 digits there: 1e-7 relative)."""
    from visualslam_amd import api
    ok, s = api.joint_compat(ctx, [1], KAT_SU, KAT_Y3, KAT_MUY[:2], KAT_SYY[:2, :2], KAT_LUT[:1])
    assert ok and abs(s - 10.105667572052072) <= 1e-7 * s
    ok, s = api.joint_compat(ctx, [-1], KAT_SU, KAT_Y3, KAT_MUY[:2], KAT_SYY[:2, :2], KAT_LUT[:1])
    assert ok and abs(s - 14.832478857591873) <= 1e-12 * s
    ok, s = api.joint_compat(ctx, [6, 7, 10], KAT_SU, KAT_Y13, KAT_MUY, KAT_SYY, KAT_LUT)
    assert ok and abs(s - 33.394861266913537) <= 1e-7 * s


def _density(L, seed, m_extra, noise):
    """Predicted-measurement density of a synthetic scene plus a feature set: the true projections of a
    shuffled subset of the landmarks with pixel noise, and clutter."""
    from oracle import pyoracle as orc
    from visualslam_amd import synthetic as syn
    sc = syn.Scene(L, seed)
    rng = np.random.default_rng(seed + 77)
    keep = rng.permutation(L)[: max(1, (3 * L) // 4)]
    feats = sc.pixel_true[keep] + noise * rng.standard_normal((keep.size, 2))
    clutter = np.stack([rng.uniform(0, 1920, m_extra), rng.uniform(0, 1080, m_extra)], axis=1)
    y = np.concatenate([feats, clutter])[rng.permutation(keep.size + m_extra)].T.copy()
    return sc, y, orc


@pytest.mark.parametrize("L,seed,enforce", [(1, 3, False), (4, 5, False), (9, 11, False), (9, 12, True), (16, 21, False), (60, 33, False), (60, 34, True)])
def test_snn_vs_oracle(ctx, gcam, ocam, L, seed, enforce):
    """snn end to end (bundle -> affineTransform -> greedy search): same association vector as the oracle's
    literal restatement (which re-factors per candidate like the reference) and the same surprisal."""
    from visualslam_amd import api
    sc, y, orc = _density(L, seed, 6, 2.0)
    s0, idx0 = orc.snn(sc.x, sc.S, y, ocam, 0, enforce)
    s, idx = api.snn(ctx, gcam, sc.x, sc.S, y, enforce_jc=enforce)
    assert idx.tolist() == idx0.tolist()
    assert abs(s - s0) <= 1e-9 * abs(s0)
    assert (idx >= 0).sum() >= 1            # the scene is built so that associations exist


def test_joint_compat_vs_oracle_random_assignments(ctx, gcam, ocam):
    from visualslam_amd import api
    sc, y, orc = _density(8, 31, 5, 1.5)
    h, SR, Cm, _ = orc.bundle_measure(sc.x, ocam)
    SYY = orc.affine_sqrt(sc.S, Cm, SR)
    c = 2 * orc.normcdf(3) - 1
    lut = [orc.chi2inv(c, 2 * (k + 1)) for k in range(8)]
    sU = np.log(1920) + np.log(1080)
    rng = np.random.default_rng(4)
    for _ in range(6):
        idx = np.where(rng.random(8) < 0.6, rng.integers(0, y.shape[1], 8), -1)
        ok0, s0 = orc.joint_compat(idx, sU, y, h, SYY, lut)
        ok, s = api.joint_compat(ctx, idx, sU, y, h, SYY, lut)
        assert ok == ok0 and abs(s - s0) <= 1e-9 * abs(s0)
    ind = api.individual_compat(ctx, y, h, SYY, lut)
    ind0 = np.array([[orc.individual_compat(i, j, y, h, SYY, lut) for j in range(8)] for i in range(y.shape[1])])
    assert (ind == ind0).all()


def test_snn_set_difference_quirk(ctx):
    """The reference offers a feature again when an earlier landmark took a higher-numbered one after it
    (std::set_difference on an unsorted range, src/dataAssociation.cpp:106).  Two landmarks predicted at the
    same pixel and features ordered so that landmark 0 takes feature 1 and landmark 1 feature 0: feature 0 is
    offered again to landmark 2 and taken twice."""
    from oracle import pyoracle as orc
    from visualslam_amd import api
    muY = np.array([500.0, 400.0, 900.0, 300.0, 901.0, 301.0])
    SYY = np.diag([10.0, 10.0, 12.0, 12.0, 10.0, 10.0])
    y = np.array([[900.5, 500.2, 1500.0], [300.5, 400.1, 900.0]])
    s0, idx0 = orc.snn_density(muY, SYY, y, 1920, 1080)
    s, idx = api.snn_density(ctx, muY, SYY, y, 1920, 1080)
    assert idx.tolist() == idx0.tolist() == [1, 0, 0]
    assert abs(s - s0) <= 1e-12 * abs(s0)


def test_snn_density_batch_dev(ctx, gcam, ocam):
    """Batch of independent instances through the device-pointer entry point, each against the oracle."""
    import torch
    from visualslam_amd import api
    L, m_extra, B = 6, 4, 5
    mus, Ss, ys, want = [], [], [], []
    for b in range(B):
        sc, y, orc = _density(L, 40 + b, m_extra, 2.0)
        h, SR, Cm, _ = orc.bundle_measure(sc.x, ocam)
        SYY = orc.affine_sqrt(sc.S, Cm, SR)
        mus.append(h); Ss.append(SYY.T.copy()); ys.append(y.T.copy())       # column-major images
        want.append(orc.snn_density(h, SYY, y, 1920, 1080))
    m = ys[0].shape[0]
    dev = torch.device("cuda:0")
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    muY = torch.tensor(np.stack(mus), device=dev)
    SYYt = torch.tensor(np.stack(Ss), device=dev)
    yt = torch.tensor(np.stack(ys), device=dev)
    idx = torch.zeros((B, L), dtype=torch.int32, device=dev)
    sur = torch.zeros(B, dtype=torch.float64, device=dev)
    api.snn_density_batch_dev(tctx, muY, SYYt, yt, L, m, 1920, 1080, idx, sur)
    torch.cuda.synchronize()
    for b in range(B):
        assert idx[b].cpu().numpy().tolist() == want[b][1].tolist()
        assert abs(float(sur[b]) - want[b][0]) <= 1e-9 * abs(want[b][0])
    tctx.close()
