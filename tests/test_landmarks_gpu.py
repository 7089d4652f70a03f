"""GPU parity of landmark removal (vslam_remove_landmark, reference src/SLAM.cpp:796-820) and the Hamming
matcher (vslam_match_hamming*, cv::BFMatcher(NORM_HAMMING).match at src/SLAM.cpp:355) against the CPU
oracle.  The reference has no test for either: parity with it is unpinned; the matcher is integer work
and must equal the oracle bit for bit, the factor to 1e-9."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _row_signed(S):
    d = np.sign(np.diag(S)).copy()
    d[d == 0] = 1.0
    return S * d[:, None]


@pytest.mark.parametrize("L,j", [(1, 0), (5, 0), (5, 4), (50, 17), (150, 149), (150, 0)])
def test_remove_landmark_matches_oracle(ctx, L, j):
    """Scene factors are sparse: after the columns leave, the pivot of a later column can be an exact zero
    of the factor, and the sign of that row of R is then decided by rounding (in Eigen as much as here).
    Rows are therefore compared after normalising the sign of their diagonal entry; the dense case
    below checks the signs themselves."""
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    sc = syn.Scene(L, 40 + L)
    mu0, S0 = orc.remove_landmark(sc.x, sc.S, j)
    mu1, S1 = api.remove_landmark(ctx, sc.x, sc.S, j)
    assert mu1.shape == (sc.n - 3,) and S1.shape == (sc.n - 3, sc.n - 3)
    assert np.array_equal(mu1, mu0)
    assert rel_err(_row_signed(S1), _row_signed(S0)) < 1e-9
    assert not np.tril(S1, -1).any()
    # the marginal covariance of the remaining states is untouched
    keep = [i for i in range(sc.n) if not 12 + 3 * j <= i < 15 + 3 * j]
    P = (sc.S.T @ sc.S)[np.ix_(keep, keep)]
    assert rel_err(S1.T @ S1, P) < 1e-12


@pytest.mark.parametrize("n,j", [(27, 2), (162, 20), (462, 75)])
def test_remove_landmark_dense_factor_signs(ctx, n, j):
    """A dense factor has no degenerate pivots: R must carry the oracle's (= Eigen's / LAPACK's) signs."""
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(n)
    mu = rng.standard_normal(n)
    S = np.triu(rng.standard_normal((n, n))) + 3.0 * np.eye(n)
    mu0, S0 = orc.remove_landmark(mu, S, j)
    mu1, S1 = api.remove_landmark(ctx, mu, S, j)
    assert np.array_equal(mu1, mu0)
    assert rel_err(S1, S0) < 1e-9


@pytest.mark.parametrize("n,j", [(162, 20), (462, 0), (462, 149), (1212, 3)])
def test_remove_landmark_structured_and_dense_paths_agree(ctx, n, j, monkeypatch):
    """Upper-triangular factors take the structured kernels (reflectors over four rows); VSLAM_REMOVE_DENSE=1
    forces the dense QR on the same input.  n = 1212 runs the launch-per-panel form."""
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(7 * n + j)
    mu = rng.standard_normal(n)
    S = np.triu(rng.standard_normal((n, n))) + 3.0 * np.eye(n)
    mu1, S1 = api.remove_landmark(ctx, mu, S, j)
    monkeypatch.setenv("VSLAM_REMOVE_DENSE", "1")
    mu2, S2 = api.remove_landmark(ctx, mu, S, j)
    monkeypatch.delenv("VSLAM_REMOVE_DENSE")
    assert np.array_equal(mu1, mu2)
    assert rel_err(S1, S2) < 1e-11
    assert not np.tril(S1, -1).any()
    if n <= 462:
        assert rel_err(S1, orc.remove_landmark(mu, S, j)[1]) < 1e-9


@pytest.mark.parametrize("upper", [True, False])
def test_remove_landmark_dev_matches_host_entry(ctx, upper):
    """Device-resident form: same result as the host-pointer entry on both routes."""
    import torch
    from visualslam_amd import api
    n, j = 162, 31
    rng = np.random.default_rng(11)
    mu = rng.standard_normal(n)
    S = np.triu(rng.standard_normal((n, n))) + 3.0 * np.eye(n)
    mu0, S0 = api.remove_landmark(ctx, mu, S, j)
    dev = torch.device("cuda:0")
    dmu = torch.tensor(mu, device=dev)
    dS = torch.tensor(np.ascontiguousarray(S.T), device=dev)            # column-major S
    dmo = torch.empty(n - 3, dtype=torch.float64, device=dev)
    dSo = torch.full((n - 3, n - 3), float("nan"), dtype=torch.float64, device=dev)
    api.remove_landmark_dev(ctx, dmu, dS, j, dmo, dSo, s_is_upper=upper)
    ctx.synchronize()
    torch.cuda.synchronize()
    assert np.array_equal(dmo.cpu().numpy(), mu0)
    assert rel_err(dSo.cpu().numpy().T, S0) < 1e-11


def test_remove_landmark_bank(ctx):
    """A bank of filters, each dropping its own landmark (first, last, in between), one with an index outside
    the map: results equal the single-filter entry bit for bit, the bad one is reported and left untouched."""
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api
    n, B = 162, 6
    L = (n - 12) // 3
    js = [0, L - 1, 7, 31, L, 20]                                       # L is outside 0..L-1
    rng = np.random.default_rng(23)
    mus = rng.standard_normal((B, n))
    Ss = np.stack([np.triu(rng.standard_normal((n, n))) + 3.0 * np.eye(n) for _ in range(B)])
    Ss[2] += np.tril(rng.standard_normal((n, n)), -1)                   # the strict lower part must not be read
    dev = torch.device("cuda:0")
    dmu = torch.tensor(mus, device=dev)
    dS = torch.tensor(np.ascontiguousarray(Ss.transpose(0, 2, 1)), device=dev)
    dj = torch.tensor(js, dtype=torch.int32, device=dev)
    dmo = torch.full((B, n - 3), float("nan"), dtype=torch.float64, device=dev)
    dSo = torch.full((B, n - 3, n - 3), float("nan"), dtype=torch.float64, device=dev)
    dst = torch.full((B,), -1, dtype=torch.int32, device=dev)
    api.remove_landmark_batch_dev(ctx, dmu, dS, dj, dmo, dSo, dst)
    ctx.synchronize()
    torch.cuda.synchronize()
    mo, So, st = dmo.cpu().numpy(), dSo.cpu().numpy(), dst.cpu().numpy()
    assert st.tolist() == [0, 0, 0, 0, 1, 0]
    for b in range(B):
        if st[b]:
            assert np.isnan(mo[b]).all() and np.isnan(So[b]).all()
            continue
        mu0, S0 = api.remove_landmark(ctx, mus[b], np.triu(Ss[b]), js[b])
        assert np.array_equal(mo[b], mu0)
        assert np.array_equal(So[b].T, S0)
        mu1, S1 = orc.remove_landmark(mus[b], np.triu(Ss[b]), js[b])        # and the CPU restatement (dense factor: signs are decided)
        assert np.array_equal(mo[b], mu1)
        assert rel_err(So[b].T, np.triu(S1)) < 1e-9


def test_remove_landmark_general_square_factor(ctx):
    """A factor that is not upper triangular goes through the dense QR; the marginal covariance is what counts."""
    from oracle import pyoracle as orc
    from visualslam_amd import api
    n, j = 45, 6
    rng = np.random.default_rng(3)
    S = rng.standard_normal((n, n))
    mu = rng.standard_normal(n)
    mu0, S0 = orc.remove_landmark(mu, S, j)
    mu1, S1 = api.remove_landmark(ctx, mu, S, j)
    assert rel_err(S1, S0) < 1e-9
    keep = [i for i in range(n) if not 12 + 3 * j <= i < 15 + 3 * j]
    assert rel_err(S1.T @ S1, (S.T @ S)[np.ix_(keep, keep)]) < 1e-12


def test_remove_landmark_reference_scenarios(ctx):
    """test/src/deleteLandmarks.cpp:11-156: a state of two landmarks (mu = 1..18, S = diag(1..18)); removing landmark 1
    leaves 15 states, removing landmark 1 then landmark 0 leaves 12 -- the dimensions that test pins, and with this
    diagonal factor also the values (the remaining entries, untouched)."""
    from oracle import pyoracle as orc
    from visualslam_amd import api
    mu = np.arange(1.0, 19.0)
    S = np.diag(np.arange(1.0, 19.0))
    mu1, S1 = api.remove_landmark(ctx, mu, S, 1)
    assert mu1.shape == (15,) and S1.shape == (15, 15)
    assert np.array_equal(mu1, np.arange(1.0, 16.0)) and np.array_equal(S1, np.diag(np.arange(1.0, 16.0)))
    o_mu, o_S = orc.remove_landmark(mu, S, 1)
    assert o_mu.shape == (15,) and o_S.shape == (15, 15) and np.array_equal(o_S, S1)
    mu2, S2 = api.remove_landmark(ctx, mu1, S1, 0)
    assert mu2.shape == (12,) and S2.shape == (12, 12)
    assert np.array_equal(mu2, np.arange(1.0, 13.0)) and np.array_equal(S2, np.diag(np.arange(1.0, 13.0)))


def test_remove_landmark_errors(ctx):
    from visualslam_amd import api
    with pytest.raises(api.VslamError):
        api.remove_landmark(ctx, np.zeros(15), np.eye(15), 1)
    with pytest.raises(api.VslamError):
        api.remove_landmark(ctx, np.zeros(12), np.eye(12), 0)
    with pytest.raises(api.VslamError):
        api.remove_landmark(ctx, np.zeros(16), np.eye(16), 0)


def _descriptors(rng, n, nb=32):
    return rng.integers(0, 256, (n, nb), dtype=np.uint8)


@pytest.mark.parametrize("nq,nt,nb", [(1, 1, 32), (7, 300, 32), (150, 10000, 32), (33, 1000, 64), (9, 50, 4), (20, 257, 36)])
def test_match_hamming_bit_exact(ctx, nq, nt, nb):
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(nq * 131 + nt)
    train = _descriptors(rng, nt, nb)
    # queries: noisy copies of train rows (a few flipped bits), as landmarks re-observed in a new frame
    pick = rng.integers(0, nt, nq)
    query = train[pick].copy()
    flips = rng.integers(0, nb * 8, (nq, 5))
    for q in range(nq):
        for f in flips[q]:
            query[q, f // 8] ^= np.uint8(1 << (f % 8))
    i0, d0 = orc.match_hamming(query, train)
    i1, d1 = api.match_hamming(ctx, query, train)
    assert np.array_equal(i1, i0) and np.array_equal(d1, d0)
    assert np.all(d1 <= 5)


def test_match_hamming_ties_take_the_first_and_empty_train(ctx):
    from visualslam_amd import api
    rng = np.random.default_rng(3)
    base = _descriptors(rng, 1)
    train = np.repeat(base, 600, axis=0)                 # every train descriptor identical
    train[:100] ^= np.uint8(0xFF)                          # the first hundred are far away
    idx, dist = api.match_hamming(ctx, base, train)
    assert idx[0] == 100 and dist[0] == 0
    idx, dist = api.match_hamming(ctx, base, np.zeros((0, 32), np.uint8))
    assert idx[0] == -1 and dist[0] == -1
    with pytest.raises(api.VslamError):
        api.match_hamming(ctx, np.zeros((2, 3), np.uint8), np.zeros((2, 3), np.uint8))


def test_match_hamming_batch_dev(ctx):
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(8)
    B, nq, nt = 5, 40, 700
    train = rng.integers(0, 256, (B, nt, 32), dtype=np.uint8)
    query = rng.integers(0, 256, (B, nq, 32), dtype=np.uint8)
    counts = np.array([700, 1, 350, 0, 699], np.int32)
    dev = torch.device("cuda:0")
    tq, tt = torch.tensor(query, device=dev), torch.tensor(train, device=dev)
    idx = torch.zeros((B, nq), dtype=torch.int32, device=dev)
    dist = torch.zeros((B, nq), dtype=torch.int32, device=dev)
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    api.match_hamming_batch_dev(tctx, tq, tt, idx, dist, torch.tensor(counts, device=dev))
    torch.cuda.synchronize()
    for b in range(B):
        i0, d0 = orc.match_hamming(query[b], train[b, :counts[b]])
        if counts[b] == 0:
            d0[:] = -1
        assert np.array_equal(idx[b].cpu().numpy(), i0) and np.array_equal(dist[b].cpu().numpy(), d0)
    tctx.close()
