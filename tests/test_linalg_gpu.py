"""GPU checks of the fp64 MFMA GEMM and the blocked Householder QR.
QR sign convention is pinned by the reference's affineTransform known answers
(test/src/affineTransform.cpp) and equals LAPACK's, so numpy.linalg.qr(mode='r')
is the independent check at sizes the oracle's unblocked QR would take long."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
BOTH_CONTEXTS = True      # every test that takes `ctx` runs on torch's stream and on a stream of the library's own


def _gemm(ctx, A, B, transA=False, transB=False, amode=0, b_upper=False, alpha=1.0, beta=0.0, Cin=None, ksplit=1):
    from visualslam_amd import _lib
    L = _lib.lib()
    A = np.asfortranarray(A, dtype=np.float64)
    B = np.asfortranarray(B, dtype=np.float64)
    M = A.shape[1] if transA else A.shape[0]
    K = A.shape[0] if transA else A.shape[1]
    N = B.shape[0] if transB else B.shape[1]
    out = np.zeros((M, N * max(ksplit, 1)), order="F") if Cin is None else np.array(Cin, dtype=np.float64, order="F")
    fn = L.vslam_debug_gemm
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                   C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int]
    st = fn(ctx.ptr, M, N, K, int(transA), int(transB), amode, int(b_upper), alpha, A.ctypes.data, A.shape[0],
            A.shape[0], A.shape[1], B.ctypes.data, B.shape[0], B.shape[0], B.shape[1], beta, out.ctypes.data, M, ksplit)
    assert st == 0, L.vslam_last_error(ctx.ptr)
    if ksplit > 1:
        return out.reshape(M, N, ksplit, order="F").sum(axis=2)
    return out


@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("M,N,K", [(16, 16, 4), (64, 64, 16), (70, 45, 33), (32, 200, 300), (129, 65, 7), (1, 1, 1)])
def test_gemm_layouts(ctx, ta, tb, M, N, K):
    """Asymmetric integer data: any row/column swap in the MFMA fragment maps shows up exactly."""
    rng = np.random.default_rng(M * 1000 + N * 10 + K)
    A = rng.integers(-9, 10, (K, M) if ta else (M, K)).astype(np.float64)
    B = rng.integers(-9, 10, (N, K) if tb else (K, N)).astype(np.float64)
    want = (A.T if ta else A) @ (B.T if tb else B)
    got = _gemm(ctx, A, B, ta, tb)
    assert np.array_equal(got, want)


def test_gemm_alpha_beta_masks_split(ctx):
    rng = np.random.default_rng(3)
    M, N, K = 150, 90, 40
    V = rng.standard_normal((M, K))
    Vm = np.tril(V, -1) + np.eye(M, K)                        # unit-lower view of V
    B = rng.standard_normal((K, N))
    C0 = rng.standard_normal((M, N))
    got = _gemm(ctx, V, B, amode=1, alpha=-1.0, beta=1.0, Cin=C0)
    assert rel_err(got, C0 - Vm @ B) <= 1e-13
    W = _gemm(ctx, V, rng.standard_normal((M, N)) * 0 + C0, transA=True, amode=1, ksplit=5)
    assert rel_err(W, Vm.T @ C0) <= 1e-13
    U = np.triu(rng.standard_normal((K, K)))
    Ufull = U + np.tril(rng.standard_normal((K, K)), -1)       # garbage below the diagonal must be ignored
    got = _gemm(ctx, Ufull, B, amode=2)
    assert rel_err(got, U @ B) <= 1e-13
    got = _gemm(ctx, V, Ufull, b_upper=True)
    assert rel_err(got, V @ U) <= 1e-13


@pytest.mark.parametrize("m,k", [(2, 2), (5, 3), (33, 32), (64, 64), (97, 31), (30, 50), (462, 462), (762, 300), (1500, 700)])
def test_qr_vs_lapack(ctx, m, k):
    from visualslam_amd import api
    rng = np.random.default_rng(m + k)
    A = rng.standard_normal((m, k))
    R = api.qr_upper(ctx, A)
    R0 = np.linalg.qr(A, mode="r")
    r = min(m, k)
    assert rel_err(R[:r], R0) <= 1e-11
    assert not np.tril(R, -1).any()


@pytest.mark.parametrize("m,k", [(1024, 32), (2500, 1200), (20000, 40), (61000, 33), (70000, 20), (4100, 4100)])
def test_qr_tall_panels_many_workgroups(ctx, m, k):
    """Panels of >= 1024 rows are factored by many workgroups meeting at a counter barrier
    (up to 256 rows each; beyond 240 x 256 rows the single-workgroup kernel takes over)."""
    from visualslam_amd import api
    rng = np.random.default_rng(m * 7 + k)
    A = rng.standard_normal((m, k))
    if k == 40:
        A[:, 7] = 0.0                      # a zero column: tau = 0, R column zero
        A[9:, 8] = 0.0                     # a column that is already triangular keeps its sign
    R = api.qr_upper(ctx, A)
    R0 = np.linalg.qr(A, mode="r")
    r = min(m, k)
    if k == 40:                            # LAPACK flips signs differently on exact zero tails; compare through the oracle
        from oracle import pyoracle as orc
        R0 = orc.qr_upper(A)[:r]
    assert rel_err(R[:r], R0) <= 1e-11
    assert not np.tril(R, -1).any()


def test_qr_config4_size_vs_lapack(ctx):
    """The 6012 x 6012 re-triangularisation of the 2,000-landmark update (column blocks resident in LDS, each launch
    applying the previous block's reflectors) against LAPACK."""
    from visualslam_amd import api
    rng = np.random.default_rng(6012)
    A = rng.standard_normal((6012, 6012))
    R = api.qr_upper(ctx, A)
    R0 = np.linalg.qr(A, mode="r")
    assert rel_err(R, R0) <= 1e-11
    assert np.array_equal(np.sign(np.diag(R)), np.sign(np.diag(R0)))
    assert not np.tril(R, -1).any()


@pytest.mark.parametrize("m,k", [(500, 64), (3000, 200)])
def test_qr_panels_that_leave_the_fast_path(ctx, m, k):
    """Panels the Cholesky-QR + reconstruction path must hand to the column-by-column path: a column that nearly
    repeats another (the Gram matrix of the panel is numerically singular), a pivot that is exactly zero (the
    reflector sign comes from the convention, not from the data), a column that is already triangular."""
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(m + 7 * k)
    A = rng.standard_normal((m, k))
    A[:, 40] = A[:, 3] + 1e-11 * rng.standard_normal(m)       # nearly dependent inside one panel
    A[0, 0] = 0.0                                             # zero pivot, non-zero tail
    A[34:, 33] = 0.0                                          # zero tail below the diagonal of column 33 ... after the
    A[:33, 33] = 0.0                                          # ... earlier reflections have nothing to mix in
    A[33, 33] = -2.5
    R = api.qr_upper(ctx, A)
    R0 = orc.qr_upper(A)[:k]
    assert rel_err(R[:k].T @ R[:k], A.T @ A) <= 1e-12
    good = np.abs(np.diag(R0)) > 1e-6
    assert np.array_equal(np.sign(np.diag(R)[good]), np.sign(np.diag(R0)[good]))
    rows = np.flatnonzero(good)
    rows = rows[rows < 40]                                    # rows before the nearly dependent column are well determined
    assert rel_err(R[rows][:, :40], R0[rows][:, :40]) <= 1e-9


def test_two_contexts_factor_large_matrices_at_once(ctx):
    """Two contexts on their own streams each queue a 4100 x 4100 factorisation without waiting for the other.  The
    workgroups of such a factorisation wait for one another; the library queues the two one after the other on the
    GPU instead of letting them compete for compute units."""
    import torch
    from visualslam_amd import api
    n = 4100
    rng = np.random.default_rng(41)
    As = [rng.standard_normal((n, n)) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ctxs = [api.Context(0, stream=s_.cuda_stream) for s_ in streams]
    ts = [torch.tensor(np.ascontiguousarray(A_.T), device="cuda:0") for A_ in As]       # column-major n x n
    torch.cuda.synchronize()
    for rep in range(2):                                     # second round: workspaces exist, launches go back to back
        for c_, t_ in zip(ctxs, ts):
            api.qr_upper_batch_dev(c_, t_, n, n, n, 1, n * n)
        if rep == 0:
            torch.cuda.synchronize()
            for A_, t_ in zip(As, ts):
                t_.copy_(torch.tensor(np.ascontiguousarray(A_.T)))
            torch.cuda.synchronize()
    for c_ in ctxs:
        c_.synchronize()
    for A_, t_ in zip(As, ts):
        R0 = np.linalg.qr(A_, mode="r")
        assert rel_err(t_.cpu().numpy().T, R0) <= 1e-11
    for c_ in ctxs:
        c_.close()


def test_two_contexts_solve_large_triangular_systems_at_once(ctx):
    """Two host threads, each with a context of its own, queue chained triangular solves of 9000 unknowns (139 workgroups
    that wait for one another per solve: two at once do not fit the chip).  The library queues such launches one after
    the other on the GPU; both threads get their solutions."""
    import threading
    import scipy.linalg as sla
    from visualslam_amd import _lib, api
    L = _lib.lib()
    fn = L.vslam_debug_trsv_reps
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    n = 9000
    rng = np.random.default_rng(90)
    jobs = []
    for t in range(2):
        U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + np.diag(1.0 + rng.random(n))
        jobs.append({"U": U, "S": np.ascontiguousarray(U.T), "b": rng.standard_normal(n), "z": np.zeros(n), "ctx": api.Context(0), "rc": None, "trans": t})
    def run(j):
        j["rc"] = fn(j["ctx"].ptr, j["S"].ctypes.data, n, j["trans"], j["b"].ctypes.data, j["z"].ctypes.data, 40)
    threads = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for j in jobs:
        assert j["rc"] == 0, L.vslam_last_error(j["ctx"].ptr)
        z0 = sla.solve_triangular(j["U"], j["b"], trans=j["trans"], lower=False)
        assert rel_err(j["z"], z0) <= 1e-11
        j["ctx"].close()


def test_small_chained_solves_beside_a_full_size_factorisation(ctx):
    """A 6012 x 6012 factorisation (94 workgroups that wait for one another, 47 launches) from one context while two more
    contexts on threads of their own run short chained solves (462 unknowns: 6 workgroups that wait for one another) --
    launches small enough to run without taking the GPU's turn when their context is alone on the device.  With several
    contexts on the device they do take it, so no launch can be left waiting for members that found no compute unit."""
    import threading
    import scipy.linalg as sla
    import torch
    from visualslam_amd import _lib, api
    L = _lib.lib()
    fn = L.vslam_debug_trsv_reps
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    n, m = 6012, 462
    rng = np.random.default_rng(7)
    A = rng.standard_normal((n, n))
    stream = torch.cuda.Stream()
    qctx = api.Context(0, stream=stream.cuda_stream)
    t = torch.tensor(np.ascontiguousarray(A.T), device="cuda:0")
    torch.cuda.synchronize()
    jobs = []
    for k in range(2):
        U = np.triu(rng.standard_normal((m, m))) / np.sqrt(m) + np.diag(1.0 + rng.random(m))
        jobs.append({"U": U, "S": np.ascontiguousarray(U.T), "b": rng.standard_normal(m), "z": np.zeros(m), "ctx": api.Context(0), "rc": None, "trans": k})
    def solve(j):
        j["rc"] = fn(j["ctx"].ptr, j["S"].ctypes.data, m, j["trans"], j["b"].ctypes.data, j["z"].ctypes.data, 400)
    qrc = {}
    def factor():
        try:
            api.qr_upper_batch_dev(qctx, t, n, n, n, 1, n * n)
            qctx.synchronize()
            qrc["ok"] = True
        except Exception as exc:           # surfaced below
            qrc["ok"] = repr(exc)
    threads = [threading.Thread(target=factor)] + [threading.Thread(target=solve, args=(j,)) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert qrc.get("ok") is True, qrc
    assert rel_err(t.cpu().numpy().T, np.linalg.qr(A, mode="r")) <= 1e-11
    for j in jobs:
        assert j["rc"] == 0, L.vslam_last_error(j["ctx"].ptr)
        assert rel_err(j["z"], sla.solve_triangular(j["U"], j["b"], trans=j["trans"], lower=False)) <= 1e-11
        j["ctx"].close()
    qctx.close()


def test_qr_tall_batched_many_workgroups(ctx):
    import torch
    from visualslam_amd import api
    rng = np.random.default_rng(19)
    B, m, k = 3, 3000, 70
    A = rng.standard_normal((B, k, m))
    t = torch.tensor(A, device="cuda:0")
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    api.qr_upper_batch_dev(tctx, t, m, k, m, B, m * k)
    tctx.synchronize()
    out = t.cpu().numpy()
    for b in range(B):
        R0 = np.linalg.qr(A[b].T, mode="r")
        assert rel_err(out[b].T[:k], R0) <= 1e-11
    tctx.close()


def test_qr_vs_oracle_and_edge_cases(ctx):
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(0)
    A = rng.standard_normal((90, 70))
    assert rel_err(api.qr_upper(ctx, A), orc.qr_upper(A)) <= 1e-11
    # already upper triangular with zero tails: columns keep their sign (tau = 0)
    U = np.triu(rng.standard_normal((40, 40)))
    assert rel_err(api.qr_upper(ctx, U), orc.qr_upper(U)) <= 1e-12
    # rank-deficient / zero matrix
    Z = np.zeros((20, 10))
    assert not api.qr_upper(ctx, Z).any()
    # sqrt-covariance-like input: R^T R = A^T A
    S = np.triu(rng.standard_normal((200, 200))) * 0.01 + np.diag(rng.uniform(0.001, 0.5, 200))
    G = rng.standard_normal((260, 200)) @ S
    R = api.qr_upper(ctx, G)
    assert rel_err(R[:200].T @ R[:200], G.T @ G) <= 1e-11


def test_affine_kats_on_gpu(ctx):
    """test/src/affineTransform.cpp:59-64,125-130 through the device QR."""
    from visualslam_amd import api
    for mux, Sxx, Syy in (([1., 3.], [10., 1.], [[-3.00171694492, -3.12886418774], [0, -1.05839921328]]),
                          ([1., 1.], [1.5, 1.8], [[-1.17166745172, 0.298734812658], [0, -1.63271476741]])):
        x = np.array(mux)
        n2 = x @ x
        J = np.array([[x[1] / n2, -x[0] / n2], [x[0] / np.sqrt(n2), x[1] / np.sqrt(n2)]])
        SR = np.diag([np.pi / 180, 0.1])
        A = np.vstack([np.diag(Sxx) @ J.T, SR])
        R = api.qr_upper(ctx, A)
        assert np.allclose(R[:2], np.array(Syy), rtol=1e-10, atol=1e-12)


def test_qr_bank_of_tall_matrices(ctx):
    """A bank of [S C^T; SR]-sized matrices (762 x 300, the predicted measurement density of 150 landmarks): two
    workgroups per matrix, the bank cut into launches whose workgroups are all resident together."""
    import torch
    from visualslam_amd import api
    rng = np.random.default_rng(23)
    B, m, k = 130, 762, 300
    A = rng.standard_normal((B, k, m))
    A[7, 5, :] = 0.0                                          # instance 7: a zero column
    t = torch.tensor(A, device="cuda:0")
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    api.qr_upper_batch_dev(tctx, t, m, k, m, B, m * k)
    tctx.synchronize()
    out = t.cpu().numpy()
    from oracle import pyoracle as orc
    for b in (0, 1, 7, 64, 119, 120, 129):
        R0 = orc.qr_upper(A[b].T)[:k] if b == 7 else np.linalg.qr(A[b].T, mode="r")
        assert rel_err(out[b].T[:k], R0) <= 1e-11
        assert not np.tril(out[b].T, -1).any()
    tctx.close()


def test_qr_batched_dev(ctx):
    import torch
    from visualslam_amd import api
    rng = np.random.default_rng(9)
    B, m, k = 5, 130, 100
    A = rng.standard_normal((B, k, m))                        # each instance column-major (m x k)
    t = torch.tensor(A, device="cuda:0")
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    api.qr_upper_batch_dev(tctx, t, m, k, m, B, m * k)
    torch.cuda.synchronize()
    out = t.cpu().numpy()
    for b in range(B):
        R0 = np.linalg.qr(A[b].T, mode="r")
        assert rel_err(out[b].T[:k], R0) <= 1e-11
    tctx.close()


def _trtri(ctx, S):
    from visualslam_amd import _lib
    L = _lib.lib()
    S = np.ascontiguousarray(S, dtype=np.float64)          # [batch][col][row]: each matrix column-major
    batch, n = (1, S.shape[0]) if S.ndim == 2 else (S.shape[0], S.shape[1])
    W = np.full_like(S, np.nan)
    fn = L.vslam_debug_trtri
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    assert fn(ctx.ptr, S.ctypes.data, n, batch, W.ctypes.data) == 0, L.vslam_last_error(ctx.ptr)
    return W


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 128, 130, 200, 462, 1000, 2100])
def test_triangular_inverse(ctx, n):
    """Every size class of the bottom-up inverse: one block, a ragged second block, full and ragged pairs on several
    levels; rectangular triangular-times-dense products with the zero stretch of K skipped."""
    rng = np.random.default_rng(n)
    U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + np.diag(1.0 + rng.random(n))
    U_low_garbage = U + np.tril(rng.standard_normal((n, n)), -1)       # the strict lower triangle must not be read
    W = _trtri(ctx, np.ascontiguousarray(U_low_garbage.T)).T           # column-major on the device
    assert not np.tril(W, -1).any()
    assert rel_err(W @ U, np.eye(n)) <= 1e-12
    assert rel_err(W, np.linalg.inv(U)) <= 1e-11


def test_triangular_inverse_bank(ctx):
    rng = np.random.default_rng(11)
    B, n = 5, 333
    U = np.triu(rng.standard_normal((B, n, n))) / np.sqrt(n) + np.eye(n) * (1.0 + rng.random((B, n, 1)))
    W = _trtri(ctx, np.ascontiguousarray(np.transpose(U, (0, 2, 1))))
    W = np.transpose(W, (0, 2, 1))
    for b in range(B):
        assert rel_err(W[b], np.linalg.inv(U[b])) <= 1e-11
        assert not np.tril(W[b], -1).any()


def _trsv(ctx, S_colmajor, trans, b):
    from visualslam_amd import _lib
    L = _lib.lib()
    S = np.ascontiguousarray(S_colmajor, dtype=np.float64)
    n = S.shape[0]
    b = np.ascontiguousarray(b, dtype=np.float64)
    z = np.full(n, np.nan)
    fn = L.vslam_debug_trsv
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    assert fn(ctx.ptr, S.ctypes.data, n, int(trans), b.ctypes.data, z.ctypes.data) == 0, L.vslam_last_error(ctx.ptr)
    return z


@pytest.mark.parametrize("n", [1, 5, 64, 65, 127, 128, 129, 200, 256, 257, 320, 462, 1000, 2100, 6012])
@pytest.mark.parametrize("trans", [False, True])
def test_triangular_vector_solve(ctx, n, trans):
    """One block, every count of blocks around the reach of the chain workgroup (2 to 5 blocks: with and without
    helper workgroups), ragged last blocks in both directions, and the sizes of the measurement update."""
    import scipy.linalg as sla
    rng = np.random.default_rng(3 * n + int(trans))
    U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + np.diag((1.0 + rng.random(n)) * rng.choice([-1.0, 1.0], n))
    garbage = U + np.tril(rng.standard_normal((n, n)), -1)             # the strict lower triangle must not be read
    b = rng.standard_normal(n)
    z = _trsv(ctx, np.ascontiguousarray(garbage.T), trans, b)
    z0 = sla.solve_triangular(U, b, trans=1 if trans else 0, lower=False)
    assert rel_err(z, z0) <= 1e-11
    res = (U.T @ z if trans else U @ z) - b
    assert np.abs(res).max() <= 1e-11 * max(1.0, np.abs(z).max()) * np.sqrt(n)


@pytest.mark.parametrize("n", [462, 1000])
@pytest.mark.parametrize("trans", [False, True])
def test_triangular_vector_solve_returns_nans_for_a_nan_right_hand_side(ctx, n, trans):
    """The chained solve marks an unpublished value with a NaN of all-ones bits.  A right-hand side that carries this
    very NaN must still give a finished call whose dependent entries are NaN (as the one-launch-per-block path returns
    them), not a reader polling to its bound and a device error."""
    rng = np.random.default_rng(n)
    U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + np.diag(1.0 + rng.random(n))
    b = rng.standard_normal(n)
    k = n // 2
    b[k] = np.frombuffer(np.array([-1], dtype=np.int64).tobytes(), dtype=np.float64)[0]
    z = _trsv(ctx, np.ascontiguousarray(U.T), trans, b)                    # asserts status 0
    dep = np.arange(n) >= k if trans else np.arange(n) <= k                # entries the NaN can reach
    assert np.isnan(z[k]) and np.isnan(z[dep]).all()
    # entries the NaN cannot reach stay finite and unchanged -- outside its own 64-wide diagonal block, inside which
    # the substitution multiplies masked-out coefficients (zeros) with every entry
    blk = k // 64
    free = np.arange(n) < 64 * blk if trans else np.arange(n) >= 64 * (blk + 1)
    assert np.isfinite(z[free]).all()
    clean = b.copy()
    clean[k] = 0.0
    z0 = _trsv(ctx, np.ascontiguousarray(U.T), trans, clean)
    assert np.array_equal(z[free], z0[free])


def test_triangular_vector_solve_many_calls_in_a_row(ctx):
    """The solutions and helper sums a call publishes go to a fresh stretch of a ring that is refilled with the
    not-ready pattern when it runs out: enough calls to go round it more than once, every result the same bits."""
    import scipy.linalg as sla
    n = 4000                                                # 2 * 63 * 64 values per call, a ring of 2^20: refilled every 130 calls
    rng = np.random.default_rng(77)
    U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + np.diag(1.0 + rng.random(n))
    S = np.ascontiguousarray(U.T)
    b = rng.standard_normal(n)
    first = [_trsv(ctx, S, t, b) for t in (False, True)]
    for t in (False, True):
        assert rel_err(first[t], sla.solve_triangular(U, b, trans=1 if t else 0, lower=False)) <= 1e-11
    for it in range(150):
        for t in (False, True):
            assert np.array_equal(_trsv(ctx, S, t, b), first[t]), (it, t)


@pytest.mark.parametrize("n", [65, 128, 200, 462, 1000, 1024])
@pytest.mark.parametrize("trans", [False, True])
def test_triangular_vector_solve_bank(ctx, n, trans):
    """One workgroup per matrix: every solution block stays in LDS, the far contributions are added up a link early.
    Masked matrices keep their output; the scale factor reaches the result only."""
    import scipy.linalg as sla
    from visualslam_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5 * n + int(trans))
    B = 7
    U = np.triu(rng.standard_normal((B, n, n))) / np.sqrt(n) + np.eye(n) * ((1.0 + rng.random((B, n, 1))) * rng.choice([-1.0, 1.0], (B, n, 1)))
    S = np.ascontiguousarray(np.transpose(U + np.tril(rng.standard_normal((B, n, n)), -1), (0, 2, 1)))
    b = rng.standard_normal((B, n))
    z = np.full((B, n), 123.0)
    active = np.array([1, 1, 0, 1, 1, 1, 0], dtype=np.int32)
    fn = L.vslam_debug_trsv_bank
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    assert fn(ctx.ptr, S.ctypes.data, n, B, int(trans), b.ctypes.data, z.ctypes.data, -2.0, active.ctypes.data) == 0, L.vslam_last_error(ctx.ptr)
    for m in range(B):
        if not active[m]:
            assert (z[m] == 123.0).all()
            continue
        z0 = -2.0 * sla.solve_triangular(U[m], b[m], trans=1 if trans else 0, lower=False)
        assert rel_err(z[m], z0) <= 1e-11, m


def _potrf(ctx, A, rhs=None):
    from visualslam_amd import _lib
    L = _lib.lib()
    A = np.ascontiguousarray(A, dtype=np.float64).copy()   # symmetric input: row-major and column-major coincide
    batch, n = (1, A.shape[0]) if A.ndim == 2 else (A.shape[0], A.shape[1])
    info = np.full(batch, -1, dtype=np.int32)
    r = None if rhs is None else np.ascontiguousarray(rhs, dtype=np.float64).copy()
    fn = L.vslam_debug_potrf
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    assert fn(ctx.ptr, A.ctypes.data, n, batch, info.ctypes.data, None if r is None else r.ctypes.data) == 0, L.vslam_last_error(ctx.ptr)
    U = np.swapaxes(A, -1, -2)                              # the device wrote column-major factors
    return (U, info) if rhs is None else (U, info, r)


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 96, 97, 128, 130, 160, 161, 192, 193, 200, 224, 225, 256, 462, 1000, 2100])
def test_cholesky_vs_lapack(ctx, n):
    """Blocked upper Cholesky: first block by the stand-alone kernel, later diagonal blocks by the workgroup of the
    trailing update that produced them, row panels by the MFMA substitution; the sizes end a factorisation in every
    position of a block step.  With a right-hand side riding along as one more column, it returns as U^-T r
    (in a tile column of its own when the trailing block fills its tiles, inside the last tile column otherwise)."""
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n + 3))
    H = G @ G.T / n + np.eye(n)
    U, info = _potrf(ctx, H)
    assert info[0] == 0
    assert not np.tril(U, -1).any()
    U0 = np.linalg.cholesky(H).T
    assert rel_err(U, U0) <= 1e-12
    assert rel_err(U.T @ U, H) <= 1e-13
    r = rng.standard_normal(n)
    U2, info2, t = _potrf(ctx, H, r)
    assert info2[0] == 0 and np.array_equal(U2, U)
    assert rel_err(t, np.linalg.solve(U0.T, r)) <= 1e-12


def test_cholesky_bank_and_indefinite_matrix(ctx):
    rng = np.random.default_rng(5)
    B, n = 6, 300
    G = rng.standard_normal((B, n, n))
    H = G @ np.transpose(G, (0, 2, 1)) / n + np.eye(n)
    H[2, 170, 170] = -1.0                                   # pivot 171 of matrix 2 is not positive
    H[4, 10, 10] = -50.0
    r = rng.standard_normal((B, n))
    U, info, t = _potrf(ctx, H, r)
    assert info.tolist()[:2] == [0, 0] and info[3] == 0 and info[5] == 0
    assert info[2] == 171 and info[4] == 11
    for b in (0, 1, 3, 5):
        U0 = np.linalg.cholesky(H[b]).T
        assert rel_err(U[b], U0) <= 1e-12
        assert rel_err(t[b], np.linalg.solve(U0.T, r[b])) <= 1e-12


@pytest.mark.parametrize("n", [40, 64, 65, 200, 462, 1000])
def test_cholesky_of_a_shifted_matrix_without_a_copy(ctx, n):
    """The factorisation reads H + shift I where H lies (first block row copied, the first trailing update takes the rest
    from H); entries below the diagonal of U are not meaningful; with a list of matrices only those are touched."""
    from visualslam_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(n)
    B = 5
    M = rng.standard_normal((B, n, n))
    H = M @ np.transpose(M, (0, 2, 1)) / n + 0.1 * np.eye(n)
    shift = np.array([0.0, 0.5, 2.0, 0.0, 7.0])
    U = np.full((B, n, n), -7.0)
    info = np.full(B, -1, dtype=np.int32)
    sel = np.array([4, 0, 2], dtype=np.int32)
    fn = L.vslam_debug_potrf_from
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    Hc = np.ascontiguousarray(H)
    assert fn(ctx.ptr, Hc.ctypes.data, n, B, shift.ctypes.data, info.ctypes.data, U.ctypes.data, sel.ctypes.data, len(sel)) == 0, L.vslam_last_error(ctx.ptr)
    for b in range(B):
        Ub = U[b].T                                         # column-major on the device
        if b not in sel:
            assert (Ub == -7.0).all()
            continue
        assert info[b] == 0
        U0 = np.linalg.cholesky(H[b] + shift[b] * np.eye(n)).T
        assert rel_err(np.triu(Ub), U0) <= 1e-12
    U2 = np.full((B, n, n), -7.0)
    assert fn(ctx.ptr, Hc.ctypes.data, n, B, shift.ctypes.data, info.ctypes.data, U2.ctypes.data, None, 0) == 0
    for b in range(B):
        assert rel_err(np.triu(U2[b].T), np.linalg.cholesky(H[b] + shift[b] * np.eye(n)).T) <= 1e-12
