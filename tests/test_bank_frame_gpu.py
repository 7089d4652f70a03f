"""One frame of a bank of device-resident filters through the *_batch_dev entry points, stage by stage:
time update -> predicted measurement density -> data association -> measurement update -> landmark removal ->
landmark initialisation.  Every stage consumes the previous stage's device tensors as they are (the layouts must
fit together) and is checked against the single-filter host entry point and against the CPU restatement of the
reference, both fed with the same inputs."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _cm(t):
    """(B, n, n) tensor holding column-major matrices -> numpy matrices."""
    return t.cpu().numpy().transpose(0, 2, 1)


def test_frame_of_a_bank(ctx, gcam, ocam):
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    B, L = 3, 12
    n, m = 12 + 3 * L, 2 * L
    dev = torch.device("cuda:0")
    scs = [syn.Scene(L, 500 + b, dense_S=True) for b in range(B)]
    PT, OT = np.diag([0.1, 0.2, 0.15]), np.diag([0.05, 0.04, 0.06])
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        tctx.set_trs_solver(api.TRS_CHOLESKY)
        mu = torch.tensor(np.stack([s.x for s in scs]), device=dev)
        S = torch.tensor(np.stack([np.ascontiguousarray(s.S.T) for s in scs]), device=dev)

        # 1. time update
        mu1, S1 = torch.empty_like(mu), torch.empty_like(S)
        api.time_update_batch_dev(tctx, mu, S, n, PT, OT, 1 / 30, mu1, S1)
        torch.cuda.synchronize()
        for b in range(B):
            wm, wS = api.time_update(tctx, scs[b].x, scs[b].S, PT, OT, 1 / 30)
            assert rel_err(mu1[b].cpu().numpy(), wm) < 1e-11 and rel_err(_cm(S1)[b], wS) < 1e-9
            om, oS = orc.time_update(scs[b].x, scs[b].S, PT, OT, 1 / 30)
            assert rel_err(mu1[b].cpu().numpy(), om) < 1e-11 and rel_err(_cm(S1)[b], oS) < 1e-9

        # 2. predicted measurement density
        muy = torch.empty((B, m), dtype=torch.float64, device=dev)
        Syy = torch.empty((B, m, m), dtype=torch.float64, device=dev)
        code = torch.empty((B, L), dtype=torch.int32, device=dev)
        api.affine_bundle_batch_dev(tctx, gcam, mu1, S1, muy, Syy, code)
        torch.cuda.synchronize()
        h_mu1, h_S1 = mu1.cpu().numpy(), _cm(S1)
        for b in range(B):
            wy, wSy, wc = api.affine_bundle(tctx, gcam, h_mu1[b], h_S1[b])
            assert np.array_equal(code[b].cpu().numpy(), wc)
            assert rel_err(muy[b].cpu().numpy(), wy) < 1e-11 and rel_err(_cm(Syy)[b], wSy) < 1e-9
            oh, oSR, oC, ocode = orc.bundle_measure(h_mu1[b], ocam)
            oSy = orc.affine_sqrt(h_S1[b], oC, oSR)
            assert np.array_equal(code[b].cpu().numpy(), ocode)
            assert rel_err(muy[b].cpu().numpy(), oh) < 1e-11 and rel_err(_cm(Syy)[b], np.triu(oSy)) < 1e-9

        # 3. data association against features = predicted pixels + noise, shuffled, plus two strays
        rng = np.random.default_rng(1)
        h_muy = muy.cpu().numpy()
        feats = []
        for b in range(B):
            f = h_muy[b].reshape(L, 2) + rng.normal(0.0, 1.5, (L, 2))
            f = np.vstack([f[rng.permutation(L)], [[40.0, 50.0], [1800.0, 1000.0]]])
            feats.append(f)
        mf = L + 2
        yt = torch.tensor(np.stack(feats), device=dev)                     # (B, m, 2) = column-major 2 x m
        idx = torch.empty((B, L), dtype=torch.int32, device=dev)
        sur = torch.empty(B, dtype=torch.float64, device=dev)
        api.snn_density_batch_dev(tctx, muy, Syy, yt, L, mf, gcam.width, gcam.height, idx, sur)
        torch.cuda.synchronize()
        h_idx, h_Syy = idx.cpu().numpy(), _cm(Syy)
        for b in range(B):
            ws, widx = api.snn_density(tctx, h_muy[b], h_Syy[b], feats[b].T, gcam.width, gcam.height)
            assert h_idx[b].tolist() == widx.tolist()
            assert abs(float(sur[b]) - ws) <= 1e-9 * abs(ws)
            os_, oidx = orc.snn_density(h_muy[b], h_Syy[b], feats[b].T, gcam.width, gcam.height)
            assert h_idx[b].tolist() == list(oidx) and abs(float(sur[b]) - os_) <= 1e-9 * abs(os_)
            assert (h_idx[b] >= 0).sum() >= L - 2                           # nearly every landmark finds its feature

        # 4. measurement update with the associated features
        ys = np.zeros((B, m))
        vis = np.zeros((B, L), dtype=np.uint8)
        for b in range(B):
            for j in range(L):
                if h_idx[b, j] >= 0:
                    ys[b, 2 * j:2 * j + 2] = feats[b][h_idx[b, j]]
                    vis[b, j] = 1
        mu2, S2 = torch.empty_like(mu1), torch.empty_like(S1)
        fcode, iters, status = api.measurement_update_iekf_batch_dev(tctx, gcam, mu1, S1, torch.tensor(ys, device=dev),
                                                                     torch.tensor(vis, device=dev), 7.0, mu2, S2)
        torch.cuda.synchronize()
        assert not status.any()
        h_mu2, h_S2 = mu2.cpu().numpy(), _cm(S2)
        for b in range(B):
            _, wm, wS, _, _ = api.measurement_update_iekf(tctx, gcam, h_mu1[b], h_S1[b], ys[b], vis[b], 7.0)
            assert rel_err(h_mu2[b], wm) < 1e-7
            assert rel_err(h_S2[b].T @ h_S2[b], wS.T @ wS) < 1e-6
            assert not np.tril(h_S2[b], -1).any()
            _, om, oS, _, _ = orc.measurement_update_iekf(h_mu1[b], h_S1[b], ys[b], vis[b], ocam, 7.0)
            assert rel_err(h_mu2[b], om) < 1e-7 and rel_err(h_S2[b].T @ h_S2[b], oS.T @ oS) < 1e-6

        # 5. every filter drops a landmark of its own
        js = [0, L - 1, 5]
        mu3 = torch.empty((B, n - 3), dtype=torch.float64, device=dev)
        S3 = torch.empty((B, n - 3, n - 3), dtype=torch.float64, device=dev)
        st = torch.empty(B, dtype=torch.int32, device=dev)
        api.remove_landmark_batch_dev(tctx, mu2, S2, torch.tensor(js, dtype=torch.int32, device=dev), mu3, S3, st)
        torch.cuda.synchronize()
        assert not st.cpu().numpy().any()
        for b in range(B):
            wm, wS = api.remove_landmark(tctx, h_mu2[b], h_S2[b], js[b])
            assert np.array_equal(mu3[b].cpu().numpy(), wm) and np.array_equal(_cm(S3)[b], wS)
            om, oS = orc.remove_landmark(h_mu2[b], h_S2[b], js[b])
            assert np.array_equal(mu3[b].cpu().numpy(), om)
            assert rel_err(np.abs(_cm(S3)[b]), np.abs(np.triu(oS))) < 1e-9      # rows with an exact-zero pivot get their sign from rounding

        # 6. and takes in two new keypoints
        M = 2
        px = np.stack([np.stack([rng.uniform(300, 1600, M), rng.uniform(150, 900, M)], axis=1) for _ in range(B)])
        n4 = n - 3 + 3 * M
        mu4 = torch.empty((B, n4), dtype=torch.float64, device=dev)
        S4 = torch.empty((B, n4, n4), dtype=torch.float64, device=dev)
        api.initialize_landmarks_batch_dev(tctx, gcam, mu3, S3, torch.tensor(px, device=dev), mu4, S4)
        torch.cuda.synchronize()
        h_mu3, h_S3 = mu3.cpu().numpy(), _cm(S3)
        for b in range(B):
            wm, wS = api.initialize_landmarks(tctx, gcam, h_mu3[b], h_S3[b], px[b])
            assert np.array_equal(mu4[b].cpu().numpy(), wm) and np.array_equal(_cm(S4)[b], wS)
    finally:
        tctx.close()
