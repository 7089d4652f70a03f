"""Dense fp64 leg of bench.py (rank 0, one GPU): the square-root re-triangularisation QR of the
2,000-landmark state (6012 x 6012, device resident) against the fp64 matrix-core peak, and the
wall time of one complete measurement update (150 landmarks) on both trust-region routes, and the
per-frame steps either side of it (time update, data association, descriptor matching)."""
import time

import numpy as np

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense fp64 matrix peak


def run_dense_leg(ctx, dev, stream):
    import torch
    from visualslam_amd import api, synthetic as syn

    n = 6012
    gen = torch.Generator(device=dev)
    gen.manual_seed(4)
    A0 = torch.randn((n, n), dtype=torch.float64, device=dev, generator=gen)      # column-major n x n
    A = A0.clone()
    api.qr_upper_batch_dev(ctx, A, n, n, n, 1, n * n)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        A.copy_(A0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        api.qr_upper_batch_dev(ctx, A, n, n, n, 1, n * n)
        e1.record(stream)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    flops = 4.0 / 3.0 * n ** 3
    tf = flops / (ms * 1e-3) / 1e12
    del A, A0
    # host LAPACK (numpy.linalg.qr, R only, whatever threads its BLAS uses) on a matrix of the same size
    ns = n
    As = np.random.default_rng(4).standard_normal((ns, ns))
    t0 = time.perf_counter()
    np.linalg.qr(As, mode="r")
    cpu_s = time.perf_counter() - t0
    cpu_tf = 4.0 / 3.0 * ns ** 3 / cpu_s / 1e12
    del As
    try:
        import threadpoolctl
        threads = max([p_.get("num_threads", 1) for p_ in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        threads = None

    cam = api.camera_from_xml(syn.CAMERA_XML)
    sc = syn.Scene(150, 3, dense_S=True)
    hctx = api.Context(dev.index or 0)
    upd = {}
    api.measurement_update_iekf(hctx, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
    t0 = time.perf_counter()
    api.measurement_update_iekf(hctx, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
    upd["default_route_ms"] = (time.perf_counter() - t0) * 1e3
    upd["default_route"] = "cholesky (device)" if hctx.last_trs_route() == api.TRS_CHOLESKY else "eigen (host)"
    for name, solver in (("cholesky_route_ms", api.TRS_CHOLESKY), ("eigen_route_ms", api.TRS_EIG)):
        hctx.set_trs_solver(solver)
        api.measurement_update_iekf(hctx, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
        t0 = time.perf_counter()
        out = api.measurement_update_iekf(hctx, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
        upd[name] = (time.perf_counter() - t0) * 1e3
        upd["iterations"] = out[3]
    # BASELINE configs[3]: the full trust-region Newton update of the 2,000-landmark scene (n = 6012), default route
    c4 = None
    try:
        sc4 = syn.Scene(2000, 4, dense_S=True)
        hctx.set_trs_solver(api.TRS_CHOLESKY)
        api.measurement_update_iekf(hctx, cam, sc4.x, sc4.S, sc4.y, sc4.visible, sc4.sigma)          # warm-up (workspace, code objects)
        t0 = time.perf_counter()
        out4 = api.measurement_update_iekf(hctx, cam, sc4.x, sc4.S, sc4.y, sc4.visible, sc4.sigma)
        wall4 = (time.perf_counter() - t0) * 1e3
        route4 = hctx.last_trs_route()
        prof = hctx.last_update_profile() if route4 == api.TRS_CHOLESKY else None
        n4 = sc4.n
        c4 = {"workload": "measurementUpdateIEKF, 2000 landmarks, n = %d, host-pointer entry (PCIe copies of the %d MB factor in and out included)" % (n4, n4 * n4 * 8 // 1000000),
              "wall_ms": wall4, "route": "cholesky (device)" if route4 == api.TRS_CHOLESKY else "eigen (host)",
              "fmin_code": int(out4[0]), "iterations": int(out4[3]), "cost_evaluations": int(out4[4]), "breakdown": prof}
        if prof is not None:
            # algorithmic flops of the pieces: prior Hessian once (triangular inverse n^3/3 + W W^T 2/3 n^3), one Cholesky
            # factorisation n^3/3 per trust-region solve, posterior = one factorisation + one triangular inverse
            fl_cost = 1.0 * n4 ** 3
            fl_trs = prof["factorisations"] * n4 ** 3 / 3.0
            fl_post = 2.0 * n4 ** 3 / 3.0
            pieces = {"cost_evaluations": (fl_cost, prof["cost_evaluations_ms"]), "trust_region": (fl_trs, prof["trust_region_ms"]),
                      "posterior_factor": (fl_post, prof["posterior_factor_ms"])}
            dom = max(pieces, key=lambda k_: pieces[k_][1])
            c4["pieces"] = {k_: {"algorithmic_flops": v_[0], "ms": v_[1], "tflops": v_[0] / (v_[1] * 1e-3) / 1e12 if v_[1] > 0 else None} for k_, v_ in pieces.items()}
            tf4 = pieces[dom][0] / (pieces[dom][1] * 1e-3) / 1e12
            c4["roofline"] = {"bound": "mfma", "piece": dom, "achieved": tf4, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": tf4 / FP64_MFMA_PEAK_TFLOPS, "traffic": None}
        # the same update with the state resident in HBM (the bank entry point with one filter): what a caller pays who
        # keeps the filter on the GPU between frames
        try:
            dv4 = torch.device("cuda:0")
            mu4 = torch.tensor(sc4.x[None], device=dv4)
            S4 = torch.tensor(np.ascontiguousarray(sc4.S.T)[None], device=dv4)
            y4 = torch.tensor(sc4.y[None], device=dv4)
            v4 = torch.tensor(sc4.visible[None].astype(np.uint8), device=dv4)
            mp4, Sp4 = torch.empty_like(mu4), torch.empty_like(S4)
            torch.cuda.synchronize()
            api.measurement_update_iekf_batch_dev(hctx, cam, mu4, S4, y4, v4, sc4.sigma, mp4, Sp4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, it4, st4 = api.measurement_update_iekf_batch_dev(hctx, cam, mu4, S4, y4, v4, sc4.sigma, mp4, Sp4)
            torch.cuda.synchronize()
            c4["device_resident_ms"] = (time.perf_counter() - t0) * 1e3
            c4["device_resident_agrees"] = bool(int(st4[0]) == 0 and int(it4[0]) == int(out4[3]) and
                                                float(np.abs(mp4[0].cpu().numpy() - out4[1]).max()) <= 1e-9 * max(1.0, float(np.abs(out4[1]).max())))
            # the two steps in front of the update, device resident as well: predicted measurement density and association
            L4 = (n4 - 12) // 3
            muy4 = torch.zeros((1, 2 * L4), dtype=torch.float64, device=dv4)
            Syy4 = torch.zeros((1, 2 * L4, 2 * L4), dtype=torch.float64, device=dv4)
            rng4 = np.random.default_rng(0)
            f4 = np.concatenate([sc4.pixel_true + rng4.standard_normal((L4, 2)), rng4.uniform(0, 1000, (100, 2))])
            yt4 = torch.tensor(f4[None], device=dv4)
            idx4 = torch.zeros((1, L4), dtype=torch.int32, device=dv4)
            sur4 = torch.zeros(1, dtype=torch.float64, device=dv4)
            for label, fn in (("affine_bundle_device_resident_ms", lambda: api.affine_bundle_batch_dev(hctx, cam, mu4, S4, muy4, Syy4)),
                              ("snn_density_device_resident_ms", lambda: api.snn_density_batch_dev(hctx, muy4, Syy4, yt4, L4, f4.shape[0], 1920, 1080, idx4, sur4))):
                fn(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn(); torch.cuda.synchronize()
                c4[label] = (time.perf_counter() - t0) * 1e3
            c4["snn_density_associated"] = int((idx4 >= 0).sum().item())
            del muy4, Syy4
            del mu4, S4, mp4, Sp4
        except Exception as exc:
            c4["device_resident_error"] = repr(exc)
        # CPU: the oracle's update (analytic derivatives, scalar C, one thread) on the 150-landmark scene; the n = 6012
        # update is O(n^3) per iteration there (host eigensolver) and is extrapolated, not run
        from oracle import pyoracle as orc
        cd = syn.read_camera_xml()
        ocam = orc.make_camera(cd["K"][0, 0], cd["K"][1, 1], cd["K"][0, 2], cd["K"][1, 2], cd["dist"], cd["flag"], cd["width"], cd["height"])
        t0 = time.perf_counter()
        o_ret = orc.measurement_update_iekf(sc.x, sc.S, sc.y, sc.visible, ocam, sc.sigma)
        o_s = time.perf_counter() - t0
        c4["cpu_baseline"] = {"value": 1.0 / o_s, "unit": "updates/s (150 landmarks, n = 462)", "cores": 1, "kind": "port",
                              "sample": "one complete oracle update at 150 landmarks, %.2f s, %d iterations; the 2000-landmark update was not run "
                                        "on the CPU: extrapolated by (6012/462)^3 it is about %.0f s" % (o_s, o_ret[3], o_s * (6012.0 / 462.0) ** 3),
                              "extrapolated_seconds_at_2000_landmarks": o_s * (6012.0 / 462.0) ** 3}
        del sc4
    except Exception as exc:
        c4 = {"error": repr(exc)}
    # the per-frame steps either side of the update: time update, data association, descriptor matching
    pt, ot = np.diag([0.5, 0.5, 0.175]), np.diag([0.05, 0.05, 0.05])
    frame = {}
    x = sc.x.copy()
    x[:6] = 0.2 * np.random.default_rng(1).standard_normal(6)
    api.time_update(hctx, x, sc.S, pt, ot, 1 / 30)
    t0 = time.perf_counter()
    api.time_update(hctx, x, sc.S, pt, ot, 1 / 30)
    frame["time_update_ms"] = (time.perf_counter() - t0) * 1e3
    B = 512
    mu_b = torch.tensor(x, device=dev).repeat(B, 1).contiguous()
    S_b = torch.tensor(np.ascontiguousarray(sc.S.T), device=dev).repeat(B, 1, 1).contiguous()
    mu_o, S_o = torch.empty_like(mu_b), torch.empty_like(S_b)
    api.time_update_batch_dev(ctx, mu_b, S_b, sc.n, pt, ot, 1 / 30, mu_o, S_o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    api.time_update_batch_dev(ctx, mu_b, S_b, sc.n, pt, ot, 1 / 30, mu_o, S_o)
    e1.record(stream)
    torch.cuda.synchronize()
    frame["time_update_512_filters_ms"] = e0.elapsed_time(e1)
    del mu_b, S_b, mu_o, S_o
    rng = np.random.default_rng(80)
    feats = np.concatenate([sc.pixel_true + 2.0 * rng.standard_normal((sc.L, 2)),
                            np.stack([rng.uniform(0, 1920, sc.L), rng.uniform(0, 1080, sc.L)], axis=1)])
    yf = feats[rng.permutation(feats.shape[0])].T.copy()
    api.snn(hctx, cam, sc.x, sc.S, yf)
    t0 = time.perf_counter()
    api.snn(hctx, cam, sc.x, sc.S, yf)
    frame["snn_ms"] = (time.perf_counter() - t0) * 1e3
    Bm, nq, nt = 64, 150, 10000                         # 64 frames: 150 landmark descriptors against 10,000 ORB features
    gen.manual_seed(6)
    tq = torch.randint(0, 256, (Bm, nq, 32), dtype=torch.uint8, device=dev, generator=gen)
    tt = torch.randint(0, 256, (Bm, nt, 32), dtype=torch.uint8, device=dev, generator=gen)
    mi = torch.zeros((Bm, nq), dtype=torch.int32, device=dev)
    api.match_hamming_batch_dev(ctx, tq, tt, mi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10):
        api.match_hamming_batch_dev(ctx, tq, tt, mi)
    e1.record(stream)
    torch.cuda.synchronize()
    mms = e0.elapsed_time(e1) / 10
    frame["match_hamming_64_frames_ms"] = mms
    # selection of new landmarks: 64 frames x 10,000 sorted keypoints against 100 map landmarks, room for 50 new ones
    Ks, Ls = 10000, 100
    rs = np.random.default_rng(8)
    fz = np.zeros((Bm, Ks, 3), dtype=np.int32)
    fz[:, :, 0] = rs.integers(0, 1920, (Bm, Ks))
    fz[:, :, 1] = rs.integers(0, 1080, (Bm, Ks))
    fz[:, :, 2] = np.sort(rs.uniform(1e-4, 1e-2, (Bm, Ks)).astype(np.float32), axis=1)[:, ::-1].copy().view(np.int32)
    sel_feats = torch.tensor(fz, device=dev)
    sel_lm = torch.tensor(np.stack([rs.uniform(0, 1920, (Bm, Ls)), rs.uniform(0, 1080, (Bm, Ls))], axis=2).astype(np.float32), device=dev)
    sel_out = torch.zeros((Bm, 64), dtype=torch.int32, device=dev)
    sel_cnt = torch.zeros(Bm, dtype=torch.int32, device=dev)
    sp = api.select_params(max_new=50)
    api.select_new_landmarks_batch_dev(ctx, sel_feats, None, Ks, sel_lm, None, Ls, sp, sel_out, sel_cnt)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(10):
        api.select_new_landmarks_batch_dev(ctx, sel_feats, None, Ks, sel_lm, None, Ls, sp, sel_out, sel_cnt)
    e1.record(stream)
    torch.cuda.synchronize()
    frame["select_new_landmarks_64_frames_ms"] = e0.elapsed_time(e1) / 10
    frame["select_new_landmarks_mean_taken"] = float(sel_cnt.float().mean().item())
    # initialisation of the selected keypoints: a device-resident state of 150 landmarks grows by 50
    n_i, M_i = 12 + 3 * 150, 50
    gen.manual_seed(9)
    i_mu = torch.randn(n_i, dtype=torch.float64, device=dev, generator=gen) * 0.1
    i_S = torch.triu(torch.randn((n_i, n_i), dtype=torch.float64, device=dev, generator=gen)).T.contiguous()
    i_px = torch.stack([torch.rand(M_i, dtype=torch.float64, device=dev, generator=gen) * 1400 + 250,
                        torch.rand(M_i, dtype=torch.float64, device=dev, generator=gen) * 880 + 100], dim=1).contiguous()
    m_i = n_i + 3 * M_i
    i_mo = torch.zeros(m_i, dtype=torch.float64, device=dev)
    i_So = torch.zeros((m_i, m_i), dtype=torch.float64, device=dev)
    api.initialize_landmarks_dev(ctx, cam, i_mu, i_S, i_px, i_mo, i_So)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(20):
        api.initialize_landmarks_dev(ctx, cam, i_mu, i_S, i_px, i_mo, i_So)
    e1.record(stream)
    torch.cuda.synchronize()
    frame["initialize_50_landmarks_n462_ms"] = e0.elapsed_time(e1) / 20
    # the bundle measurement model on its own: dense C (2L x n) and SR (2L x 2L) are write-only outputs
    bundle = {}
    for Lb in (150, 2000):
        sb = syn.Scene(Lb, 40 + Lb, dense_S=False)
        nb_, mb_ = 12 + 3 * Lb, 2 * Lb
        bx = torch.tensor(sb.x, device=dev)
        bh = torch.zeros(mb_, dtype=torch.float64, device=dev)
        bC = torch.zeros(mb_ * nb_, dtype=torch.float64, device=dev)
        bS = torch.zeros(mb_ * mb_, dtype=torch.float64, device=dev)
        bcode = torch.zeros(Lb, dtype=torch.int32, device=dev)
        api.bundle_measure_dev(ctx, cam, bx, bh, bS, bC, bcode)
        torch.cuda.synchronize()
        reps = 50
        e0.record(stream)
        for _ in range(reps):
            api.bundle_measure_dev(ctx, cam, bx, bh, bS, bC, bcode)
        e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        nbytes = 16.0 * Lb * nb_ + 32.0 * Lb * Lb          # C and SR written once
        gbs = nbytes / (us * 1e-6) / 1e9
        bundle["L%d" % Lb] = {"us": us, "algorithmic_bytes": nbytes,
                              "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                                           "traffic": None}}
        del bx, bh, bC, bS, bcode
    frame["bundle_measure_dev"] = bundle
    # landmark removal in a bank of 512 filters of 150 landmarks, each dropping its own landmark
    Bb = 512
    gen.manual_seed(10)
    b_mu = torch.randn((Bb, n_i), dtype=torch.float64, device=dev, generator=gen)
    b_S = (torch.triu(torch.randn((n_i, n_i), dtype=torch.float64, device=dev, generator=gen)) + 3.0 * torch.eye(n_i, dtype=torch.float64, device=dev))
    b_S = b_S.T.contiguous().unsqueeze(0).repeat(Bb, 1, 1).contiguous()
    b_j = torch.randint(0, 150, (Bb,), dtype=torch.int32, device=dev, generator=gen)
    b_mo = torch.zeros((Bb, n_i - 3), dtype=torch.float64, device=dev)
    b_So = torch.zeros((Bb, n_i - 3, n_i - 3), dtype=torch.float64, device=dev)
    b_st = torch.zeros(Bb, dtype=torch.int32, device=dev)
    api.remove_landmark_batch_dev(ctx, b_mu, b_S, b_j, b_mo, b_So, b_st)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(5):
        api.remove_landmark_batch_dev(ctx, b_mu, b_S, b_j, b_mo, b_So, b_st)
    e1.record(stream)
    torch.cuda.synchronize()
    frame["remove_landmark_512_filters_ms"] = e0.elapsed_time(e1) / 5
    # predicted measurement density (affineTransform through the bundle) for the same bank
    a_mu = torch.tensor(np.tile(sc.x, (Bb, 1)), device=dev)
    a_S = torch.tensor(np.ascontiguousarray(sc.S.T), device=dev).unsqueeze(0).repeat(Bb, 1, 1).contiguous()
    a_my = torch.zeros((Bb, 300), dtype=torch.float64, device=dev)
    a_Sy = torch.zeros((Bb, 300, 300), dtype=torch.float64, device=dev)
    api.affine_bundle_batch_dev(ctx, cam, a_mu, a_S, a_my, a_Sy)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(3):
        api.affine_bundle_batch_dev(ctx, cam, a_mu, a_S, a_my, a_Sy)
    e1.record(stream)
    torch.cuda.synchronize()
    frame["affine_bundle_512_filters_ms"] = e0.elapsed_time(e1) / 3
    # ... and its data association against the frame's features (the density computed above stays on the device)
    a_y = torch.tensor(np.ascontiguousarray(yf.T), device=dev).unsqueeze(0).repeat(Bb, 1, 1).contiguous()   # (B, m, 2)
    a_idx = torch.zeros((Bb, 150), dtype=torch.int32, device=dev)
    a_sur = torch.zeros(Bb, dtype=torch.float64, device=dev)
    api.snn_density_batch_dev(ctx, a_my, a_Sy, a_y, 150, a_y.shape[1], 1920, 1080, a_idx, a_sur)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(3):
        api.snn_density_batch_dev(ctx, a_my, a_Sy, a_y, 150, a_y.shape[1], 1920, 1080, a_idx, a_sur)
    e1.record(stream)
    torch.cuda.synchronize()
    frame["snn_density_512_filters_ms"] = e0.elapsed_time(e1) / 3
    frame["snn_density_512_filters_mean_associated"] = float((a_idx >= 0).float().sum(dim=1).mean().item())
    del a_y
    del b_mu, b_S, b_mo, b_So, a_mu, a_S, a_my, a_Sy
    frame["match_hamming_pairs_per_s"] = Bm * nq * nt / (mms * 1e-3)
    hctx.close()
    # a bank of independent filters on one GPU: one context (stream + workspace) per host thread
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    from run_iekf_bank import bank_throughput, batched_throughput
    rate8, _ = bank_throughput(150, 8, 4, api.TRS_CHOLESKY, dev.index or 0)
    upd["bank_8_contexts_updates_per_s"] = rate8
    rate512, dt512, nfail, _ = batched_throughput(150, 512, dev.index or 0, reps=1)
    upd["bank_512_lock_step_updates_per_s"] = rate512
    upd["bank_512_lock_step_ms"] = dt512 * 1e3
    upd["bank_512_handed_back_unfinished"] = nfail          # filters the device route could not finish (hard case); not counted in the rate
    try:        # the bank at the size of BASELINE configs[4] on one GPU: 4 096 filters, one pass of 56 GB of work space
        import torch
        torch.cuda.empty_cache()
        rate4k, dt4k, nfail4k, _ = batched_throughput(150, 4096, dev.index or 0, reps=1)
        upd["bank_4096_lock_step_updates_per_s"] = rate4k
        upd["bank_4096_lock_step_ms"] = dt4k * 1e3
        upd["bank_4096_handed_back_unfinished"] = nfail4k
        torch.cuda.empty_cache()
    except Exception as exc:
        upd["bank_4096_error"] = repr(exc)
    return {
        "frame_steps_150_landmarks": frame,
        "qr": {"workload": "in-place Householder QR (R only), %d x %d fp64, device resident" % (n, n), "ms": ms,
               "roofline": {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": tf / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "algorithmic_flops": flops},
               "cpu_baseline": {"value": cpu_tf, "unit": "TFLOP/s", "kind": "lapack", "cores": threads,
                                "sample": "numpy.linalg.qr(mode='r') of a %d x %d matrix of the same kind on the host, run at full size: %.2f s "
                                          "(BLAS threads: %s)" % (ns, ns, cpu_s, threads),
                                "seconds_at_full_size": cpu_s}},
        "measurement_update_150_landmarks": upd,
        "measurement_update_2000_landmarks": c4,
    }
