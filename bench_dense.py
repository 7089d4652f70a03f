"""Dense fp64 leg of bench.py (rank 0, one GPU): the square-root re-triangularisation QR of the
2,000-landmark state (6012 x 6012, device resident) against the fp64 matrix-core peak, and the
wall time of one complete measurement update (150 landmarks) on both trust-region routes."""
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

    cam = api.camera_from_xml(syn.CAMERA_XML)
    sc = syn.Scene(150, 3, dense_S=True)
    hctx = api.Context(dev.index or 0)
    upd = {}
    for name, solver in (("cholesky_route_ms", api.TRS_CHOLESKY), ("eigen_route_ms", api.TRS_EIG)):
        hctx.set_trs_solver(solver)
        api.measurement_update_iekf(hctx, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
        t0 = time.perf_counter()
        out = api.measurement_update_iekf(hctx, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
        upd[name] = (time.perf_counter() - t0) * 1e3
        upd["iterations"] = out[3]
    hctx.close()
    return {
        "qr": {"workload": "in-place Householder QR (R only), %d x %d fp64, device resident" % (n, n), "ms": ms,
               "roofline": {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": tf / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "algorithmic_flops": flops}},
        "measurement_update_150_landmarks": upd,
    }
