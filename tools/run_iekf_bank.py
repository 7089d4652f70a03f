#!/usr/bin/env python3
"""Throughput of complete measurement updates for a bank of independent filters on one GPU: one context
(stream + workspace) per host thread, every thread running updates back to back.
python tools/run_iekf_bank.py [L] [threads] [updates_per_thread] [--eig]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualslam_amd import api, synthetic as syn  # noqa: E402


def bank_throughput(L=150, threads=8, per_thread=4, solver=None, device=0):
    solver = api.TRS_CHOLESKY if solver is None else solver
    cam = api.camera_from_xml(syn.CAMERA_XML)
    scenes = [syn.Scene(L, 9000 + t, dense_S=True) for t in range(threads)]
    ctxs = [api.Context(device) for _ in range(threads)]
    for c in ctxs:
        c.set_trs_solver(solver)
    for c, sc in zip(ctxs, scenes):                              # warm-up: workspaces, attributes
        api.measurement_update_iekf(c, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
    start = threading.Barrier(threads + 1)

    def work(c, sc):
        start.wait()
        for _ in range(per_thread):
            api.measurement_update_iekf(c, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)

    ts = [threading.Thread(target=work, args=(c, sc)) for c, sc in zip(ctxs, scenes)]
    for t in ts:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    return threads * per_thread / dt, dt


def batched_throughput(L=150, B=256, device=0, reps=2):
    """The whole bank through vslam_measurement_update_iekf_batch_dev (lock step, state on the device)."""
    import torch
    cam = api.camera_from_xml(syn.CAMERA_XML)
    uniq = min(B, 16)
    scs = [syn.Scene(L, 9000 + t, dense_S=True) for t in range(uniq)]
    rng = np.random.default_rng(1)
    idx = [b % uniq for b in range(B)]
    dev = torch.device("cuda", device)
    mus = np.stack([scs[i].x + np.concatenate([np.zeros(6), 1e-4 * rng.standard_normal(6), 1e-2 * rng.standard_normal(3 * L)])
                    for i in idx])
    mu = torch.tensor(mus, device=dev)
    S = torch.tensor(np.stack([np.ascontiguousarray(scs[i].S.T) for i in idx]), device=dev)
    y = torch.tensor(np.stack([scs[i].y for i in idx]), device=dev)
    vis = torch.tensor(np.stack([scs[i].visible for i in idx]).astype(np.uint8), device=dev)
    mup, Sp = torch.empty_like(mu), torch.empty_like(S)
    ctx = api.Context(device)
    out = api.measurement_update_iekf_batch_dev(ctx, cam, mu, S, y, vis, scs[0].sigma, mup, Sp)     # warm-up: workspace
    t0 = time.perf_counter()
    for _ in range(reps):
        out = api.measurement_update_iekf_batch_dev(ctx, cam, mu, S, y, vis, scs[0].sigma, mup, Sp)
    dt = (time.perf_counter() - t0) / reps
    ctx.close()
    nfail = int(out[2].sum())
    return (B - nfail) / dt, dt, nfail, int(out[1].max())        # the rate counts finished filters only


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    L = int(args[0]) if len(args) > 0 else 150
    threads = int(args[1]) if len(args) > 1 else 8
    per = int(args[2]) if len(args) > 2 else 4
    solver = api.TRS_EIG if "--eig" in sys.argv else api.TRS_CHOLESKY
    if "--batched" in sys.argv:
        for B in (threads,):
            rate, dt, nfail, itmax = batched_throughput(L, B)
            print("L=%d, bank of %d filters in lock step: %.1f measurement updates/s (%.1f ms per bank; %d not positive definite, "
                  "%d iterations at most)" % (L, B, rate, dt * 1e3, nfail, itmax), flush=True)
        return
    for th in sorted({1, threads}):
        rate, dt = bank_throughput(L, th, per, solver)
        print("L=%d, %d concurrent contexts: %.1f measurement updates/s (%.1f ms per update per context)"
              % (L, th, rate, dt / per * 1e3), flush=True)


if __name__ == "__main__":
    main()
