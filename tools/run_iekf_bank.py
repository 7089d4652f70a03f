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


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    L = int(args[0]) if len(args) > 0 else 150
    threads = int(args[1]) if len(args) > 1 else 8
    per = int(args[2]) if len(args) > 2 else 4
    solver = api.TRS_EIG if "--eig" in sys.argv else api.TRS_CHOLESKY
    for th in sorted({1, threads}):
        rate, dt = bank_throughput(L, th, per, solver)
        print("L=%d, %d concurrent contexts: %.1f measurement updates/s (%.1f ms per update per context)"
              % (L, th, rate, dt / per * 1e3), flush=True)


if __name__ == "__main__":
    main()
