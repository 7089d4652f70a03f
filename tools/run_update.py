#!/usr/bin/env python3
"""Runs the complete measurement update (default device route) of an L-landmark scene a few times and
prints the library's own phase breakdown; run under rocprofv3 --kernel-trace --stats for kernel times.

    python tools/run_update.py [L] [reps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualslam_amd import api, synthetic as syn  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    sc = syn.Scene(L, 4, dense_S=True)
    ctx = api.Context(0)
    cam = api.camera_from_xml(syn.CAMERA_XML)
    ctx.set_trs_solver(api.TRS_CHOLESKY)
    api.measurement_update_iekf(ctx, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
    for _ in range(reps):
        t0 = time.perf_counter()
        out = api.measurement_update_iekf(ctx, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
        wall = (time.perf_counter() - t0) * 1e3
        print("L=%d n=%d wall %.1f ms code %d iters %d evals %d route %d" % (L, sc.n, wall, out[0], out[3], out[4], ctx.last_trs_route()), flush=True)
        print(ctx.last_update_profile(), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
