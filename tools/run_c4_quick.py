"""Quick look at the measurement update timings: 150-landmark single, 2000-landmark single, bank of 512."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from visualslam_amd import api, synthetic as syn
cam = api.camera_from_xml(syn.CAMERA_XML)
ctx = api.Context(0)
for L, seed in ((150, 3), (2000, 4)):
    sc = syn.Scene(L, seed, dense_S=True)
    api.measurement_update_iekf(ctx, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
    t0 = time.perf_counter()
    out = api.measurement_update_iekf(ctx, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
    dt = time.perf_counter() - t0
    print("L=%d update %.2f ms, route %d, iters %d evals %d, profile %s" % (L, dt * 1e3, ctx.last_trs_route(), out[3], out[4], ctx.last_update_profile()), flush=True)
from run_iekf_bank import batched_throughput
rate, dt, nfail, itmax = batched_throughput(150, 512, 0, reps=2)
print("bank 512: %.1f updates/s, %.2f ms, %d handed back, %d iterations at most" % (rate, dt * 1e3, nfail, itmax), flush=True)
