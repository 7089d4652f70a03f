"""Times the predicted measurement density of one filter (host-pointer entry): python tools/run_affine2000.py [L] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visualslam_amd import api, synthetic as syn

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc = syn.Scene(L, 3, dense_S=True)
ctx = api.Context(0)
cam = api.camera_from_xml(syn.CAMERA_XML)
api.affine_bundle(ctx, cam, sc.x, sc.S, 0)
t0 = time.perf_counter()
for _ in range(reps):
    muy, Syy, _ = api.affine_bundle(ctx, cam, sc.x, sc.S, 0)
print("L=%d affine_bundle (host pointers) %.2f ms, checksum %.12g" % (L, (time.perf_counter() - t0) / reps * 1e3, float(np.abs(Syy).sum())))
