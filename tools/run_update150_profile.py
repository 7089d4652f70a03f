import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from visualslam_amd import api, synthetic as syn
sc = syn.Scene(150, 3, dense_S=True)
ctx = api.Context(0)
cam = api.camera_from_xml(syn.CAMERA_XML)
for _ in range(3):
    t0 = time.perf_counter(); out = api.measurement_update_iekf(ctx, cam, sc.x, sc.S, sc.y, sc.visible, sc.sigma); dt = (time.perf_counter() - t0) * 1e3
    print("wall %.2f ms" % dt, ctx.last_update_profile())
