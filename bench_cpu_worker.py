#!/usr/bin/env python3
"""One worker of bench.py's all-core CPU baselines: runs the CPU oracle on its own synthetic units for a fixed time
and prints {"units": N, "seconds": T}.  bench.py starts one such process per host core it may use (unit-level
parallelism: the reference has no threads of its own, so all cores means independent frames / filter instances side
by side) and adds the rates.  Never imports torch or touches the GPU.

    python bench_cpu_worker.py loglik_analytic SECONDS SEED     units = landmark updates (150-landmark instances)
    python bench_cpu_worker.py corner SECONDS SEED              units = 1080p frames, both detectors
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    kind, seconds, seed = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    from oracle import pyoracle as orc
    from visualslam_amd import synthetic as syn
    c = syn.read_camera_xml()
    cam = orc.make_camera(c["K"][0, 0], c["K"][1, 1], c["K"][0, 2], c["K"][1, 2], c["dist"], c["flag"], c["width"], c["height"])
    units = 0
    if kind == "loglik_analytic":
        scs = [syn.Scene(150, 9000 + seed * 4 + i, cam=c, dense_S=False) for i in range(4)]
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            s = scs[units // 150 % 4]
            orc.point_loglik_analytic(s.x, s.y, s.visible, cam, 7.0)
            units += 150
    elif kind == "corner":
        frames = [syn.frame(5000 + seed * 2 + i) for i in range(2)]
        op = orc.corner_params(harris_thresh=1e-4, shi_thresh=1e-3, max_features=10000)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            orc.corner_detect(frames[units % 2], op, 0)
            orc.corner_detect(frames[units % 2], op, 1)
            units += 1
    else:
        raise SystemExit("unknown kind %r" % kind)
    print(json.dumps({"units": units, "seconds": time.perf_counter() - t0}))


if __name__ == "__main__":
    main()
