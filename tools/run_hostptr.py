#!/usr/bin/env python3
"""Wall time of the host-pointer (drop-in, PCIe-inclusive, synchronising) entry points at C3 size."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualslam_amd import api, synthetic as syn  # noqa: E402


def timed(f, reps=50):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    ctx = api.Context(0)
    cam = api.camera_from_xml(syn.CAMERA_XML)
    sc = syn.Scene(150, 3, dense_S=True)
    img = syn.frame(2)
    params = api.corner_params(harris_thresh=1e-4, shi_thresh=1e-3, max_features=10000)
    rows = [
        ("point_loglik value", lambda: api.point_loglik(ctx, cam, sc.x, sc.y, sc.visible, sc.sigma, want_g=False, want_H=False)),
        ("point_loglik value+g+H", lambda: api.point_loglik(ctx, cam, sc.x, sc.y, sc.visible, sc.sigma)),
        ("bundle_measure h+SR+C", lambda: api.bundle_measure(ctx, cam, sc.x)),
        ("project (150 landmarks)", lambda: api.project(ctx, cam, sc.x)),
        ("log_gaussian value+g+H", lambda: api.log_gaussian(ctx, sc.x, sc.x + 1e-3, sc.S)),
        ("cost_joint_density", lambda: api.cost_joint_density(ctx, cam, sc.x, sc.x, sc.S, sc.y, sc.visible, sc.sigma)),
        ("affine_bundle", lambda: api.affine_bundle(ctx, cam, sc.x, sc.S)),
        ("qr_upper 462x462", lambda: api.qr_upper(ctx, sc.S + 0.0)),
        ("corner_detect 1080p harris", lambda: api.corner_detect(ctx, img, params, api.HARRIS)),
        ("corner_detect 1080p shi-tomasi", lambda: api.corner_detect(ctx, img, params, api.SHI_TOMASI)),
    ]
    print("host-pointer entry points, n=462 / 1920x1080, wall time per call (us):")
    for name, f in rows:
        print("  %-30s %10.1f" % (name, timed(f)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
