#!/usr/bin/env python3
"""Times snn (bundle -> affineTransform -> greedy association) at L landmarks against the CPU oracle:
python tools/run_snn.py [L] [--oracle]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualslam_amd import api, synthetic as syn  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 150
    sc = syn.Scene(L, 3, dense_S=True)
    rng = np.random.default_rng(80)
    keep = rng.permutation(L)[: (3 * L) // 4]
    feats = sc.pixel_true[keep] + 2.0 * rng.standard_normal((keep.size, 2))
    clutter = np.stack([rng.uniform(0, 1920, L), rng.uniform(0, 1080, L)], axis=1)
    y = np.concatenate([feats, clutter])[rng.permutation(keep.size + L)].T.copy()
    ctx = api.Context(0)
    cam = api.camera_from_xml(syn.CAMERA_XML)
    s, idx = api.snn(ctx, cam, sc.x, sc.S, y)
    t0 = time.perf_counter()
    for _ in range(5):
        s, idx = api.snn(ctx, cam, sc.x, sc.S, y)
    dt = (time.perf_counter() - t0) / 5
    print("L=%d m=%d: snn %.2f ms, surprisal %.6f, %d associated" % (L, y.shape[1], dt * 1e3, s, int((idx >= 0).sum())), flush=True)
    if "--oracle" in sys.argv:
        from oracle import pyoracle as orc
        c = syn.read_camera_xml()
        ocam = orc.make_camera(c["K"][0, 0], c["K"][1, 1], c["K"][0, 2], c["K"][1, 2], c["dist"], c["flag"], c["width"], c["height"])
        t0 = time.perf_counter()
        s0, idx0 = orc.snn(sc.x, sc.S, y, ocam, 0, False)
        print("oracle (re-factors per candidate like the reference, 1 core): %.1f ms, same association: %s, surprisal diff %.2e"
              % ((time.perf_counter() - t0) * 1e3, idx0.tolist() == idx.tolist(), abs(s - s0) / abs(s0)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
