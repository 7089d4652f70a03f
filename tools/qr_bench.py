#!/usr/bin/env python3
"""Device-resident timing of the square-root QR: python tools/qr_bench.py [rows] [cols] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualslam_amd import api  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6012
    cols = int(sys.argv[2]) if len(sys.argv) > 2 else rows
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    torch.manual_seed(0)
    A0 = torch.randn((cols, rows), dtype=torch.float64, device="cuda:0")     # column-major rows x cols
    stream = torch.cuda.current_stream()
    ctx = api.Context(0, stream=stream.cuda_stream)
    A = A0.clone()
    api.qr_upper_batch_dev(ctx, A, rows, cols, rows, 1, rows * cols)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        A.copy_(A0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        api.qr_upper_batch_dev(ctx, A, rows, cols, rows, 1, rows * cols)
        e1.record(stream)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    k = min(rows, cols)
    flops = 2.0 * rows * cols * cols - 2.0 / 3.0 * cols ** 3 if rows >= cols else 2.0 * cols * rows * rows - 2.0 / 3.0 * rows ** 3
    ms = float(np.median(ts))
    print("qr %dx%d: %.2f ms  (%.2f TFLOP/s fp64, %s)" % (rows, cols, ms, flops / ms / 1e9, ts), flush=True)
    if "--check" in sys.argv:
        R = A.cpu().numpy().T[:k]
        R0 = np.linalg.qr(A0.cpu().numpy().T, mode="r")
        print("max rel err vs LAPACK: %.2e" % (np.abs(R - R0).max() / np.abs(R0).max()))
    ctx.close()


if __name__ == "__main__":
    main()
