#!/usr/bin/env python3
"""Timing of the time update: python tools/run_time_update.py [L] [batch] [reps] [--check]
Device-resident filters (vslam_time_update_batch_dev); --check compares instance 0 with the CPU oracle."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualslam_amd import api, synthetic as syn  # noqa: E402

PT = np.diag([0.5, 0.5, 0.175])
OT = np.diag([0.05, 0.05, 0.05])


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    L = int(args[0]) if len(args) > 0 else 150
    B = int(args[1]) if len(args) > 1 else 1
    reps = int(args[2]) if len(args) > 2 else 5
    n = 12 + 3 * L
    sc = syn.Scene(L, 3)
    x = sc.x.copy()
    x[:6] = 0.2 * np.random.default_rng(1).standard_normal(6)
    dev = torch.device("cuda:0")
    mu0 = torch.tensor(x, device=dev).repeat(B, 1).contiguous()
    S0 = torch.tensor(np.ascontiguousarray(sc.S.T), device=dev).repeat(B, 1, 1).contiguous()
    mu, S = torch.empty_like(mu0), torch.empty_like(S0)
    stream = torch.cuda.current_stream()
    ctx = api.Context(0, stream=stream.cuda_stream)
    api.time_update_batch_dev(ctx, mu0, S0, n, PT, OT, 1 / 30, mu, S)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        api.time_update_batch_dev(ctx, mu0, S0, n, PT, OT, 1 / 30, mu, S)
        e1.record(stream)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    print("time update n=%d batch=%d: %.3f ms  (%.1f filters/s)" % (n, B, ms, B / ms * 1e3), flush=True)
    if B == 1:
        t0 = time.perf_counter()
        for _ in range(reps):
            api.time_update(ctx, x, sc.S, PT, OT, 1 / 30)
        print("  host-pointer entry point: %.3f ms" % ((time.perf_counter() - t0) / reps * 1e3), flush=True)
    if "--check" in sys.argv:
        from oracle import pyoracle as orc
        t0 = time.perf_counter()
        m0, Sr = orc.time_update(x, sc.S, PT, OT, 1 / 30)
        t1 = time.perf_counter()
        err = np.abs(S[0].cpu().numpy().T - Sr).max() / np.abs(Sr).max()
        print("  oracle %.1f ms; max rel err S %.2e, mu %.2e" % ((t1 - t0) * 1e3, err, np.abs(mu[0].cpu().numpy() - m0).max()), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
