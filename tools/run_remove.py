"""Times landmark removal through the host-pointer entry: structured kernels against the dense QR."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visualslam_amd import api

ctx = api.Context(0)
for n, j in ((462, 10), (462, 140), (1512, 10), (6012, 10)):
    rng = np.random.default_rng(n)
    mu = rng.standard_normal(n)
    S = np.triu(rng.standard_normal((n, n))) + 3.0 * np.eye(n)
    out = {}
    for mode in ("structured", "dense"):
        if mode == "dense":
            os.environ["VSLAM_REMOVE_DENSE"] = "1"
        else:
            os.environ.pop("VSLAM_REMOVE_DENSE", None)
        api.remove_landmark(ctx, mu, S, j)
        t0 = time.perf_counter()
        reps = 5 if n < 2000 else 2
        for _ in range(reps):
            r = api.remove_landmark(ctx, mu, S, j)
        out[mode] = ((time.perf_counter() - t0) / reps, r[1])
    err = np.abs(out["structured"][1] - out["dense"][1]).max() / np.abs(out["dense"][1]).max()
    print(f"n={n} j={j}: structured {out['structured'][0]*1e3:.2f} ms, dense {out['dense'][0]*1e3:.2f} ms (host-pointer entry, PCIe included), rel diff {err:.1e}", flush=True)

import torch
dev = torch.device("cuda:0")
for n, j in ((462, 10), (462, 140), (1512, 10), (6012, 10)):
    rng = np.random.default_rng(n)
    dmu = torch.tensor(rng.standard_normal(n), device=dev)
    dS = torch.tensor(np.ascontiguousarray((np.triu(rng.standard_normal((n, n))) + 3.0 * np.eye(n)).T), device=dev)
    dmo = torch.empty(n - 3, dtype=torch.float64, device=dev)
    dSo = torch.empty((n - 3, n - 3), dtype=torch.float64, device=dev)
    res = {}
    for upper in (True, False):
        api.remove_landmark_dev(ctx, dmu, dS, j, dmo, dSo, s_is_upper=upper)
        ctx.synchronize()
        reps = 10 if n < 2000 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            api.remove_landmark_dev(ctx, dmu, dS, j, dmo, dSo, s_is_upper=upper)
        ctx.synchronize()
        res[upper] = (time.perf_counter() - t0) / reps
    print(f"device resident n={n} j={j}: structured {res[True]*1e3:.3f} ms, dense {res[False]*1e3:.3f} ms", flush=True)
