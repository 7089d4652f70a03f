"""Times the device-resident measurement update of ONE filter through the bank entry point: python tools/run_update_dev1.py [L] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from visualslam_amd import api, synthetic as syn

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sc = syn.Scene(L, 3, dense_S=True)
dev = torch.device("cuda:0")
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
cam = api.camera_from_xml(syn.CAMERA_XML)
mu = torch.tensor(sc.x[None], device=dev)
S = torch.tensor(np.ascontiguousarray(sc.S.T)[None], device=dev)
y = torch.tensor(sc.y[None], device=dev)
vis = torch.tensor(sc.visible[None].astype(np.uint8), device=dev)
mup, Sp = torch.empty_like(mu), torch.empty_like(S)
torch.cuda.synchronize()
for r in range(reps + 1):
    t0 = time.perf_counter()
    code, iters, status = api.measurement_update_iekf_batch_dev(ctx, cam, mu, S, y, vis, 7.0, mup, Sp)
    torch.cuda.synchronize()
    print("L=%d device-resident update through the bank entry: %.2f ms, iterations %d, status %d" % (L, (time.perf_counter() - t0) * 1e3, int(iters[0]), int(status[0])))
