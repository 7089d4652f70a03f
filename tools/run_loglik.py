"""Runs the batched point log-likelihood kernel a few times (profiling target):
B instances of L landmarks, value + gradient + Hessian."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from visualslam_amd import api, synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n = 12 + 3 * L
dev = torch.device("cuda:0")
scs = [syn.Scene(L, 9000 + i, dense_S=False) for i in range(min(B, 8))]
rep = (B + len(scs) - 1) // len(scs)
x = torch.tensor(np.tile(np.stack([s.x for s in scs]), (rep, 1))[:B], device=dev)
y = torch.tensor(np.tile(np.stack([s.y for s in scs]), (rep, 1))[:B], device=dev)
vis = torch.tensor(np.tile(np.stack([s.visible for s in scs]), (rep, 1))[:B], device=dev)
val = torch.empty(B, dtype=torch.float64, device=dev)
g = torch.empty((B, n), dtype=torch.float64, device=dev)
H = torch.empty((B, n, n), dtype=torch.float64, device=dev)
code = torch.empty((B, L), dtype=torch.int32, device=dev)
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
cam = api.camera_from_xml(syn.CAMERA_XML)
for _ in range(reps):
    api.point_loglik_batch_dev(ctx, cam, n, x, y, vis, 7.0, val, g, H, code)
torch.cuda.synchronize()
print("value[0]", float(val[0]))
