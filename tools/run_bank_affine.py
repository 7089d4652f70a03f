"""Runs the predicted measurement density of a bank of filters (profiling target):
python tools/run_bank_affine.py [B] [L] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from visualslam_amd import api, synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
sc = syn.Scene(L, 3, dense_S=True)
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
cam = api.camera_from_xml(syn.CAMERA_XML)
mu = torch.tensor(np.tile(sc.x, (B, 1)), device=dev)
S = torch.tensor(np.ascontiguousarray(sc.S.T), device=dev).unsqueeze(0).repeat(B, 1, 1).contiguous()
my = torch.zeros((B, 2 * L), dtype=torch.float64, device=dev)
Sy = torch.zeros((B, 2 * L, 2 * L), dtype=torch.float64, device=dev)
for _ in range(reps):
    api.affine_bundle_batch_dev(ctx, cam, mu, S, my, Sy)
torch.cuda.synchronize()
print("muY[0,:2]", my[0, :2].tolist())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    api.affine_bundle_batch_dev(ctx, cam, mu, S, my, Sy)
e1.record()
torch.cuda.synchronize()
print("ms per bank", e0.elapsed_time(e1) / reps)
