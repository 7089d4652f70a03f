"""Runs the whole ORB front end (all pyramid levels, keypoints + orientation + descriptors) for a batch of
device-resident 1080p frames a few times (profiling target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from visualslam_amd import api, synthetic as syn

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
base = torch.tensor(np.stack([syn.frame(5000 + i) for i in range(4)]), device=dev)
frames = torch.stack([torch.roll(base[i % 4], (i // 4) * 7, 1) for i in range(F)])
cap = 12288
out = torch.zeros((F, cap, 4), dtype=torch.int32, device=dev)
cnt = torch.zeros(F, dtype=torch.int32, device=dev)
ang = torch.zeros((F, cap), dtype=torch.float32, device=dev)
desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device=dev)
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    api.orb_detect_and_compute_batch_dev(ctx, frames, out, cnt, angle=ang, desc=desc)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
api.orb_detect_and_compute_batch_dev(ctx, frames, out, cnt, angle=ang, desc=desc)
e1.record()
torch.cuda.synchronize()
print("mean keypoints", float(cnt.float().mean()), "ms per batch", e0.elapsed_time(e1))
