"""Runs the batched FAST-9 detector a few times (profiling target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from visualslam_amd import api, synthetic as syn

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
base = torch.tensor(np.stack([syn.frame(5000 + i) for i in range(4)]), device=dev)
frames = torch.stack([torch.roll(base[i % 4], (i // 4) * 7, 1) for i in range(F)])
out = torch.zeros((F, 10000, 3), dtype=torch.int32, device=dev)
cnt = torch.zeros(F, dtype=torch.int32, device=dev)
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    api.fast9_detect_batch_dev(ctx, frames, 1920, 1080, 1920, 1920 * 1080, F, 20, True, 10000, out, 10000, cnt)
torch.cuda.synchronize()
print("mean keypoints", float(cnt.float().mean()))
