"""Times the batched corner detector (64 frames 1080p, both detectors): python tools/time_corner.py [frames] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from visualslam_amd import api, synthetic as syn

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
base = torch.tensor(np.stack([syn.frame(5000 + i) for i in range(4)]), device=dev)
frames = torch.stack([torch.roll(base[i % 4], (i // 4) * 7, 1) for i in range(F)])
params = api.corner_params(harris_thresh=1e-4, shi_thresh=1e-3, max_features=10000)
out = torch.zeros((F, 2, 10000, 3), dtype=torch.int32, device=dev)
cnt = torch.zeros((F, 2), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream()
ctx = api.Context(0, stream=stream.cuda_stream)
for _ in range(5):
    api.corner_detect_batch_dev(ctx, frames, 1920, 1080, 1920, 1920 * 1080, F, params, 2, out, 10000, cnt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(reps):
    api.corner_detect_batch_dev(ctx, frames, 1920, 1080, 1920, 1920 * 1080, F, params, 2, out, 10000, cnt)
e1.record(stream)
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("VSLAM_STRIP_WAVES=%s: %.4f ms per %d frames, %.0f frames/s, checksum %d" % (os.environ.get("VSLAM_STRIP_WAVES", "-"), ms, F, F / ms * 1e3, int(out.sum().item()) + int(cnt.sum().item())))
