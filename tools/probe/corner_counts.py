import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from visualslam_amd import api, synthetic as syn
F=64
dev = torch.device("cuda:0")
base = np.stack([syn.frame(5000 + i) for i in range(8)])
tb = torch.tensor(base, device=dev)
frames = torch.empty((F, 1080, 1920), dtype=torch.uint8, device=dev)
for i in range(F): frames[i] = torch.roll(tb[i % 8], shifts=(i // 8) * 7, dims=1)
params = api.corner_params(harris_thresh=1e-4, shi_thresh=1e-3, max_features=16384)
out = torch.zeros((F, 2, 16384, 3), dtype=torch.int32, device=dev)
cnt = torch.zeros((F, 2), dtype=torch.int32, device=dev)
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
api.corner_detect_batch_dev(ctx, frames, 1920, 1080, 1920, 1920 * 1080, F, params, 2, out, 16384, cnt)
torch.cuda.synchronize()
c = cnt.cpu().numpy()
print("harris min/mean/max", c[:,0].min(), c[:,0].mean(), c[:,0].max(), " shi", c[:,1].min(), c[:,1].mean(), c[:,1].max())
