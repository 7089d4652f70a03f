"""Times the two kernels of the batched point log-likelihood call apart (library events) and the call as a whole."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from visualslam_amd import api, synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
n = 12 + 3 * L
dev = torch.device("cuda:0")
scs = [syn.Scene(L, 9000 + i, dense_S=False) for i in range(min(B, 8))]
rep = (B + len(scs) - 1) // len(scs)
x = torch.tensor(np.tile(np.stack([s.x for s in scs]), (rep, 1))[:B], device=dev)
y = torch.tensor(np.tile(np.stack([s.y for s in scs]), (rep, 1))[:B], device=dev)
vis = torch.tensor(np.tile(np.stack([s.visible for s in scs]), (rep, 1))[:B], device=dev)
if os.environ.get("TIME_LOGLIK_INVISIBLE"):
    vis.zero_()          # every landmark culled: what is left of the terms kernel is its fixed cost
val = torch.empty(B, dtype=torch.float64, device=dev)
g = torch.empty((B, n), dtype=torch.float64, device=dev)
H = torch.empty((B, n, n), dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream()
ctx = api.Context(0, stream=stream.cuda_stream)
cam = api.camera_from_xml(syn.CAMERA_XML)
for _ in range(5):
    api.point_loglik_batch_dev(ctx, cam, n, x, y, vis, 7.0, val, g, H)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in ev:
    a.record(stream); api.point_loglik_batch_dev(ctx, cam, n, x, y, vis, 7.0, val, g, H); b.record(stream)
torch.cuda.synchronize()
whole = np.array([a.elapsed_time(b) for a, b in ev])
ctx.set_kernel_timing(True)
sp = []
for _ in range(reps):
    api.point_loglik_batch_dev(ctx, cam, n, x, y, vis, 7.0, val, g, H)
    sp.append(ctx.last_point_loglik_times())
sp = np.array(sp)
alg = (8 * n * n + 16 * n + 17 * L + 168) * B
print("B=%d L=%d: call median %.4f ms (%.3f of 8 TB/s), terms %.4f ms, assemble %.4f ms (%.3f); checksum %.12g" % (
    B, L, np.median(whole), alg / np.median(whole) / 1e6 / 8000, np.median(sp[:, 0]), np.median(sp[:, 1]), alg / np.median(sp[:, 1]) / 1e6 / 8000,
    float(val.sum()) + float(g.sum()) + float(H.sum())))
