"""Times the device-resident steps of one filter's frame at a large map: python tools/run_frame2000.py [L]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from visualslam_amd import api, synthetic as syn

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n, m = 12 + 3 * L, 2 * L
sc = syn.Scene(L, 3, dense_S=True)
dev = torch.device("cuda:0")
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
cam = api.camera_from_xml(syn.CAMERA_XML)
mu = torch.tensor(sc.x[None], device=dev)
S = torch.tensor(np.ascontiguousarray(sc.S.T)[None], device=dev)
muy = torch.zeros((1, m), dtype=torch.float64, device=dev)
Syy = torch.zeros((1, m, m), dtype=torch.float64, device=dev)
rng = np.random.default_rng(0)
feats = np.concatenate([sc.pixel_true + rng.standard_normal((L, 2)), rng.uniform(0, 1000, (100, 2))])
yt = torch.tensor(feats[None], device=dev)
idx = torch.zeros((1, L), dtype=torch.int32, device=dev)
sur = torch.zeros(1, dtype=torch.float64, device=dev)

def timed(label, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print("L=%d %s: %.2f ms" % (L, label, (time.perf_counter() - t0) / reps * 1e3))

timed("predicted measurement density (affine_bundle_batch_dev, 1 filter)", lambda: api.affine_bundle_batch_dev(ctx, cam, mu, S, muy, Syy))
timed("association (snn_density_batch_dev, 1 filter, %d features)" % feats.shape[0],
      lambda: api.snn_density_batch_dev(ctx, muy, Syy, yt, L, feats.shape[0], 1920, 1080, idx, sur), reps=1)
print("associated", int((idx >= 0).sum()))
