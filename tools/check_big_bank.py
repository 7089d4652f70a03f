import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from visualslam_amd import api, synthetic as syn
L = 150; n = 12 + 3 * L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
scs = [syn.Scene(L, 9000 + i, dense_S=True) for i in range(3)]
dev = torch.device("cuda:0")
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
cam = api.camera_from_xml(syn.CAMERA_XML)
def run(B):
    mu = torch.tensor(np.stack([scs[b % 3].x for b in range(B)]), device=dev)
    S = torch.tensor(np.stack([np.ascontiguousarray(sc.S.T) for sc in scs]), device=dev)[torch.arange(B, device=dev) % 3].contiguous()
    y = torch.tensor(np.stack([scs[b % 3].y for b in range(B)]), device=dev)
    vis = torch.tensor(np.stack([scs[b % 3].visible for b in range(B)]).astype(np.uint8), device=dev)
    mup, Sp = torch.empty_like(mu), torch.empty_like(S)
    torch.cuda.synchronize()
    code, iters, status = api.measurement_update_iekf_batch_dev(ctx, cam, mu, S, y, vis, 7.0, mup, Sp)
    torch.cuda.synchronize()
    return mup, Sp, iters, status
m3, S3, it3, st3 = run(3)
mB, SB, itB, stB = run(B)
ok = True
for b in range(B):
    r = b % 3
    # the same scene: bit-identical within the large bank, and equal to the small bank's (one-level factorisation) to rounding
    same = torch.equal(mB[b], mB[r]) and torch.equal(SB[b], SB[r]) and itB[b] == itB[r] and stB[b] == 0
    close = float((mB[b] - m3[r]).abs().max()) <= 1e-9 * max(1.0, float(m3[r].abs().max())) and \
        float((SB[b] - S3[r]).abs().max()) <= 1e-9 * float(S3[r].abs().max()) and itB[b] == it3[r]
    if not (same and close):
        ok = False; print("mismatch at", b, float((mB[b] - m3[r]).abs().max()), int(itB[b]), int(it3[r])); break
print("B =", B, "every filter equals its scene's first copy bit for bit and the bank of 3 to 1e-9:", ok)
