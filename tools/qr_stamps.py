#!/usr/bin/env python3
"""Phase timing inside qr_block_kernel (library built with EXTRA=-DQB_STAMPS): python tools/qr_stamps.py rows cols"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualslam_amd import api, _lib  # noqa: E402

rows, cols = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
A0 = torch.randn((cols, rows), dtype=torch.float64, device="cuda:0")
stream = torch.cuda.current_stream()
ctx = api.Context(0, stream=stream.cuda_stream)
L = _lib.lib()
L.vslam_debug_ws_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for rep in range(3):
    A = A0.clone()
    api.qr_upper_batch_dev(ctx, A, rows, cols, rows, 1, rows * cols)
    torch.cuda.synchronize()
    out = np.zeros(1056)
    assert L.vslam_debug_ws_read(ctx.ptr, out.ctypes.data, 1056) == 0
    st = out[1024:1036]
    print("path fast=%d  phase cycles:" % out[1055], np.diff(st).astype(np.int64).tolist(), "total", int(st[-1] - st[0]), flush=True)
names = ["gram1", "allreduce1", "load", "chol1", "trsm1", "gram2", "allreduce2", "load+check", "chol2", "lu||rr", "trsm2||tt", "Tout"]
print(names)
ctx.close()
