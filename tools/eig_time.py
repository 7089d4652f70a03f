#!/usr/bin/env python3
"""Times the host symmetric eigensolver of the default trust-region route (test hook vslam_debug_sym_eig)
against numpy's LAPACK: python tools/eig_time.py [n]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualslam_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 462
L = C.CDLL(_lib.LIB_PATH)
L.vslam_debug_sym_eig.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(0)
M = rng.standard_normal((n, n))
H = np.asfortranarray(M + M.T)
Q = np.zeros((n, n), order="F")
v = np.zeros(n)
ts = []
for _ in range(3):
    t = time.time()
    L.vslam_debug_sym_eig(H.ctypes.data, n, Q.ctypes.data, v.ctypes.data)
    ts.append((time.time() - t) * 1e3)
print("n=%d: sym_eig_host %s ms, residual %.2e" % (n, ["%.1f" % x for x in ts], np.abs(Q @ np.diag(v) @ Q.T - H).max()), flush=True)
t = time.time()
np.linalg.eigh(H)
print("numpy eigh %.1f ms" % ((time.time() - t) * 1e3))
