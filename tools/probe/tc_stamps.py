"""Stamps of the chained triangular solve (library built with the stamp edits, VSLAM_HIP_LIB pointing at it):
python tools/probe/tc_stamps.py n trans"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from visualslam_amd import _lib, api

n, trans = int(sys.argv[1]), int(sys.argv[2])
ctx = api.Context()
L = _lib.lib()
rng = np.random.default_rng(0)
U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + np.diag(1.0 + rng.random(n))
S = np.ascontiguousarray(U.T)
b = rng.standard_normal(n)
z = np.zeros(n)
fn = L.vslam_debug_trsv
fn.restype = C.c_int32
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
for rep in range(3):
    assert fn(ctx.ptr, S.ctypes.data, n, trans, b.ctypes.data, z.ctypes.data) == 0
st = np.loadtxt(os.environ["TC_STAMPS"], dtype=np.int64)
nblk = (n + 63) // 64
st = st[:nblk]
t0 = st[0, 0]
print("total %.1f us for %d links" % ((st[nblk - 1, 1] - t0) / 100.0, nblk))
print("link: chain start, chain loop end | workers: start, terms done, stage done, loads issued   (us, relative to the link's chain start; 10 ns ticks)")
for i in list(range(0, 8)) + list(range(40, 48)):
    if i >= nblk - 1: break
    r = st[i]
    print(i, "%.2f" % ((r[0] - t0) / 100.0), " ".join("%.2f" % ((x - r[0]) / 100.0) for x in r[1:6]), "| waves at barrier", " ".join("%.2f" % ((x - r[0]) / 100.0) for x in r[6:14]), "| next link after %.2f" % ((st[i + 1, 0] - r[0]) / 100.0))
