#!/usr/bin/env python3
"""Golden TV-L1 flows from the REFERENCE's own native code, compiled from its sources by
oracle/Makefile into oracle/_ref/libBridge.so (build container only):

    make -C oracle && OMP_NUM_THREADS=1 python3 tools/make_golden_tvl1.py

Writes tests/golden/tvl1_*.npz (inputs + the reference's flow)."""
import ctypes
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(REPO, "oracle", "_ref", "libBridge.so"))
lib.tvl1flow.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2
lib.tvl1flow.restype = None


def ref(I0, I1):
    h, w = I0.shape
    u = np.zeros(2 * h * w, np.float32)
    a, b = np.ascontiguousarray(I0, np.float32), np.ascontiguousarray(I1, np.float32)
    lib.tvl1flow(a.ctypes.data, b.ctypes.data, u.ctypes.data, w, h)
    return u.reshape(2, h, w)


def scene(h, w, dx, dy, seed, noise=0.02):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    f = lambda x, y: (np.sin(0.2 * x) + np.cos(0.15 * y) + np.sin(0.05 * x + 0.1 * y) * np.cos(0.0035 * x * y)).astype(np.float32)
    return (f(xx, yy) + noise * rng.standard_normal((h, w)).astype(np.float32),
            f(xx + dx, yy + dy) + noise * rng.standard_normal((h, w)).astype(np.float32))


cases = {"a_48x64": (48, 64, 1.5, -0.7, 1), "b_40x72": (40, 72, -2.3, 1.1, 2), "c_33x47": (33, 47, 0.4, 0.3, 3),
         "d_90x160": (90, 160, 3.1, 2.2, 4)}
for name, (h, w, dx, dy, seed) in cases.items():
    I0, I1 = scene(h, w, dx, dy, seed)
    u = ref(I0, I1)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", f"tvl1_{name}.npz"), I0=I0, I1=I1, flow=u)
    print(name, "median flow", float(np.median(u[0])), float(np.median(u[1])))
