#!/usr/bin/env python3
"""Generates tests/golden/dubins_golden.npz from the reference's own src/dubins.cpp.

oracle/Makefile compiles that file, where it lies in the reference tree, into
oracle/_ref/libdubins_ref.so (nothing of the reference is copied).  This script
calls dubins_init / dubins_path_length / dubins_path_sample through the C++
mangled names and stores inputs and outputs as raw float32 / int32 arrays.
Run it in the container that holds the reference tree:

    make -C oracle && python tests/golden/make_dubins_golden.py
"""
import ctypes as C
import math
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]


class DubinsPath(C.Structure):  # include/dubins.h:52-58
    _fields_ = [("qi", C.c_float * 3), ("param", C.c_float * 3), ("rho", C.c_float), ("type", C.c_int)]


def main():
    lib = C.CDLL(str(ROOT / "oracle" / "_ref" / "libdubins_ref.so"))
    init = getattr(lib, "_ZN11HybridAStar11dubins_initEPfS0_fPNS_10DubinsPathE")
    init.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.POINTER(DubinsPath)]
    init.restype = C.c_int
    length = getattr(lib, "_ZN11HybridAStar18dubins_path_lengthEPNS_10DubinsPathE")
    length.argtypes = [C.POINTER(DubinsPath)]
    length.restype = C.c_float
    sample = getattr(lib, "_ZN11HybridAStar18dubins_path_sampleEPNS_10DubinsPathEfPf")
    sample.argtypes = [C.POINTER(DubinsPath), C.c_float, C.POINTER(C.c_float)]
    sample.restype = C.c_int

    rng = np.random.default_rng(4)
    n = 4096
    q0 = np.stack([rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), rng.uniform(0, 2 * math.pi, n)], 1).astype(np.float32)
    q1 = np.stack([rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), rng.uniform(0, 2 * math.pi, n)], 1).astype(np.float32)
    rho = np.where(rng.random(n) < 0.5, np.float32(0.35), np.float32(6.0)).astype(np.float32)
    # a few structured cases: near-colocated, axis aligned, opposite headings
    q0[:8] = [[0, 0, 0]] * 8
    q1[:8] = [[1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0], [5, 5, math.pi], [0.01, 0, 0], [3, 2, 1], [0, 0, math.pi]]
    NS = 8
    rc = np.zeros(n, np.int32)
    typ = np.zeros(n, np.int32)
    par = np.zeros((n, 3), np.float32)
    ln = np.zeros(n, np.float32)
    smp = np.zeros((n, NS, 3), np.float32)
    smp_rc = np.zeros((n, NS), np.int32)
    fp = C.POINTER(C.c_float)
    for i in range(n):
        p = DubinsPath()
        p.type = -1
        a = q0[i].copy()
        b = q1[i].copy()
        rc[i] = init(a.ctypes.data_as(fp), b.ctypes.data_as(fp), C.c_float(float(rho[i])), C.byref(p))
        if rc[i] != 0:
            continue
        typ[i] = p.type
        par[i] = list(p.param)
        ln[i] = length(C.byref(p))
        for k in range(NS):
            t = np.float32(ln[i] * np.float32(k) / np.float32(NS))
            q = np.zeros(3, np.float32)
            smp_rc[i, k] = sample(C.byref(p), C.c_float(float(t)), q.ctypes.data_as(fp))
            smp[i, k] = q
    out = ROOT / "tests" / "golden" / "dubins_golden.npz"
    np.savez_compressed(out, q0=q0, q1=q1, rho=rho, rc=rc, type=typ, param=par, length=ln, samples=smp, sample_rc=smp_rc)
    print("wrote", out, "ok cases", int((rc == 0).sum()), "of", n)


if __name__ == "__main__":
    main()
