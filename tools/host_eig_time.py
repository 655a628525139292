"""The host eigensolver of si_construct_finish alone (si_host_sym_eig_top: no GPU): first call and steady state, at the K x M of
cfg2 / cfg4 / the README toy's K > N route, on the CPU it runs on."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.environ.get("SI_PROBE_LIB") or os.path.join(ROOT, "subspaceinference.jl_amd", "libsubspace_hip.so"))   # SI_PROBE_LIB: another build, for A/B
f = lib.si_host_sym_eig_top
f.restype = ctypes.c_int32
f.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
rng = np.random.default_rng(0)
with open("/proc/cpuinfo") as fh:
    print([ln.split(":")[1].strip() for ln in fh if ln.startswith("model name")][0])
for k, m in [(100, 20), (200, 20), (682, 3)]:
    a = rng.standard_normal((4 * k, k))
    g = np.ascontiguousarray(a.T @ a)
    w, v = np.empty(m), np.empty((k, m))
    t0 = time.perf_counter()
    rc = f(k, g.ctypes.data, m, w.ctypes.data, v.ctypes.data)
    first = time.perf_counter() - t0
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        f(k, g.ctypes.data, m, w.ctypes.data, v.ctypes.data)
        ts.append(time.perf_counter() - t0)
    wl = np.linalg.eigvalsh(g)[::-1][:m]
    print("K %4d M %2d: rc %d  first call %.3f ms  steady state %.3f ms (min of 20)  max eigenvalue error %.1e of the largest"
          % (k, m, rc, first * 1e3, min(ts) * 1e3, np.max(np.abs(w - wl)) / wl[0]))
