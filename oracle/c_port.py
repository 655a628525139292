"""ctypes loader of the C/OpenMP restatement (oracle/subspace_oracle_c.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__ (build / smoke) and bench.py's cpu_baseline leg may import this module.  The shared object
is built with gcc next to the source (oracle/_build/, git-ignored; it travels to the GPU box with the snapshot, and is
rebuilt there on demand -- the image has gcc)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "subspace_oracle_c.c")
LIB = os.path.join(HERE, "_build", "libsubspace_oracle_c.so")


class SoLayer(ctypes.Structure):
    _fields_ = [("in_", ctypes.c_int32), ("out", ctypes.c_int32), ("act", ctypes.c_int32),
                ("w_off", ctypes.c_int64), ("b_off", ctypes.c_int64)]


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.run(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-Wall", "-Wno-psabi", "-o", LIB, SRC, "-lm"], check=True)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        lib.so_logdensity.restype = ctypes.c_int
        lib.so_logdensity.argtypes = [ctypes.POINTER(SoLayer), ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                      ctypes.c_double, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
        lib.so_forward.restype = ctypes.c_int
        lib.so_forward.argtypes = [ctypes.POINTER(SoLayer), ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                   ctypes.c_void_p, ctypes.c_int]
        lib.so_isa.restype = ctypes.c_int
        lib.so_max_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def _table(table):
    arr = (SoLayer * len(table))()
    for i, (fin, fout, act, w_off, b_off) in enumerate(table):
        arr[i] = SoLayer(int(fin), int(fout), int(act), int(w_off), int(b_off))
    return arr


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def isa():
    return {2: "avx512f", 1: "avx2+fma", 0: "scalar"}[load().so_isa()]


def cpu_budget():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a container that sees 256
    logical CPUs but is throttled to 16 runs SLOWER with 256 threads than with 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(per)
    except (OSError, ValueError):
        try:                                                 # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def max_threads():
    return min(int(load().so_max_threads()), cpu_budget()[0])


def forward(table, wflat, x, threads=0):
    lib = load()
    w = np.ascontiguousarray(wflat, dtype=np.float64)
    x = np.asfortranarray(x, dtype=np.float64)
    out = np.empty((table[-1][1], x.shape[1]), dtype=np.float64, order="F")
    if lib.so_forward(_table(table), len(table), _p(w), _p(x), x.shape[1], _p(out), int(threads)) != 0:
        raise MemoryError("so_forward")
    return out


def logdensity(table, w_swa, p, x, y, sigma_m, z, threads=0):
    """oracle.subspace_oracle.logdensity in C (src/space_inference.jl:90-95)."""
    lib = load()
    w_swa = np.ascontiguousarray(w_swa, dtype=np.float64)
    p = np.asfortranarray(p, dtype=np.float64)
    x = np.asfortranarray(x, dtype=np.float64)
    y = np.asfortranarray(y, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64)
    lp = ctypes.c_double()
    rc = lib.so_logdensity(_table(table), len(table), w_swa.size, p.shape[1], _p(w_swa), _p(p), _p(x), _p(y), y.shape[0],
                           x.shape[1], float(sigma_m), _p(z), ctypes.byref(lp), int(threads))
    if rc != 0:
        raise MemoryError("so_logdensity")
    return lp.value
