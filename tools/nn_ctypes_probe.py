"""the C-ABI probe again, from Python WITHOUT numpy (ctypes + array only), then with numpy imported: which import costs the 9 %?"""
import array
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "numpy":
    import numpy  # noqa: F401
if len(sys.argv) > 1 and sys.argv[1] == "package":
    sys.path.insert(0, ROOT)
    import subspaceinference_jl_amd  # noqa: F401
lib = ctypes.CDLL(os.path.join(ROOT, "subspaceinference.jl_amd", "libsubspace_hip.so"))


class Layer(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("in_", ctypes.c_int32), ("out", ctypes.c_int32), ("act", ctypes.c_int32), ("w_off", ctypes.c_int64),
                ("b_off", ctypes.c_int64)] + [(n, ctypes.c_int32) for n in ("kw", "kh", "cin", "cout", "wi", "hi", "sw", "sh", "pw", "ph", "dw", "dh")]


dims, acts, B, M = [2, 200, 50, 50, 50, 1], [1, 1, 1, 1, 0], 1000, 20
lay = (Layer * 5)()
off = 0
for l in range(5):
    lay[l].in_, lay[l].out, lay[l].act, lay[l].w_off = dims[l], dims[l + 1], acts[l], off
    off += dims[l] * dims[l + 1]
    lay[l].b_off = off
    off += dims[l + 1]
N = off
s = 1


def rnd(n, scale):
    global s
    out = array.array("d", bytes(8 * n))
    for i in range(n):
        s = (s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        out[i] = scale * ((s >> 11) / 9007199254740992.0 - 0.5)
    return out


w, P, X, Y = rnd(N, 0.6), rnd(N * M, 0.1), rnd(2 * B, 2.0), rnd(B, 2.0)
ptr = lambda a: ctypes.c_void_p(a.buffer_info()[0])
if len(sys.argv) > 1 and sys.argv[1] == "tooldata":   # the data of tools/nn_example_modes.py (normal draws; its chain rejects nearly every proposal)
    import numpy as np
    rng = np.random.default_rng(0)
    wn, pn = 0.3 * rng.standard_normal(N), np.asfortranarray(0.05 * rng.standard_normal((N, M)))
    xn, yn = np.asfortranarray(rng.standard_normal((2, B))), np.asfortranarray(rng.standard_normal((1, B)))
    keep = (wn, pn, xn, yn)
    w, P, X, Y = [array.array("d", a.tobytes(order="F")) for a in keep]
ctx = ctypes.c_void_p()
lib.si_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32]
assert lib.si_create(ctypes.byref(ctx), 0) == 0
lib.si_infer_setup.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_int32]
assert lib.si_infer_setup(ctx, ctypes.cast(lay, ctypes.c_void_p), 5, N, M, ptr(w), ptr(P), ptr(X), ptr(Y), 2, 1, B, 1.0, 1) == 0
itr = 20000
Z, lp, acc = array.array("d", bytes(8 * M * itr)), array.array("d", bytes(8 * itr)), array.array("d", bytes(8))
lib.si_sample_rwmh.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 3
lib.si_sample_rwmh(ctx, 20, 0.1, 1, 0, 1, ptr(Z), ptr(lp), ptr(acc))
for _ in range(3):
    t0 = time.perf_counter()
    assert lib.si_sample_rwmh(ctx, itr, 0.1, 1, 0, 1, ptr(Z), ptr(lp), ptr(acc)) == 0
    print("python ctypes (%s): %.2f us per transition (lp[last] %.6f)" % (sys.argv[1] if len(sys.argv) > 1 else "no numpy", (time.perf_counter() - t0) / itr * 1e6, lp[itr - 1]), flush=True)
