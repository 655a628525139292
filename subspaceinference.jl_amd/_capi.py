"""ctypes binding of libsubspace_hip.so -- the same C ABI the Julia `ccall` wrapper binds
(include/subspace_hip.h; julia/SubspaceInferenceHIP.jl).  No arithmetic happens in this file.

Errors follow the reference's convention (`throw(::String)`, e.g. src/space_inference.jl:42,103,162):
a non-zero status raises `SubspaceError(message)`.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_int, c_int32, c_int64, c_uint64,
                    c_void_p)

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsubspace_hip.so")

SI_OK, SI_ERR_INVALID, SI_ERR_STATE, SI_ERR_HIP, SI_ERR_NOMEM, SI_ERR_BOUNDS, SI_ERR_NODEVICE = 0, -1, -2, -3, -4, -5, -6
SI_ERR_COMM = -7
SI_COMM_ID_BYTES, SI_COMM_SUM, SI_COMM_MAX = 128, 0, 1
SI_F32, SI_F64 = 0, 1
SI_DTYPE_OF_DATA = -1
ACT_IDENTITY, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
ACT_LEAKYRELU, ACT_ELU, ACT_SOFTPLUS, ACT_SELU = 4, 5, 6, 7
K_NAMES = ["push", "gram", "gram_reduce", "project", "reconstruct", "dense", "sse", "rwmh", "dense_main", "eig_host", "backward",
           "conv", "conv_aux"]
LAYER_DENSE, LAYER_CONV, LAYER_MAXPOOL, LAYER_FLATTEN = 0, 1, 2, 3
K_COUNT = len(K_NAMES)


class SubspaceError(RuntimeError):
    """The host-side image of the reference's `throw("...")`."""

    def __init__(self, msg, code=SI_ERR_INVALID):
        super().__init__(msg)
        self.code = code


class BoundsError(SubspaceError, IndexError):
    """Julia's BoundsError at `U[:,1:M]` (src/subspace_construction.jl:65) when rank(A) < M."""


class SiLayer(Structure):
    _fields_ = [("kind", c_int32), ("in_", c_int32), ("out", c_int32), ("act", c_int32),
                ("w_off", c_int64), ("b_off", c_int64),
                ("kw", c_int32), ("kh", c_int32), ("cin", c_int32), ("cout", c_int32), ("wi", c_int32), ("hi", c_int32),
                ("sw", c_int32), ("sh", c_int32), ("pw", c_int32), ("ph", c_int32), ("dw", c_int32), ("dh", c_int32)]


class SiStats(Structure):
    _fields_ = [("ms", c_double * K_COUNT), ("launches", c_int64 * K_COUNT),
                ("flops", c_double * K_COUNT), ("bytes", c_double * K_COUNT)]


# every symbol include/subspace_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "si_version": (c_int32, []),
    "si_create": (c_int32, [POINTER(c_void_p), c_int32]),
    "si_destroy": (c_int32, [c_void_p]),
    "si_last_error": (c_char_p, [c_void_p]),
    "si_set_stream": (c_int32, [c_void_p, c_void_p]),
    "si_synchronize": (c_int32, [c_void_p]),
    "si_set_profiling": (c_int32, [c_void_p, c_int32]),
    "si_set_profiling_classes": (c_int32, [c_void_p, ctypes.c_uint32]),
    "si_get_stats": (c_int32, [c_void_p, POINTER(SiStats)]),
    "si_reset_stats": (c_int32, [c_void_p]),
    "si_device_name": (c_int32, [c_void_p, c_char_p, c_int32]),
    "si_construct_begin": (c_int32, [c_void_p, c_int64, c_int64, c_int32]),
    "si_construct_set_mean": (c_int32, [c_void_p, c_void_p, c_int32]),
    "si_construct_push": (c_int32, [c_void_p, c_void_p, c_int32, c_double]),
    "si_construct_push_dev": (c_int32, [c_void_p, c_void_p, c_int32, c_double]),
    "si_construct_push_batch_dev": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p]),
    "si_construct_gram": (c_int32, [c_void_p]),
    "si_construct_gram_get": (c_int32, [c_void_p, c_void_p, POINTER(c_int64)]),
    "si_construct_gram_set": (c_int32, [c_void_p, c_void_p]),
    "si_construct_gram_ptr": (c_int32, [c_void_p, POINTER(c_void_p), POINTER(c_int64)]),
    "si_construct_result_ptr": (c_int32, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64),
                                          POINTER(c_int32)]),
    "si_construct_needs_refine": (c_int32, [c_void_p, c_int32, POINTER(c_int32)]),
    "si_construct_refine": (c_int32, [c_void_p]),
    "si_construct_finish": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, POINTER(c_int64)]),
    "si_construct_get_A": (c_int32, [c_void_p, c_int64, c_int64, c_void_p]),
    "si_construct_get_result": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int64), POINTER(c_int32)]),
    "si_infer_setup": (c_int32, [c_void_p, POINTER(SiLayer), c_int32, c_int64, c_int32, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_int32, c_int32, c_int64, c_double, c_int32]),
    "si_infer_setup_dev": (c_int32, [c_void_p, POINTER(SiLayer), c_int32, c_int64, c_int32, c_void_p, c_void_p, c_int64,
                                     c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int64, c_double, c_int32]),
    "si_infer_set_prior": (c_int32, [c_void_p, c_double]),
    "si_logdensity": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "si_logdensity_grad": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "si_forward": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "si_predict": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    "si_sample_rwmh": (c_int32, [c_void_p, c_int64, c_double, c_uint64, c_int32, c_int32, c_void_p, c_void_p,
                                 c_void_p]),
    "si_sample_rwmh_weights": (c_int32, [c_void_p, c_int64, c_double, c_uint64, c_int32, c_int32, c_void_p, c_void_p,
                                         c_void_p, c_void_p]),
    "si_rwmh_begin": (c_int32, [c_void_p, c_int64, c_double, c_uint64, c_int32, c_int32, c_int64]),
    "si_rwmh_step_eval": (c_int32, [c_void_p, c_void_p]),
    "si_rwmh_step_accept": (c_int32, [c_void_p, c_void_p]),
    "si_rwmh_sse_ptr": (c_int32, [c_void_p, POINTER(c_void_p), POINTER(c_int32)]),
    "si_rwmh_end": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "si_rwmh_abort": (c_int32, [c_void_p]),
    "si_reconstruct": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "si_train_setup": (c_int32, [c_void_p, POINTER(SiLayer), c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_int32,
                                 c_int32, c_int64, c_int64, c_int32, c_double, c_double, c_double]),
    "si_train_setup_ex": (c_int32, [c_void_p, POINTER(SiLayer), c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                    c_int32, c_int64, c_int64, c_int32, c_double, c_double, c_double, c_int32]),
    "si_train_compute_dtype": (c_int32, [c_void_p, c_void_p]),
    "si_train_step": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "si_train_push": (c_int32, [c_void_p, c_double]),
    "si_train_get_weights": (c_int32, [c_void_p, c_void_p]),
    "si_train_get_opt_state": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "si_train_grad": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "si_train_grad_ptr": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "si_train_grad_get": (c_int32, [c_void_p, c_void_p]),
    "si_train_grad_set": (c_int32, [c_void_p, c_void_p]),
    "si_train_apply": (c_int32, [c_void_p]),
    "si_comm_unique_id": (c_int32, [c_void_p]),
    "si_comm_init_rank": (c_int32, [c_void_p, c_int32, c_int32, c_void_p]),
    "si_comm_destroy": (c_int32, [c_void_p]),
    "si_comm_info": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "si_comm_allreduce_host": (c_int32, [c_void_p, c_void_p, c_int64, c_int32]),
    "si_comm_allgather_host": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "si_comm_barrier": (c_int32, [c_void_p]),
    "si_row_shard": (c_int32, [c_int64, c_int32, c_int32, POINTER(c_int64), POINTER(c_int64)]),
    "si_construct_allreduce_gram": (c_int32, [c_void_p]),
    "si_construct_allgather": (c_int32, [c_void_p, c_int64]),
    "si_bcast_subspace": (c_int32, [c_void_p, c_int32, c_int64, c_int32]),
    "si_rwmh_allreduce_sse": (c_int32, [c_void_p]),
    "si_sample_rwmh_sharded": (c_int32, [c_void_p, c_int64, c_double, c_uint64, c_int32, c_int32, c_int64, c_void_p, c_void_p,
                                         c_void_p]),
    "si_train_allreduce_grad": (c_int32, [c_void_p, c_void_p]),
    "si_train_step_dp": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "si_host_sym_eig": (c_int, [c_int, c_void_p, c_void_p]),
    "si_host_sym_eig_top": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "si_host_jacobi_eig_psd": (c_int, [c_int, c_void_p, c_void_p, c_void_p]),
    "si_set_chain_loop": (c_int32, [c_void_p, c_int32]),
    "si_chain_kernel_info": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_int32)]),
    "si_chain_spec_message": (c_char_p, [c_void_p]),
    "si_construct_set_storage": (c_int32, [c_void_p, c_int32]),
    "si_host_cpu_budget": (c_int, []),
    "si_host_parse_cpu_max": (c_double, [c_char_p]),
    "si_host_copy_plan": (c_int, [c_int, c_int, c_char_p]),
}

_lib = None


def load():
    """dlopen the in-tree library.  Fails loudly when it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SubspaceError(
            "libsubspace_hip.so is missing (%s): build it with `python __graft_entry__.py` -- "
            "this package has no CPU or PyTorch fallback" % LIB_PATH, SI_ERR_NODEVICE)
    if os.environ.get("SI_PRELOAD_TORCH", "1") != "0":
        # torch wheels bundle their own libamdhip64.so.7; when both live in one process the HIP runtime
        # must be loaded once.  Importing torch first makes the library bind to the already loaded runtime.
        try:
            import torch  # noqa: F401
        except Exception:  # torch is plumbing only; the library itself does not need it
            pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    # (a.ctypes builds a helper object per call -- a quarter of a millisecond in a loop of training steps)
    return None if a is None else c_void_p(a.__array_interface__["data"][0])


def conv_out_size(wi, k, s, p, d):
    return (wi + 2 * p - d * (k - 1) - 1) // s + 1


def _layer_array(table):
    """table rows: Dense (in, out, act, w_off, b_off), or
         ("conv", (kw, kh, cin, cout), (wi, hi), (sw, sh), (pw, ph), (dw, dh), act, w_off, b_off)
         ("maxpool", (pw, ph), c, (wi, hi), (sw, sh))        ("flatten", c, (wi, hi))"""
    arr = (SiLayer * len(table))()
    for i, row in enumerate(table):
        if row[0] == "conv":
            _, (kw, kh, cin, cout), (wi, hi), (sw, sh), (pw, ph), (dw, dh), act, w_off, b_off = row
            wo, ho = conv_out_size(wi, kw, sw, pw, dw), conv_out_size(hi, kh, sh, ph, dh)
            arr[i] = SiLayer(LAYER_CONV, wi * hi * cin, wo * ho * cout, int(act), int(w_off), int(b_off), kw, kh, cin, cout,
                             wi, hi, sw, sh, pw, ph, dw, dh)
        elif row[0] == "maxpool":
            _, (kw, kh), c, (wi, hi), (sw, sh) = row
            wo, ho = (wi - kw) // sw + 1, (hi - kh) // sh + 1
            arr[i] = SiLayer(LAYER_MAXPOOL, wi * hi * c, wo * ho * c, 0, 0, 0, kw, kh, c, c, wi, hi, sw, sh, 0, 0, 1, 1)
        elif row[0] == "flatten":
            _, c, (wi, hi) = row
            arr[i] = SiLayer(LAYER_FLATTEN, wi * hi * c, wi * hi * c, 0, 0, 0, 0, 0, c, c, wi, hi, 1, 1, 0, 0, 1, 1)
        else:
            fin, fout, act, w_off, b_off = row[:5]
            arr[i] = SiLayer(LAYER_DENSE, int(fin), int(fout), int(act), int(w_off), int(b_off))
    return arr


def _f64(a, order="F"):
    return np.require(a, dtype=np.float64, requirements=[order, "A"])


class Context:
    """Owns one `si_ctx` (one GPU).  Thin: argument marshalling + status -> exception."""

    def __init__(self, device=0):
        self.lib = load()
        h = c_void_p()
        rc = self.lib.si_create(byref(h), int(device))
        if rc != SI_OK:
            msg = self.lib.si_last_error(None).decode()
            raise SubspaceError(msg, rc)
        self.h = h
        self.device = int(device)

    # -- plumbing
    def _check(self, rc):
        if rc == SI_OK:
            return
        msg = self.lib.si_last_error(self.h).decode()
        if rc == SI_ERR_BOUNDS:
            raise BoundsError(msg, rc)
        raise SubspaceError(msg, rc)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.si_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        self._check(self.lib.si_device_name(self.h, buf, 256))
        return buf.value.decode()

    def set_stream(self, stream_handle):
        self._check(self.lib.si_set_stream(self.h, c_void_p(stream_handle) if stream_handle else None))

    def synchronize(self):
        self._check(self.lib.si_synchronize(self.h))

    def set_profiling(self, on, classes=None):
        """classes: iterable of class names (K_NAMES) that get event pairs; None = all."""
        self._check(self.lib.si_set_profiling(self.h, 1 if on else 0))
        mask = 0xFFFFFFFF if classes is None else sum(1 << K_NAMES.index(k) for k in classes)
        self._check(self.lib.si_set_profiling_classes(self.h, mask))

    def reset_stats(self):
        self._check(self.lib.si_reset_stats(self.h))

    def stats(self):
        st = SiStats()
        self._check(self.lib.si_get_stats(self.h, byref(st)))
        return {K_NAMES[i]: {"ms": st.ms[i], "launches": st.launches[i], "flops": st.flops[i], "bytes": st.bytes[i]}
                for i in range(K_COUNT)}

    # -- R1: the RCCL communicator of this ctx (one rank per ctx / GPU / process)
    def comm_init_rank(self, world, rank, uid):
        if len(uid) != SI_COMM_ID_BYTES:
            raise SubspaceError("comm_init_rank: the id must be %d bytes" % SI_COMM_ID_BYTES)
        buf = ctypes.create_string_buffer(bytes(uid), SI_COMM_ID_BYTES)
        self._check(self.lib.si_comm_init_rank(self.h, int(world), int(rank), buf))

    def comm_destroy(self):
        self._check(self.lib.si_comm_destroy(self.h))

    def comm_info(self):
        """(world, rank, rccl_version); world = 0 without a communicator"""
        w, r, v = c_int32(), c_int32(), c_int32()
        self._check(self.lib.si_comm_info(self.h, byref(w), byref(r), byref(v)))
        return int(w.value), int(r.value), int(v.value)

    def comm_world(self):
        return self.comm_info()[0]

    def comm_allreduce_host(self, values, op="sum"):
        a = np.array(values, dtype=np.float64, order="C").reshape(-1)
        self._check(self.lib.si_comm_allreduce_host(self.h, _ptr(a), a.size, SI_COMM_SUM if op == "sum" else SI_COMM_MAX))
        return a.reshape(np.shape(values))

    def comm_allgather_host(self, values):
        """(world, n) array: row r = rank r's `values`"""
        a = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        out = np.empty((self.comm_world(), a.size), dtype=np.float64)
        self._check(self.lib.si_comm_allgather_host(self.h, _ptr(a), a.size, _ptr(out)))
        return out

    def comm_barrier(self):
        self._check(self.lib.si_comm_barrier(self.h))

    def construct_allreduce_gram(self):
        self._check(self.lib.si_construct_allreduce_gram(self.h))

    def construct_allgather(self, n_total):
        """full (W_swa, P) on every rank, device to device; the ctx then holds a finished construction of n_total rows"""
        self._check(self.lib.si_construct_allgather(self.h, int(n_total)))
        self._n = int(n_total)

    def bcast_subspace(self, root, n, m):
        self._check(self.lib.si_bcast_subspace(self.h, int(root), int(n), int(m)))
        self._n = int(n)

    def rwmh_allreduce_sse(self):
        self._check(self.lib.si_rwmh_allreduce_sse(self.h))

    def sample_rwmh_sharded(self, itr, sigma_z, seed, d_total, chain_id0=0, nchains=1):
        z = np.empty((self._m, int(itr), int(nchains)), dtype=np.float64, order="F")
        lp = np.empty((int(itr), int(nchains)), dtype=np.float64, order="F")
        acc = np.empty(int(nchains), dtype=np.float64)
        self._check(self.lib.si_sample_rwmh_sharded(self.h, int(itr), float(sigma_z), int(seed), int(chain_id0), int(nchains),
                                                    int(d_total), _ptr(z), _ptr(lp), _ptr(acc)))
        return z, lp, acc

    def train_allreduce_grad(self, want_sse=True):
        sse = np.empty(1, dtype=np.float64) if want_sse else None
        self._check(self.lib.si_train_allreduce_grad(self.h, _ptr(sse)))
        return float(sse[0]) if want_sse else None

    def train_step_dp(self, idx, nb_total, want_loss=True):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        loss = np.empty(1, dtype=np.float64) if want_loss else None
        self._check(self.lib.si_train_step_dp(self.h, _ptr(idx) if idx.size else None, idx.size, int(nb_total), _ptr(loss)))
        return float(loss[0]) if want_loss else None

    def construct_get_result(self, want_p=True):
        """(W_swa, P, s) of the FINISHED construction the ctx holds -- its own, or one received through bcast_subspace /
        construct_allgather -- copied to the host."""
        n, m = c_int64(), c_int32()
        self._check(self.lib.si_construct_get_result(self.h, None, None, None, byref(n), byref(m)))
        w = np.empty(n.value, dtype=np.float64)
        p = np.empty((n.value, m.value), dtype=np.float64, order="F") if want_p else None
        s = np.empty(m.value, dtype=np.float64)
        self._check(self.lib.si_construct_get_result(self.h, _ptr(w), _ptr(p), _ptr(s), byref(n), byref(m)))
        return w, p, s

    # -- construction
    def construct_begin(self, n, k_capacity, max_cols=0):
        self._check(self.lib.si_construct_begin(self.h, int(n), int(k_capacity), int(max_cols)))
        self._n = int(n)

    def construct_set_storage(self, a_dtype):
        """SI_F32: keep the deviation matrix in fp32 (opt-in); right after construct_begin."""
        self._check(self.lib.si_construct_set_storage(self.h, int(a_dtype)))

    def construct_set_mean(self, w):
        """non-default init = :pretrained (Q1): W_swa starts at w instead of zeros"""
        w = np.ascontiguousarray(w)
        if w.dtype not in (np.float32, np.float64) or w.size != self._n:
            raise SubspaceError("DimensionMismatch / dtype: initial mean needs %d Float32 or Float64 values" % self._n)
        self._check(self.lib.si_construct_set_mean(self.h, _ptr(w), SI_F32 if w.dtype == np.float32 else SI_F64))

    def construct_push(self, w, n):
        w = np.ascontiguousarray(w)
        if w.dtype == np.float32:
            dt = SI_F32
        elif w.dtype == np.float64:
            dt = SI_F64
        else:
            raise SubspaceError("weights must be Float32 or Float64, got %s" % w.dtype)
        if w.size != self._n:
            raise SubspaceError("DimensionMismatch: snapshot has %d elements, expected %d" % (w.size, self._n))
        self._check(self.lib.si_construct_push(self.h, _ptr(w), dt, float(n)))

    def construct_push_dev(self, dev_ptr, dtype, n):
        self._check(self.lib.si_construct_push_dev(self.h, c_void_p(int(dev_ptr)), int(dtype), float(n)))

    def construct_push_batch_dev(self, dev_ptr, dtype, ld, ns):
        ns = np.ascontiguousarray(ns, dtype=np.float64)
        self._check(self.lib.si_construct_push_batch_dev(self.h, c_void_p(int(dev_ptr)), int(dtype), int(ld), ns.size,
                                                         _ptr(ns)))

    def construct_gram(self):
        self._check(self.lib.si_construct_gram(self.h))

    def construct_gram_get(self):
        k = c_int64()
        self._check(self.lib.si_construct_gram_get(self.h, None, byref(k)))
        g = np.empty((k.value, k.value), dtype=np.float64, order="F")
        self._check(self.lib.si_construct_gram_get(self.h, _ptr(g), byref(k)))
        return g

    def construct_gram_set(self, g):
        g = _f64(g)
        self._check(self.lib.si_construct_gram_set(self.h, _ptr(g)))

    def construct_needs_refine(self, m):
        """True when the Gram route cannot resolve s_M (ill-conditioned A): the two-stage route is needed."""
        flag = c_int32()
        self._check(self.lib.si_construct_needs_refine(self.h, int(m), byref(flag)))
        return bool(flag.value)

    def construct_refine(self):
        self._check(self.lib.si_construct_refine(self.h))

    def construct_gram_ptr(self):
        """(device address, K) of the K x K fp64 Gram matrix, for an in-place RCCL all-reduce."""
        ptr, k = c_void_p(), c_int64()
        self._check(self.lib.si_construct_gram_ptr(self.h, byref(ptr), byref(k)))
        return int(ptr.value), int(k.value)

    def construct_result_ptr(self):
        """(W_swa device address, P device address, ld, M) of the finished construction."""
        ws, pp, ld, m = c_void_p(), c_void_p(), c_int64(), c_int32()
        self._check(self.lib.si_construct_result_ptr(self.h, byref(ws), byref(pp), byref(ld), byref(m)))
        return int(ws.value), int(pp.value), int(ld.value), int(m.value)

    def construct_finish(self, m, want_swa=True, want_p=True):
        k = c_int64()
        w_swa = np.empty(self._n, dtype=np.float64) if want_swa else None
        p = np.empty((self._n, int(m)), dtype=np.float64, order="F") if want_p else None
        s = np.empty(int(m), dtype=np.float64)
        self._check(self.lib.si_construct_finish(self.h, int(m), _ptr(w_swa), _ptr(p), _ptr(s), byref(k)))
        return w_swa, p, s, k.value

    def construct_get_A(self, k0, nk):
        a = np.empty((self._n, int(nk)), dtype=np.float64, order="F")
        self._check(self.lib.si_construct_get_A(self.h, int(k0), int(nk), _ptr(a)))
        return a

    # -- on-device training (f1)
    def train_setup(self, table, n, w0, x, y, batch_max, opt_kind, eta, p1=0.0, p2=0.0, compute_dtype=None):
        """si_train_setup_ex.  compute_dtype None: the reference's own arithmetic for the data handed in -- Float32 (x, y) give
        the all-Float32 pass (SI_F32), anything else is promoted to Float64 as Julia would; SI_F32 / SI_F64 override."""
        arr = _layer_array(table)
        x, y = np.asarray(x), np.asarray(y)
        data_f32 = x.dtype == np.float32 and y.dtype == np.float32
        dt = np.float32 if data_f32 else np.float64
        x, y = np.asfortranarray(x, dtype=dt), np.asfortranarray(y, dtype=dt)
        w0 = np.ascontiguousarray(w0, dtype=np.float32)
        if w0.size != n or x.ndim != 2 or y.ndim != 2 or x.shape[1] != y.shape[1]:
            raise SubspaceError("DimensionMismatch: w0 %s, X %s, Y %s" % (w0.shape, x.shape, y.shape))
        self._check(self.lib.si_train_setup_ex(self.h, arr, len(table), int(n), _ptr(w0), _ptr(x), _ptr(y),
                                               SI_F32 if data_f32 else SI_F64, x.shape[0], y.shape[0], x.shape[1], int(batch_max),
                                               int(opt_kind), float(eta), float(p1), float(p2),
                                               SI_DTYPE_OF_DATA if compute_dtype is None else int(compute_dtype)))
        self._tn = int(n)
        self._tout = int(y.shape[0])

    def train_compute_dtype(self):
        out = np.zeros(1, dtype=np.int32)
        self._check(self.lib.si_train_compute_dtype(self.h, _ptr(out)))
        return int(out[0])

    def train_step(self, idx, want_loss=True):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        loss = np.empty(1, dtype=np.float64) if want_loss else None
        self._check(self.lib.si_train_step(self.h, _ptr(idx), idx.size, _ptr(loss)))
        return float(loss[0]) if want_loss else None

    # data-parallel form: grad -> (caller's all-reduce) -> apply
    def train_grad(self, idx, nb_total):
        """Gradient of mse over the WHOLE batch (nb_total observations) restricted to this rank's idx; returns local SSE."""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        sse = np.empty(1, dtype=np.float64)
        self._check(self.lib.si_train_grad(self.h, _ptr(idx), idx.size, int(nb_total), _ptr(sse)))
        return float(sse[0])

    def train_grad_ptr(self):
        """(device address, element count) of the fp64 gradient buffer, for an in-place RCCL all-reduce."""
        ptr, n = c_void_p(), c_int64()
        self._check(self.lib.si_train_grad_ptr(self.h, byref(ptr), byref(n)))
        return int(ptr.value), int(n.value)

    def train_grad_get(self):
        g = np.empty(self._tn, dtype=np.float64)
        self._check(self.lib.si_train_grad_get(self.h, _ptr(g)))
        return g

    def train_grad_set(self, g):
        g = np.ascontiguousarray(g, dtype=np.float64)
        if g.size != self._tn:
            raise SubspaceError("DimensionMismatch: gradient of %d elements for %d weights" % (g.size, self._tn))
        self._check(self.lib.si_train_grad_set(self.h, _ptr(g)))

    def train_apply(self):
        self._check(self.lib.si_train_apply(self.h))

    def train_out_dim(self):
        return self._tout

    def train_push(self, n):
        self._check(self.lib.si_train_push(self.h, float(n)))

    def train_get_weights(self):
        w = np.empty(self._tn, dtype=np.float32)
        self._check(self.lib.si_train_get_weights(self.h, _ptr(w)))
        return w

    def train_get_opt_state(self):
        """(m, v, (beta1^t, beta2^t)) -- Float32 flat state of the device optimiser."""
        m = np.empty(self._tn, dtype=np.float32)
        v = np.empty(self._tn, dtype=np.float32)
        bp = np.empty(2, dtype=np.float64)
        self._check(self.lib.si_train_get_opt_state(self.h, _ptr(m), _ptr(v), _ptr(bp)))
        return m, v, (float(bp[0]), float(bp[1]))

    # -- density + sampling
    def infer_setup(self, table, n, m, w_swa, p, x, y, sigma_m, compute_dtype=SI_F64):
        """table: list of (in, out, act, w_off, b_off); w_swa/p None => reuse the finished construction.
        compute_dtype: SI_F64 (the reference's arithmetic) or SI_F32 (Dense chains: fp32 X / weights / activations on the
        fp32 matrix instruction, head + SSE in fp64; the measured option of SURVEY section 0 Q6)."""
        arr = _layer_array(table)
        x = _f64(x)
        y = _f64(y)
        if x.ndim != 2 or y.ndim != 2 or x.shape[1] != y.shape[1]:
            raise SubspaceError("DimensionMismatch: X is %s, Y is %s" % (x.shape, y.shape))
        if w_swa is not None:
            w_swa = _f64(w_swa)
            p = _f64(p)
            if w_swa.shape != (n,) or p.shape != (n, m):
                raise SubspaceError("DimensionMismatch: W_swa %s, P %s, expected (%d,), (%d, %d)"
                                    % (w_swa.shape, p.shape, n, n, m))
        self._check(self.lib.si_infer_setup(self.h, arr, len(table), int(n), int(m), _ptr(w_swa), _ptr(p), _ptr(x),
                                            _ptr(y), x.shape[0], y.shape[0], x.shape[1], float(sigma_m), int(compute_dtype)))
        self._m, self._in, self._out, self._b, self._ni = int(m), x.shape[0], y.shape[0], x.shape[1], int(n)

    def infer_setup_dev(self, table, n, m, w_swa_ptr, p_ptr, ld_p, x_ptr, y_ptr, in_dim, out_dim, b, sigma_m, borrow=False,
                        compute_dtype=SI_F64):
        """si_infer_setup with device addresses (ints); w_swa_ptr / p_ptr 0 or None => the finished construction."""
        arr = _layer_array(table)
        self._check(self.lib.si_infer_setup_dev(
            self.h, arr, len(table), int(n), int(m), c_void_p(int(w_swa_ptr)) if w_swa_ptr else None,
            c_void_p(int(p_ptr)) if p_ptr else None, int(ld_p), 1 if borrow else 0, c_void_p(int(x_ptr)), c_void_p(int(y_ptr)),
            int(in_dim), int(out_dim), int(b), float(sigma_m), int(compute_dtype)))
        self._m, self._in, self._out, self._b, self._ni = int(m), int(in_dim), int(out_dim), int(b), int(n)

    def set_prior(self, sigma_p):
        """non-default include_prior (Q4): sigma_p > 0 adds logpdf(MvNormal(zeros(N), sigma_p), W_swa + P z); 0 = off"""
        self._check(self.lib.si_infer_set_prior(self.h, float(sigma_p)))

    def logdensity(self, z):
        z = _f64(z)
        if z.ndim == 1:
            z = z.reshape(-1, 1, order="F")
        if z.shape[0] != self._m:
            raise SubspaceError("DimensionMismatch: z has %d rows, M = %d" % (z.shape[0], self._m))
        lp = np.empty(z.shape[1], dtype=np.float64)
        self._check(self.lib.si_logdensity(self.h, _ptr(z), z.shape[1], _ptr(lp)))
        return lp

    def logdensity_grad(self, z):
        z = _f64(z).reshape(-1)
        if z.size != self._m:
            raise SubspaceError("DimensionMismatch: z has %d elements, M = %d" % (z.size, self._m))
        lp = np.empty(1, dtype=np.float64)
        g = np.empty(self._m, dtype=np.float64)
        self._check(self.lib.si_logdensity_grad(self.h, _ptr(z), _ptr(lp), _ptr(g)))
        return float(lp[0]), g

    def forward(self, z):
        z = _f64(z).reshape(-1)
        yhat = np.empty((self._out, self._b), dtype=np.float64, order="F")
        self._check(self.lib.si_forward(self.h, _ptr(z), _ptr(yhat)))
        return yhat

    def predict(self, z, xnew):
        z = _f64(z)
        if z.ndim == 1:
            z = z.reshape(-1, 1, order="F")
        xnew = _f64(xnew)
        if z.shape[0] != self._m or xnew.ndim != 2 or xnew.shape[0] != self._in:
            raise SubspaceError("DimensionMismatch: Z %s, Xnew %s" % (z.shape, xnew.shape))
        out = np.empty((self._out, xnew.shape[1], z.shape[1]), dtype=np.float64, order="F")
        self._check(self.lib.si_predict(self.h, _ptr(z), z.shape[1], _ptr(xnew), xnew.shape[1], _ptr(out)))
        return out

    def sample_rwmh(self, itr, sigma_z, seed, chain_id0=0, nchains=1, want_z=True):
        z = np.empty((self._m, int(itr), int(nchains)), dtype=np.float64, order="F") if want_z else None
        lp = np.empty((int(itr), int(nchains)), dtype=np.float64, order="F")
        acc = np.empty(int(nchains), dtype=np.float64)
        self._check(self.lib.si_sample_rwmh(self.h, int(itr), float(sigma_z), int(seed), int(chain_id0), int(nchains),
                                            _ptr(z), _ptr(lp), _ptr(acc)))
        return z, lp, acc

    def sample_rwmh_weights(self, itr, sigma_z, seed, chain_id0=0, nchains=1, out=None):
        """si_sample_rwmh + the output map (:125) streamed while the chain runs: returns (z, lp, acc, W) with
        W of shape (N, itr, nchains), Fortran order -- W[:, t, c] == W_swa + P z[:, t, c]."""
        z = np.empty((self._m, int(itr), int(nchains)), dtype=np.float64, order="F")
        lp = np.empty((int(itr), int(nchains)), dtype=np.float64, order="F")
        acc = np.empty(int(nchains), dtype=np.float64)
        if out is None:
            w = np.empty((self._ni, int(itr), int(nchains)), dtype=np.float64, order="F")
        else:
            w = out
            if w.dtype != np.float64 or w.shape != (self._ni, int(itr), int(nchains)) or not w.flags.f_contiguous:
                raise ValueError("sample_rwmh_weights: `out` must be a Fortran-ordered float64 array of shape (N, itr, nchains)")
        self._check(self.lib.si_sample_rwmh_weights(self.h, int(itr), float(sigma_z), int(seed), int(chain_id0), int(nchains),
                                                    _ptr(z), _ptr(lp), _ptr(acc), _ptr(w)))
        return z, lp, acc, w

    def set_chain_loop(self, on):
        """False / 0: one launch per layer and per step of si_sample_rwmh; True / 1 (default): small and narrow Dense chains run
        fused and device-resident; 2: the one-launch density for narrow chains, but no device-resident loop; 3 / 4: as 1 / 2 with
        the generic kernels only (no run-time specialisation to the chain's shapes)."""
        self._check(self.lib.si_set_chain_loop(self.h, int(on)))

    def chain_kernel_info(self):
        """(density_specialised, loop_specialised, message): did the last density evaluation / si_sample_rwmh* call run the kernels
        compiled for this chain's shapes at run time (hiprtc), and if not, why"""
        d, l = c_int32(0), c_int32(0)
        self._check(self.lib.si_chain_kernel_info(self.h, byref(d), byref(l)))
        return bool(d.value), bool(l.value), (self.lib.si_chain_spec_message(self.h) or b"").decode()

    def rwmh_begin(self, itr, sigma_z, seed, chain_id0=0, nchains=1, d_total=0):
        self._check(self.lib.si_rwmh_begin(self.h, int(itr), float(sigma_z), int(seed), int(chain_id0), int(nchains),
                                           int(d_total)))
        self._sw = (int(itr), int(nchains))

    def rwmh_step_eval(self, on_device=False):
        """on_device: leave the partial sums in the library's device buffer (rwmh_sse_ptr) -- no copy, no sync."""
        if on_device:
            self._check(self.lib.si_rwmh_step_eval(self.h, None))
            return None
        sse = np.empty(self._sw[1], dtype=np.float64)
        self._check(self.lib.si_rwmh_step_eval(self.h, _ptr(sse)))
        return sse

    def rwmh_step_accept(self, sse_total=None):
        """sse_total None: the device buffer already holds the totals (all-reduced in place)."""
        if sse_total is None:
            self._check(self.lib.si_rwmh_step_accept(self.h, None))
            return
        sse_total = np.ascontiguousarray(sse_total, dtype=np.float64)
        self._check(self.lib.si_rwmh_step_accept(self.h, _ptr(sse_total)))

    def rwmh_sse_ptr(self):
        ptr, c = c_void_p(), c_int32()
        self._check(self.lib.si_rwmh_sse_ptr(self.h, byref(ptr), byref(c)))
        return int(ptr.value), int(c.value)

    def rwmh_abort(self):
        self._check(self.lib.si_rwmh_abort(self.h))

    def rwmh_end(self):
        itr, c = self._sw
        z = np.empty((self._m, itr, c), dtype=np.float64, order="F")
        lp = np.empty((itr, c), dtype=np.float64, order="F")
        acc = np.empty(c, dtype=np.float64)
        self._check(self.lib.si_rwmh_end(self.h, _ptr(z), _ptr(lp), _ptr(acc)))
        return z, lp, acc

    def reconstruct(self, z, out=None):
        """W_swa + P z for every column of z (space_inference.jl:125); `out` (N x C, Fortran order, float64) is filled in
        place when given (e.g. a buffer the caller reuses), else a new array is returned."""
        z = _f64(z)
        if z.ndim == 1:
            z = z.reshape(-1, 1, order="F")
        if out is None:
            w = np.empty((self._ni, z.shape[1]), dtype=np.float64, order="F")
        else:
            w = out
            if w.dtype != np.float64 or w.shape != (self._ni, z.shape[1]) or not w.flags.f_contiguous:
                raise ValueError("reconstruct: `out` must be a Fortran-ordered float64 array of shape (N, C)")
        self._check(self.lib.si_reconstruct(self.h, _ptr(z), z.shape[1], _ptr(w)))
        return w


def comm_unique_id():
    """128-byte RCCL id (rank 0 creates it and ships it to the other ranks; needs no ctx)."""
    lib = load()
    buf = ctypes.create_string_buffer(SI_COMM_ID_BYTES)
    rc = lib.si_comm_unique_id(buf)
    if rc != SI_OK:
        raise SubspaceError(lib.si_last_error(None).decode(), rc)
    return buf.raw


def row_shard(n_total, rank, world):
    """si_row_shard: rows [r0, r1) of rank `rank` (32-element aligned boundaries) -- the partition the library assumes."""
    r0, r1 = c_int64(), c_int64()
    if load().si_row_shard(int(n_total), int(rank), int(world), byref(r0), byref(r1)) != SI_OK:
        raise SubspaceError("si_row_shard: bad argument")
    return int(r0.value), int(r1.value)


def host_sym_eig(g):
    """The host eigensolver of si_construct_finish (needs no GPU)."""
    lib = load()
    a = np.array(g, dtype=np.float64, order="F")
    n = a.shape[0]
    w = np.empty(n, dtype=np.float64)
    rc = lib.si_host_sym_eig(n, _ptr(a), _ptr(w))
    if rc != 0:
        raise SubspaceError("eigensolver did not converge")
    return w, a


def host_sym_eig_top(g, m):
    """The m largest eigenpairs by the fast host route of si_construct_finish (needs no GPU).  Returns
    (w_top descending, V n x m), or None when the route declines / fails its own verification."""
    lib = load()
    a = np.array(g, dtype=np.float64, order="F")
    n = a.shape[0]
    w = np.empty(int(m), dtype=np.float64)
    v = np.empty((n, int(m)), dtype=np.float64, order="F")
    rc = lib.si_host_sym_eig_top(n, _ptr(a), int(m), _ptr(w), _ptr(v))
    return (w, v) if rc == 0 else None


def host_jacobi_eig_psd(g):
    """Scaled-criterion Jacobi of the ill-conditioned route (needs no GPU): (w descending, V)."""
    lib = load()
    a = np.array(g, dtype=np.float64, order="F")
    n = a.shape[0]
    w = np.empty(n, dtype=np.float64)
    v = np.empty((n, n), dtype=np.float64, order="F")
    lib.si_host_jacobi_eig_psd(n, _ptr(a), _ptr(w), _ptr(v))
    return w, v
