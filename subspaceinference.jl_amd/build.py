"""Build libsubspace_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and the tests.

The library is plain HIP + C++ (no torch, no Triton, no hipify); it is built next to its sources so that it
travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsubspace_hip.so")
ARCH = "gfx950"

# (source, extra flags).  kernels_stream.hip must not contract a*b+c: K1 is bit-exact against the reference's
# three rounded operations.
SOURCES = [
    ("capi.hip", []),
    ("capi_infer.hip", []),
    ("capi_sample.hip", []),
    ("capi_train.hip", []),
    ("comm.hip", []),
    ("host_copy.cpp", []),
    ("kernels_stream.hip", ["-ffp-contract=off"]),
    ("kernels_chain.hip", ["-ffp-contract=off"]),
    # (-amdgpu-mfma-vgpr-form: accumulators stay in VGPRs; the default moved them between the two register files at every
    #  block boundary of the layer loops, 32 copies + a pipeline drain per 16 MFMAs)
    ("kernels_chain_grid.hip", ["-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("kernels_gemm.hip", []),
    ("kernels_gemm_small.hip", []),
    ("kernels_gemm_f32.hip", []),
    ("kernels_gram.hip", []),
    ("kernels_gram_wave.hip", ["-DSI_GW_PART=0"], "kernels_gram_wave0"),   # same source, three slices of the tile counts
    ("kernels_gram_wave.hip", ["-DSI_GW_PART=1"], "kernels_gram_wave1"),
    ("kernels_gram_wave.hip", ["-DSI_GW_PART=2"], "kernels_gram_wave2"),
    ("kernels_project.hip", []),
    ("kernels_bwd.hip", []),
    ("kernels_bwd_f32.hip", []),
    ("kernels_conv.hip", []),
    ("capi_net.hip", []),
    ("eig.cpp", ["-DSI_EIG_NS=base"]),
    ("eig.cpp", ["-DSI_EIG_NS=avx2", "-mavx2", "-mfma"], "eig_avx2"),   # same source, other ISAs (eig_dispatch.cpp chooses)
    ("eig.cpp", ["-DSI_EIG_NS=avx512", "-mavx512f", "-mavx512vl", "-mavx512dq", "-mfma"], "eig_avx512"),
    ("eig_dispatch.cpp", []),
]
HEADERS = ["si_internal.h", "philox.h", "kernels_gemm.h", "gemm_pipeline.h", "chain_common.h", "capi_common.h", os.path.join("..", "..", "include", "subspace_hip.h")]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libsubspace_hip.so")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


DEV_LIB = os.path.join(os.path.dirname(HERE), "tools", "bin", "libsubspace_hip_dev.so")


def build(force=False, verbose=False, dev=False):
    """dev=True: the DEVELOPMENT build (-DSI_DEV_KNOBS: the alternative kernels and SI_* environment knobs behind the
    measurements quoted in DESIGN.md) into tools/bin/libsubspace_hip_dev.so -- never loaded by the package."""
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build_dev" if dev else "build")
    os.makedirs(objdir, exist_ok=True)
    lib = DEV_LIB if dev else LIB
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs, jobs = [], []
    for entry in SOURCES + ([("guard_alloc.hip", [])] if dev else []):   # (guard-page allocator: development build only)
        src, extra = entry[0], entry[1]
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, (entry[2] if len(entry) > 2 else os.path.splitext(src)[0]) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
                         "-c", s, "-o", o] + extra + (["-DSI_DEV_KNOBS"] if dev else []))

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    if jobs:  # the translation units are independent: compile them side by side (the fully unrolled Gram kernels take minutes)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1), 8)) as ex:
            list(ex.map(run, jobs))
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    import sys
    print(build(verbose=True, dev="--dev" in sys.argv))
