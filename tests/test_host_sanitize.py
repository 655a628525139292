"""The library's host-only code built with sanitizers and run on the CPU (GPU-side sanitizers are not available on the MI355X
pool): csrc/eig.cpp (both eigensolver routes) under AddressSanitizer + UBSan, csrc/host_copy.cpp (the persistent copy pool
of si_construct_push / the output map: atomics, a spin-then-sleep hand-over, two caller threads) under ThreadSanitizer and
under AddressSanitizer."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.timeout(300)
def test_eigensolver_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "eig_sanitize")
    src = os.path.join(HERE, "native", "eig_sanitize_main.cpp")
    build = subprocess.run([gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-fno-omit-frame-pointer", src, "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=240,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0 and "EIG_SANITIZE_OK" in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr


@pytest.mark.timeout(600)
@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_host_copy_pool_under_sanitizers(tmp_path, san):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "host_copy_san")
    srcs = [os.path.join(HERE, "native", "host_copy_tsan_main.cpp"),
            os.path.join(root, "subspaceinference.jl_amd", "csrc", "host_copy.cpp")]
    build = subprocess.run([gxx, "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=" + san, "-fno-omit-frame-pointer",
                            "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"] + srcs + ["-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and ("cannot find" in build.stderr or "unrecognized" in build.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=500, env=dict(os.environ, SI_HOST_COPY_THREADS="4"))
    assert run.returncode == 0 and "HOST_COPY_OK" in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]
    assert "ThreadSanitizer" not in run.stderr and "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-4000:]
