"""The library's host-only numerics (csrc/eig.cpp: both eigensolver routes) built with AddressSanitizer + UBSan and run
on the CPU.  (GPU-side sanitizers are not available on the MI355X pool.)"""
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
