"""The steady-state loop of dense_f64_panel_kernel (csrc/kernels_gemm_panel.hip) as the compiler emits it for gfx950.

The kernel's gain over the tile kernel is ONE property of its tile loop: nothing in it waits for the stores of the tiles before
(the only vector-memory wait is the hand-counted `s_waitcnt vmcnt(16)`).  Three things the compiler can do silently take that
away, each seen while the kernel was written and each leaving results unchanged -- so this is checked on the
assembly, for every instantiation, without a GPU:
  * a register spilled across the loop: its `scratch_load` behind the barrier is a vector-memory load, and the wait for it is a
    wait for every store in front of it;
  * an `s_waitcnt vmcnt(0)` of its own (a pending load entering the loop, LDS reads it can see behind an LDS-DMA);
  * a global load inside the loop (the bias went through LDS for that reason).
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "subspaceinference.jl_amd", "csrc")


@pytest.fixture(scope="module")
def panel_asm():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "panel.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", "-I", CSRC,
                        "-I", os.path.join(ROOT, "include"), os.path.join(CSRC, "kernels_gemm_panel.hip"), "-o", out], check=True, timeout=600)
        return open(out).read()


def _kernels(asm):
    """{mangled name: body lines} of every dense_f64_panel_kernel instantiation"""
    found, cur, name = {}, None, None
    for ln in asm.splitlines():
        m = re.match(r"^(_ZN2si22dense_f64_panel_kernelI\w+):", ln)
        if m:
            name, cur = m.group(1), []
        elif cur is not None:
            cur.append(ln.strip())
            if ".end_amdhsa_kernel" in ln:
                found[name], cur = cur, None
    return found


def test_tile_loop_waits_for_nothing_but_its_counted_wait(panel_asm):
    kernels = _kernels(panel_asm)
    assert len(kernels) == 16, sorted(kernels)       # KT = 1 .. 8, whole / ragged reduction
    for name, body in kernels.items():
        # the tile loop: from its barrier (the last s_barrier of the kernel) to the fourth output store behind it
        b = max(i for i, ln in enumerate(body) if ln.startswith("s_barrier"))
        stores = [i for i, ln in enumerate(body) if i > b and ln.startswith("global_store_dwordx2")]
        assert len(stores) >= 4, name
        loop = body[b:stores[3] + 1]
        assert sum(ln.startswith("v_mfma_f64_16x16x4") for ln in loop) == 4 * int(re.search(r"ILi(\d)E", name).group(1)), name
        assert not [ln for ln in loop if ln.startswith("scratch_")], (name, "a spill inside the tile loop")
        assert not [ln for ln in loop if ln.startswith("s_waitcnt") and "vmcnt" in ln], (name, "a vector-memory wait inside the tile loop")
        assert not [ln for ln in loop if ln.startswith("global_load_dword")], (name, "a global load inside the tile loop")
        assert [ln for ln in loop if ln.startswith("global_load_lds_dwordx4")], name          # the W tile three ahead
        assert sum(ln.startswith("ds_read_b64") for ln in loop) >= 4 * int(re.search(r"ILi(\d)E", name).group(1)), name   # hand-written fragment reads
        # the counted wait sits right in front of the barrier
        head = body[max(0, b - 40):b]
        waits = [ln for ln in head if ln.startswith("s_waitcnt") and "vmcnt" in ln]
        assert waits and all("vmcnt(0)" not in w for w in waits[-1:]), (name, waits)
