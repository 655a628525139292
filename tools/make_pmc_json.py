#!/usr/bin/env python3
"""Turn the PMC summary of tools/profile_round.sh (pmc_dense_summary.txt: medians per kernel of the separate
--pmc FETCH_SIZE / WRITE_SIZE / MFMA-busy / LDS passes) into profiles/<round>_pmc_dense_main.json, the record bench.py's
`roofline.traffic` is read from.  The record carries the sha256 of the kernel source it was measured on; bench.py reports
`traffic: null` as soon as that source changes.

    python3 tools/make_pmc_json.py gpurun_out/r3prof/pmc_dense_summary.txt profiles/r03_pmc_dense_main.json
    python3 tools/make_pmc_json.py gpurun_out/r4prof/pmc_dense_f32_summary.txt profiles/r04_pmc_dense_f32_main.json f32
"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = sys.argv[1], sys.argv[2]
F32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
KERNEL = "2, 4, true, 16>" if F32 else "true, false, true>"   # the FUSE = true instantiation: layer 2 + fused head
KNAME = "dense_f32_dma_kernel<192, 128" if F32 else "dense_f64_kernel<96, 128"
KSRC = "kernels_gemm_f32.hip" if F32 else "kernels_gemm.hip"
vals, launches, cur = {}, None, None
for line in open(src):
    if line.startswith("void si::") or line.startswith("si::"):
        cur = KERNEL in line and KNAME in line
    elif cur:
        m = re.match(r"\s+launches (\d+)\s+duration ms min ([\d.]+) med ([\d.]+)", line)
        if m:
            launches = int(m.group(1))
            vals.setdefault("duration_ms_med", []).append(float(m.group(3)))
        m = re.match(r"\s+(\w+)\s+med ([\d.e+]+)", line)
        if m:
            vals[m.group(1)] = float(m.group(2))
fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
_code = "\n".join(ln for ln in (re.sub(r"//.*", "", raw).rstrip() for raw in open(os.path.join(ROOT, "subspaceinference.jl_amd", "csrc", KSRC))) if ln)
sha = hashlib.sha256(_code.encode()).hexdigest()[:16]   # code only: `//` comments and blank lines dropped (bench.py _sha16)
# SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
busy = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
dur = sorted(vals["duration_ms_med"])[len(vals["duration_ms_med"]) // 2]
rec = {
    "kernel": ("si::dense_f32_dma_kernel<192,128,2,4,2,4,true,16> (fp32, layer 960x960 + fused 960->1 head), cfg2" if F32 else
               "si::dense_f64_kernel<96,128,2,4,4,true,false,true> (layer 960x960 + fused 960->1 tail), cfg2"),
    "source": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
              "tools/f32_after_f64.py f64 (the cfg2 construct, a 50-step fp64 chain and a 12-step fp32 chain: bench.py itself makes more dispatches "
              "than rocprofv3's counter collection survives); medians over %d launches; %s" % (launches, os.path.basename(src)),
    "FETCH_SIZE_KB": fetch_kb,
    "WRITE_SIZE_KB": write_kb,
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) "
                  "-> doubled; WRITE_SIZE taken as read; unit KB",
    "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    # W 960x960 + bias + head weights, the H panel 960 x B read once, the head-partial slots x B written (fp64 partials)
    "algorithmic_bytes_per_launch": (960 * 960 + 960 + 960) * 4 + 960 * 100000 * 4 + 10 * 100000 * 8 if F32 else 791380480,
    "kernel_code_sha16": {KSRC: sha},
    "mfma_busy_fraction": busy,
    "effective_clock_ghz": vals["GRBM_GUI_ACTIVE"] / 8.0 / (dur * 1e-3) / 1e9,
    "lds_bank_conflict_cycles": vals.get("SQ_LDS_BANK_CONFLICT"),
}
json.dump(rec, open(dst, "w"), indent=1)
print(json.dumps(rec, indent=1))
