"""CPU: the register budgets the reconstruction kernels were tuned to still hold.
Occupancy on CDNA4 steps at 64 / 72 / 80 VGPRs (8 / 7 / 6 waves per SIMD); a harmless-looking refactor of
recon_task has cost the B path a wave more than once (a lambda around the reference fetch: 70 -> 73).
Compiles the device code only (hipcc cross-compiles without a GPU) and reads -Rpass-analysis."""
import os
import re
import shutil
import subprocess

import pytest

from helpers import ROOT

HIPCC = "/opt/rocm/bin/hipcc"

# kernel -> most VGPRs it may use (and the waves per SIMD that buys), no scratch
BUDGET = {
    "k_reconILi1ELb0E": 64, "k_reconILi2ELb0E": 64, "k_reconILi3ELb0E": 72,
    "k_reconILi1ELb1E": 64, "k_reconILi2ELb1E": 64, "k_reconILi3ELb1E": 72,
    # the B display kernels are held to 7 waves; the macroblock maps carried from the chroma part to the luma parts are one
    # register (flags) -- the vectors wait in LDS: carried in registers they went to scratch (12 bytes per lane, +6 % traffic)
    "k_recon_displayILi1ELb0ELb0E": 64, "k_recon_displayILi2ELb0ELb0E": 64, "k_recon_displayILi3ELb0ELb0E": 72,
    "k_recon_displayILi1ELb1ELb0E": 64, "k_recon_displayILi2ELb1ELb0E": 64, "k_recon_displayILi3ELb1ELb0E": 72,
    # yuva (A part + Y part per side): one occupancy step below the three-component kernels
    "k_recon_displayILi1ELb0ELb1E": 64, "k_recon_displayILi2ELb0ELb1E": 72, "k_recon_displayILi3ELb0ELb1E": 88,
    "k_recon_displayILi1ELb1ELb1E": 64, "k_recon_displayILi2ELb1ELb1E": 72, "k_recon_displayILi3ELb1ELb1E": 88,
    # the GPU parser: LDS allows 4 waves per SIMD = 128 registers; NO scratch -- a DC predictor (or anything else of the
    # lane's context) read through a computed address puts the whole context there and costs a quarter of the speed
    "k_vlc_parse": 128, "k_vlc_blocks": 64, "k_vlc_index": 64,
}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_register_budgets_and_no_scratch(tmp_path):
    src = os.path.join(ROOT, "mpeg1video-decoder-webgl_amd", "csrc", "leon_hip.cpp")
    out = subprocess.run([HIPCC, "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
                          "--cuda-device-only", "-o", str(tmp_path / "k.s"), src, "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    cur, seen = None, {}
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
        m = re.search(r"\bVGPRs: (\d+)", line)
        if m and cur:
            seen.setdefault(cur, {})["vgpr"] = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and cur:
            seen.setdefault(cur, {})["scratch"] = int(m.group(1))
    for key, limit in BUDGET.items():
        hit = [(n, v) for n, v in seen.items() if key in n]
        assert hit, "kernel %s not found in the resource report" % key
        name, v = hit[0]
        limit, scratch = limit if isinstance(limit, tuple) else (limit, 0)
        assert v["scratch"] <= scratch, "%s spills %d bytes per lane" % (name, v["scratch"])
        assert v["vgpr"] <= limit, "%s uses %d VGPRs, budget %d" % (name, v["vgpr"], limit)
