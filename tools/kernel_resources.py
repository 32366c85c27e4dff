#!/usr/bin/env python3
"""VGPRs / scratch / LDS of every reconstruction kernel, from hipcc -Rpass-analysis=kernel-resource-usage
(device code only; no GPU needed).  Extra arguments go to hipcc:  python tools/kernel_resources.py -DLEON_X=1"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(extra=()):
    src = os.path.join(ROOT, "mpeg1video-decoder-webgl_amd", "csrc", "leon_hip.cpp")
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
                          "--cuda-device-only", "-o", "/dev/null", src, "-Rpass-analysis=kernel-resource-usage"] + list(extra),
                         capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-2000:])
    cur, seen = None, {}
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("sgpr", r"\bSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                seen.setdefault(cur, {})[key] = int(m.group(1))
    return seen


if __name__ == "__main__":
    for name, v in report(sys.argv[1:]).items():
        m = re.search(r"(k_\w+?)I(.*?)EEv", name)
        short = (m.group(1) + "<" + m.group(2).replace("Li", "").replace("Lb", "").replace("E", ",").rstrip(",") + ">") if m else name
        print("%-34s vgpr %3d  scratch %3d  static LDS %5d  occupancy(regs) %d" % (short, v.get("vgpr", -1), v.get("scratch", -1), v.get("lds", -1), v.get("occupancy", -1)))
