#!/usr/bin/env python3
"""A/B of kernel variants on ONE box: runs bench.py once per variant built by tools/ab_build.sh (or the in-tree
library for the name `tree`), optionally with environment settings (name@VAR=VALUE), and prints one line each.
    python tools/ab_run.py base lut base@LEON_DEBUG_LDS_PAD=5120 [-- extra bench.py flags]"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
for spec in args:
    name, *envs = spec.split("@")
    env = dict(os.environ)
    if name != "tree":
        env["LEON_DEBUG_LIB"] = os.path.join(root, "build", "ab", name, "libleon_hip.so")
    for e in envs:
        k, v = e.split("=", 1)
        env[k] = v
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-second-recipe", "--steps", "20"] + extra,
                         env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        pt = d["roofline"]["per_picture_type"]
        print("%-36s %.3f ms/step  %.4g MB/s  copy %.0f  " % (spec, d["ms_per_step"], d["value"], d["roofline"]["measured_copy_gbps"]) +
              "  ".join("%s %.3f ms (%.3f)" % (k, v["avg_launch_ms"], v["frac_of_measured_copy"]) for k, v in pt.items()) +
              "  parity %s" % (d.get("parity_vs_oracle", {}).get("differing_rgba_bytes")), flush=True)
    except Exception as e:
        print(spec, "FAILED", e, out.stderr[-800:], flush=True)
