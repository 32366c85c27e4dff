#!/usr/bin/env python3
"""A/B of library variants (tools/ab_build.sh) on the end-to-end pipeline bench, one box:
    python tools/ab_pipe.py ring noring [-- extra tools/pipeline_bench.py flags]"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = ["--loop", "7680", "--threads", "16", "--window", "128", "--gpu-parser"]
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
for spec in args:
    name, *envs = spec.split("@")                   # name@VAR=VALUE: an environment setting for that run
    env = dict(os.environ)
    if name != "tree":
        env["LEON_DEBUG_LIB"] = os.path.join(root, "build", "ab", name, "libleon_hip.so")
    for e in envs:
        k, v = e.split("=", 1)
        env[k] = v
    name = spec
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "pipeline_bench.py")] + extra, env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print("%-34s %7.0f pictures/s  %.2f s  scan %5.0f /thread/s  window %d  device %.1f GB" % (name, d["value"], d["seconds"], d["parser_pictures_per_s_per_thread"], d["gops_per_window"],
                                                                                                    d.get("device_gb_held_by_the_pipeline") or 0.0), flush=True)
    except Exception as e:
        print(name, "FAILED", e, out.stderr[-500:], flush=True)
