#!/bin/bash
# the FETCH_SIZE pass of tools/probe/profile_round.sh on its own (it has failed at start-up once or twice when it directly followed the
# trace pass), merged into an existing <tag>_pmc.json:  bash tools/probe/fetch_pass.sh <outdir> <tag> <pmc.json to complete>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; tag=$2; have=$3; mkdir -p $out
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $out/pf --output-format csv -- python3 bench.py --no-cpu-baseline --no-second-recipe --no-end-to-end --no-placement --steps 2 --warmup 1 > $out/pf.log 2>&1 || echo "fetch failed"
python3 tools/summarize_profile.py ${tag}_fetch --out $out/summary --pmc $out/pf
python3 - $out/summary/${tag}_fetch_pmc.json $have <<'PY'
import json, sys
f, h = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
for k, v in f["kernels"].items():
    h["kernels"].setdefault(k, {}).update(v)
json.dump(h, open(sys.argv[2], "w"), indent=1)
print({k: v.get("hbm_read_bytes_corrected") for k, v in h["kernels"].items() if "display" in k})
PY
rm -rf $out/pf
