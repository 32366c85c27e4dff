#!/bin/bash
# occupancy sweep of k_recon via extra dynamic LDS per workgroup (debug knob LEON_DEBUG_LDS_PAD)
for pad in 0 7000 13000 27000 40000 67000; do
  LEON_DEBUG_LDS_PAD=$pad python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-rgba 2>/dev/null > /tmp/occ.json
  python - "$pad" <<'PY'
import json,sys
d=json.load(open('/tmp/occ.json'))
lds=4*3328+int(sys.argv[1]); wg=min(8, 163840//lds)
print("pad", sys.argv[1], "-> WGs/CU", wg, "waves/SIMD<=", wg, " MB/s(M)", round(d["value"]/1e6), " recon GB/s", round(d["roofline"]["achieved"]), " avg launch ms", round(d["roofline"]["avg_launch_ms"],4))
PY
done
