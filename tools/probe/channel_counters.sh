#!/bin/bash
# Per-channel fabric requests of the reconstruction launches BY POSITION in the step: TCC_EA0_WRREQ and
# TCC_EA0_RDREQ per TCC instance (16 L2 channels x 8 XCDs), one --pmc pass, program directly after `--`.  The three P launches
# of a step (and the four B launches) are the same kernel on the same statistics of content but on different buffers; is the
# slower position's traffic spread less evenly over the channels?
#   bash tools/probe/channel_counters.sh <outdir> [tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/chan}; tag=${2:-r04}
mkdir -p $out
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_RDREQ -d $out/raw --output-format json -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-second-recipe > $out/log.txt 2>&1 || echo "pass failed"
python3 - $out $tag <<'PY'
import glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/raw/**/*_results.json", recursive=True)[0]
d = json.load(open(f))["rocprofiler-sdk-tool"][0]
names = {c["id"]["handle"]: c["name"] for c in d["counters"]}
order = {c["id"]["handle"]: [(i["dimensions"][1]["index"], i["dimensions"][0]["index"]) if i["dimensions"][0]["dimension_name"] == "DIMENSION_INSTANCE"
                             else (i["dimensions"][0]["index"], i["dimensions"][1]["index"]) for i in c["instances"]] for c in d["counters"]}     # (xcc, channel)
kern = {k["kernel_id"]: k["formatted_kernel_name"] for k in d["kernel_symbols"]}
disp = []
for r in d["callback_records"]["counter_collection"]:
    info = r["dispatch_data"]["dispatch_info"]
    name = kern.get(info["kernel_id"], "?")
    if "k_recon_display" not in name:
        continue
    per = collections.defaultdict(list)
    for x in r["records"]:
        per[x["counter_id"]["handle"]].append(x["value"])
    disp.append({"dispatch": info["dispatch_id"], "kernel": name.split("(")[0].replace("void leon::", ""), "grid": info["grid_size"]["x"],
                 "us": (r["dispatch_data"]["end_timestamp"] - r["dispatch_data"]["start_timestamp"]) / 1e3,
                 "v": {names[h]: vals for h, vals in per.items()}, "order": {names[h]: order[h] for h in per}})
disp.sort(key=lambda x: x["dispatch"])
per_step = 8
disp = disp[-per_step * 2:]                      # the last two steps (the first ones are warm-up)
res = {"note": "per reconstruction launch of the last two steps of `bench.py --steps 3 --warmup 1` under rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_RDREQ: requests per "
               "TCC instance (16 channels x 8 XCDs = 128 cells); imbalance = busiest cell / mean cell; by_channel = the 16 channels summed over the XCDs, "
               "by_xcd = the 8 XCDs summed over their channels", "launches": []}
seen = collections.Counter()
for k, x in enumerate(disp):
    t = {"k_recon_display<1": "I", "k_recon_display<2": "P", "k_recon_display<3": "B"}[x["kernel"][:17]]
    if k % per_step == 0:
        seen.clear()
    seen[t] += 1
    e = {"position": "%s%d" % (t, seen[t]), "step": k // per_step, "duration_us_under_counters": round(x["us"], 1)}
    for c, vals in x["v"].items():
        cells = collections.defaultdict(float)
        for (xcc, ch), v in zip(x["order"][c], vals):
            cells[(xcc, ch)] += v
        tot = sum(cells.values())
        if not tot:
            continue
        mean = tot / len(cells)
        by_ch = [sum(v for (xc, ch), v in cells.items() if ch == k2) for k2 in range(16)]
        by_x = [sum(v for (xc, ch), v in cells.items() if xc == k2) for k2 in range(8)]
        e[c] = {"total": tot, "cell_imbalance": max(cells.values()) / mean, "cell_min_over_mean": min(cells.values()) / mean,
                "channel_imbalance": max(by_ch) / (sum(by_ch) / 16), "xcd_imbalance": max(by_x) / (sum(by_x) / 8),
                "by_channel_share": [round(v / tot, 4) for v in by_ch], "by_xcd_share": [round(v / tot, 4) for v in by_x]}
    res["launches"].append(e)
json.dump(res, open(out + "/%s_channel_counters.json" % tag, "w"), indent=1)
for e in res["launches"]:
    print(e["position"], e["step"], e["duration_us_under_counters"], {c: (round(e[c]["cell_imbalance"], 3), round(e[c]["channel_imbalance"], 3), round(e[c]["xcd_imbalance"], 3)) for c in ("TCC_EA0_WRREQ", "TCC_EA0_RDREQ") if c in e})
PY
rm -rf $out/raw
