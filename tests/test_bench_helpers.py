"""CPU: the parts of bench.py that do not need a GPU -- the committed PMC summary it quotes as
`roofline.traffic`, and the stream index / shard arithmetic of the multi-GPU path."""
import json
import os

from helpers import ROOT

import bench


def test_pmc_traffic_comes_from_the_committed_profile():
    """Fabric bytes per launch of the committed PMC passes against the bytes SURVEY.md 8d counts when every picture
    is charged its full set of reference planes -- per GOP and step 1 I, 3 P, 6 bidirectional and 2 backward-only B
    pictures.  That is what moves at 128-byte line granularity: with vectors spread over +-15 samples the macroblocks
    that do use a reference touch every line of it, whatever their neighbours skip (profiles/r02_fetch_calibration.json:
    a line is fetched whole even for 4 of its bytes).  Fused launches since round 3: the two B pictures between two
    anchors run with their workgroups interleaved (leon_kernels.h pic_of_wg), the second one finds the reference lines
    in L2 -- a B PAIR is charged its references once."""
    mbs = (bench.CW // 16) * (bench.CH // 16)
    for fused in (True, False):
        traffic, src = bench.pmc_traffic(128, fused)
        assert traffic is not None and "profiles/" in src
        path = os.path.join(ROOT, src.split(" ")[0])
        assert os.path.exists(path), path
        k = json.load(open(path))["kernels"]
        names = ("void leon::k_recon_display<%d, false, false>", "void leon::k_recon_display<%d, false>") if fused else ("void leon::k_recon<%d, false>",)
        assert all(any(n % t in k for n in names) for t in (1, 2, 3))
        # fused display conversion: + RGBA of the 1080 displayed rows, - the planes of the B pictures
        rgba = 1024.0 * bench.FH / bench.CH if fused else 0.0
        b_planes = 384.0 if fused else 0.0
        if fused:      # 8 B pictures without their references + three pairs with two reference planes + the leading pair with one
            planes = 128 * mbs * ((1154 + rgba) + 3 * (1542 + rgba) + 8 * (1162 + rgba - b_planes) + (3 * 768 + 384)) / 8.0
        else:
            planes = 128 * mbs * ((1154 + rgba) + 3 * (1542 + rgba) + 6 * (1930 + rgba - b_planes) + 2 * (1542 + rgba - b_planes)) / 8.0
        assert 0.97 < traffic / planes < 1.06, (fused, traffic, planes)


def test_kernel_stats_of_the_same_round_are_committed():
    tag = bench.pmc_traffic(1)[1].split("/")[1].split("_")[0]          # e.g. r01i
    for suffix in ("_kernel_stats.csv", "_pmc.json", "_bench_line_under_rocprofv3.json"):
        assert os.path.exists(os.path.join(ROOT, "profiles", tag + suffix)), tag + suffix
    line = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench_line_under_rocprofv3.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
