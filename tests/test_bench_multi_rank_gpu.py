"""GPU: bench.py's multi-rank path end to end on the one-GPU box -- `--gpus 2` starts two ranks itself, both on
device 0 over gloo (the rehearsal knobs; RCCL needs one device per rank), each decodes its key-map slice, the
per-GOP checksums are all-gathered and rank 0 re-decodes a GOP of the other rank and compares."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def test_two_ranks_decode_disjoint_gops_with_matching_checksums():
    env = dict(os.environ, LEON_BENCH_BACKEND="gloo", LEON_BENCH_ONE_DEVICE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gops", "4", "--unique", "2", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline", "--no-second-recipe", "--e2e-window", "8", "--e2e-loop", "4"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    # the stream-bytes leg of N ranks: every rank a pipeline on its GOP shards (g = r mod N) of the 16-GOP 1080p stream, its
    # frames checked against rank 0's oracle pass over the whole stream
    e = d["end_to_end_sharded"]
    assert e["verified"] is True and e["frames_equal_rank0_reference"] is True, e
    assert len(e["per_rank"]) == 2 and all(x and x > 0 for x in e["per_rank"]) and abs(e["value"] - sum(e["per_rank"])) < 1e-6
    assert [r["shard"]["index"] for r in e["ranks"]] == [0, 1] and all(r["shard"]["gops_of_the_stream"] == 8 for r in e["ranks"])
    for r in e["ranks"]:
        v = r["verified"]
        assert v["ok"] and v["gpu_parser_one_pass"]["frames"] == 96 and v["host_parser_one_pass"]["frames"] == 96 and r["gpu_parser_checked"]["ok"]
    assert d["n_gpus"] == 2 and d["config"]["pictures_per_step"] == 2 * 4 * 12
    c = d["gop_checksums"]
    assert c["gops"] == 8 and c["distinct"] == 8 and c["cross_rank_ok"] is True
    assert 0.2 < d["efficiency"] < 1.2 and d["single_rank_same_run"]["value"] > 0
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "roofline"):
        assert k in d
    # the line says who ran: backend, the world size the process group saw, every rank's device
    assert d["backend"] == "gloo" and d["world_size_seen"] == 2
    assert [x["rank"] for x in d["devices"]] == [0, 1] and all(x["name"] and x["pci"] for x in d["devices"])
    assert d["distinct_devices"] == 1              # the rehearsal: both ranks on device 0 (LEON_BENCH_ONE_DEVICE)
    # the one-sided and the mixed B launches apart, each with the spread of its launch times
    pt = d["roofline"]["per_picture_type"]
    assert {"I", "P", "B", "B_leading", "B_mixed"} <= set(pt)
    for v in pt.values():
        assert v["launch_ms"]["min"] <= v["launch_ms"]["median"] <= v["launch_ms"]["max"] and v["launches"] > 0
    assert pt["B_leading"]["bytes_per_macroblock"] < pt["B_mixed"]["bytes_per_macroblock"]
    assert pt["B_leading"]["launches"] + pt["B_mixed"]["launches"] == pt["B"]["launches"]


def test_ranks_without_distinct_devices_are_refused():
    """two ranks on a one-GPU box are an error unless the rehearsal knob says so: rank 1 finds no device of its own"""
    env = dict(os.environ, LEON_BENCH_BACKEND="gloo")
    env.pop("LEON_BENCH_ONE_DEVICE", None)
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box with exactly one GPU")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gops", "2", "--unique", "1", "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline", "--no-second-recipe", "--no-end-to-end"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], "a result line although a rank had no device"
