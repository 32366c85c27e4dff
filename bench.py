#!/usr/bin/env python3
"""bench.py -- 1080p macroblock reconstruction throughput on MI355X.

Workload (BASELINE.json config 4): G independent closed GOPs of a
synthetic 1920x1080 IBBP stream (12 pictures each: I B B P B B P B B P B B in coded
order), boundary tensors resident in HBM in the reference's own layout (dense int16
coefficient planes + per-macroblock maps, decoders/jsv.js:1204-1298).  One "step"
decodes all G GOPs: 5 dependency levels, each one launch of the fused dequant+IDCT+MC kernel
per picture type present (8 launches: every picture of a level and type, across all GOPs),
then one YCbCr->RGBA launch over all 12*G pictures.  value = macroblocks/s over the
whole job (all ranks); N>1 = frame-parallel GOP shards, one process per GPU, the
stream index broadcast once over RCCL before the timed region (weak scaling).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "mpeg1video-decoder-webgl_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

CW, CH, FW, FH = 1920, 1088, 1920, 1080
GOP_LEN = 12
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def build_workload(L, S, dec, torch, gops, seed, sparse=False):
    """One unique synthetic GOP on the host, replicated into `gops` independent device-resident
    GOPs (own coefficient planes, own slots).  Returns (levels, slot ids in display order)."""
    rng = np.random.default_rng(seed)
    gop = S.gop_ibbp(GOP_LEN)
    levels = S.dependency_levels(gop)
    host = {}
    for ptype, disp, f, b in gop:
        force = 2 if (ptype == S.PIC_B and f is None) else None
        host[disp] = S.make_picture(rng, CW, CH, ptype, force_dir=force)
        if sparse:   # the same coefficients as per-group entry lists (include/leon_vlc.h)
            import leon_vlc_ctypes as V
            t = host[disp]
            t["grp_off"], t["entries"] = V.sparsify(t["coef_y"], t["coef_cb"], t["coef_cr"], CW, CH)
    dense_keys = ("coef_y", "coef_cb", "coef_cr")
    as_dev = lambda v: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).cuda()
    dev_unique = {d: {k: as_dev(v) for k, v in t.items()
                      if isinstance(v, np.ndarray) and not (sparse and k in dense_keys)} for d, t in host.items()}
    keep = [dev_unique]
    batches = []
    for lv in levels:
        pics = []
        for g in range(gops):
            for ptype, disp, f, b in lv:
                src = dev_unique[disp]
                # every GOP owns its coefficient planes (the working set must be real HBM
                # traffic, far beyond the 256 MiB Infinity Cache); the small maps are cloned too
                d = {k: (v.clone() if g else v) for k, v in src.items()}
                keep.append(d)
                ptr = lambda k: d[k].data_ptr() if k in d else None
                base = g * GOP_LEN
                fwd = f if f is not None else b
                slots = dict(ref_fwd_slot=-1 if fwd is None else base + fwd, ref_bwd_slot=-1 if b is None else base + b)
                if sparse:
                    pics.append(L.make_sparse_picture(
                        ptype, base + disp, ptr("grp_off"), ptr("entries"), len(host[disp]["entries"]), ptr("qscale"),
                        ptr("intra"), ptr("repadd"), ptr("mv_fwd"), ptr("mv_bwd"), ptr("mb_dir"), device=True, **slots))
                else:
                    pics.append(L.make_picture(
                        ptype, base + disp, ptr("coef_y"), ptr("coef_cb"), ptr("coef_cr"), ptr("qscale"),
                        ptr("intra"), ptr("repadd"), ptr("mv_fwd"), ptr("mv_bwd"), ptr("mb_dir"), device=True, **slots))
        batches.append(dec.batch_create_sparse(pics) if sparse else dec.batch_create(pics))
    torch.cuda.synchronize()
    return batches, keep, host, gop


def pmc_traffic(gops):
    """HBM bytes per k_recon launch from the committed rocprofv3 PMC passes of this same workload
    (profiles/r01j_pmc.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950, WRITE_SIZE exact).
    Counters cannot be read from inside this process, so this is the profiled figure of the
    identical launches, scaled by the GOP count; None when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01j_pmc.json")
    if not os.path.exists(path):
        return None, None
    k = json.load(open(path))["kernels"]
    per_type = {}
    wg_threads = ((CW // 64 + 0) * (CH // 16) + ((CW // 16 + 7) // 8) * (CH // 16) + 3) // 4 * 256   # threads per picture
    for t in (1, 2, 3):
        e = k.get("void leon::k_recon<%d, false>" % t)
        if not e or "hbm_read_bytes_corrected" not in e:
            return None, None
        # pictures in the profiled launches, from their grid size (one workgroup = 4 tasks of one picture)
        n_pics = e["FETCH_SIZE"]["grid_size"] / float(wg_threads)
        per_type[t] = (e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]) / n_pics
    # one step = 1 I launch, 3 P launches, 4 B launches (2 B pictures per GOP each)
    step_bytes = gops * (per_type[1] + 3 * per_type[2] + 8 * per_type[3])
    return step_bytes / 8.0, "profiles/r01j_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 FETCH x2 correction)"


def _cpu_gops(O, host, gop, budget_s, counter, slot):
    """decode + RGBA of whole GOPs with the oracle until the budget is used; pictures done -> counter[slot]"""
    t0 = time.perf_counter()
    done = 0
    outs = {}
    while time.perf_counter() - t0 < budget_s:
        for ptype, disp, f, b in gop:
            t = host[disp]
            fwd = f if f is not None else b
            outs[disp] = O.decode_picture(ptype, CW, CH, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"],
                                          t["intra"], repadd=t.get("repadd"), mb_dir=t.get("mb_dir"),
                                          mv_fwd=t.get("mv_fwd"), mv_bwd=t.get("mv_bwd"),
                                          ref_fwd=None if fwd is None else outs[fwd],
                                          ref_bwd=None if b is None else outs[b])
            y, cb, cr = O.split_planes(outs[disp], CW, CH)
            O.ycbcr_to_rgba(y, cb, cr, CW, FW, FH, "cpu")
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
    counter[slot] = (done, time.perf_counter() - t0)


def cpu_baseline_all_cores(host, gop, budget_s=8.0):
    """The same oracle on every host core this process may use: one independent GOP stream per
    thread (closed GOPs shard on a CPU exactly as they do across GPUs; ctypes releases the GIL).
    Reported next to the single-core figure, which is the like-for-like one (the reference is
    single-threaded)."""
    import threading
    from oracle import oracle_py as O
    O.lib()
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = max(1, min(n, 64))
    res = [None] * n
    th = [threading.Thread(target=_cpu_gops, args=(O, host, gop, budget_s, res, i)) for i in range(n)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    mbs_per_pic = (CW // 16) * (CH // 16)
    return {"value": done * mbs_per_pic / dt, "unit": "macroblocks/s", "cores": n, "kind": "port",
            "sample": "%d threads x whole 1080p IBBP GOPs (decode + RGBA), %d pictures in %.1f s" % (n, done, dt),
            "fps": done / dt}


def cpu_baseline_js(host, gop, budget_s=6.0):
    """SURVEY.md 8d's CPU baseline in the reference's own language: the plain-JavaScript oracle
    (oracle/leon_oracle.js, bit-identical to the C oracle) on the same GOP, timed by process.hrtime in
    Node on one thread and on worker_threads = hardware threads.  None when node is not installed."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("node") is None:
        return None
    from oracle import oracle_py as O
    d = tempfile.mkdtemp(prefix="leon_js_gop_")
    try:
        O.dump_gop(d, CW, CH, FW, FH, gop, host)
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        n = max(1, min(n, 64))
        out = subprocess.run(["node", os.path.join(ROOT, "oracle", "js_baseline.js"), "time", d, str(budget_s), str(n)],
                             capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            return {"error": out.stderr[-300:]}
        r = json.loads(out.stdout)
        return {"kind": "port", "language": "JavaScript (%s)" % r["node"], "unit": "macroblocks/s",
                "value": r["one_thread"]["macroblocks_per_s"], "cores": 1,
                "sample": "%d 1080p pictures of the same IBBP GOP (decode + RGBA), oracle/leon_oracle.js, %.1f s"
                          % (r["one_thread"]["pictures"], r["one_thread"]["seconds"]),
                "workers": {"value": r["workers"]["macroblocks_per_s"], "cores": r["workers"]["threads"],
                            "sample": "%d pictures in %.1f s" % (r["workers"]["pictures"], r["workers"]["seconds"])}}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(S, host, gop, budget_s=12.0):
    """The oracle (a scalar C port of the reference's path) on ONE host core, on a bounded
    sample of the same workload: whole 1080p GOPs, decode + RGBA, until ~budget_s."""
    from oracle import oracle_py as O
    O.lib()
    mbs_per_pic = (CW // 16) * (CH // 16)
    t0 = time.perf_counter()
    done = 0
    outs = {}
    while True:
        for ptype, disp, f, b in gop:
            t = host[disp]
            fwd = f if f is not None else b
            outs[disp] = O.decode_picture(ptype, CW, CH, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"],
                                          t["intra"], repadd=t.get("repadd"), mb_dir=t.get("mb_dir"),
                                          mv_fwd=t.get("mv_fwd"), mv_bwd=t.get("mv_bwd"),
                                          ref_fwd=None if fwd is None else outs[fwd],
                                          ref_bwd=None if b is None else outs[b])
            y, cb, cr = O.split_planes(outs[disp], CW, CH)
            O.ycbcr_to_rgba(y, cb, cr, CW, FW, FH, "cpu")
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done * mbs_per_pic / dt, "unit": "macroblocks/s", "cores": 1, "kind": "port",
            "sample": "%d 1080p pictures of the same IBBP GOP (decode + RGBA), oracle/leon_oracle.c, %.1f s" % (done, dt),
            "fps": done / dt}, outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gops", type=int, default=128,
                    help="independent GOPs per GPU per step (every launch holds that many pictures of a type per "
                         "GOP position; 128 GOPs = 32 GB of tensors, slots and RGBA output on a 288 GB part)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--boundary", choices=("dense", "sparse"), default="dense",
                    help="dense: the int16 planes the reference uploads (the BASELINE metric); sparse: the same "
                         "pictures as per-group entry lists, the output format of the native front end")
    ap.add_argument("--no-rgba", action="store_true", help="leave the RGBA conversion out of the step (diagnostic)")
    ap.add_argument("--rgba-lag", type=int, default=0,
                    help="issue the RGBA conversion of a dependency level this many levels late (0 = right after it)")
    ap.add_argument("--overlap", action="store_true",
                    help="run the RGBA conversions on the decoder's second stream (measured: no gain, both kernels want the VALU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the reconstruction path has no CPU fallback")
    # rehearsal knobs (a 1-GPU box cannot host two RCCL ranks): LEON_BENCH_BACKEND=gloo runs the
    # same multi-rank code path over gloo, LEON_BENCH_ONE_DEVICE=1 puts every rank on cuda:0
    backend = os.environ.get("LEON_BENCH_BACKEND", "nccl")
    if os.environ.get("LEON_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    import leon_ctypes as L
    import synth as S
    import shards

    # ---- stream index: rank 0 owns it, everyone else learns it over RCCL (xGMI) ----
    total_gops = args.gops * world
    index = shards.make_index(CW, CH, FW, FH, rate_idx=3, n_gops=total_gops, gop_len=GOP_LEN) if rank == 0 else None
    index = shards.broadcast_index(index, dist if world > 1 else None, torch, src=0)
    my_gops = shards.shard_gops(index, rank, world)
    assert len(my_gops) == args.gops

    stream = torch.cuda.Stream()
    n_slots = args.gops * GOP_LEN
    dec = L.Decoder(index["coded_w"], index["coded_h"], index["frame_w"], index["frame_h"], n_slots=n_slots,
                    device_id=local_rank, stream=stream.cuda_stream)
    sparse = args.boundary == "sparse"
    batches, keep, host, gop = build_workload(L, S, dec, torch, args.gops, seed=0x4C454F4E, sparse=sparse)
    # display conversion per dependency level (with --overlap on the decoder's second stream)
    levels = S.dependency_levels(gop)
    level_slots = [np.array([g * GOP_LEN + e[1] for g in range(args.gops) for e in lv], dtype=np.int32) for lv in levels]
    rgba_lv = [torch.empty((len(s), FH, FW, 4), dtype=torch.uint8, device="cuda") for s in level_slots] if not args.no_rgba else None
    rgba = rgba_lv[0] if rgba_lv else None            # level 0 = the I pictures, GOP order
    if args.overlap:
        dec.set_overlap_convert(True)

    defer = 0 if args.rgba_lag is None else args.rgba_lag

    def step():
        # --rgba-lag N issues the display conversion of level k N levels late, so that the next
        # level's reconstruction reads its references before 1.2 GB of RGBA output per level
        # passes through the caches.  Measured: no difference (the anchors are re-fetched from
        # HBM either way at 48 GOPs per launch); default 0 = convert right after the level.
        for k, b in enumerate(batches):
            dec.batch_run(b)
            if rgba_lv is not None and k >= defer:
                dec.convert_rgba_batch(level_slots[k - defer], rgba_lv[k - defer].data_ptr())
        if rgba_lv is not None:
            for k in range(len(batches) - defer, len(batches)):
                dec.convert_rgba_batch(level_slots[k], rgba_lv[k].data_ptr())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # clocks: ~30 ms of streaming copies before the W warm-up steps, so that a short run (small W) is
    # not timed on an idle-clocked device; not part of the step, not timed
    dec.measure_copy_bandwidth(1 << 30, 40)
    dec.timing_enable(True)              # on during the warm-up too: the event pool is filled before the timed region
    for _ in range(args.warmup):
        step()
    fence()
    dec.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    dec.timing_enable(False)
    recon = dec.timing_get(0)
    conv = dec.timing_get(1)
    per_type = {n: dec.timing_get(k) for n, k in (("I", 2), ("P", 3), ("B", 4))}

    if world > 1:
        cdev = "cuda" if backend == "nccl" else "cpu"
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # per-rank output checksum, gathered for the report (frame-parallel shards are identical work)
        crc = torch.tensor([int(rgba[0].to(torch.int64).sum().item()) if rgba is not None else 0],
                           dtype=torch.int64, device=cdev)
        crcs = [torch.zeros_like(crc) for _ in range(world)]
        dist.all_gather(crcs, crc)
        crcs = [int(c.item()) for c in crcs]
    else:
        crcs = None

    mbs_per_pic = (CW // 16) * (CH // 16)
    pics_per_step = args.gops * GOP_LEN * world
    value = pics_per_step * mbs_per_pic * args.steps / dt

    if rank == 0:
        copy_gbps = dec.measure_copy_bandwidth(1 << 31, 5)
        # The library prices every B picture at 1930 B/MB.  The two leading B pictures of
        # a closed GOP use the backward reference only, and the kernel does not fetch the other one:
        # their forward reference (384 B/MB) and forward vectors (4 B/MB) are taken out again, so that
        # no byte that is not moved counts as achieved bandwidth.
        one_sided_b = sum(1 for ptype, disp, f, b in gop if ptype == S.PIC_B and f is None)
        unread = 388.0 * one_sided_b * args.gops * mbs_per_pic * args.steps
        recon["algorithmic_bytes"] -= unread
        per_type["B"]["algorithmic_bytes"] -= unread
        achieved = recon["algorithmic_bytes"] / (recon["total_ms"] * 1e-3) / 1e9 if recon["total_ms"] else 0.0
        traffic, traffic_src = pmc_traffic(args.gops) if not sparse else (None, None)   # PMC passes exist for the dense boundary
        out = {
            "metric": "1080p macroblocks/s", "value": value, "unit": "macroblocks/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32/u8 (fp64 RGBA)",
            "data": "synthetic",
            "config": {"workload": "1920x1080 IBBP closed GOPs (12 pictures), %d GOPs/GPU/step, %s boundary "
                                   "tensors resident in HBM, decode + RGBA of every picture"
                                   % (args.gops, "sparse group-list" if sparse else "dense-int16"),
                       "boundary": args.boundary,
                       "coded": [CW, CH], "gops_per_gpu": args.gops, "pictures_per_step": pics_per_step,
                       "parallelism": "gop-shards x%d" % world, "rgba_in_step": rgba is not None,
                       "rgba_overlapped_on_second_stream": (rgba is not None and args.overlap)},
            "fps": pics_per_step * args.steps / dt,
            "roofline": {"bound": "hbm", "kernel": "leon::k_recon<I|P|B> (fused dequant+IDCT+MC; one launch per picture type and dependency level)",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_src, "launches": recon["launches"],
                         "avg_launch_ms": recon["total_ms"] / max(1, recon["launches"]),
                         "algorithmic_bytes_per_launch": recon["algorithmic_bytes"] / max(1, recon["launches"]),
                         "measured_copy_gbps": copy_gbps, "frac_of_measured_copy": achieved / copy_gbps if copy_gbps else None,
                         # the motion-compensated launches reported separately
                         "per_picture_type": {n: {"achieved": (t["algorithmic_bytes"] / (t["total_ms"] * 1e-3) / 1e9) if t["total_ms"] else None,
                                                  "launches": t["launches"],
                                                  "avg_launch_ms": t["total_ms"] / max(1, t["launches"])}
                                              for n, t in per_type.items()}},
            "rgba_kernel": {"achieved_gbps": conv["algorithmic_bytes"] / (conv["total_ms"] * 1e-3) / 1e9 if conv["total_ms"] else None,
                            "launches": conv["launches"]},
            "stream_index_bytes": int(index["blob_bytes"]),
            "rank_checksums": crcs,
        }
        if not args.no_cpu_baseline and world == 1:
            cb, outs = cpu_baseline(S, host, gop)
            out["cpu_baseline"] = cb
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(host, gop)
            out["cpu_baseline_js"] = cpu_baseline_js(host, gop)
            # the bench doubles as a parity check of the timed workload: GOP 0 against the oracle
            bad = 0
            for disp in outs:
                y, cbp, crp = dec.read_planes(disp)
                got = np.concatenate([y.ravel(), cbp.ravel(), crp.ravel()])
                bad += int((got != outs[disp]).sum())
            out["parity_vs_oracle"] = {"pictures_checked": len(outs), "differing_samples": bad}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
