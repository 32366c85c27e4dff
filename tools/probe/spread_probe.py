#!/usr/bin/env python3
"""Where does the spread of the write-bound launches come from?  (the I launch of bench.py took
0.407-0.530 ms on boxes -- and, as round 3 saw, on RUNS of one box -- with the same copy bandwidth.)

One process, one I-picture launch (k_recon_display<I>, 128 pictures of 1080p: reads 1.2 GB of coefficients, writes 0.4 GB
of planes and 1.06 GB of RGBA), timed by the library's HIP events, 12 launches per arrangement:
  * the same buffers again and again             -> is a launch reproducible at all?
  * the RGBA frames freed and allocated again    -> does it follow the allocation (virtual/physical placement)?
  * frame stride padded to 2 MiB, to 8 MiB + 4 KiB, base shifted by 64 KiB / 1 MiB
  * the coefficient planes allocated again
Prints one JSON object.   python tools/probe/spread_probe.py [--pictures 128]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "mpeg1video-decoder-webgl_amd"), os.path.join(ROOT, "tools")]
CW, CH, FW, FH = 1920, 1088, 1920, 1080


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pictures", type=int, default=128)
    ap.add_argument("--launches", type=int, default=12)
    ap.add_argument("--realloc-only", type=int, default=0, help="only N rounds of: free everything, allocate again, time the launch")
    ap.add_argument("--contig", type=int, default=0, help="with --hold: this many more RGBA buffers from hipExtMallocWithFlags(hipDeviceMallocContiguous)")
    ap.add_argument("--lib-alloc", action="store_true", help="RGBA frames and coefficient planes from leon_device_malloc (contiguous unless LEON_CONTIGUOUS=0) instead of torch")
    ap.add_argument("--junk-gb", type=float, default=0.0, help="allocate (and keep) this much device memory before anything else")
    ap.add_argument("--hold", type=int, default=0, help="N RGBA buffers (and N sets of coefficient planes) alive at once: every combination of the first three timed")
    a = ap.parse_args()
    import torch
    import leon_ctypes as L
    import synth as S
    n = a.pictures
    rng = np.random.default_rng(0x4C454F4E)
    t = S.make_picture(rng, CW, CH, S.PIC_I)
    junk = torch.empty(int(a.junk_gb * (1 << 30)), dtype=torch.uint8, device="cuda") if a.junk_gb > 0 else None
    stream = torch.cuda.Stream()
    dec = L.Decoder(CW, CH, FW, FH, n_slots=n, device_id=0, stream=stream.cuda_stream)
    dec.measure_copy_bandwidth(1 << 30, 40)
    copy = dec.measure_copy_bandwidth(1 << 31, 5)
    keys = ("coef_y", "coef_cb", "coef_cr", "qscale", "intra")

    held = []

    def upload():
        if not a.lib_alloc:
            return [{k: torch.from_numpy(np.ascontiguousarray(t[k])).cuda() for k in keys} for _ in range(n)]
        per = sum((t[k].nbytes + 255) // 256 * 256 for k in keys)
        big = L.DeviceBuffer(n * per)
        held.append(big)
        res = []
        for i in range(n):
            o, d = i * per, {}
            for k in keys:
                src = np.ascontiguousarray(t[k])
                d[k] = big.as_tensor({np.dtype("int16"): torch.int16, np.dtype("uint8"): torch.uint8}[src.dtype], src.shape, o)
                d[k].copy_(torch.from_numpy(src))
                o += (src.nbytes + 255) // 256 * 256
            res.append(d)
        return res

    frame_bytes = FW * FH * 4

    def frames(stride, shift=0):
        if a.lib_alloc:
            buf = L.DeviceBuffer(n * stride + shift + 4096)
            base = (buf.ptr + 4095) // 4096 * 4096 + shift
            return buf, [base + i * stride for i in range(n)]
        buf = torch.empty(n * stride + shift + 4096, dtype=torch.uint8, device="cuda")
        base = (buf.data_ptr() + 4095) // 4096 * 4096 + shift
        return buf, [base + i * stride for i in range(n)]

    def run(tensors, ptrs):
        pics = [L.make_picture(S.PIC_I, i, d["coef_y"].data_ptr(), d["coef_cb"].data_ptr(), d["coef_cr"].data_ptr(), d["qscale"].data_ptr(),
                               d["intra"].data_ptr(), device=True, rgba_out=ptrs[i]) for i, d in enumerate(tensors)]
        b = dec.batch_create(pics)
        for _ in range(3):
            dec.batch_run(b)
        torch.cuda.synchronize()
        dec.timing_enable(True)
        dec.timing_reset()
        for _ in range(a.launches):
            dec.batch_run(b)
        ms = sorted(l["ms"] for l in dec.timing_launches())
        dec.timing_enable(False)
        dec.batch_destroy(b)
        return {"min": round(ms[0], 4), "median": round(ms[len(ms) // 2], 4), "max": round(ms[-1], 4)}

    out = {"pictures": n, "copy_gbps": copy, "runs": [], "lib": os.environ.get("LEON_DEBUG_LIB", "tree"), "junk_gb": a.junk_gb, "lib_alloc": a.lib_alloc, "LEON_CONTIGUOUS": os.environ.get("LEON_CONTIGUOUS", "default")}
    if a.realloc_only:
        ms = []
        for rep in range(a.realloc_only):
            tensors = upload()
            buf, ptrs = frames(frame_bytes)
            ms.append(run(tensors, ptrs)["median"])
            del tensors, buf
            torch.cuda.empty_cache()
        out["medians_ms"] = ms
        out["min"], out["mean"], out["max"] = min(ms), sum(ms) / len(ms), max(ms)
        print(json.dumps(out))
        return
    if a.hold:
        # placements side by side: is a slow one slow whatever it is paired with?
        bufs = [frames(frame_bytes) for _ in range(a.hold)]
        sets = [upload() for _ in range(min(a.hold, 3))]
        out["rgba_x_coef_median_ms"] = [[run(t_, ptrs_)["median"] for t_ in sets] for _, ptrs_ in bufs]
        out["rgba_base"] = [hex(ptrs_[0]) for _, ptrs_ in bufs]
        if a.contig:
            import ctypes as C
            hip = C.CDLL("libamdhip64.so")
            hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
            out["contiguous_median_ms"], out["contiguous_rc"] = [], []
            for _ in range(a.contig):
                ptr = C.c_void_p()
                rc = hip.hipExtMallocWithFlags(C.byref(ptr), n * frame_bytes + 4096, 0x4)      # hipDeviceMallocContiguous
                out["contiguous_rc"].append(rc)
                if rc == 0:
                    base = (ptr.value + 4095) // 4096 * 4096
                    out["contiguous_median_ms"].append([run(t_, [base + i * frame_bytes for i in range(n)])["median"] for t_ in sets[:2]])
        print(json.dumps(out))
        return
    tensors = upload()
    buf, ptrs = frames(frame_bytes)
    for rep in range(3):
        out["runs"].append({"what": "same buffers, run %d" % rep, "rgba_base_mod_2MiB": ptrs[0] % (2 << 20), **run(tensors, ptrs)})
    for rep in range(4):
        del buf
        torch.cuda.empty_cache()
        buf, ptrs = frames(frame_bytes)
        out["runs"].append({"what": "RGBA frames allocated again (%d)" % rep, "rgba_base_mod_2MiB": ptrs[0] % (2 << 20), **run(tensors, ptrs)})
    for name, stride, shift in (("frame stride 8 MiB (2 MiB multiple)", 8 << 20, 0), ("frame stride 8 MiB + 4 KiB", (8 << 20) + 4096, 0),
                                ("frame stride 8 MiB + 64 KiB", (8 << 20) + 65536, 0), ("tight stride, base + 64 KiB", frame_bytes, 65536),
                                ("tight stride, base + 1 MiB", frame_bytes, 1 << 20)):
        del buf
        torch.cuda.empty_cache()
        buf, ptrs = frames(stride, shift)
        out["runs"].append({"what": name, "rgba_base_mod_2MiB": ptrs[0] % (2 << 20), **run(tensors, ptrs)})
    del buf
    torch.cuda.empty_cache()
    buf, ptrs = frames(frame_bytes)
    for rep in range(3):
        del tensors
        torch.cuda.empty_cache()
        tensors = upload()
        out["runs"].append({"what": "coefficient planes allocated again (%d)" % rep, "coef_base_mod_2MiB": tensors[0]["coef_y"].data_ptr() % (2 << 20), **run(tensors, ptrs)})
    # one big allocation for all coefficient planes instead of 3 x n small ones
    del tensors
    torch.cuda.empty_cache()
    ny, nc = CW * CH * 2, CW * CH // 2
    per = (ny + 2 * nc + 4095) // 4096 * 4096
    big = torch.empty(n * per, dtype=torch.uint8, device="cuda")
    small = {k: torch.from_numpy(np.ascontiguousarray(t[k])).cuda() for k in ("qscale", "intra")}
    tensors = []
    for i in range(n):
        o = i * per
        y = big[o:o + ny].view(torch.int16); y.copy_(torch.from_numpy(t["coef_y"].ravel()).cuda())
        cb = big[o + ny:o + ny + nc].view(torch.int16); cb.copy_(torch.from_numpy(t["coef_cb"].ravel()).cuda())
        cr = big[o + ny + nc:o + ny + 2 * nc].view(torch.int16); cr.copy_(torch.from_numpy(t["coef_cr"].ravel()).cuda())
        tensors.append({"coef_y": y, "coef_cb": cb, "coef_cr": cr, "qscale": small["qscale"], "intra": small["intra"]})
    out["runs"].append({"what": "coefficients in ONE allocation", **run(tensors, ptrs)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
