"""GPU: leon_device_malloc / leon_device_free (include/leon.h) -- the allocator bench.py uses for its large buffers
(physically contiguous where the device grants it) -- through the ctypes wrapper L.DeviceBuffer, and the process-lifetime
POOL behind it: contiguous memory is recycled, never handed back to the driver (round 3 saw wrong B pictures in later
pipelines of processes that had hipFree'd contiguous slot rings; DESIGN.md section 9, leon_hip.cpp big_alloc)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_buffer_round_trip_and_views():
    import torch
    import leon_ctypes as L
    buf = L.DeviceBuffer(8 << 20)
    assert buf.ptr and buf.ptr % 256 == 0 and isinstance(buf.contiguous, bool)
    a = buf.as_tensor(torch.int16, (1024, 512))
    b = buf.as_tensor(torch.uint8, (4096,), offset=4 << 20)
    assert a.data_ptr() == buf.ptr and b.data_ptr() == buf.ptr + (4 << 20) and a.is_cuda
    src = torch.arange(1024 * 512, dtype=torch.int32).to(torch.int16).reshape(1024, 512)
    a.copy_(src)
    b.fill_(7)
    torch.cuda.synchronize()
    assert torch.equal(a.cpu(), src) and int(b.sum()) == 7 * 4096
    with pytest.raises(ValueError):
        buf.as_tensor(torch.uint8, (16,), offset=(8 << 20) - 8)
    del a, b
    buf.free()
    buf.free()                      # idempotent
    with pytest.raises(L.LeonError):
        L.DeviceBuffer(0)


def test_frames_written_into_a_library_buffer():
    """the fused launch writes its RGBA frame into memory from leon_device_malloc exactly as into a torch tensor"""
    import torch
    import leon_ctypes as L
    import synth as S
    from oracle import oracle_py as O
    cw, ch = 96, 64
    rng = np.random.default_rng(5)
    t = S.make_picture(rng, cw, ch, S.PIC_I)
    dec = L.Decoder(cw, ch, n_slots=2)
    try:
        buf = L.DeviceBuffer(cw * ch * 4 + 256)
        frame = buf.as_tensor(torch.uint8, (ch, cw, 4))
        frame.zero_()
        keep = []
        p = L.make_picture(S.PIC_I, 0, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"], keep=keep, rgba_out=buf.ptr, no_planes=False)
        dec.submit_picture(p)
        dec.sync()
        planes = O.decode_picture(S.PIC_I, cw, ch, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"])
        y, cb, cr = O.split_planes(planes, cw, ch)
        want = O.ycbcr_to_rgba(y, cb, cr, cw, cw, ch)
        assert np.array_equal(frame.cpu().numpy().reshape(-1), np.asarray(want).reshape(-1))
        del frame
        buf.free()
    finally:
        dec.close()


def test_contiguous_memory_is_recycled_not_returned():
    """free -> the range is handed to the next request (smallest range that fits, split at 2 MiB granules, merged with free
    neighbours); the pool's holdings never shrink"""
    import leon_ctypes as L
    MB = 1 << 20
    a = L.DeviceBuffer(64 * MB)
    if not a.contiguous:
        a.free()
        pytest.skip("this device grants no physically contiguous memory: nothing is pooled")
    s0 = L.pool_stats()
    p0 = a.ptr
    a.free()
    s1 = L.pool_stats()
    assert s1["held_bytes"] == s0["held_bytes"] and s1["in_use_bytes"] == s0["in_use_bytes"] - 64 * MB
    b = L.DeviceBuffer(48 * MB)                    # the freed range, split
    c = L.DeviceBuffer(15 * MB + 5)                # the tail of it (rounded up to 16 MiB)
    assert b.contiguous and c.contiguous and b.ptr == p0 and c.ptr == p0 + 48 * MB
    assert L.pool_stats()["held_bytes"] == s0["held_bytes"] and L.pool_stats()["segments"] == s0["segments"]
    b.free()
    c.free()
    d = L.DeviceBuffer(64 * MB)                    # merged again
    assert d.ptr == p0 and L.pool_stats()["held_bytes"] == s0["held_bytes"]
    e = L.DeviceBuffer(64 * MB)                    # nothing free that fits: a new segment
    assert e.ptr != p0 and L.pool_stats()["held_bytes"] == s0["held_bytes"] + 64 * MB
    d.free()
    e.free()
    assert L.pool_stats()["in_use_bytes"] == s1["in_use_bytes"]
    small = L.DeviceBuffer(4096)                   # below 1 MiB: an ordinary allocation, freed for real
    assert not small.contiguous
    small.free()


def test_decoders_with_contiguous_slot_rings_come_and_go_and_a_later_pipeline_is_right():
    """leon_config.contiguous_slots in a process that creates and destroys decoders (what round 3 saw go wrong, and shipped
    with a warning): two decoders with 75 MB contiguous slot rings, written by reconstruction launches, destroyed -- the
    second takes the first one's ring back from the pool --, then the 5-GOP 96 x 64 stream of the round-3 failure through
    both front ends against the oracle."""
    import leon_ctypes as L
    import synth as S
    from test_pipeline_gpu import assert_frames, end_capture, ibbp_stream, oracle_frames, run_pipeline, start_capture
    cw, ch = 1920, 1088
    rng = np.random.default_rng(17)
    t = S.make_picture(rng, cw, ch, S.PIC_I)
    held = []
    for k in range(2):
        before = L.pool_stats()
        dec = L.Decoder(cw, ch, 1920, 1080, n_slots=24, contiguous_slots=True)
        try:
            held.append(L.pool_stats()["held_bytes"])
            keep = []
            for slot in (0, 11, 23):               # kernels write the ring, first and last slot included
                dec.submit_picture(L.make_picture(S.PIC_I, slot, t["coef_y"], t["coef_cb"], t["coef_cr"], t["qscale"], t["intra"], keep=keep))
            dec.sync()
            y0, _, _ = dec.read_planes(0)
            y23, _, _ = dec.read_planes(23)
            assert np.array_equal(y0, y23)
        finally:
            dec.close()
        after = L.pool_stats()
        assert after["in_use_bytes"] == before["in_use_bytes"], "the ring went back to the pool"
        assert after["held_bytes"] >= before["held_bytes"], "nothing contiguous goes back to the driver"
    assert held[1] == held[0], "the second decoder's ring is the first one's, recycled"
    data = ibbp_stream(96, 64, [6, 9, 3, 12, 6], seed=77)
    detail = {}
    want = oracle_frames(data, detail)
    for gpu_parser in (True, False):
        cap = start_capture()
        try:
            got, _, stats = run_pipeline(L, data, parser_threads=2, gops_per_window=1, gpu_parser=gpu_parser)
        except BaseException:
            end_capture(cap)
            raise
        assert set(got) == set(want) and stats["pictures"] == len(want)
        assert_frames(got, want, detail, cap, "after_contiguous_rings_%s" % ("gpu" if gpu_parser else "host"))
