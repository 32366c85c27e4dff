"""GPU: leon_device_malloc / leon_device_free (include/leon.h) -- the allocator bench.py and the pipeline use for their
large buffers (physically contiguous where the device grants it) -- through the ctypes wrapper L.DeviceBuffer.
(Named to run last: freed contiguous memory in a process's history is what round 3 saw go with wrong frames in later
pipelines -- DESIGN.md section 4, "spread" -- and nothing suggests a caller's 8 MB do that, but the suite need not find out.)"""
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
