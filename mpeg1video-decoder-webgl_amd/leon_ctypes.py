"""ctypes binding of libleon_hip.so (include/leon.h) for the Python-side plumbing:
tests, bench.py and the multi-GPU launcher.  The product's host language is
JavaScript (js/ + the N-API addon); this file only moves pointers around.

Fails loudly when the library is missing or no gfx950 device is usable: there is
no CPU fallback anywhere in this package.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LEON_LIB", os.path.join(_HERE, "lib", "libleon_hip.so"))

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_NO_FREE_SLOT, ERR_NOMEM = -1, -2, -3, -4, -5
PIC_I, PIC_P, PIC_B = 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1
RGB_CPU_TWIN, RGB_GL = 0, 1

# every symbol include/leon.h declares (tests check the library exports all of them)
SYMBOLS = [
    "leon_abi_version", "leon_last_error", "leon_create", "leon_destroy", "leon_set_quant_matrices", "leon_add_quant_matrices",
    "leon_acquire_slot", "leon_release_slot", "leon_free_decoded_slots", "leon_submit_picture",
    "leon_submit_batch", "leon_batch_create", "leon_batch_run", "leon_batch_destroy",
    "leon_submit_sparse", "leon_batch_create_sparse",
    "leon_convert_rgba", "leon_convert_rgba_batch", "leon_read_planes", "leon_write_planes",
    "leon_read_alpha_plane", "leon_write_alpha_plane", "leon_slot_device_ptr", "leon_sync", "leon_set_overlap_convert", "leon_timing_enable", "leon_timing_reset", "leon_timing_get",
    "leon_timing_get_launches",
    "leon_measure_copy_bandwidth", "leon_measure_stream_bandwidth", "leon_device_malloc", "leon_device_free", "leon_device_pool_stats",
]
ABI_VERSION = 3        # LEON_ABI_VERSION of include/leon.h
# include/leon_pipeline.h (same library)
PIPELINE_SYMBOLS = [
    "leon_pipeline_create", "leon_pipeline_create_partial", "leon_pipeline_feed", "leon_pipeline_get_info", "leon_pipeline_release_window", "leon_pipeline_wait",
    "leon_pipeline_get_stats", "leon_pipeline_read_frame", "leon_pipeline_error", "leon_pipeline_destroy",
]


class LeonError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("leon error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("coded_width", C.c_int32), ("coded_height", C.c_int32), ("frame_width", C.c_int32),
                ("frame_height", C.c_int32), ("n_slots", C.c_int32), ("device_id", C.c_int32),
                ("stream", C.c_void_p), ("alpha", C.c_int32), ("contiguous_slots", C.c_int32)]


class Picture(C.Structure):
    _fields_ = [("type", C.c_int32), ("out_slot", C.c_int32), ("ref_fwd_slot", C.c_int32),
                ("ref_bwd_slot", C.c_int32), ("coef_y", C.c_void_p), ("coef_cb", C.c_void_p),
                ("coef_cr", C.c_void_p), ("qscale", C.c_void_p), ("intra", C.c_void_p),
                ("repadd", C.c_void_p), ("mv_fwd", C.c_void_p), ("mv_bwd", C.c_void_p),
                ("mb_dir", C.c_void_p),
                # ABI 2: fused display conversion (device pointer to the RGBA frame, or NULL)
                ("rgba_out", C.c_void_p), ("no_planes", C.c_int32), ("qm_set", C.c_int32),      # qm_set: ABI 3
                ("coef_a", C.c_void_p)]


class SparsePicture(C.Structure):
    _fields_ = [("type", C.c_int32), ("out_slot", C.c_int32), ("ref_fwd_slot", C.c_int32),
                ("ref_bwd_slot", C.c_int32), ("grp_off", C.c_void_p), ("entries", C.c_void_p),
                ("n_entries", C.c_uint32), ("reserved", C.c_int32), ("qscale", C.c_void_p), ("intra", C.c_void_p),
                ("repadd", C.c_void_p), ("mv_fwd", C.c_void_p), ("mv_bwd", C.c_void_p), ("mb_dir", C.c_void_p),
                ("rgba_out", C.c_void_p), ("no_planes", C.c_int32), ("qm_set", C.c_int32)]


class KernelStats(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("total_ms", C.c_double), ("algorithmic_bytes", C.c_double),
                ("macroblocks", C.c_uint64)]


class LaunchTime(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pic_type", C.c_int32), ("ms", C.c_double), ("algorithmic_bytes", C.c_double),
                ("macroblocks", C.c_uint64)]


def read_frame(frame):
    """one frame of a window (a dict handed to on_window) as a host array, through the pipeline that delivered it"""
    return frame["_pipe"].read_frame(frame)


class PipelineConfig(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("parser_threads", C.c_int32), ("gops_per_window", C.c_int32),
                ("windows_in_flight", C.c_int32), ("max_gop_pictures", C.c_int32), ("loop", C.c_int32),
                ("shard_index", C.c_int32), ("shard_count", C.c_int32), ("start_seconds", C.c_double),
                ("gpu_parser", C.c_int32), ("display_flavour", C.c_int32)]


class PipelineFrame(C.Structure):
    _fields_ = [("gop", C.c_uint64), ("display_index", C.c_int32), ("type", C.c_int32), ("ts_ms", C.c_double),
                ("rgba", C.c_void_p)]


class PipelineInfo(C.Structure):
    _fields_ = [("coded_width", C.c_int32), ("coded_height", C.c_int32), ("frame_width", C.c_int32), ("frame_height", C.c_int32),
                ("picture_rate", C.c_double), ("duration", C.c_double), ("gops", C.c_uint32), ("shard_gops", C.c_uint32),
                ("first_gop", C.c_uint32), ("parser_threads", C.c_int32), ("gops_per_window", C.c_int32),
                ("gpu_parser", C.c_int32), ("display_flavour", C.c_int32)]


class PipelineStats(C.Structure):
    _fields_ = [("pictures", C.c_uint64), ("gops", C.c_uint64), ("windows", C.c_uint64), ("stream_bytes", C.c_uint64),
                ("seconds", C.c_double), ("parse_seconds_sum", C.c_double), ("upload_bytes", C.c_double), ("entries", C.c_uint64)]


PIPELINE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.POINTER(PipelineFrame), C.c_int32, C.c_int32)

_lib = None


def load():
    """dlopen the library (no GPU needed for this); raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7; two HIP runtimes in one process cannot both
    # own the GPU ("No HIP GPUs are available").  Whichever is loaded first serves both,
    # so when torch is going to be used in this process let it load first.
    if os.environ.get("LEON_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = os.environ.get("LEON_DEBUG_LIB") or LIB_PATH      # A/B runs of kernel variants (tools/ab_bench.sh)
    if not os.path.exists(path):
        raise ImportError("libleon_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C mpeg1video-decoder-webgl_amd/csrc` (no CPU fallback exists)")
    lib = C.CDLL(path)
    if lib.leon_abi_version() != ABI_VERSION:
        raise ImportError("libleon_hip.so (%s) speaks ABI %d, this binding ABI %d: rebuild it" % (path, lib.leon_abi_version(), ABI_VERSION))
    lib.leon_last_error.restype = C.c_char_p
    lib.leon_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.leon_destroy.argtypes = [C.c_void_p]
    lib.leon_destroy.restype = None
    lib.leon_set_quant_matrices.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.leon_add_quant_matrices.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    lib.leon_acquire_slot.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.leon_release_slot.argtypes = [C.c_void_p, C.c_int32]
    lib.leon_free_decoded_slots.argtypes = [C.c_void_p]
    lib.leon_submit_picture.argtypes = [C.c_void_p, C.POINTER(Picture)]
    lib.leon_submit_batch.argtypes = [C.c_void_p, C.POINTER(Picture), C.c_int32, C.c_int32]
    lib.leon_batch_create.argtypes = [C.c_void_p, C.POINTER(Picture), C.c_int32, C.POINTER(C.c_void_p)]
    lib.leon_submit_sparse.argtypes = [C.c_void_p, C.POINTER(SparsePicture), C.c_int32, C.c_int32]
    lib.leon_batch_create_sparse.argtypes = [C.c_void_p, C.POINTER(SparsePicture), C.c_int32, C.POINTER(C.c_void_p)]
    lib.leon_batch_run.argtypes = [C.c_void_p, C.c_void_p]
    lib.leon_batch_destroy.argtypes = [C.c_void_p, C.c_void_p]
    lib.leon_batch_destroy.restype = None
    lib.leon_convert_rgba.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
    lib.leon_convert_rgba_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
    lib.leon_read_planes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.leon_write_planes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.leon_read_alpha_plane.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.leon_write_alpha_plane.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.leon_slot_device_ptr.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.leon_sync.argtypes = [C.c_void_p]
    lib.leon_set_overlap_convert.argtypes = [C.c_void_p, C.c_int32]
    lib.leon_timing_enable.argtypes = [C.c_void_p, C.c_int32]
    lib.leon_timing_reset.argtypes = [C.c_void_p]
    lib.leon_timing_get.argtypes = [C.c_void_p, C.c_int32, C.POINTER(KernelStats)]
    lib.leon_timing_get_launches.argtypes = [C.c_void_p, C.POINTER(LaunchTime), C.c_int32, C.POINTER(C.c_int32)]
    lib.leon_measure_copy_bandwidth.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.POINTER(C.c_double)]
    lib.leon_measure_stream_bandwidth.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    lib.leon_device_pool_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    lib.leon_device_malloc.argtypes = [C.c_int32, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
    lib.leon_device_free.argtypes = [C.c_void_p]
    lib.leon_pipeline_create.argtypes = [C.POINTER(PipelineConfig), C.c_void_p, C.c_size_t, PIPELINE_CB, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.leon_pipeline_create_partial.argtypes = [C.POINTER(PipelineConfig), C.c_void_p, C.c_size_t, C.c_size_t, PIPELINE_CB, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.leon_pipeline_feed.argtypes = [C.c_void_p, C.c_size_t]
    lib.leon_pipeline_get_info.argtypes = [C.c_void_p, C.POINTER(PipelineInfo)]
    lib.leon_pipeline_release_window.argtypes = [C.c_void_p, C.c_int64]
    lib.leon_pipeline_wait.argtypes = [C.c_void_p]
    lib.leon_pipeline_get_stats.argtypes = [C.c_void_p, C.POINTER(PipelineStats)]
    lib.leon_pipeline_read_frame.argtypes = [C.c_void_p, C.POINTER(PipelineFrame), C.c_void_p]
    lib.leon_pipeline_error.argtypes = [C.c_void_p]
    lib.leon_pipeline_error.restype = C.c_char_p
    lib.leon_pipeline_destroy.argtypes = [C.c_void_p]
    lib.leon_pipeline_destroy.restype = None
    _lib = lib
    return lib


def _chk(rc):
    if rc != OK:
        raise LeonError(rc, load().leon_last_error().decode("utf-8", "replace"))


def _hostptr(a, dtype, keep):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    keep.append(a)
    return a.ctypes.data


def make_picture(ptype, out_slot, coef_y, coef_cb, coef_cr, qscale, intra, repadd=None, mv_fwd=None,
                 mv_bwd=None, mb_dir=None, ref_fwd_slot=-1, ref_bwd_slot=-1, keep=None, device=False,
                 rgba_out=None, no_planes=False, coef_a=None, qm_set=0):
    """Fill a Picture from numpy arrays (host) or raw device addresses (device=True: ints).
    rgba_out: device address of the RGBA frame for the fused display conversion (always a device address);
    coef_a: the A plane's levels for a yuva decoder."""
    p = Picture()
    p.type, p.out_slot, p.ref_fwd_slot, p.ref_bwd_slot = ptype, out_slot, ref_fwd_slot, ref_bwd_slot
    p.rgba_out = None if rgba_out is None else int(rgba_out)
    p.no_planes = 1 if no_planes else 0
    p.qm_set = int(qm_set)
    if device:
        vals = (coef_y, coef_cb, coef_cr, qscale, intra, repadd, mv_fwd, mv_bwd, mb_dir)
        (p.coef_y, p.coef_cb, p.coef_cr, p.qscale, p.intra, p.repadd, p.mv_fwd, p.mv_bwd, p.mb_dir) = \
            [None if v is None else int(v) for v in vals]
        p.coef_a = None if coef_a is None else int(coef_a)
        return p
    keep = keep if keep is not None else []
    p.coef_a = _hostptr(coef_a, np.int16, keep)
    p.coef_y = _hostptr(coef_y, np.int16, keep)
    p.coef_cb = _hostptr(coef_cb, np.int16, keep)
    p.coef_cr = _hostptr(coef_cr, np.int16, keep)
    p.qscale = _hostptr(qscale, np.uint8, keep)
    p.intra = _hostptr(intra, np.uint8, keep)
    p.repadd = _hostptr(repadd, np.uint8, keep)
    p.mv_fwd = _hostptr(mv_fwd, np.int16, keep)
    p.mv_bwd = _hostptr(mv_bwd, np.int16, keep)
    p.mb_dir = _hostptr(mb_dir, np.uint8, keep)
    p._keep = keep
    return p


def make_sparse_picture(ptype, out_slot, grp_off, entries, n_entries, qscale, intra, repadd=None, mv_fwd=None,
                        mv_bwd=None, mb_dir=None, ref_fwd_slot=-1, ref_bwd_slot=-1, keep=None, device=False,
                        rgba_out=None, no_planes=False, qm_set=0):
    """The sparse-boundary twin of make_picture (lists in the format of include/leon_vlc.h)."""
    p = SparsePicture()
    p.qm_set = int(qm_set)
    p.type, p.out_slot, p.ref_fwd_slot, p.ref_bwd_slot = ptype, out_slot, ref_fwd_slot, ref_bwd_slot
    p.rgba_out = None if rgba_out is None else int(rgba_out)
    p.no_planes = 1 if no_planes else 0
    p.n_entries = int(n_entries)
    if device:
        vals = (grp_off, entries, qscale, intra, repadd, mv_fwd, mv_bwd, mb_dir)
        (p.grp_off, p.entries, p.qscale, p.intra, p.repadd, p.mv_fwd, p.mv_bwd, p.mb_dir) = \
            [None if v is None else int(v) for v in vals]
        return p
    keep = keep if keep is not None else []
    p.grp_off = _hostptr(grp_off, np.uint32, keep)
    p.entries = _hostptr(entries, np.uint32, keep)
    p.qscale = _hostptr(qscale, np.uint8, keep)
    p.intra = _hostptr(intra, np.uint8, keep)
    p.repadd = _hostptr(repadd, np.uint8, keep)
    p.mv_fwd = _hostptr(mv_fwd, np.int16, keep)
    p.mv_bwd = _hostptr(mv_bwd, np.int16, keep)
    p.mb_dir = _hostptr(mb_dir, np.uint8, keep)
    p._keep = keep
    return p


class Decoder:
    """Thin object wrapper over the C ABI; method names follow include/leon.h."""

    def __init__(self, coded_w, coded_h, frame_w=None, frame_h=None, n_slots=13, device_id=0, stream=None, alpha=False, contiguous_slots=False):
        self.lib = load()
        cfg = Config(coded_w, coded_h, frame_w or coded_w, frame_h or coded_h, n_slots, device_id, stream, 1 if alpha else 0,
                     1 if contiguous_slots else 0)
        h = C.c_void_p()
        _chk(self.lib.leon_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.cw, self.ch = coded_w, coded_h
        self.fw, self.fh = cfg.frame_width, cfg.frame_height
        self.n_slots = n_slots

    def close(self):
        if self.h:
            self.lib.leon_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_quant_matrices(self, intra=None, non_intra=None):
        keep = []
        _chk(self.lib.leon_set_quant_matrices(self.h, _hostptr(intra, np.uint8, keep), _hostptr(non_intra, np.uint8, keep)))

    def add_quant_matrices(self, intra=None, non_intra=None):
        """a further set of matrices beside set 0; returns the id pictures name in qm_set"""
        keep, s = [], C.c_int32()
        _chk(self.lib.leon_add_quant_matrices(self.h, _hostptr(intra, np.uint8, keep), _hostptr(non_intra, np.uint8, keep), C.byref(s)))
        return s.value

    def acquire_slot(self):
        s = C.c_int32()
        _chk(self.lib.leon_acquire_slot(self.h, C.byref(s)))
        return s.value

    def release_slot(self, slot):
        _chk(self.lib.leon_release_slot(self.h, slot))

    def free_decoded_slots(self):
        _chk(self.lib.leon_free_decoded_slots(self.h))

    def submit_picture(self, pic):
        _chk(self.lib.leon_submit_picture(self.h, C.byref(pic)))

    def submit_batch(self, pics, mem=MEM_DEVICE):
        arr = (Picture * len(pics))(*pics)
        _chk(self.lib.leon_submit_batch(self.h, arr, len(pics), mem))

    def batch_create(self, pics):
        arr = (Picture * len(pics))(*pics)
        b = C.c_void_p()
        _chk(self.lib.leon_batch_create(self.h, arr, len(pics), C.byref(b)))
        return b

    def submit_sparse(self, pics, mem=MEM_HOST):
        arr = (SparsePicture * len(pics))(*pics)
        _chk(self.lib.leon_submit_sparse(self.h, arr, len(pics), mem))

    def batch_create_sparse(self, pics):
        arr = (SparsePicture * len(pics))(*pics)
        b = C.c_void_p()
        _chk(self.lib.leon_batch_create_sparse(self.h, arr, len(pics), C.byref(b)))
        return b

    def batch_run(self, b):
        _chk(self.lib.leon_batch_run(self.h, b))

    def batch_destroy(self, b):
        self.lib.leon_batch_destroy(self.h, b)

    def convert_rgba(self, slot, flavour=RGB_CPU_TWIN):
        out = np.empty((self.fh, self.fw, 4), dtype=np.uint8)
        _chk(self.lib.leon_convert_rgba(self.h, slot, out.ctypes.data, MEM_HOST, flavour))
        return out

    def convert_rgba_batch(self, slots, rgba_device_ptr, flavour=RGB_CPU_TWIN):
        s = np.ascontiguousarray(slots, dtype=np.int32)
        _chk(self.lib.leon_convert_rgba_batch(self.h, s.ctypes.data, len(s), int(rgba_device_ptr), flavour))

    def read_planes(self, slot):
        n = self.cw * self.ch
        y = np.empty((self.ch, self.cw), dtype=np.uint8)
        cb = np.empty((self.ch // 2, self.cw // 2), dtype=np.uint8)
        cr = np.empty((self.ch // 2, self.cw // 2), dtype=np.uint8)
        _chk(self.lib.leon_read_planes(self.h, slot, y.ctypes.data, cb.ctypes.data, cr.ctypes.data))
        return y, cb, cr

    def read_alpha_plane(self, slot):
        a = np.empty((self.ch, self.cw), dtype=np.uint8)
        _chk(self.lib.leon_read_alpha_plane(self.h, slot, a.ctypes.data))
        return a

    def write_alpha_plane(self, slot, a):
        keep = []
        _chk(self.lib.leon_write_alpha_plane(self.h, slot, _hostptr(a, np.uint8, keep)))

    def write_planes(self, slot, y, cb, cr):
        keep = []
        _chk(self.lib.leon_write_planes(self.h, slot, _hostptr(y, np.uint8, keep), _hostptr(cb, np.uint8, keep),
                                        _hostptr(cr, np.uint8, keep)))

    def slot_device_ptr(self, slot):
        p, n = C.c_void_p(), C.c_size_t()
        _chk(self.lib.leon_slot_device_ptr(self.h, slot, C.byref(p), C.byref(n)))
        return p.value, n.value

    def sync(self):
        _chk(self.lib.leon_sync(self.h))

    def set_overlap_convert(self, on=True):
        _chk(self.lib.leon_set_overlap_convert(self.h, 1 if on else 0))

    def timing_enable(self, on=True):
        _chk(self.lib.leon_timing_enable(self.h, 1 if on else 0))

    def timing_reset(self):
        _chk(self.lib.leon_timing_reset(self.h))

    def timing_get(self, kind=0):
        s = KernelStats()
        _chk(self.lib.leon_timing_get(self.h, kind, C.byref(s)))
        return {"launches": s.launches, "total_ms": s.total_ms, "algorithmic_bytes": s.algorithmic_bytes,
                "macroblocks": s.macroblocks}

    def timing_launches(self):
        """the timed launches one by one, in submission order"""
        n = C.c_int32()
        _chk(self.lib.leon_timing_get_launches(self.h, None, 0, C.byref(n)))
        arr = (LaunchTime * max(1, n.value))()
        _chk(self.lib.leon_timing_get_launches(self.h, arr, n.value, C.byref(n)))
        return [{"kind": a.kind, "pic_type": a.pic_type, "ms": a.ms, "algorithmic_bytes": a.algorithmic_bytes,
                 "macroblocks": a.macroblocks} for a in arr[:n.value]]

    def measure_copy_bandwidth(self, nbytes=1 << 31, iters=10):
        g = C.c_double()
        _chk(self.lib.leon_measure_copy_bandwidth(self.h, nbytes, iters, C.byref(g)))
        return g.value

    def measure_stream_bandwidth(self, nbytes=1 << 31, iters=10, reads=1, writes=1):
        """GB/s over all streams of the 16-byte-per-lane kernel with `reads` source and `writes` destination streams"""
        g = C.c_double()
        _chk(self.lib.leon_measure_stream_bandwidth(self.h, nbytes, iters, reads, writes, C.byref(g)))
        return g.value


class _Frames:
    """the frames of a delivered window as a read-only sequence of dicts, made when asked for (a 128-GOP window has 1536 of
    them; a callback that looks at a dozen should not pay for the rest on the pipeline's notify thread)"""

    def __init__(self, frames, n, pipe):
        self._f, self._n, self._pipe = frames, n, pipe

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        f = self._f[i]
        return {"gop": int(f.gop), "display_index": f.display_index, "type": f.type, "ts_ms": f.ts_ms, "rgba": f.rgba,
                "_i": i, "_frames": self._f, "_pipe": self._pipe}

    def __iter__(self):
        return (self[i] for i in range(self._n))


class Pipeline:
    """leon_pipeline_* (include/leon_pipeline.h): stream bytes in, RGBA frames in device memory out.
    on_window(window_id, frames) runs on the pipeline's notify thread with a list of dicts
    (gop, display_index, type, ts_ms, rgba = device address); unless it returns False the window is
    released right after.  read_frame(frame) works until the frame's window is released."""

    def __init__(self, data, device_id=0, parser_threads=0, gops_per_window=0, windows_in_flight=0, max_gop_pictures=0,
                 loop=0, on_window=None, shard_index=0, shard_count=0, start_seconds=0.0, gpu_parser=None, valid_bytes=None, display_flavour=0):
        self.lib = load()
        self._data = (C.c_uint8 * len(data)).from_buffer_copy(data)      # must outlive the pipeline
        self._on_window = on_window
        self.windows = 0
        self.frames = 0
        self.ended = False
        self.error = None

        import threading
        ready = threading.Event()          # set once self.h exists: decoding starts inside leon_pipeline_create

        def _release(window):
            rc = self.lib.leon_pipeline_release_window(self.h, window)
            if rc != OK and self.error is None:
                self.error = LeonError(rc, "leon_pipeline_release_window(%d): %s" % (window, self.lib.leon_last_error().decode()))

        def _cb(_user, window, frames, n, status):
            try:
                ready.wait()
                if window < 0:
                    self.ended = True
                    return
                if status != OK:          # a failed window is delivered too and must be given back like any other
                    self.error = status
                    _release(window)
                    return
                self.windows += 1
                self.frames += n
                keep = None
                if self._on_window is not None:
                    # `_pipe`: a callback may read frames through the dicts alone (read_frame(f) below), without the
                    # variable its caller assigns the pipeline to -- which does not exist yet while the constructor runs
                    fl = _Frames(frames, n, self)
                    keep = self._on_window(window, fl)
                if keep is not False:
                    _release(window)
            except Exception as e:      # never let an exception cross the C boundary
                self.error = e
        self._cb = PIPELINE_CB(_cb)
        self._ready = ready
        cfg = PipelineConfig(device_id, parser_threads, gops_per_window, windows_in_flight, max_gop_pictures, loop,
                             shard_index, shard_count, float(start_seconds), 0 if gpu_parser is None else (1 if gpu_parser else -1), int(display_flavour))      # None: the library's default (the GPU)
        h = C.c_void_p()
        self.h = None
        # valid_bytes: the stream is still arriving (leon_pipeline_create_partial); feed() reports progress
        if valid_bytes is None:
            rc = self.lib.leon_pipeline_create(C.byref(cfg), self._data, len(data), self._cb, None, C.byref(h))
        else:
            rc = self.lib.leon_pipeline_create_partial(C.byref(cfg), self._data, len(data), int(valid_bytes), self._cb, None, C.byref(h))
        _chk(rc)
        self.h = h
        info = PipelineInfo()
        rc = self.lib.leon_pipeline_get_info(self.h, C.byref(info))
        self.info = info
        ready.set()
        _chk(rc)

    def read_frame(self, frame):
        out = np.empty((self.info.frame_height, self.info.frame_width, 4), dtype=np.uint8)
        _chk(self.lib.leon_pipeline_read_frame(self.h, C.byref(frame["_frames"][frame["_i"]]), out.ctypes.data))
        return out

    def feed(self, valid_bytes, chunk=None, offset=None):
        """more of the stream has arrived: optionally copy `chunk` to `offset` of the pipeline's buffer first"""
        if chunk is not None:
            if offset is None:
                raise ValueError("feed(chunk=...) needs the offset the chunk belongs at")
            offset, chunk = int(offset), bytes(chunk)
            if offset < 0 or offset + len(chunk) > len(self._data):
                raise ValueError("chunk of %d bytes at offset %d does not fit the stream buffer of %d bytes" % (len(chunk), offset, len(self._data)))
            C.memmove(C.addressof(self._data) + offset, chunk, len(chunk))
        _chk(self.lib.leon_pipeline_feed(self.h, int(valid_bytes)))

    def release_window(self, window):
        _chk(self.lib.leon_pipeline_release_window(self.h, window))

    def wait(self):
        _chk(self.lib.leon_pipeline_wait(self.h))
        if isinstance(self.error, Exception):
            raise self.error

    def stats(self):
        s = PipelineStats()
        _chk(self.lib.leon_pipeline_get_stats(self.h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in PipelineStats._fields_}

    def close(self):
        if self.h:
            self._ready.set()
            self.lib.leon_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pool_stats():
    """leon_device_pool_stats: contiguous memory this process holds (never returned to the driver), how much is handed out"""
    held, used, seg = C.c_uint64(), C.c_uint64(), C.c_int32()
    _chk(load().leon_device_pool_stats(C.byref(held), C.byref(used), C.byref(seg)))
    return {"held_bytes": held.value, "in_use_bytes": used.value, "segments": seg.value}


def device_view(ptr, nbytes, device_id=0):
    """`nbytes` of device memory at `ptr` as a torch uint8 tensor, without copying (the memory must outlive the tensor):
    how bench.py and the tests checksum the pipeline's frames where they lie"""
    import torch

    class _View:
        pass
    v = _View()
    v.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(v, device="cuda:%d" % device_id)


class DeviceBuffer:
    """leon_device_malloc / leon_device_free: device memory allocated the way the library allocates its own large buffers
    (physically contiguous where the device grants it).  `ptr` is the device address; `contiguous` says which it got.
    as_tensor(dtype, shape) wraps (a part of) it as a torch tensor without copying -- the buffer must outlive the tensor
    (the tensor keeps a reference to it)."""

    def __init__(self, nbytes, device_id=0):
        self.lib = load()
        p, c = C.c_void_p(), C.c_int32(0)
        _chk(self.lib.leon_device_malloc(device_id, nbytes, C.byref(p), C.byref(c)))
        self.ptr, self.nbytes, self.contiguous, self.device_id = p.value, nbytes, bool(c.value), device_id

    def as_tensor(self, dtype, shape, offset=0):
        import numpy as np
        import torch
        typestr = {torch.uint8: "|u1", torch.int8: "|i1", torch.int16: "<i2", torch.int32: "<i4", torch.int64: "<i8",
                   torch.float32: "<f4", torch.float64: "<f8"}[dtype]
        n = int(np.prod(shape)) * int(typestr[2:])
        if offset < 0 or offset + n > self.nbytes:
            raise ValueError("tensor does not fit in the buffer")

        class _View:          # __cuda_array_interface__: torch.as_tensor wraps the memory and keeps this object (and the buffer) alive
            pass
        v = _View()
        v.owner = self
        v.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": typestr, "data": (self.ptr + offset, False), "version": 2}
        return torch.as_tensor(v, device="cuda:%d" % self.device_id)

    def free(self):
        if self.ptr:
            _chk(self.lib.leon_device_free(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
