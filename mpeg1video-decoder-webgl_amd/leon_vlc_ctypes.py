"""ctypes binding of libleon_vlc.so (include/leon_vlc.h): the native bitstream front end.
No torch, no HIP: usable on any host.  Arrays returned by Stream.next_picture() are COPIES
(numpy), so they survive the next call."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 2        # LEON_VLC_ABI_VERSION of include/leon_vlc.h
SYMBOLS = ["leon_vlc_abi_version", "leon_vlc_last_error", "leon_vlc_open", "leon_vlc_open_shard", "leon_vlc_open_scan", "leon_vlc_close", "leon_vlc_get_info",
           "leon_vlc_next_picture", "leon_vlc_next_picture_sync", "leon_vlc_seek", "leon_vlc_densify", "leon_vlc_densify_alpha", "leon_vlc_get_keymap",
           "leon_vlc_scan_picture", "leon_vlc_get_gpu_tables"]


class PictureScan(C.Structure):
    _fields_ = [("type", C.c_int32), ("temporal_reference", C.c_int32), ("ts_ms", C.c_double), ("new_sequence", C.c_int32),
                ("full_pel_fwd", C.c_int32), ("fwd_rsize", C.c_int32), ("full_pel_bwd", C.c_int32), ("bwd_rsize", C.c_int32),
                ("n_slices", C.c_uint32), ("slice_code", C.POINTER(C.c_int32)), ("slice_bit_pos", C.POINTER(C.c_uint64)),
                ("end_byte", C.c_uint64), ("open_gop", C.c_int32), ("reserved", C.c_int32)]


class GpuTables(C.Structure):
    _fields_ = [("fast12", C.c_uint32 * 4096), ("coef16", C.c_int32 * 65536), ("motion_s", C.c_int32 * 2048), ("mba", C.c_int32 * 2048),
                ("mbtype", (C.c_int32 * 64) * 4), ("cbp", C.c_int32 * 512), ("dc_lum", C.c_int32 * 128), ("dc_chr", C.c_int32 * 256),
                ("zz_off", C.c_uint16 * 64)]


class Info(C.Structure):
    _fields_ = [("frame_width", C.c_int32), ("frame_height", C.c_int32), ("coded_width", C.c_int32),
                ("coded_height", C.c_int32), ("mb_width", C.c_int32), ("mb_height", C.c_int32),
                ("groups_y", C.c_int32), ("groups_c", C.c_int32), ("n_groups", C.c_int32), ("has_alpha", C.c_int32),
                ("picture_rate", C.c_double), ("duration", C.c_double), ("keymap_count", C.c_uint32),
                ("threads", C.c_uint32), ("intra_qm", C.c_uint8 * 64), ("non_intra_qm", C.c_uint8 * 64)]


class Picture(C.Structure):
    _fields_ = [("type", C.c_int32), ("temporal_reference", C.c_int32), ("ts_ms", C.c_double),
                ("new_sequence", C.c_int32), ("n_groups", C.c_int32), ("n_entries", C.c_uint32),
                ("grp_off", C.POINTER(C.c_uint32)), ("entries", C.POINTER(C.c_uint32)),
                ("qscale", C.POINTER(C.c_uint8)), ("intra", C.POINTER(C.c_uint8)), ("repadd", C.POINTER(C.c_uint8)),
                ("mv_fwd", C.POINTER(C.c_int16)), ("mv_bwd", C.POINTER(C.c_int16)), ("mb_dir", C.POINTER(C.c_uint8)),
                ("n_slices", C.c_uint32), ("open_gop", C.c_int32)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("LEON_VLC_LIB", os.path.join(_HERE, "lib", "libleon_vlc.so"))
    if not os.path.exists(path):
        raise RuntimeError("libleon_vlc.so is missing (%s): run `make -C mpeg1video-decoder-webgl_amd/csrc`" % path)
    lib = C.CDLL(path)
    if not hasattr(lib, "leon_vlc_abi_version") or lib.leon_vlc_abi_version() != ABI_VERSION:
        raise RuntimeError("libleon_vlc.so (%s) does not speak ABI %d of include/leon_vlc.h: rebuild it" % (path, ABI_VERSION))
    lib.leon_vlc_last_error.restype = C.c_char_p
    lib.leon_vlc_open.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.POINTER(C.c_void_p)]
    lib.leon_vlc_open_shard.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.leon_vlc_open_scan.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int32, C.POINTER(C.c_void_p)]
    lib.leon_vlc_close.argtypes = [C.c_void_p]
    lib.leon_vlc_close.restype = None
    lib.leon_vlc_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
    lib.leon_vlc_next_picture.argtypes = [C.c_void_p, C.POINTER(Picture)]
    lib.leon_vlc_densify_alpha.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.leon_vlc_get_keymap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.leon_vlc_seek.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_uint64)]
    lib.leon_vlc_densify.argtypes = [C.POINTER(Info), C.POINTER(Picture), C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


class VlcError(RuntimeError):
    pass


def _arr(ptr, n, dtype):
    if not ptr or n == 0:
        return None if not ptr else np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class Stream:
    """One JSV / MPEG-1 video elementary stream; mirrors the decoder object's parsing half
    (decodeFrame / seek of decoders/jsv.js) and yields sparse boundary tensors."""

    def __init__(self, data, threads=0, has_alpha=-1, scan_only=False):
        """has_alpha: for a GOP shard (no container header of its own), the whole stream's info.has_alpha.
        scan_only: leon_vlc_open_scan -- the bytes are read in place, only scan_picture() is served."""
        self.h = None
        self.lib = load()
        h = C.c_void_p()
        if scan_only:
            self._bytes = bytes(data) + bytes(16)
            rc = self.lib.leon_vlc_open_scan(self._bytes, len(self._bytes) - 16, len(self._bytes), has_alpha, C.byref(h))
        else:
            self._bytes = bytes(data)
            rc = self.lib.leon_vlc_open_shard(self._bytes, len(self._bytes), threads, has_alpha, C.byref(h))
        if rc != 0:
            raise VlcError(self.lib.leon_vlc_last_error().decode())
        self.h = h
        self.info = Info()
        self.lib.leon_vlc_get_info(self.h, C.byref(self.info))

    def close(self):
        if self.h:
            self.lib.leon_vlc_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def refresh_info(self):
        self.lib.leon_vlc_get_info(self.h, C.byref(self.info))
        return self.info

    def next_picture(self, dense=False):
        """dict of numpy arrays for the next picture in coded order, or None at the end."""
        p = Picture()
        rc = self.lib.leon_vlc_next_picture(self.h, C.byref(p))
        if rc < 0:
            raise VlcError(self.lib.leon_vlc_last_error().decode())
        if rc == 0:
            return None
        I = self.refresh_info()
        nmb = I.mb_width * I.mb_height
        out = {"type": p.type, "temporal_reference": p.temporal_reference, "ts": p.ts_ms, "new_sequence": bool(p.new_sequence),
               "n_slices": p.n_slices, "open_gop": bool(p.open_gop),
               "grp_off": _arr(p.grp_off, p.n_groups + 1, np.uint32), "entries": _arr(p.entries, p.n_entries, np.uint32),
               "qscale": _arr(p.qscale, nmb, np.uint8), "intra": _arr(p.intra, nmb, np.uint8),
               "repadd": _arr(p.repadd, nmb, np.uint8), "mv_fwd": _arr(p.mv_fwd, 2 * nmb, np.int16),
               "mv_bwd": _arr(p.mv_bwd, 2 * nmb, np.int16), "mb_dir": _arr(p.mb_dir, nmb, np.uint8)}
        if dense:
            cw, ch = I.coded_width, I.coded_height
            y = np.empty(cw * ch, np.int16)
            cb = np.empty(cw * ch // 4, np.int16)
            cr = np.empty(cw * ch // 4, np.int16)
            if self.lib.leon_vlc_densify(C.byref(I), C.byref(p), y.ctypes.data, cb.ctypes.data, cr.ctypes.data) != 0:
                raise VlcError(self.lib.leon_vlc_last_error().decode())
            out["coef_y"], out["coef_cb"], out["coef_cr"] = y, cb, cr
            if I.has_alpha == 1:
                a = np.empty(cw * ch, np.int16)
                if self.lib.leon_vlc_densify_alpha(C.byref(I), C.byref(p), a.ctypes.data) != 0:
                    raise VlcError(self.lib.leon_vlc_last_error().decode())
                out["coef_a"] = a
        return out

    def scan_picture(self):
        """leon_vlc_scan_picture: the next picture's header fields and slice positions (nothing below a slice start
        code is read), or None at the end.  Not to be mixed with next_picture on one stream."""
        p = PictureScan()
        self.lib.leon_vlc_scan_picture.argtypes = [C.c_void_p, C.POINTER(PictureScan)]
        rc = self.lib.leon_vlc_scan_picture(self.h, C.byref(p))
        if rc < 0:
            raise VlcError(self.lib.leon_vlc_last_error().decode())
        if rc == 0:
            return None
        return {"type": p.type, "temporal_reference": p.temporal_reference, "ts": p.ts_ms, "new_sequence": bool(p.new_sequence),
                "full_pel_fwd": p.full_pel_fwd, "fwd_rsize": p.fwd_rsize, "full_pel_bwd": p.full_pel_bwd, "bwd_rsize": p.bwd_rsize,
                "slice_code": [int(p.slice_code[i]) for i in range(p.n_slices)],
                "slice_bit_pos": [int(p.slice_bit_pos[i]) for i in range(p.n_slices)], "end_byte": int(p.end_byte)}

    def keymap(self):
        """byte offsets of the GOP shards (leon_vlc_get_keymap)"""
        n = self.lib.leon_vlc_get_keymap(self.h, None, None, 0)
        offs = (C.c_uint32 * max(n, 1))()
        self.lib.leon_vlc_get_keymap(self.h, offs, None, n)
        return [int(offs[i]) for i in range(n)]

    def shard_ranges(self):
        """(begin, end) of every GOP shard: from a key-map entry up to the next one's offset plus three -- the start
        code prefix of what follows stays in, see leon_vlc_get_keymap in include/leon_vlc.h"""
        offs, n = self.keymap(), len(self._bytes)
        return [(b, min((offs[g + 1] if g + 1 < len(offs) else n) + 3, n)) for g, b in enumerate(offs)]

    def seek(self, seconds):
        off = C.c_uint64()
        if self.lib.leon_vlc_seek(self.h, float(seconds), C.byref(off)) != 0:
            raise VlcError(self.lib.leon_vlc_last_error().decode())
        return int(off.value)


def sparsify(coef_y, coef_cb, coef_cr, cw, ch, coef_a=None):
    """Dense int16 planes -> (grp_off, entries) in the format of include/leon_vlc.h (numpy; tests
    and synthetic workloads).  coef_a: the A plane of a yuva picture; its groups follow the Cr groups."""
    mbw, mbh = cw // 16, ch // 16
    gy, gc = (2 * mbw + 7) // 8, (mbw + 7) // 8
    gids, ents = [], []
    base = 0
    planes = [(coef_y, cw, ch, gy), (coef_cb, cw // 2, ch // 2, gc), (coef_cr, cw // 2, ch // 2, gc)]
    if coef_a is not None:
        planes.append((coef_a, cw, ch, gy))
    for plane, W, H, G in planes:
        p = np.asarray(plane, dtype=np.int16).reshape(H, W)
        ys, xs = np.nonzero(p)
        R, r = ys >> 3, ys & 7
        q, c = xs >> 3, xs & 7
        gids.append(base + R * G + (q >> 3))
        off = (r * 128 + (q & 7) * 16 + c * 2).astype(np.uint32)
        ents.append((off << 16) | p[ys, xs].astype(np.uint16).astype(np.uint32))
        base += (H // 8) * G
    gid = np.concatenate(gids)
    ent = np.concatenate(ents)
    order = np.argsort(gid, kind="stable")
    grp_off = np.zeros(base + 1, np.uint32)
    np.cumsum(np.bincount(gid, minlength=base), out=grp_off[1:])
    return grp_off, ent[order].astype(np.uint32)
