#!/usr/bin/env python3
"""jsv_writer.py -- synthetic JSV stream writer (test tooling).

Turns per-picture boundary tensors (the dicts of mpeg1video-decoder-webgl_amd/synth.py)
into a JSV byte stream: the container header + GOP key map the reference reads in
decoders/jsv.js:237-350, then MPEG-1 video layers (ISO/IEC 11172-2) with the one
start-code difference of the format (sequence header 0xC3, decoders/jsv.js:2440).
The VLC tables are those of ISO/IEC 11172-2 Annex B, written here as (code, length).

The output is validated by feeding it to the UNMODIFIED reference parser under Node
(tools/make_golden.js parser) and to the product's own parser (js/jsv_parser.js).
"""
import numpy as np

PIC_I, PIC_P, PIC_B = 1, 2, 3

ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
    28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
    54, 47, 55, 62, 63])

# Table B.1 macroblock_address_increment: index = increment - 1
MBA = [(0x1, 1), (0x3, 3), (0x2, 3), (0x3, 4), (0x2, 4), (0x3, 5), (0x2, 5), (0x7, 7), (0x6, 7), (0xb, 8), (0xa, 8),
       (0x9, 8), (0x8, 8), (0x7, 8), (0x6, 8), (0x17, 10), (0x16, 10), (0x15, 10), (0x14, 10), (0x13, 10), (0x12, 10),
       (0x23, 11), (0x22, 11), (0x21, 11), (0x20, 11), (0x1f, 11), (0x1e, 11), (0x1d, 11), (0x1c, 11), (0x1b, 11),
       (0x1a, 11), (0x19, 11), (0x18, 11)]
MBA_ESCAPE = (0x8, 11)
MBA_STUFFING = (0xf, 11)

# flags: 0x10 quant, 0x08 forward, 0x04 backward, 0x02 pattern, 0x01 intra
MBTYPE_I = {0x01: (0x1, 1), 0x11: (0x1, 2)}
MBTYPE_P = {0x0a: (0x1, 1), 0x02: (0x1, 2), 0x08: (0x1, 3), 0x01: (0x3, 5), 0x1a: (0x2, 5), 0x12: (0x1, 5),
            0x11: (0x1, 6)}
MBTYPE_B = {0x0c: (0x2, 2), 0x0e: (0x3, 2), 0x04: (0x2, 3), 0x06: (0x3, 3), 0x08: (0x2, 4), 0x0a: (0x3, 4),
            0x01: (0x3, 5), 0x1e: (0x2, 5), 0x1a: (0x3, 6), 0x16: (0x2, 6), 0x11: (0x1, 6)}

# Table B.3 coded_block_pattern: index = cbp (1..63)
CBP = [(0x1, 9), (0xb, 5), (0x9, 5), (0xd, 6), (0xd, 4), (0x17, 7), (0x13, 7), (0x1f, 8), (0xc, 4), (0x16, 7),
       (0x12, 7), (0x1e, 8), (0x13, 5), (0x1b, 8), (0x17, 8), (0x13, 8), (0xb, 4), (0x15, 7), (0x11, 7), (0x1d, 8),
       (0x11, 5), (0x19, 8), (0x15, 8), (0x11, 8), (0xf, 6), (0xf, 8), (0xd, 8), (0x3, 9), (0xf, 5), (0xb, 8),
       (0x7, 8), (0x7, 9), (0xa, 4), (0x14, 7), (0x10, 7), (0x1c, 8), (0xe, 6), (0xe, 8), (0xc, 8), (0x2, 9),
       (0x10, 5), (0x18, 8), (0x14, 8), (0x10, 8), (0xe, 5), (0xa, 8), (0x6, 8), (0x6, 9), (0x12, 5), (0x1a, 8),
       (0x16, 8), (0x12, 8), (0xd, 5), (0x9, 8), (0x5, 8), (0x5, 9), (0xc, 5), (0x8, 8), (0x4, 8), (0x4, 9),
       (0x7, 3), (0xa, 5), (0x8, 5), (0xc, 6)]

# Table B.4 motion vector code magnitude 0..16 (sign bit follows for non-zero)
MOTION = [(0x1, 1), (0x1, 2), (0x1, 3), (0x1, 4), (0x3, 6), (0x5, 7), (0x4, 7), (0x3, 7), (0xb, 9), (0xa, 9),
          (0x9, 9), (0x11, 10), (0x10, 10), (0xf, 10), (0xe, 10), (0xd, 10), (0xc, 10)]

# Table B.5a / B.5b dct_dc_size
DC_LUM = [(0x4, 3), (0x0, 2), (0x1, 2), (0x5, 3), (0x6, 3), (0xe, 4), (0x1e, 5), (0x3e, 6), (0x7e, 7)]
DC_CHR = [(0x0, 2), (0x1, 2), (0x2, 2), (0x6, 3), (0xe, 4), (0x1e, 5), (0x3e, 6), (0x7e, 7), (0xfe, 8)]

# Table B.5c-g dct coefficients: (code, length) without the sign bit, in the order of
# (run 0: levels 1..40) (run 1: 1..18) (run 2: 1..5) (run 3: 1..4) (run 4: 1..3) (run 5: 1..3)
# (run 6: 1..3) (run 7..10: 1..2) (run 11..16: 1..2) (run 17..31: 1)
_COEF_CODES = [
    (0x3, 2), (0x4, 4), (0x5, 5), (0x6, 7), (0x26, 8), (0x21, 8), (0xa, 10), (0x1d, 12), (0x18, 12), (0x13, 12),
    (0x10, 12), (0x1a, 13), (0x19, 13), (0x18, 13), (0x17, 13), (0x1f, 14), (0x1e, 14), (0x1d, 14), (0x1c, 14),
    (0x1b, 14), (0x1a, 14), (0x19, 14), (0x18, 14), (0x17, 14), (0x16, 14), (0x15, 14), (0x14, 14), (0x13, 14),
    (0x12, 14), (0x11, 14), (0x10, 14), (0x18, 15), (0x17, 15), (0x16, 15), (0x15, 15), (0x14, 15), (0x13, 15),
    (0x12, 15), (0x11, 15), (0x10, 15),
    (0x3, 3), (0x6, 6), (0x25, 8), (0xc, 10), (0x1b, 12), (0x16, 13), (0x15, 13), (0x1f, 15), (0x1e, 15), (0x1d, 15),
    (0x1c, 15), (0x1b, 15), (0x1a, 15), (0x19, 15), (0x13, 16), (0x12, 16), (0x11, 16), (0x10, 16),
    (0x5, 4), (0x4, 7), (0xb, 10), (0x14, 12), (0x14, 13),
    (0x7, 5), (0x24, 8), (0x1c, 12), (0x13, 13),
    (0x6, 5), (0xf, 10), (0x12, 12),
    (0x7, 6), (0x9, 10), (0x12, 13),
    (0x5, 6), (0x1e, 12), (0x14, 16),
    (0x4, 6), (0x15, 12), (0x7, 7), (0x11, 12), (0x5, 7), (0x11, 13), (0x27, 8), (0x10, 13),
    (0x23, 8), (0x1a, 16), (0x22, 8), (0x19, 16), (0x20, 8), (0x18, 16), (0xe, 10), (0x17, 16), (0xd, 10), (0x16, 16),
    (0x8, 10), (0x15, 16),
    (0x1f, 12), (0x1a, 12), (0x19, 12), (0x17, 12), (0x16, 12), (0x1f, 13), (0x1e, 13), (0x1d, 13), (0x1c, 13),
    (0x1b, 13), (0x1f, 16), (0x1e, 16), (0x1d, 16), (0x1c, 16), (0x1b, 16)]
_RUN_LEVELS = ([(0, l) for l in range(1, 41)] + [(1, l) for l in range(1, 19)] + [(2, l) for l in range(1, 6)] +
               [(3, l) for l in range(1, 5)] + [(4, l) for l in range(1, 4)] + [(5, l) for l in range(1, 4)] +
               [(6, l) for l in range(1, 4)] + [(r, l) for r in range(7, 17) for l in (1, 2)] +
               [(r, 1) for r in range(17, 32)])
assert len(_COEF_CODES) == len(_RUN_LEVELS) == 111
COEF = dict(zip(_RUN_LEVELS, _COEF_CODES))
COEF_ESCAPE = (0x1, 6)
COEF_EOB = (0x2, 2)

START_PICTURE, START_SEQUENCE, START_GOP, START_MAP, START_END = 0x00, 0xC3, 0xB8, 0xC4, 0xB7


class BitWriter:
    def __init__(self):
        self.buf = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, bits):
        assert 0 <= value < (1 << bits), (value, bits)
        self.acc = (self.acc << bits) | value
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.buf.append((self.acc >> self.n) & 0xff)
        self.acc &= (1 << self.n) - 1

    def vlc(self, cl):
        self.put(cl[0], cl[1])

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def start_code(self, code):
        self.align()
        self.buf += bytes([0, 0, 1, code])

    def tell(self):
        assert self.n == 0
        return len(self.buf)


def _put_mba(bw, incr):
    while incr > 33:
        bw.vlc(MBA_ESCAPE)
        incr -= 33
    bw.vlc(MBA[incr - 1])


def _put_motion(bw, delta, f_code):
    """One motion component: delta = vector - predictor, already wrapped to [-16f, 16f-1]."""
    f = 1 << (f_code - 1)
    if delta == 0:
        bw.vlc(MOTION[0])
        return
    a = abs(delta) + f - 1
    code, resid = a // f, a % f          # |delta| = (code-1)*f + resid + 1
    bw.vlc(MOTION[code])
    bw.put(1 if delta < 0 else 0, 1)
    if f_code > 1:
        bw.put(resid, f_code - 1)


def _wrap(delta, f_code):
    r = 16 << (f_code - 1)
    if delta < -r:
        delta += 2 * r
    elif delta > r - 1:
        delta -= 2 * r
    return delta


def _put_block(bw, levels_zz, intra, is_chroma, dc_pred):
    """levels_zz: 64 levels in zig-zag scan order.  Returns the new DC predictor (intra)."""
    start = 0
    if intra:
        dc = int(levels_zz[0])
        diff = dc - dc_pred
        size = 0 if diff == 0 else abs(diff).bit_length()
        bw.vlc((DC_CHR if is_chroma else DC_LUM)[size])
        if size:
            bw.put(diff if diff > 0 else diff + (1 << size) - 1, size)
        dc_pred = dc
        start = 1
    run = 0
    first = not intra
    for k in range(start, 64):
        lv = int(levels_zz[k])
        if lv == 0:
            run += 1
            continue
        a = abs(lv)
        if first and run == 0 and a == 1:
            bw.put(0x2 | (1 if lv < 0 else 0), 2)          # '1s' for the first coefficient
        elif (run, a) in COEF:
            bw.vlc(COEF[(run, a)])
            bw.put(1 if lv < 0 else 0, 1)
        else:
            bw.vlc(COEF_ESCAPE)
            bw.put(run, 6)
            if -127 <= lv <= 127:
                bw.put(lv & 0xff, 8)
            elif lv > 0:
                bw.put(0x00, 8)
                bw.put(lv, 8)                              # 128..255
            else:
                bw.put(0x80, 8)
                bw.put(lv + 256, 8)                        # -255..-128
        first = False
        run = 0
    bw.vlc(COEF_EOB)
    return dc_pred


def _mb_blocks(t, mbx, mby):
    """The six 8x8 blocks of a macroblock as zig-zag level vectors -- ten in a yuva stream: the four
    blocks of the A component (laid out like the luma blocks) follow Cr."""
    out = []
    for b in range(4):
        y0, x0 = mby * 16 + 8 * (b >> 1), mbx * 16 + 8 * (b & 1)
        out.append(t["coef_y"][y0:y0 + 8, x0:x0 + 8].reshape(64)[ZIGZAG])
    for k in ("coef_cb", "coef_cr"):
        out.append(t[k][mby * 8:mby * 8 + 8, mbx * 8:mbx * 8 + 8].reshape(64)[ZIGZAG])
    if "coef_a" in t:
        for b in range(4):
            y0, x0 = mby * 16 + 8 * (b >> 1), mbx * 16 + 8 * (b & 1)
            out.append(t["coef_a"][y0:y0 + 8, x0:x0 + 8].reshape(64)[ZIGZAG])
    return out


def write_picture(bw, t, cw, ch, temporal_ref, f_code=(2, 2), full_pel=(0, 0), slice_mbs=None):
    """One picture.  slice_mbs None: one slice per macroblock row; else a new slice every slice_mbs
    macroblocks in raster order (slices then start mid-row and may span rows, as MPEG-1 allows).

    yuva ("coef_a" in t; container flag `a`, decoders/jsv.js:256-259).  The reference defines no syntax
    for the fourth component (its slice loop reads six blocks, jsv.js:817-828), so this is the repo's:
    a macroblock carries four more blocks, A0..A3, placed like the luma blocks and coded like them
    (luminance DC table, own DC predictor, same AC codes), after Cr.  Which of them are coded:
    all four in an intra macroblock; otherwise a 4-bit alpha_pattern (A0 = MSB) that follows
    coded_block_pattern -- or the vectors, when the macroblock type has no pattern -- in EVERY
    non-intra macroblock.  A macroblock is skipped only if its alpha blocks are empty too."""
    ptype = t["type"]
    alpha = "coef_a" in t
    mbw, mbh = cw // 16, ch // 16
    bw.start_code(START_PICTURE)
    bw.put(temporal_ref & 1023, 10)
    bw.put(ptype, 3)
    bw.put(0xffff, 16)                                     # vbv_delay
    if ptype in (PIC_P, PIC_B):
        bw.put(full_pel[0], 1)
        bw.put(f_code[0], 3)
    if ptype == PIC_B:
        bw.put(full_pel[1], 1)
        bw.put(f_code[1], 3)
    bw.put(0, 1)                                           # extra_bit_picture
    types = {PIC_I: MBTYPE_I, PIC_P: MBTYPE_P, PIC_B: MBTYPE_B}[ptype]
    per_slice = slice_mbs or mbw
    for first in range(0, mbw * mbh, per_slice):
        last = min(first + per_slice, mbw * mbh) - 1
        bw.start_code(first // mbw + 1)
        qcur = int(t["qscale"][first])
        if qcur < 1:
            qcur = 1
        bw.put(qcur, 5)
        bw.put(0, 1)                                       # extra_bit_slice
        dc_pred = [128, 128, 128, 128]
        pmv_f = [0, 0]
        pmv_b = [0, 0]
        last_coded = first - (first % mbw) - 1            # the slice's address origin: row start - 1
        prev_intra = False
        for mb in range(first, last + 1):
            mbx, mby = mb % mbw, mb // mbw
            intra = bool(t["intra"][mb])
            blocks = _mb_blocks(t, mbx, mby)
            cbp = 0
            for b in range(6):
                if np.any(blocks[b] != 0):
                    cbp |= 1 << (5 - b)
            if intra:
                cbp = 0x3f
            apat = 0
            if alpha:
                for b in range(4):
                    if intra or np.any(blocks[6 + b] != 0):
                        apat |= 1 << (3 - b)
            mvf = [int(v) for v in t["mv_fwd"][2 * mb:2 * mb + 2]] if ptype != PIC_I else [0, 0]
            mvb = [int(v) for v in t["mv_bwd"][2 * mb:2 * mb + 2]] if ptype == PIC_B else [0, 0]
            d = int(t["mb_dir"][mb]) & 3 if ptype == PIC_B else 1
            # a P macroblock without coefficients and with a zero vector may be skipped, but never
            # the first or last one of a slice
            if ptype == PIC_P and not intra and cbp == 0 and apat == 0 and mvf == [0, 0] and first < mb < last:
                continue
            flags = 0
            if intra:
                flags = 0x01
            else:
                if ptype == PIC_P:
                    if mvf != [0, 0] or cbp == 0:
                        flags |= 0x08                      # "MC": vectors are transmitted
                if ptype == PIC_B:
                    flags |= (0x08 if d & 1 else 0) | (0x04 if d & 2 else 0)
                    if not flags:
                        flags = 0x08
                if cbp:
                    flags |= 0x02
            q = int(t["qscale"][mb])
            if (flags & 0x03) and q != qcur and q >= 1:
                flags |= 0x10
            if flags not in types:                         # e.g. P: quant without pattern does not exist
                flags &= ~0x10
            skipped = mb - last_coded - 1
            if mb == first:
                skipped = mbx                              # address increment from the row start, nothing is skipped
            elif skipped and ptype == PIC_P:
                pmv_f = [0, 0]                             # skipped P macroblocks reset the predictor
            if skipped and mb != first:
                dc_pred = [128, 128, 128, 128]
            _put_mba(bw, skipped + 1)
            bw.vlc(types[flags])
            if flags & 0x10:
                bw.put(q, 5)
                qcur = q
            if not intra and (prev_intra or True):
                dc_pred = [128, 128, 128, 128]             # a non-intra macroblock resets the DC predictors
            if ptype == PIC_P and not intra and not (flags & 0x08):
                pmv_f = [0, 0]                             # no-MC macroblock: vector and predictor are zero
                assert mvf == [0, 0]
            if intra and ptype != PIC_I:
                pmv_f = [0, 0]
                pmv_b = [0, 0]
            if flags & 0x08:
                for k in range(2):
                    v = mvf[k] >> 1 if full_pel[0] else mvf[k]
                    _put_motion(bw, _wrap(v - pmv_f[k], f_code[0]), f_code[0])
                    pmv_f[k] = v
            if flags & 0x04:
                for k in range(2):
                    v = mvb[k] >> 1 if full_pel[1] else mvb[k]
                    _put_motion(bw, _wrap(v - pmv_b[k], f_code[1]), f_code[1])
                    pmv_b[k] = v
            if (flags & 0x02) and not intra:
                bw.vlc(CBP[cbp])
            if alpha and not intra:
                bw.put(apat, 4)
            for b in range(6):
                if cbp & (1 << (5 - b)):
                    comp = 0 if b < 4 else b - 3
                    dc_pred[comp] = _put_block(bw, blocks[b], intra, b >= 4, dc_pred[comp])
            for b in range(4):
                if apat & (1 << (3 - b)):
                    dc_pred[3] = _put_block(bw, blocks[6 + b], intra, False, dc_pred[3])
            last_coded = mb
            prev_intra = intra


def write_stream(pictures, cw, ch, frame_w=None, frame_h=None, rate_idx=3, gop_starts=None,
                 qm_intra=None, qm_non_intra=None, key_map=True, f_code=(2, 2), slice_mbs=None, alpha=None, gop_qm=None):
    """pictures: tensors dicts in CODED order, each with 'display' (temporal reference inside
    its GOP).  gop_starts: indices into `pictures` where a sequence header + GOP header go.
    gop_qm: {index into `pictures`: (qm_intra, qm_non_intra)} -- matrices of that GOP's own sequence header
    instead of the stream's (a sequence header may reload them, decoders/jsv.js:540-558).
    Returns bytes."""
    frame_w, frame_h = frame_w or cw, frame_h or ch
    gop_starts = sorted(set(gop_starts or [0]))
    body = BitWriter()
    offsets = []
    rate = [0, 23.976, 24, 25, 29.97, 30, 50, 59.94, 60][rate_idx]
    frame_no = 0
    for i, t in enumerate(pictures):
        if i in gop_starts:
            offsets.append((body.tell() if body.n == 0 else None, frame_no))
            body.start_code(START_SEQUENCE)
            offsets[-1] = (len(body.buf) - 4, frame_no)
            body.put(frame_w, 12)
            body.put(frame_h, 12)
            body.put(1, 4)                                 # aspect
            body.put(rate_idx, 4)
            body.put(0x3ffff, 18)                          # bit rate (variable)
            body.put(1, 1)
            body.put(1, 10)                                # vbv buffer size: 16 KiB look-ahead
            body.put(0, 1)
            for qm in (gop_qm[i] if gop_qm and i in gop_qm else (qm_intra, qm_non_intra)):
                if qm is None:
                    body.put(0, 1)
                else:
                    body.put(1, 1)
                    for v in np.asarray(qm, dtype=np.uint8).reshape(64)[ZIGZAG]:
                        body.put(int(v), 8)
            body.start_code(START_GOP)
            sec = int(frame_no / rate)
            body.put(0, 1)
            body.put(sec // 3600, 5)
            body.put((sec // 60) % 60, 6)
            body.put(1, 1)
            body.put(sec % 60, 6)
            body.put(int(frame_no - sec * rate) & 63, 6)
            body.put(1, 1)                                 # closed_gop
            body.put(0, 1)                                 # broken_link
        write_picture(body, t, cw, ch, t.get("display", 0), f_code=f_code, slice_mbs=slice_mbs)
        frame_no += 1
    body.start_code(START_END)
    body.buf += bytes(8)                                   # tail so the last start-code scan terminates
    # container header (decoders/jsv.js:248-268): magic, w, h, 0, a=0, duration in 1/100 s, key map
    hdr = BitWriter()
    hdr.put(0x6A73, 16)
    hdr.put(frame_w, 16)
    hdr.put(frame_h, 16)
    hdr.put(0, 16)
    if alpha is None:
        alpha = any("coef_a" in t for t in pictures)
    hdr.put(1 if alpha else 0, 1)                          # `a`: yuva (decoders/jsv.js:256)
    hdr.put(int(round(len(pictures) / rate * 100)), 23)
    n_hdr = 11 + (8 + 8 * len(offsets) if key_map else 0)
    if key_map:
        hdr.buf += bytes([0, 0, 1, START_MAP])
        hdr.put(len(offsets), 32)
        for off, fno in offsets:
            hdr.put(off + n_hdr, 32)                       # absolute byte offset of the sequence header
            sec = int(fno / rate)
            tc = ((sec // 3600) << 26) | (((sec // 60) % 60) << 20) | (1 << 19) | ((sec % 60) << 13) | ((int(fno - sec * rate) & 63) << 7)
            hdr.put(tc & 0xffffffff, 32)
    assert len(hdr.buf) == n_hdr, (len(hdr.buf), n_hdr)
    return bytes(hdr.buf) + bytes(body.buf), [o + n_hdr for o, _ in offsets]


def merge_gops(streams, frame_w, frame_h, rate_idx=3, alpha=False):
    """One stream from several single-GOP streams (each written by write_stream with key_map=True and gop_starts=[0],
    same picture size and rate): the bodies back to back, the GOP headers' time codes counted on, one container header
    with a key map over all of them.  Lets the GOPs of a long stream be written by parallel processes."""
    rate = [0, 23.976, 24, 25, 29.97, 30, 50, 59.94, 60][rate_idx]
    bodies, offsets, frame_no = [], [], 0
    at = 0
    for i, data in enumerate(streams):
        data = bytes(data)
        if data[:2] != b"\x6a\x73" or data[11:15] != bytes([0, 0, 1, START_MAP]) or int.from_bytes(data[15:19], "big") != 1:
            raise ValueError("stream %d is not a single-GOP stream with a key map" % i)
        body = bytearray(data[27:])
        tail = bytes([0, 0, 1, START_END]) + bytes(8)
        if bytes(body[-12:]) != tail:
            raise ValueError("stream %d does not end with the end code" % i)
        if i + 1 < len(streams):
            del body[-12:]
        g = bytes(body[:256]).find(bytes([0, 0, 1, START_GOP]))
        if g < 0:
            raise ValueError("stream %d: no GOP header behind the sequence header" % i)
        sec = int(frame_no / rate)
        tc = ((sec // 3600) << 26) | (((sec // 60) % 60) << 20) | (1 << 19) | ((sec % 60) << 13) | ((int(frame_no - sec * rate) & 63) << 7) | (1 << 6)
        body[g + 4:g + 8] = tc.to_bytes(4, "big")             # time code, closed_gop = 1, broken_link = 0, padding
        offsets.append((at, frame_no))
        at += len(body)
        bodies.append(bytes(body))
        # pictures of this GOP: picture start codes 00 00 01 00
        frame_no += bytes(body).count(bytes([0, 0, 1, START_PICTURE]))
    hdr = BitWriter()
    hdr.put(0x6A73, 16)
    hdr.put(frame_w, 16)
    hdr.put(frame_h, 16)
    hdr.put(0, 16)
    hdr.put(1 if alpha else 0, 1)
    hdr.put(int(round(frame_no / rate * 100)), 23)
    n_hdr = 11 + 8 + 8 * len(offsets)
    hdr.buf += bytes([0, 0, 1, START_MAP])
    hdr.put(len(offsets), 32)
    for off, fno in offsets:
        hdr.put(off + n_hdr, 32)
        sec = int(fno / rate)
        tc = ((sec // 3600) << 26) | (((sec // 60) % 60) << 20) | (1 << 19) | ((sec % 60) << 13) | ((int(fno - sec * rate) & 63) << 7)
        hdr.put(tc & 0xffffffff, 32)
    assert len(hdr.buf) == n_hdr
    return bytes(hdr.buf) + b"".join(bodies), [o + n_hdr for o, _ in offsets]
