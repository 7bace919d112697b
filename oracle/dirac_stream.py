"""dirac_stream.py -- a Dirac core-syntax front end for the oracle.  TEST INFRASTRUCTURE ONLY
(see oracle/schro_oracle.h): it exists so that a REAL stream -- the reference's own
testsuite/test_stream.drc (BASELINE config 2) -- can be pushed through the oracle's pixel path
(inverse wavelet, OBMC, combine) and the decoded pictures compared with what the reference
decoder produced for the same stream (schro_frame_md5 of its output frames).
It also yields real-stream coefficients and motion vectors as inputs for the GPU parity tests.

Restated from dschleef/schroedinger 1.0.11.1 (pure Python: the stream is 320x240):
  parse units, sequence header      schrodecoder.c:2214-2374, schrobitstream.h:39-48
  picture header / parse            schrodecoder.c:2376-2403, 1389-1470
  prediction parameters             schrodecoder.c:2405-2515, schroparams.c:166-198, 210-223
  block data: split, mode, DC, MVs  schrodecoder.c:2532-2815, schromotion.c:163-430
  transform parameters / data       schrodecoder.c:2817-2874, 2939-2988
  sub-bands, codeblocks, coefficients
                                    schrodecoder.c:3005-3075 (generic line decoder), 3280-3352,
                                    3452-3638; schroparams.c:319-368
  binary arithmetic decoder         schroarith.c:211-239, 642-670, schroarith.h:147-210
  schro_frame_md5                   schroframe.c:1715-1852
Tables that are numbers of the Dirac specification (arithmetic-coder probability LUT,
quantiser factors and offsets) are not restated here: quantisers come from the C oracle's
formulas (pinned against the reference's constants), the LUT from tests/golden/arith_lut.json
(the reference's 256 numbers, extracted by tests/golden/make_stream_golden.py).
Not handled (not in the stream): low delay, VLC-coded core syntax, global motion, interlaced
coding, custom signal ranges above 8 bits.
"""
import numpy as np

MV_DTYPE = np.dtype([("flags", "<u4"), ("metric", "<u4"), ("chroma_metric", "<u4"), ("v", "<i2", (4,))])

# contexts: own numbering; `follow` is the continuation-bin chain of the reference's next_list
(CTX_ZERO_CODEBLOCK, CTX_Q_CONT, CTX_Q_VALUE, CTX_Q_SIGN, CTX_ZPZN_F1, CTX_ZPNN_F1, CTX_ZP_F2, CTX_ZP_F3,
 CTX_ZP_F4, CTX_ZP_F5, CTX_ZP_F6, CTX_NPZN_F1, CTX_NPNN_F1, CTX_NP_F2, CTX_NP_F3, CTX_NP_F4, CTX_NP_F5,
 CTX_NP_F6, CTX_SIGN_POS, CTX_SIGN_NEG, CTX_SIGN_ZERO, CTX_COEFF_DATA, CTX_SB_F1, CTX_SB_F2, CTX_SB_DATA,
 CTX_MODE_REF1, CTX_MODE_REF2, CTX_GLOBAL, CTX_DC_CONT1, CTX_DC_CONT2, CTX_DC_VALUE, CTX_DC_SIGN,
 CTX_MV_CONT1, CTX_MV_CONT2, CTX_MV_CONT3, CTX_MV_CONT4, CTX_MV_CONT5, CTX_MV_VALUE, CTX_MV_SIGN, CTX_LAST) = range(40)
FOLLOW = {CTX_Q_CONT: CTX_Q_CONT,
          CTX_ZPZN_F1: CTX_ZP_F2, CTX_ZPNN_F1: CTX_ZP_F2, CTX_ZP_F2: CTX_ZP_F3, CTX_ZP_F3: CTX_ZP_F4,
          CTX_ZP_F4: CTX_ZP_F5, CTX_ZP_F5: CTX_ZP_F6, CTX_ZP_F6: CTX_ZP_F6,
          CTX_NPZN_F1: CTX_NP_F2, CTX_NPNN_F1: CTX_NP_F2, CTX_NP_F2: CTX_NP_F3, CTX_NP_F3: CTX_NP_F4,
          CTX_NP_F4: CTX_NP_F5, CTX_NP_F5: CTX_NP_F6, CTX_NP_F6: CTX_NP_F6,
          CTX_SB_F1: CTX_SB_F2, CTX_SB_F2: CTX_SB_F2,
          CTX_DC_CONT1: CTX_DC_CONT2, CTX_DC_CONT2: CTX_DC_CONT2,
          CTX_MV_CONT1: CTX_MV_CONT2, CTX_MV_CONT2: CTX_MV_CONT3, CTX_MV_CONT3: CTX_MV_CONT4,
          CTX_MV_CONT4: CTX_MV_CONT5, CTX_MV_CONT5: CTX_MV_CONT5}
SUBBAND_POSITION = [0, 1, 2, 3, 5, 6, 7, 9, 10, 11, 13, 14, 15, 17, 18, 19, 21, 22, 23, 25, 26, 27]
BLOCK_PARAMS = {1: (8, 8, 4, 4), 2: (12, 12, 8, 8), 3: (16, 16, 12, 12), 4: (24, 24, 16, 16)}   # schroparams.c:192-198


class Bits:
    """schrounpack.c by bit position; the guard bit of the parse layer is 1 (schrodecoder.c:936)."""

    def __init__(self, data):
        self.d, self.p = data, 0

    def bit(self):
        B = self.p >> 3
        v = (self.d[B] >> (7 - (self.p & 7))) & 1 if B < len(self.d) else 1
        self.p += 1
        return v

    def bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v

    def uint(self):
        c = v = 0
        while not self.bit():
            c += 1
            v = (v << 1) | self.bit()
        return (1 << c) - 1 + v

    def sint(self):
        v = self.uint()
        return -v if v and self.bit() else v

    def sync(self):
        self.p = (self.p + 7) & ~7

    def take(self, nbytes):
        assert self.p & 7 == 0
        b = self.d[self.p >> 3:(self.p >> 3) + nbytes]
        self.p += 8 * nbytes
        return b


class Arith:
    """schroarith.h:147-196: 16-bit probabilities of a 0, range kept in the top half of 32 bits."""

    def __init__(self, data, lut):
        n = len(data)
        self.d, self.n, self.lut = data, n, lut
        self.range = 0xffff0000
        self.code = 0
        for k in range(4):
            self.code = (self.code << 8) | (data[k] if k < n else 0xff)
        self.offset, self.cntr = 3, 16
        self.prob = [0x8000] * CTX_LAST

    def bit(self, ctx):
        rng, code, d, n = self.range, self.code, self.d, self.n
        while rng <= 0x40000000:
            rng = (rng << 1) & 0xffffffff
            code = (code << 1) & 0xffffffff
            self.cntr -= 1
            if not self.cntr:
                self.offset += 1
                code |= (d[self.offset] if self.offset < n else 0xff) << 8
                self.offset += 1
                code |= d[self.offset] if self.offset < n else 0xff
                self.cntr = 16
        p = self.prob[ctx]
        rxp = ((rng >> 16) * p) & 0xffff0000
        if code >= rxp:
            self.prob[ctx] = p - self.lut[p >> 8]
            self.code, self.range = code - rxp, rng - rxp
            return 1
        self.prob[ctx] = p + self.lut[255 - (p >> 8)]
        self.code, self.range = code, rxp
        return 0

    def uint(self, cont, value):
        bits = 1
        while not self.bit(cont):
            bits = (bits << 1) | self.bit(value)
            cont = FOLLOW[cont]
        return bits - 1

    def sint(self, cont, value, sign):
        bits, count = 1, 0
        while not self.bit(cont):
            bits = (bits << 1) | self.bit(value)
            cont = FOLLOW[cont]
            count += 1
            if count == 30:
                break
        v = bits - 1
        return -v if v and self.bit(sign) else v


def parse_units(data):
    """(parse code, payload) of every unit (schroparse.c: 'BBCD', code, next, previous offsets)."""
    pos, out = 0, []
    while pos + 13 <= len(data) and data[pos:pos + 4] == b"BBCD":
        code = data[pos + 4]
        nxt = int.from_bytes(data[pos + 5:pos + 9], "big")
        end = pos + nxt if nxt else len(data)
        out.append((code, data[pos + 13:end]))
        if not nxt:
            break
        pos = end
    return out


def parse_sequence_header(payload):
    b = Bits(payload)
    fmt = dict(major=b.uint(), minor=b.uint(), profile=b.uint(), level=b.uint())
    index = b.uint()
    assert index == 0, "only the custom base format (640x480 4:2:0 8 bit progressive) is tabulated here"
    fmt.update(width=640, height=480, chroma_format=2, interlaced=0, luma_offset=0, luma_excursion=255)
    if b.bit():
        fmt["width"], fmt["height"] = b.uint(), b.uint()
    if b.bit():
        fmt["chroma_format"] = b.uint()          # 0 4:4:4, 1 4:2:2, 2 4:2:0
    if b.bit():
        fmt["interlaced"] = b.uint()
    if b.bit():
        if b.uint() == 0:
            b.uint(), b.uint()
    if b.bit():
        if b.uint() == 0:
            b.uint(), b.uint()
    if b.bit():
        b.uint(), b.uint(), b.uint(), b.uint()
    if b.bit():
        idx = b.uint()
        assert idx in (1, 2), "8-bit signal ranges only"
    if b.bit():
        if b.uint() == 0:
            for _ in range(3):
                if b.bit():
                    b.uint()
    fmt["interlaced_coding"] = b.uint()
    assert fmt["interlaced_coding"] == 0
    fmt["h_shift"] = 0 if fmt["chroma_format"] == 0 else 1
    fmt["v_shift"] = 1 if fmt["chroma_format"] == 2 else 0
    return fmt


class Picture:
    pass


def subband_view(plane, depth, index):
    """schro_subband_get_frame_data on a numpy plane of the iwt size."""
    position = SUBBAND_POSITION[index]
    shift = depth - (position >> 2)
    h, w = plane.shape[0] >> shift, plane.shape[1] >> shift
    rows = plane[(1 << shift) >> 1 if position & 2 else 0::1 << shift]
    return rows[:h, w if position & 1 else 0:][:, :w]


def round_up_pow2(x, p):
    return (x + (1 << p) - 1) & ~((1 << p) - 1)


class Decoder:
    def __init__(self, lut, quant_factor, quant_offset_1_2, quant_offset_3_8):
        self.lut = lut
        self.qf, self.qo12, self.qo38 = quant_factor, quant_offset_1_2, quant_offset_3_8
        self.fmt = None
        self.compat_quant_offset = False        # SchroDecoderInstance.compat_quant_offset

    # ---- parse ---------------------------------------------------------------------------
    def parse_picture(self, code, payload):
        fmt = self.fmt
        pic = Picture()
        pic.num_refs = code & 3
        pic.is_ref = (code & 0x0c) == 0x0c
        assert (code & 0x88) == 0x08 and (code & 0x48) == 0x08, "core syntax with arithmetic coding only"
        b = Bits(payload)
        b.sync()
        pic.number = b.bits(32)
        pic.refs = [pic.number + b.sint() for _ in range(pic.num_refs)]
        pic.retired = pic.number + b.sint() if pic.is_ref else None
        pic.width, pic.height = fmt["width"], fmt["height"]
        pic.cw = -(-pic.width >> fmt["h_shift"])
        pic.ch = -(-pic.height >> fmt["v_shift"])
        if pic.num_refs:
            b.sync()
            index = b.uint()
            if index == 0:
                pic.xblen, pic.yblen, pic.xbsep, pic.ybsep = b.uint(), b.uint(), b.uint(), b.uint()
            else:
                pic.xblen, pic.yblen, pic.xbsep, pic.ybsep = BLOCK_PARAMS[index]
            pic.mv_precision = b.uint()
            assert not b.bit(), "global motion"
            assert b.uint() == 0, "picture prediction mode"
            pic.weight_bits, pic.weight1, pic.weight2 = 1, 1, 1
            if b.bit():
                pic.weight_bits = b.uint()
                pic.weight1 = b.sint()
                if pic.num_refs > 1:
                    pic.weight2 = b.sint()
            pic.x_num_blocks = 4 * -(-pic.width // (4 * pic.xbsep))
            pic.y_num_blocks = 4 * -(-pic.height // (4 * pic.ybsep))
            b.sync()
            pic.motion_buffers = []
            for i in range(9):
                if pic.num_refs < 2 and i in (4, 5):
                    pic.motion_buffers.append(None)
                    continue
                length = b.uint()
                b.sync()
                pic.motion_buffers.append(b.take(length))
        b.sync()
        pic.zero_residual = bool(b.bit()) if pic.num_refs else False
        if not pic.zero_residual:
            pic.wavelet = b.uint()
            pic.depth = b.uint()
            pic.horiz_cb = [1] * (pic.depth + 1)
            pic.vert_cb = [1] * (pic.depth + 1)
            pic.cb_mode = 0
            if b.bit():
                for i in range(pic.depth + 1):
                    pic.horiz_cb[i], pic.vert_cb[i] = b.uint(), b.uint()
                pic.cb_mode = b.uint()
            pic.iwt = [(round_up_pow2(pic.height, pic.depth), round_up_pow2(pic.width, pic.depth)),
                       (round_up_pow2(pic.ch, pic.depth), round_up_pow2(pic.cw, pic.depth))]
            b.sync()
            pic.subbands = []
            for comp in range(3):
                row = []
                for i in range(1 + 3 * pic.depth):
                    b.sync()
                    length = b.uint()
                    if length == 0:
                        b.sync()
                        row.append((0, b""))
                    else:
                        q = b.uint()
                        assert 0 <= q <= 60
                        b.sync()
                        row.append((q, b.take(length)))
                pic.subbands.append(row)
        return pic

    # ---- block data ----------------------------------------------------------------------
    def decode_block_data(self, pic):
        nx, ny = pic.x_num_blocks, pic.y_num_blocks
        mode = np.zeros((ny, nx), np.int32)
        split = np.zeros((ny, nx), np.int32)
        vec = np.zeros((ny, nx, 4), np.int32)          # dx0 dx1 dy0 dy1 | dc0 dc1 dc2 -
        ar = [Arith(buf, self.lut) if buf is not None else None for buf in pic.motion_buffers]
        (A_SB, A_MODE, A_X1, A_Y1, A_X2, A_Y2, A_DC0, A_DC1, A_DC2) = range(9)

        def short(v):
            return ((v + 32768) & 0xffff) - 32768

        def mode_pred(x, y):
            if y == 0:
                return 0 if x == 0 else mode[0, x - 1]
            if x == 0:
                return mode[y - 1, 0]
            a, b_, c = mode[y, x - 1], mode[y - 1, x], mode[y - 1, x - 1]
            return (a & b_) | (b_ & c) | (c & a)

        def dc_pred(x, y, i):
            s = n = 0
            for (xx, yy, ok) in ((x - 1, y, x > 0), (x, y - 1, y > 0), (x - 1, y - 1, x > 0 and y > 0)):
                if ok and mode[yy, xx] == 0:
                    s += vec[yy, xx, i]
                    n += 1
            if n == 0:
                return 0
            if n == 1:
                return short(s)
            if n == 2:
                return (s + 1) >> 1
            return ((s + 1) * 21845 + 10922) >> 16         # schro_divide3

        def median3(a, b_, c):
            return sorted((a, b_, c))[1]

        def mv_pred(x, y, m):
            vx, vy = [], []
            for (xx, yy, ok) in ((x - 1, y, x > 0), (x, y - 1, y > 0), (x - 1, y - 1, x > 0 and y > 0)):
                if ok and (mode[yy, xx] & m):
                    vx.append(vec[yy, xx, m - 1])
                    vy.append(vec[yy, xx, 2 + m - 1])
            if not vx:
                return 0, 0
            if len(vx) == 1:
                return vx[0], vy[0]
            if len(vx) == 2:
                return (vx[0] + vx[1] + 1) >> 1, (vy[0] + vy[1] + 1) >> 1
            return median3(*vx), median3(*vy)

        def unit(x, y):
            m = mode_pred(x, y)
            m ^= ar[A_MODE].bit(CTX_MODE_REF1)
            if pic.num_refs > 1:
                m ^= ar[A_MODE].bit(CTX_MODE_REF2) << 1
            mode[y, x] = m
            if m == 0:
                for i, a in enumerate((A_DC0, A_DC1, A_DC2)):
                    vec[y, x, i] = short(dc_pred(x, y, i) + ar[a].sint(CTX_DC_CONT1, CTX_DC_VALUE, CTX_DC_SIGN))
                return
            if m & 1:
                px, py = mv_pred(x, y, 1)
                vec[y, x, 0] = short(px + ar[A_X1].sint(CTX_MV_CONT1, CTX_MV_VALUE, CTX_MV_SIGN))
                vec[y, x, 2] = short(py + ar[A_Y1].sint(CTX_MV_CONT1, CTX_MV_VALUE, CTX_MV_SIGN))
            if m & 2:
                px, py = mv_pred(x, y, 2)
                vec[y, x, 1] = short(px + ar[A_X2].sint(CTX_MV_CONT1, CTX_MV_VALUE, CTX_MV_SIGN))
                vec[y, x, 3] = short(py + ar[A_Y2].sint(CTX_MV_CONT1, CTX_MV_VALUE, CTX_MV_SIGN))

        def copy(ys, xs, y0, x0):
            mode[ys, xs] = mode[y0, x0]
            split[ys, xs] = split[y0, x0]
            vec[ys, xs] = vec[y0, x0]

        for j in range(0, ny, 4):
            for i in range(0, nx, 4):
                if j == 0:
                    sp = 0 if i == 0 else split[0, i - 4]
                elif i == 0:
                    sp = split[j - 4, 0]
                else:
                    sp = (split[j - 4, i] + split[j, i - 4] + split[j - 4, i - 4] + 1) // 3
                s = (sp + ar[A_SB].uint(CTX_SB_F1, CTX_SB_DATA)) % 3
                split[j, i] = s
                if s == 0:
                    unit(i, j)
                    copy(slice(j, j + 4), slice(i, i + 4), j, i)
                elif s == 1:
                    for (yy, xx) in ((j, i), (j, i + 2), (j + 2, i), (j + 2, i + 2)):
                        split[yy, xx] = 1
                        unit(xx, yy)
                        copy(slice(yy, yy + 2), slice(xx, xx + 2), yy, xx)
                else:
                    for l in range(4):
                        for k in range(4):
                            split[j + l, i + k] = 2
                            unit(i + k, j + l)
        mv = np.zeros(ny * nx, MV_DTYPE)
        mv["flags"] = (mode | (split << 3)).astype(np.uint32).ravel()
        mv["v"] = vec.astype(np.int16).reshape(-1, 4)
        return mv

    # ---- coefficients ------------------------------------------------------------------------
    def decode_coefficients(self, pic, quantised=False):
        """Three int16 planes of the iwt size in the interleaved sub-band layout.
        quantised: stop BEFORE dequantisation and intra DC prediction (what a host decoder hands
        to schro_hip_dequant_batch): int32 planes of quantised values, and self.codeblocks[comp]
        = [(sub-band index, xmin, ymin, xmax, ymax, zero codeblock?, quantiser index)]."""
        intra = pic.num_refs == 0
        planes = []
        self.codeblocks = [[], [], []]
        for comp in range(3):
            h, w = pic.iwt[1 if comp else 0]
            plane = np.zeros((h, w), np.int32)
            for index in range(1 + 3 * pic.depth):
                q0, buf = pic.subbands[comp][index]
                band = subband_view(plane, pic.depth, index)
                if not buf:
                    # schro_decoder_decode_subband: subband_length == 0 -> zero_block over the sub-band
                    self.codeblocks[comp].append((index, 0, 0, band.shape[1], band.shape[0], True, q0))
                    continue
                position = SUBBAND_POSITION[index]
                parent = subband_view(plane, pic.depth, index - 3) if position >= 4 else None
                self.decode_subband(pic, band, parent, position, index, q0, buf, intra, quantised, self.codeblocks[comp])
                if position == 0 and intra and not quantised:
                    dc_predict_s16(band)
            planes.append(plane.astype(np.int32 if quantised else np.int16))
        return planes

    def decode_subband(self, pic, band, parent, position, index, quant_index, buf, intra, quantised=False,
                       records=None):
        ar = Arith(buf, self.lut)
        bh, bw = band.shape
        level = 0 if position == 0 else (position >> 2) + 1
        ncx, ncy = pic.horiz_cb[level], pic.vert_cb[level]
        zero_flags = ncx > 1 or ncy > 1
        # schro_decoder_setup_codeblocks: per-codeblock quantiser offsets, except that streams of
        # early encoders do not code one for sub-bands of a single codeblock ("compatibility
        # mode", found by schro_decoder_test_quant_offset_compat: a first offset that leaves
        # 0..60 switches it on for the rest of the stream)
        quant_delta = pic.cb_mode == 1
        single = ncx == 1 and ncy == 1
        if quant_delta and single and self.compat_quant_offset:
            quant_delta = False
        if quant_delta and single and index == 0:
            import copy
            peek = copy.copy(ar)
            peek.prob = list(ar.prob)
            q = quant_index + peek.sint(CTX_Q_CONT, CTX_Q_VALUE, CTX_Q_SIGN)
            if q < 0 or q > 60:
                self.compat_quant_offset = True
                quant_delta = False
        horiz, vert = (position & 3) == 2, (position & 3) == 1
        cbw = bw // ncx
        inc = bw - ncx * cbw
        rows = [[int(v) for v in r] for r in band]          # Python ints: the hot loop
        prow = [[int(v) for v in r] for r in parent] if parent is not None else None
        qoff = self.qo12 if intra else self.qo38
        bit = ar.bit
        for cy in range(ncy):
            ymin, ymax = (bh * cy) // ncy, (bh * (cy + 1)) // ncy
            xmin = acc = 0
            for cx in range(ncx):
                x0 = xmin
                xmin += cbw
                acc += inc
                if acc >= ncx:
                    acc -= ncx
                    xmin += 1
                x1 = xmin
                if zero_flags and bit(CTX_ZERO_CODEBLOCK):
                    if records is not None:
                        records.append((index, x0, ymin, x1, ymax, True, quant_index))
                    continue                                  # the plane starts as zeros
                if quant_delta:
                    quant_index = min(max(quant_index + ar.sint(CTX_Q_CONT, CTX_Q_VALUE, CTX_Q_SIGN), 0), 60)
                if records is not None:
                    records.append((index, x0, ymin, x1, ymax, False, quant_index))
                factor, offset = self.qf[quant_index], qoff[quant_index]
                for j in range(ymin, ymax):
                    line = rows[j]
                    prev = rows[j - 1] if j > 0 else None
                    par = prow[j >> 1] if prow is not None else None
                    for i in range(x0, x1):
                        nhood = 0
                        if prev is not None:
                            nhood = prev[i]
                            if i > 0:
                                nhood |= prev[i - 1]
                        if i > 0:
                            nhood |= line[i - 1]
                        if par is not None and par[i >> 1] != 0:
                            cont = CTX_NPNN_F1 if nhood else CTX_NPZN_F1
                        else:
                            cont = CTX_ZPNN_F1 if nhood else CTX_ZPZN_F1
                        # _schro_arith_decode_uint
                        bits = 1
                        while not bit(cont):
                            bits = (bits << 1) | bit(CTX_COEFF_DATA)
                            cont = FOLLOW[cont]
                        v = bits - 1
                        if v:
                            pv = 0
                            if horiz:
                                pv = line[i - 1] if i > 0 else 0
                            elif vert:
                                pv = prev[i] if prev is not None else 0
                            sign = CTX_SIGN_NEG if pv < 0 else (CTX_SIGN_POS if pv > 0 else CTX_SIGN_ZERO)
                            # quantised: keep the decoded value itself.  The contexts only look at
                            # whether neighbours / parent are zero and at a neighbour's sign, which
                            # dequantisation preserves (factor >= 4: a non-zero value stays non-zero)
                            if not quantised:
                                v = (offset + factor * v + 2) >> 2
                            if bit(sign):
                                v = -v
                            line[i] = ((v + 32768) & 0xffff) - 32768      # stored as int16_t
        band[...] = np.array(rows, np.int32)


def dc_predict_s16(band):
    """schro_decoder_subband_dc_predict, schrodecoder.c:3219-3247 (in place, int16 stores)."""
    h, w = band.shape
    x = [[int(v) for v in r] for r in band]

    def short(v):
        return ((v + 32768) & 0xffff) - 32768
    for i in range(1, w):
        x[0][i] = short(x[0][i] + x[0][i - 1])
    for j in range(1, h):
        x[j][0] = short(x[j][0] + x[j - 1][0])
        for i in range(1, w):
            s = x[j][i - 1] + x[j - 1][i] + x[j - 1][i - 1] + 1
            x[j][i] = short(x[j][i] + ((s * 21845 + 10922) >> 16))
    band[...] = np.array(x, np.int32)


# ---- schro_frame_md5 ---------------------------------------------------------------------
_S = [7, 12, 17, 22] * 4 + [5, 9, 14, 20] * 4 + [4, 11, 16, 23] * 4 + [6, 10, 15, 21] * 4
_K = [int(abs(__import__("math").sin(i + 1)) * 2 ** 32) & 0xffffffff for i in range(64)]


def _md5_block(state, words):
    a, b, c, d = state
    for i in range(64):
        if i < 16:
            f, g = (b & c) | (~b & d), i
        elif i < 32:
            f, g = (d & b) | (~d & c), (5 * i + 1) % 16
        elif i < 48:
            f, g = b ^ c ^ d, (3 * i + 5) % 16
        else:
            f, g = c ^ (b | (~d & 0xffffffff)), (7 * i) % 16
        f = (f + a + _K[i] + words[g]) & 0xffffffff
        a, d, c = d, c, b
        b = (b + ((f << _S[i]) | (f >> (32 - _S[i])))) & 0xffffffff
    return [(state[0] + a) & 0xffffffff, (state[1] + b) & 0xffffffff,
            (state[2] + c) & 0xffffffff, (state[3] + d) & 0xffffffff]


def frame_md5(planes):
    """schro_frame_md5: the MD5 compression function over every 64-byte piece of every row
    (the last piece of a row zero-padded), no length block; returns the four state words as
    the reference's testsuite prints them (%08x each)."""
    state = [0x67452301, 0xefcdab89, 0x98badcfe, 0x10325476]
    for p in planes:
        h, w = p.shape
        for y in range(h):
            row = p[y].tobytes()
            for x in range(0, w, 64):
                chunk = row[x:x + 64].ljust(64, b"\0")
                state = _md5_block(state, np.frombuffer(chunk, "<u4").tolist())
    return "".join("%08x" % s for s in state)
