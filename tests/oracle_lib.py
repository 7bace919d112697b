"""ctypes binding of oracle/libschro_oracle.so (the CPU checker).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None
_REFDRV = None
_REFORC = None


def build_oracle():
    """(Re)build the oracle; also _ref/ when /root/reference is present."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        # (SCHRO_ORACLE_LIB: tests/test_sanitizers.py points the oracle's own tests at its sanitizer build)
        path = os.environ.get("SCHRO_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libschro_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        vp, i = C.c_void_p, C.c_int
        for name in ("oracle_iiwt_2d", "oracle_iwt_2d"):
            getattr(L, name).argtypes = [vp, i, i, i, i, i]
            getattr(L, name).restype = i
        for name in ("oracle_inverse_iwt_component", "oracle_forward_iwt_component"):
            getattr(L, name).argtypes = [vp, i, i, i, i, i, i]
            getattr(L, name).restype = i
        L.oracle_upcomp_new.argtypes = [i, i, i]
        L.oracle_upcomp_new.restype = vp
        L.oracle_upcomp_free.argtypes = [vp]
        L.oracle_upcomp_set_plane0.argtypes = [vp, vp, i]
        L.oracle_upcomp_edgeextend.argtypes = [vp]
        L.oracle_upcomp_upsample.argtypes = [vp]
        L.oracle_upcomp_get_plane.argtypes = [vp, i, vp, i]
        L.oracle_upcomp_get.argtypes = [vp, i, i, i]
        L.oracle_upcomp_get.restype = i
        L.oracle_convert_u8_from_signed.argtypes = [vp, i, vp, i, i, i, i]
        L.oracle_motion_render_u8.argtypes = [vp, vp, i, vp, vp, vp, i, i, vp, i, vp, i, i, i]
        L.oracle_motion_render_u8.restype = i
        L.oracle_pack_u8.argtypes = [vp, i, i, i, i, vp]
        L.oracle_pack_u8.restype = i
        L.oracle_pack_v210.argtypes = [vp, i, i, i, vp, i]
        L.oracle_pack_v210.restype = i
        L.oracle_pack_wide.argtypes = [vp, i, i, i, i, vp, i]
        L.oracle_pack_wide.restype = i
        L.oracle_shift_right.argtypes = [vp, i, i, i, i, i]
        L.oracle_shift_right.restype = i
        L.oracle_lowdelay_arith.argtypes = [C.POINTER(LowDelayParams), i]
        L.oracle_lowdelay_arith.restype = i
        L.oracle_lowdelay_decode.argtypes = [vp, C.c_int64, C.c_void_p * 3, C.c_int * 3, C.POINTER(LowDelayParams), i]
        L.oracle_lowdelay_decode.restype = i
        L.oracle_lowdelay_write.argtypes = [vp, C.c_int64, C.c_void_p * 3, C.c_int * 3, C.POINTER(LowDelayParams), i,
                                            vp, i, i]
        L.oracle_lowdelay_write.restype = i
        L.oracle_dc_predict.argtypes = [vp, i, i, i, i]
        L.oracle_dc_predict.restype = None
        L.oracle_quant_factor.argtypes = [i]
        L.oracle_quant_factor.restype = C.c_uint32
        L.oracle_quant_offset_1_2.argtypes = [i]
        L.oracle_quant_offset_1_2.restype = C.c_uint32
        L.oracle_quant_offset_3_8.argtypes = [i]
        L.oracle_quant_offset_3_8.restype = i
        L.oracle_dequant_codeblock.argtypes = [vp, i, i, vp, i, i, i, i, i, i]
        L.oracle_dequant_codeblock.restype = None
        L.oracle_dequantise_var_s16.argtypes = [C.c_int16, i, i]
        L.oracle_dequantise_var_s16.restype = C.c_int16
        _LIB = L
    return _LIB


class LowDelayParams(C.Structure):
    _fields_ = [("transform_depth", C.c_int),
                ("iwt_luma_width", C.c_int), ("iwt_luma_height", C.c_int),
                ("iwt_chroma_width", C.c_int), ("iwt_chroma_height", C.c_int),
                ("n_horiz_slices", C.c_int), ("n_vert_slices", C.c_int),
                ("slice_bytes_num", C.c_int), ("slice_bytes_denom", C.c_int),
                ("quant_matrix", C.c_int * 19)]


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libschro_refdrv.so"))


def refdrv():
    global _REFDRV
    if _REFDRV is None:
        L = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libschro_refdrv.so"))
        L.refdrv_iiwt_2d.argtypes = [C.c_void_p] + [C.c_int] * 5
        L.refdrv_iiwt_2d.restype = C.c_int
        _REFDRV = L
    return _REFDRV


def reforc():
    """The reference's own kernels (schroorc-dist.c compiled unmodified)."""
    global _REFORC
    if _REFORC is None:
        _REFORC = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libschroorc_ref.so"))
    return _REFORC


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _bpp(a):
    assert a.dtype in (np.int16, np.int32)
    return a.dtype.itemsize


# ---- wavelet ---------------------------------------------------------------

def iiwt_2d(a, filt):
    """One inverse level, returns a new array."""
    a = np.ascontiguousarray(a).copy()
    h, w = a.shape
    r = lib().oracle_iiwt_2d(_ptr(a), a.strides[0], w, h, filt, _bpp(a))
    assert r == 0
    return a


def iwt_2d(a, filt):
    a = np.ascontiguousarray(a).copy()
    h, w = a.shape
    r = lib().oracle_iwt_2d(_ptr(a), a.strides[0], w, h, filt, _bpp(a))
    assert r == 0
    return a


def inverse_iwt(a, depth, filt):
    a = np.ascontiguousarray(a).copy()
    h, w = a.shape
    r = lib().oracle_inverse_iwt_component(_ptr(a), a.strides[0], w, h, depth, filt, _bpp(a))
    assert r == 0
    return a


def forward_iwt(a, depth, filt):
    a = np.ascontiguousarray(a).copy()
    h, w = a.shape
    r = lib().oracle_forward_iwt_component(_ptr(a), a.strides[0], w, h, depth, filt, _bpp(a))
    assert r == 0
    return a


def refdrv_iiwt_2d(a, filt):
    a = np.ascontiguousarray(a).copy()
    h, w = a.shape
    r = refdrv().refdrv_iiwt_2d(_ptr(a), a.strides[0], w, h, filt, _bpp(a))
    assert r == 0
    return a


# ---- frames ----------------------------------------------------------------

class UpComp:
    """One component of an upsampled reference (4 planes + 32-px aprons)."""

    def __init__(self, pic, ext=32, upsample=True):
        pic = np.ascontiguousarray(pic, dtype=np.uint8)
        self.h, self.w = pic.shape
        self.ext = ext
        self.c = lib().oracle_upcomp_new(self.w, self.h, ext)
        lib().oracle_upcomp_set_plane0(self.c, _ptr(pic), pic.strides[0])
        lib().oracle_upcomp_edgeextend(self.c)
        if upsample:
            lib().oracle_upcomp_upsample(self.c)

    def plane(self, i):
        out = np.empty((self.h, self.w), np.uint8)
        lib().oracle_upcomp_get_plane(self.c, i, _ptr(out), out.strides[0])
        return out

    def get(self, i, x, y):
        return lib().oracle_upcomp_get(self.c, i, x, y)

    def __del__(self):
        try:
            lib().oracle_upcomp_free(self.c)
        except Exception:
            pass


def convert_u8(src, width, height):
    src = np.ascontiguousarray(src)
    out = np.empty((height, width), np.uint8)
    lib().oracle_convert_u8_from_signed(_ptr(out), out.strides[0], _ptr(src), src.strides[0],
                                        _bpp(src), width, height)
    return out


FORMAT_YUYV, FORMAT_UYVY, FORMAT_AYUV = 0x100, 0x101, 0x102


class PackSrc(C.Structure):
    _fields_ = [("data", C.c_void_p * 3), ("stride", C.c_int * 3), ("width", C.c_int),
                ("height", C.c_int), ("h_shift", C.c_int), ("v_shift", C.c_int)]


def pack_u8(planes, h_shift, v_shift, fmt, width, height):
    """schro_frame_convert (packed dest, planar u8 src): rows of 4-byte groups."""
    planes = [np.ascontiguousarray(p, np.uint8) for p in planes]
    src = PackSrc()
    for k in range(3):
        src.data[k] = planes[k].ctypes.data
        src.stride[k] = planes[k].strides[0]
    src.height, src.width = planes[0].shape
    src.h_shift, src.v_shift = h_shift, v_shift
    groups = width if fmt == FORMAT_AYUV else width // 2
    out = np.zeros((height, 4 * groups), np.uint8)
    if groups:
        r = lib().oracle_pack_u8(_ptr(out), out.strides[0], fmt, width, height, C.byref(src))
        assert r == 0, "oracle_pack_u8 refused the arguments"
    return out


FORMAT_V210 = 0x106


def pack_v210(planes, h_shift, v_shift, width, height):
    """schro_frame_convert (v210 dest, planar u8 / s16 / s32 src): rows of 16-byte groups."""
    dt = np.asarray(planes[0]).dtype
    planes = [np.ascontiguousarray(p, dt) for p in planes]
    src = PackSrc()
    for k in range(3):
        src.data[k] = planes[k].ctypes.data
        src.stride[k] = planes[k].strides[0]
    src.height, src.width = planes[0].shape
    src.h_shift, src.v_shift = h_shift, v_shift
    out = np.zeros((height, 16 * (-(-width // 6))), np.uint8)
    r = lib().oracle_pack_v210(_ptr(out), out.strides[0], width, height, C.byref(src), dt.itemsize)
    assert r == 0, "oracle_pack_v210 refused the arguments"
    return out


WIDE_ROW_BYTES = {0x105: lambda w: 8 * (w // 2), 0x103: lambda w: 4 * w, 0x107: lambda w: 8 * w}


def pack_wide(planes, h_shift, v_shift, width, height, fmt):
    """schro_frame_convert (v216 0x105 / ARGB 0x103 / AY64 0x107 dest, planar u8 / s16 / s32 src)."""
    dt = np.asarray(planes[0]).dtype
    planes = [np.ascontiguousarray(p, dt) for p in planes]
    src = PackSrc()
    for k in range(3):
        src.data[k] = planes[k].ctypes.data
        src.stride[k] = planes[k].strides[0]
    src.height, src.width = planes[0].shape
    src.h_shift, src.v_shift = h_shift, v_shift
    out = np.zeros((height, WIDE_ROW_BYTES[fmt](width)), np.uint8)
    r = lib().oracle_pack_wide(_ptr(out), out.strides[0], fmt, width, height, C.byref(src), dt.itemsize)
    assert r == 0, "oracle_pack_wide refused the arguments"
    return out


def shift_right(a, shift):
    a = np.ascontiguousarray(a).copy()
    r = lib().oracle_shift_right(_ptr(a), a.strides[0], a.shape[1], a.shape[0], a.dtype.itemsize, shift)
    assert r == 0
    return a


MV_DTYPE = np.dtype([("flags", "<u4"), ("metric", "<u4"), ("chroma_metric", "<u4"),
                     ("v", "<i2", (4,))])
assert MV_DTYPE.itemsize == 20


class MotionParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "x_num_blocks", "y_num_blocks", "xblen_luma", "yblen_luma", "xbsep_luma", "ybsep_luma",
        "mv_precision", "picture_weight_bits", "picture_weight_1", "picture_weight_2",
        "chroma_h_shift", "chroma_v_shift")]


def motion_render(mvs, params, k, ref1, ref2, residual, width, height, return_acc=False):
    """schro_motion_render_u8 (add=TRUE) for component k -> u8 (height, width); return_acc: (out, the s16
    accumulator the blocks were scattered into -- what orc_rrshift6_*'s read)."""
    mvs = np.ascontiguousarray(mvs)
    assert mvs.dtype == MV_DTYPE
    residual = np.ascontiguousarray(residual)
    acc = np.zeros((height, width), np.int16)
    out = np.zeros((height, width), np.uint8)
    r = lib().oracle_motion_render_u8(
        _ptr(mvs), C.byref(params), k, ref1.c, ref2.c if ref2 is not None else None,
        _ptr(residual), residual.strides[0], _bpp(residual),
        _ptr(acc), acc.strides[0], _ptr(out), out.strides[0], width, height)
    assert r == 0
    return (out, acc) if return_acc else out


def rrshift6_s16(acc):
    """orc_rrshift6_s16_ip_2d (schroorc.orc:676-682: subw 8160, shrsw 6): the s16 frame schro_motion_render (add = FALSE)
    leaves in its dest (schromotion8.c:896-899) -- the prediction - 128.  The reference's own compiled kernel where
    oracle/_ref is there, else its restatement (tests/test_oracle_motion.py compares the two)."""
    acc = np.ascontiguousarray(acc, np.int16)
    if ref_available():
        d = acc.copy()
        reforc().orc_rrshift6_s16_ip_2d(_ptr(d), C.c_int(d.strides[0]), C.c_int(d.shape[1]), C.c_int(d.shape[0]))
        return d
    return rrshift6_s16_restated(acc)


def rrshift6_s16_restated(acc):
    return ((acc.astype(np.int32) - 8160).astype(np.int16) >> 6).astype(np.int16)


def frame_add(dst, src):
    """schro_frame_add's two cases (schroframe.c:1082-1135): dst (s16) += src (s16 | u8) over the common size,
    16-bit wrap -- orc_add_s16_2d / orc_add_s16_u8_2d compiled from the reference where oracle/_ref is there."""
    d = np.ascontiguousarray(dst, np.int16).copy()
    s = np.ascontiguousarray(src)
    h, w = min(d.shape[0], s.shape[0]), min(d.shape[1], s.shape[1])
    if ref_available():
        f = reforc().orc_add_s16_2d if s.dtype == np.int16 else reforc().orc_add_s16_u8_2d
        assert s.dtype in (np.int16, np.uint8)
        f(_ptr(d), C.c_int(d.strides[0]), _ptr(s), C.c_int(s.strides[0]), C.c_int(w), C.c_int(h))
        return d
    return frame_add_restated(dst, src)


def frame_add_restated(dst, src):
    d = np.ascontiguousarray(dst, np.int16).copy()
    h, w = min(d.shape[0], src.shape[0]), min(d.shape[1], src.shape[1])
    d[:h, :w] = (d[:h, :w].astype(np.int32) + src[:h, :w].astype(np.int32)).astype(np.int16)
    return d


# ---- VC-2 low-delay transform data ----------------------------------------------

LOWDELAY_FAST16, LOWDELAY_SLOW16, LOWDELAY_S32 = 0, 1, 2


def _ld_params(P):
    lp = LowDelayParams()
    for name in ("transform_depth", "iwt_luma_width", "iwt_luma_height", "iwt_chroma_width",
                 "iwt_chroma_height", "n_horiz_slices", "n_vert_slices", "slice_bytes_num",
                 "slice_bytes_denom"):
        setattr(lp, name, int(P[name]))
    for k, q in enumerate(P["quant_matrix"]):
        lp.quant_matrix[k] = int(q)
    return lp


def _ld_planes(planes):
    comp = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
    stride = (C.c_int * 3)(*[p.strides[0] for p in planes])
    return comp, stride


def lowdelay_arith(P, bpp):
    return lib().oracle_lowdelay_arith(C.byref(_ld_params(P)), bpp)


def lowdelay_slice_bytes(P):
    """schrodecoder.c:2931-2932"""
    return (P["slice_bytes_num"] * P["n_horiz_slices"] * P["n_vert_slices"]) // P["slice_bytes_denom"]


def lowdelay_decode(data, planes, P):
    """Decodes into the three planes (C-contiguous s16 / s32 arrays) in place."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    comp, stride = _ld_planes(planes)
    r = lib().oracle_lowdelay_decode(_ptr(data), data.size, comp, stride, C.byref(_ld_params(P)),
                                     planes[0].dtype.itemsize)
    assert r == 0, r


def lowdelay_write(planes, P, bpp, base_index, pad_bit=1, y_length_bias=0, nbytes=None):
    """Slice bytes for QUANTISED values (int32 planes in the coefficient frame layout) of a
    picture whose samples are bpp bytes."""
    assert all(p.dtype == np.int32 for p in planes)
    data = np.zeros(lowdelay_slice_bytes(P) if nbytes is None else nbytes, np.uint8)
    base_index = np.ascontiguousarray(base_index, dtype=np.uint8)
    assert base_index.size == P["n_horiz_slices"] * P["n_vert_slices"]
    comp, stride = _ld_planes(planes)
    r = lib().oracle_lowdelay_write(_ptr(data), data.size, comp, stride, C.byref(_ld_params(P)),
                                    bpp, _ptr(base_index), pad_bit, y_length_bias)
    assert r == 0, r
    return data


def dc_predict(a):
    a = np.ascontiguousarray(a).copy()
    lib().oracle_dc_predict(_ptr(a), a.strides[0], a.shape[1], a.shape[0], _bpp(a))
    return a


def quant_tables():
    L = lib()
    return ([L.oracle_quant_factor(q) for q in range(61)], [L.oracle_quant_offset_1_2(q) for q in range(61)])


def quant_offset_3_8():
    return [lib().oracle_quant_offset_3_8(q) for q in range(61)]


def dequant_codeblock(dst, q, quant_index, is_intra, arith):
    """dst: 2-D view (a codeblock of a sub-band, any strides) of int16 / int32, written in place.
    q: the codeblock's quantised values (2-D, int8 / int16 / int32) or None for a zero codeblock."""
    h, w = dst.shape
    assert dst.strides[1] == dst.itemsize
    src = None
    if q is not None:
        src = np.ascontiguousarray(q)
        assert src.shape == (h, w)
    lib().oracle_dequant_codeblock(dst.ctypes.data_as(C.c_void_p), dst.strides[0], dst.itemsize,
                                   _ptr(src) if src is not None else None, src.itemsize if src is not None else 0,
                                   w, h, quant_index, 1 if is_intra else 0, arith)
