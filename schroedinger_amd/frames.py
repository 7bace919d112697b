"""SchroFrame-shaped views for the stage-level C ABI (include/schro_hip.h, frame layer).

SchroHipFrame IS SchroFrame (same layout, tests/test_ref_layout.py): HostFrame wraps three numpy
planes as a frame with domain == NULL (a decoder's host SchroFrame); DeviceFrame owns a frame
allocated by schro_hip_frame_new_and_alloc in the context's memory domain."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

_DEPTH = {np.dtype(np.uint8): 0x00, np.dtype(np.int16): 0x04, np.dtype(np.int32): 0x08}
_DTYPE = {0x00: np.uint8, 0x04: np.int16, 0x08: np.int32}


def frame_format(dtype, h_shift, v_shift):
    """SchroFrameFormat (schroframe.h:22-35) from depth and chroma shifts."""
    return _DEPTH[np.dtype(dtype)] | (1 if h_shift else 0) | (2 if v_shift else 0)


class HostFrame:
    def __init__(self, planes, h_shift, v_shift):
        self.planes = [np.ascontiguousarray(p) for p in planes]
        self.c = _lib.Frame()
        f = self.c
        f.refcount, f.domain = 1, None
        f.format = frame_format(self.planes[0].dtype, h_shift, v_shift)
        f.height, f.width = self.planes[0].shape
        for k, p in enumerate(self.planes):
            d = f.components[k]
            d.format, d.data, d.stride = f.format, p.ctypes.data, p.strides[0]
            d.height, d.width = p.shape
            d.length = p.strides[0] * p.shape[0]
            d.h_shift, d.v_shift = (h_shift, v_shift) if k else (0, 0)

    def ptr(self):
        return C.byref(self.c)


def packed_row_bytes(fmt, width):
    """Bytes of the pixels of one row of a packed frame (schroframe.c:233-330 layouts)."""
    return {0x100: 4 * (width // 2), 0x101: 4 * (width // 2), 0x102: 4 * width, 0x103: 4 * width,
            0x105: 8 * (width // 2), 0x106: 16 * (-(-width // 6)), 0x107: 8 * width}[fmt]


class PackedHostFrame:
    """A packed (YUYV / UYVY / AYUV / ARGB / v216 / v210 / AY64) host frame: one component."""

    def __init__(self, fmt, width, height):
        self.rows = np.zeros((height, max(packed_row_bytes(fmt, width), 4)), np.uint8)
        self.c = _lib.Frame()
        f = self.c
        f.refcount, f.domain, f.format, f.width, f.height = 1, None, fmt, width, height
        d = f.components[0]
        d.format, d.data, d.stride = fmt, self.rows.ctypes.data, self.rows.strides[0]
        d.width, d.height, d.length = width, height, self.rows.nbytes

    def ptr(self):
        return C.byref(self.c)


class DeviceFrame:
    def __init__(self, ctx, fmt, width, height, upsampled=False):
        self.ctx = ctx
        self.p = ctx.lib.schro_hip_frame_new_and_alloc(ctx.h, fmt, width, height, 1 if upsampled else 0)
        if not self.p:
            raise _lib.SchroHipError(ctx.lib.schro_hip_last_error().decode())

    @property
    def c(self):
        return self.p.contents

    def ptr(self):
        return self.p

    def upload(self, host):
        check(self.ctx.lib.schro_frame_to_hip(self.p, host.ptr()))
        return self

    def download(self):
        f = self.c
        if f.format & 0x100:            # packed: rows of 4-byte groups
            host = PackedHostFrame(f.format, f.width, f.height)
            check(self.ctx.lib.schro_hipframe_to_cpu(host.ptr(), self.p))
            return host.rows[:, :packed_row_bytes(f.format, f.width)]
        dt = _DTYPE[f.format & 0xc]
        mul = 2 if f.is_upsampled else 1
        planes = [np.zeros((f.components[k].height * mul, f.components[k].width * mul), dt)
                  for k in range(3)]
        host = HostFrame(planes, f.format & 1, (f.format >> 1) & 1)
        if f.is_upsampled:      # copy the full half-pel images
            for k in range(1 if f.is_upsampled == 2 else 3):
                comp = f.components[k]
                check(self.ctx.lib.schro_hip_upsampled_download(
                    self.ctx.h, planes[k].ctypes.data_as(C.c_void_p), planes[k].strides[0],
                    comp.data, comp.stride, comp.width, comp.height))
            if f.is_upsampled == 2:     # the chroma components are one (U, V) pair image in components[1]
                comp = f.components[1]
                check(self.ctx.lib.schro_hip_upsampled_pair_download(
                    self.ctx.h, planes[1].ctypes.data_as(C.c_void_p), planes[2].ctypes.data_as(C.c_void_p),
                    planes[1].strides[0], comp.data, comp.stride, comp.width, comp.height))
            return planes
        check(self.ctx.lib.schro_hipframe_to_cpu(host.ptr(), self.p))
        return host.planes

    def unref(self):
        if self.p:
            self.ctx.lib.schro_hip_frame_unref(self.p)
            self.p = None


def make_params(**kw):
    p = _lib.Params()
    for k, v in kw.items():
        setattr(p, k, int(v))
    return p
