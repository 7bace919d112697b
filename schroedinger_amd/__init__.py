"""schroedinger_amd -- MI355X (gfx950) execution domain for the Dirac/VC-2 decode
pixel path (inverse wavelet -> half-pel upsample -> OBMC + residual add).

The product is libschro_hip.so (C ABI: include/schro_hip.h; kernels:
schroedinger_amd/csrc/*.hip).  This package is only the thin Python view of
that ABI used by tests/ and bench.py: device planes, batched launches and the
SchroFrame-shaped stage calls.  No CPU fallback exists: every operator calls
the HIP library or raises.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SchroHipError, check  # noqa: F401

# SchroFrameFormat values, schroedinger/schroframe.h:22-35
FORMAT_U8_444, FORMAT_U8_422, FORMAT_U8_420 = 0x00, 0x01, 0x03
FORMAT_S16_444, FORMAT_S16_422, FORMAT_S16_420 = 0x04, 0x05, 0x07
FORMAT_S32_444, FORMAT_S32_422, FORMAT_S32_420 = 0x08, 0x09, 0x0b
FORMAT_YUYV, FORMAT_UYVY, FORMAT_AYUV = 0x100, 0x101, 0x102       # packed, schroframe.h:36-38
FORMAT_ARGB, FORMAT_V216, FORMAT_AY64 = 0x103, 0x105, 0x107
FORMAT_V210 = 0x106

# SchroMotionVector, schroedinger/schromotion.h:20-37 (20 bytes)
MV_DTYPE = np.dtype([("flags", "<u4"), ("metric", "<u4"), ("chroma_metric", "<u4"),
                     ("v", "<i2", (4,))])


def device_count():
    return _lib.load().schro_hip_device_count()


class DevicePlane:
    """A 2-D array in the context's memory domain."""

    def __init__(self, ctx, height, width, dtype, stride=None):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self.height, self.width = int(height), int(width)
        row = self.width * self.dtype.itemsize
        self.stride = int(stride) if stride else (row + 63) // 64 * 64
        self.nbytes = self.stride * self.height
        self.ptr = ctx.alloc(self.nbytes)

    def level_view(self, level=1):
        """The level-`level` view of a coefficient plane in the reference's in-place layout (schroparams.c:319-352): the
        same memory, {width >> level, height >> level, stride << level} -- the `src` of a call that runs only the levels
        from `level` up (Context.iiwt_batch)."""
        v = DevicePlane.__new__(DevicePlane)
        v.ctx, v.dtype = self.ctx, self.dtype
        v.height, v.width, v.stride = self.height >> level, self.width >> level, self.stride << level
        v.nbytes, v.ptr = 0, self.ptr
        v.free = lambda: None           # (the memory is the parent's)
        return v

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.shape == (self.height, self.width), (a.shape, self.height, self.width)
        check(self.ctx.lib.schro_hip_upload_2d(
            self.ctx.h, self.ptr, self.stride, a.ctypes.data_as(C.c_void_p), a.strides[0],
            self.width * self.dtype.itemsize, self.height))
        return self

    def download(self):
        out = np.empty((self.height, self.width), self.dtype)
        check(self.ctx.lib.schro_hip_download_2d(
            self.ctx.h, out.ctypes.data_as(C.c_void_p), out.strides[0], self.ptr, self.stride,
            self.width * self.dtype.itemsize, self.height))
        return out

    def upload_async(self, a):
        """Enqueue the copy of `a` (a pinned host array, Context.host_array) on the selected queue."""
        assert a.shape == (self.height, self.width) and a.dtype == self.dtype, (a.shape, a.dtype)
        check(self.ctx.lib.schro_hip_upload_2d_async(
            self.ctx.h, self.ptr, self.stride, a.ctypes.data_as(C.c_void_p), a.strides[0],
            self.width * self.dtype.itemsize, self.height))
        return self

    def download_async(self, out):
        """Enqueue the copy into `out` (a pinned host array) on the selected queue; not waited for."""
        assert out.shape == (self.height, self.width) and out.dtype == self.dtype
        check(self.ctx.lib.schro_hip_download_2d_async(
            self.ctx.h, out.ctypes.data_as(C.c_void_p), out.strides[0], self.ptr, self.stride,
            self.width * self.dtype.itemsize, self.height))
        return out

    def fill(self, byte):
        check(self.ctx.lib.schro_hip_memset(self.ctx.h, self.ptr, byte, self.nbytes))
        return self

    def free(self):
        if self.ptr:
            self.ctx.free(self.ptr)
            self.ptr = None


class HpPlane(DevicePlane):
    """The half-pel (2x upsampled) image of a width x height u8 component: 2*height x
    2*width samples as the four tiled half-pel planes of include/schro_hip.h.
    pair=True: the PAIR image of two such components (the U and V planes of a picture,
    samples interleaved)."""

    def __init__(self, ctx, height, width, pair=False):
        self.ctx = ctx
        self.dtype = np.dtype(np.uint8)
        self.pair = bool(pair)
        self.comp_height, self.comp_width = int(height), int(width)
        self.height, self.width = 2 * self.comp_height, 2 * self.comp_width
        st = C.c_int(0)
        size_of = ctx.lib.schro_hip_upsampled_pair_bytes if pair else ctx.lib.schro_hip_upsampled_bytes
        self.nbytes = size_of(self.comp_width, self.comp_height, C.byref(st))
        self.stride = st.value
        self.ptr = ctx.alloc(self.nbytes)

    def upload(self, a):
        raise SchroHipError("half-pel images are produced by upsample_batch")

    def download(self):
        """Linear (2*height, 2*width) array; a pair image: the two components' arrays."""
        if self.pair:
            u, v = (np.empty((self.height, self.width), np.uint8) for _ in range(2))
            check(self.ctx.lib.schro_hip_upsampled_pair_download(
                self.ctx.h, u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), u.strides[0], self.ptr,
                self.stride, self.comp_width, self.comp_height))
            return u, v
        out = np.empty((self.height, self.width), np.uint8)
        check(self.ctx.lib.schro_hip_upsampled_download(
            self.ctx.h, out.ctypes.data_as(C.c_void_p), out.strides[0], self.ptr, self.stride,
            self.comp_width, self.comp_height))
        return out


class ArenaPlane(DevicePlane):
    """A height x width plane carved out of a larger device allocation (an arena: all planes of a
    picture batch in ONE block, so that they cross the host boundary as one copy); owns nothing."""

    def __init__(self, arena, offset, height, width, dtype, stride=None):
        self.ctx = arena.ctx
        self.dtype = np.dtype(dtype)
        self.height, self.width = int(height), int(width)
        row = self.width * self.dtype.itemsize
        self.stride = int(stride) if stride else (row + 63) // 64 * 64
        self.nbytes = self.stride * self.height
        assert offset % 256 == 0 and offset + self.nbytes <= arena.nbytes
        self.ptr = arena.ptr + offset

    def free(self):
        self.ptr = None


class Arena:
    """One device block handed out in 256-byte aligned pieces (ArenaPlane)."""

    def __init__(self, ctx, nbytes):
        self.block = DevicePlane(ctx, 1, int(nbytes), np.uint8)
        self.ctx, self.ptr, self.nbytes, self.used = ctx, self.block.ptr, self.block.nbytes, 0

    @staticmethod
    def size_of(shapes_dtypes):
        return sum(((w * np.dtype(d).itemsize + 63) // 64 * 64 * h + 255) // 256 * 256 for (h, w), d in shapes_dtypes)

    def plane(self, height, width, dtype):
        p = ArenaPlane(self, self.used, height, width, dtype)
        self.used += (p.nbytes + 255) // 256 * 256
        return p


class SubPlane:
    """A strided view into a DevicePlane (e.g. the LL band of a coefficient frame in the
    in-place sub-band layout: rows 2^depth apart, stride << depth); owns nothing."""

    def __init__(self, plane, y0, x0, height, width, stride=None):
        self.ctx, self.dtype = plane.ctx, plane.dtype
        self.ptr = plane.ptr + y0 * plane.stride + x0 * plane.dtype.itemsize
        self.height, self.width = int(height), int(width)
        self.stride = int(stride) if stride else plane.stride


class Context:
    """One exec-domain context: device, stream, memory domain."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        self.h = self.lib.schro_hip_context_new(device)
        if not self.h:
            raise SchroHipError("cannot create context on device %d: %s" % (
                device, self.lib.schro_hip_last_error().decode()))
        self.device = device

    def close(self):
        if self.h:
            self.lib.schro_hip_context_free(self.h)
            self.h = None

    def alloc(self, nbytes):
        p = self.lib.schro_hip_domain_alloc(self.h, nbytes)
        if not p:
            raise SchroHipError(self.lib.schro_hip_last_error().decode())
        return p

    def free(self, ptr):
        check(self.lib.schro_hip_domain_free(self.h, ptr))

    def domain_bytes(self):
        return self.lib.schro_hip_domain_bytes(self.h)

    def synchronize(self):
        check(self.lib.schro_hip_synchronize(self.h))

    QUEUE_H2D, QUEUE_D2H = 2, 3

    def queue_synchronize(self, q):
        check(self.lib.schro_hip_queue_synchronize(self.h, q))

    def queue_set_cu_mask(self, q, bits):
        """bits: iterable of 0 / 1 per compute unit (hipExtStreamCreateWithCUMask's bit order)."""
        bits = list(bits)
        words = (len(bits) + 31) // 32
        arr = (C.c_uint32 * words)()
        for n, b in enumerate(bits):
            if b:
                arr[n // 32] |= 1 << (n % 32)
        check(self.lib.schro_hip_queue_set_cu_mask(self.h, q, arr, words))

    def host_array(self, shape, dtype):
        """A numpy array in pinned host memory (schro_hip_host_alloc): what the asynchronous copies
        read and write at full rate.  Freed with the array."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = self.lib.schro_hip_host_alloc(max(n, 1))
        if not p:
            raise SchroHipError(self.lib.schro_hip_last_error().decode())
        buf = (C.c_char * max(n, 1)).from_address(p)
        a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        lib = self.lib
        import weakref
        weakref.finalize(buf, lib.schro_hip_host_free, p)
        return a

    def select_queue(self, q):
        """Calls that follow are enqueued on in-order queue q (0 or 1)."""
        check(self.lib.schro_hip_context_select_queue(self.h, q))

    def queue(self):
        """The queue selected at the moment."""
        return self.lib.schro_hip_context_queue(self.h)

    def queue_mark(self, mark):
        """Record mark (0..15) behind the work enqueued so far on the selected queue."""
        check(self.lib.schro_hip_queue_mark(self.h, mark))

    def queue_wait_mark(self, mark):
        """Later work on the selected queue waits for the latest recording of `mark`."""
        check(self.lib.schro_hip_queue_wait_mark(self.h, mark))

    def queue_mark_synchronize(self, mark):
        """The calling thread waits for the latest recording of `mark` (and for nothing else)."""
        check(self.lib.schro_hip_queue_mark_synchronize(self.h, mark))

    def queue_wait(self, waiter, signaller):
        """Later work on `waiter` starts after everything enqueued so far on `signaller`."""
        check(self.lib.schro_hip_queue_wait(self.h, waiter, signaller))

    def timer_begin(self):
        check(self.lib.schro_hip_timer_begin(self.h))

    def timer_end(self):
        ms = self.lib.schro_hip_timer_end(self.h)
        if ms < 0:
            raise SchroHipError(self.lib.schro_hip_last_error().decode())
        return ms

    KERNEL_CLASSES = ("iiwt_finest", "iiwt_coarse", "upsample", "obmc", "convert", "slices", "dc_predict",
                      "dequant")

    def profile_enable(self, on=True):
        check(self.lib.schro_hip_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        check(self.lib.schro_hip_profile_reset(self.h))

    def profile_read(self):
        """{kernel class: (total_ms, launches)} from the per-launch HIP events."""
        out = {}
        for k, name in enumerate(self.KERNEL_CLASSES):
            ms, n = C.c_double(), C.c_int()
            check(self.lib.schro_hip_profile_read(self.h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def plane(self, height, width, dtype, stride=None):
        return DevicePlane(self, height, width, dtype, stride)

    def hp_plane(self, height, width, pair=False):
        """Half-pel image buffer for a height x width u8 component (upsample_batch's dst); pair: for
        the (U, V) components of a picture together."""
        return HpPlane(self, height, width, pair)

    def upload(self, a, stride=None):
        a = np.ascontiguousarray(a)
        return DevicePlane(self, a.shape[0], a.shape[1], a.dtype, stride).upload(a)

    def upload_bytes(self, a):
        """1-D blob (e.g. a SchroMotionVector array) -> device pointer."""
        raw = np.ascontiguousarray(a).view(np.uint8).reshape(1, -1)
        return DevicePlane(self, 1, raw.shape[1], np.uint8).upload(raw)

    # ---- batched plane-level launches (asynchronous on the context stream) ----

    def iiwt_batch(self, pairs, depth, filt, ll=None):
        """pairs: [(src, dst DevicePlane)], all s16 or all s32 -- dst: the residual plane; or, the combine
        form (r04), [(src, out u8 DevicePlane, pred)]: the transform's last step writes the picture
        out = sat_u8 (residual + pred), pred a u8 DevicePlane (the prediction of obmc_batch (prediction_only)) or
        None for a picture without references (+ 128).  ll: per plane, the DevicePlane that holds the LL band of
        this call's coarsest level (the output of an earlier call on the planes' level views, `level_view`)."""
        n = len(pairs)
        arr = (_lib.IwtPlane * n)()
        bpp = pairs[0][0].dtype.itemsize
        for k, t in enumerate(pairs):
            s, d = t[0], t[1]
            assert s.dtype.itemsize == bpp
            if len(t) == 2:
                assert s.dtype == d.dtype
                arr[k] = _lib.IwtPlane(s.ptr, s.stride, d.ptr, d.stride, s.width, s.height, None, 0, 0, 0, 0)
            else:
                pred = t[2]
                assert d.dtype == np.uint8 and (pred is None or (pred.dtype == np.uint8 and (pred.height, pred.width) == (d.height, d.width)))
                arr[k] = _lib.IwtPlane(s.ptr, s.stride, d.ptr, d.stride, s.width, s.height,
                                       pred.ptr if pred is not None else None, pred.stride if pred is not None else 0,
                                       d.width, d.height, 1 if pred is not None else 2)
            if ll is not None:
                q = ll[k]
                assert q.dtype.itemsize == bpp and (q.height, q.width) == (s.height >> depth, s.width >> depth)
                arr[k].ll, arr[k].ll_stride = q.ptr, q.stride
        check(self.lib.schro_hip_iiwt_batch(self.h, arr, n, depth, filt, bpp))

    def pack_u8_batch(self, jobs):
        """jobs: (planes [Y, U, V] DevicePlanes, h_shift, v_shift, dst DevicePlane of 4-byte
        groups, width, height, format) per picture."""
        n = len(jobs)
        arr = (_lib.PackPlane * n)()
        for a, (planes, hs, vs, dst, w, h, fmt) in zip(arr, jobs):
            for k in range(3):
                a.src[k] = planes[k].ptr
                a.src_stride[k] = planes[k].stride
            a.src_width, a.src_height = planes[0].width, planes[0].height
            a.src_h_shift, a.src_v_shift = hs, vs
            a.dst, a.dst_stride = dst.ptr, dst.stride
            a.width, a.height, a.format = w, h, fmt
        check(self.lib.schro_hip_pack_u8_batch(self.h, arr, n))

    def pack_v210_batch(self, jobs):
        """jobs: (planes [Y, U, V] DevicePlanes of one dtype (u8 / s16 / s32), h_shift, v_shift,
        dst DevicePlane of bytes, width, height) per picture."""
        n = len(jobs)
        arr = (_lib.PackPlane * n)()
        bpp = jobs[0][0][0].dtype.itemsize
        for a, (planes, hs, vs, dst, w, h) in zip(arr, jobs):
            assert planes[0].dtype.itemsize == bpp
            for k in range(3):
                a.src[k] = planes[k].ptr
                a.src_stride[k] = planes[k].stride
            a.src_width, a.src_height = planes[0].width, planes[0].height
            a.src_h_shift, a.src_v_shift = hs, vs
            a.dst, a.dst_stride = dst.ptr, dst.stride
            a.width, a.height, a.format = w, h, FORMAT_V210
        check(self.lib.schro_hip_pack_v210_batch(self.h, arr, n, bpp))

    def iiwt_pack_v210_batch(self, jobs, depth, filt):
        """r05: the inverse wavelet and the v210 copy-out in one call.  jobs: (coefficient planes [Y, U, V] DevicePlanes of one
        dtype, h_shift, v_shift, dst DevicePlane of bytes, picture width, picture height) per picture."""
        n = len(jobs)
        arr = (_lib.IwtPackPicture * n)()
        bpp = jobs[0][0][0].dtype.itemsize
        for a, (planes, hs, vs, dst, w, h) in zip(arr, jobs):
            for k in range(3):
                assert planes[k].dtype.itemsize == bpp
                a.src[k] = planes[k].ptr
                a.src_stride[k] = planes[k].stride
            a.width, a.height = planes[0].width, planes[0].height
            a.h_shift, a.v_shift = hs, vs
            a.dst, a.dst_stride = dst.ptr, dst.stride
            a.out_width, a.out_height = w, h
        check(self.lib.schro_hip_iiwt_pack_v210_batch(self.h, arr, n, depth, filt, bpp))

    def pack_wide_batch(self, jobs):
        """jobs: (planes [Y, U, V] DevicePlanes of one dtype, h_shift, v_shift, dst DevicePlane of
        bytes, width, height, format (FORMAT_V216 / FORMAT_ARGB / FORMAT_AY64)) per picture."""
        n = len(jobs)
        arr = (_lib.PackPlane * n)()
        bpp = jobs[0][0][0].dtype.itemsize
        for a, (planes, hs, vs, dst, w, h, fmt) in zip(arr, jobs):
            assert planes[0].dtype.itemsize == bpp
            for k in range(3):
                a.src[k] = planes[k].ptr
                a.src_stride[k] = planes[k].stride
            a.src_width, a.src_height = planes[0].width, planes[0].height
            a.src_h_shift, a.src_v_shift = hs, vs
            a.dst, a.dst_stride = dst.ptr, dst.stride
            a.width, a.height, a.format = w, h, fmt
        check(self.lib.schro_hip_pack_wide_batch(self.h, arr, n, bpp))

    def shift_right_batch(self, planes, shift):
        """schro_frame_shift_right on DevicePlanes (s16 / s32), in place."""
        n = len(planes)
        arr = (_lib.DcPlane * n)()
        for a, p in zip(arr, planes):
            a.data, a.stride, a.width, a.height = p.ptr, p.stride, p.width, p.height
        check(self.lib.schro_hip_shift_right_batch(self.h, arr, n, planes[0].dtype.itemsize, shift))

    @staticmethod
    def lowdelay_params(P):
        """dict with the SchroParams members of the slice decode -> the ABI struct."""
        lp = _lib.LowDelayParams()
        for name in ("transform_depth", "iwt_luma_width", "iwt_luma_height", "iwt_chroma_width",
                     "iwt_chroma_height", "n_horiz_slices", "n_vert_slices", "slice_bytes_num",
                     "slice_bytes_denom"):
            setattr(lp, name, int(P[name]))
        for k, q in enumerate(P["quant_matrix"]):
            lp.quant_matrix[k] = int(q)
        return lp

    def lowdelay_arith(self, P, bpp):
        r = self.lib.schro_hip_lowdelay_arith(C.byref(self.lowdelay_params(P)), bpp)
        check(min(r, 0))
        return r

    def lowdelay_batch(self, pictures, P):
        """pictures: (slice bytes on the device (upload_bytes), [Y, U, V] coefficient
        DevicePlanes) per picture; P: the parameter dict.  Mirrors
        schro_decoder_decode_lowdelay_transform_data."""
        n = len(pictures)
        arr = (_lib.LowDelayPicture * n)()
        bpp = pictures[0][1][0].dtype.itemsize
        for a, (sl, planes) in zip(arr, pictures):
            a.slices, a.slices_bytes = sl.ptr, sl.width
            for k in range(3):
                assert planes[k].dtype.itemsize == bpp
                a.comp[k], a.stride[k] = planes[k].ptr, planes[k].stride
        check(self.lib.schro_hip_lowdelay_batch(self.h, arr, n, C.byref(self.lowdelay_params(P)), bpp))

    @staticmethod
    def codeblock_table(cbs):
        """The C table (SchroHipCodeblock array) of a list of (dst_offset, dst_stride, width, height,
        src_offset, src_bytes, quant_index): build it once per picture, not per call."""
        tab = (_lib.Codeblock * len(cbs))()
        for t, cb in zip(tab, cbs):
            (t.dst_offset, t.dst_stride, t.width, t.height, t.src_offset, t.src_bytes, t.quant_index) = cb
        return tab

    def codeblock_layout(self, width, height, depth, horiz_codeblocks, vert_codeblocks, stride, itemsize):
        """schro_hip_codeblock_layout: the geometry of every codeblock record of a component (C table)."""
        hc = (C.c_int * (depth + 1))(*horiz_codeblocks)
        vc = (C.c_int * (depth + 1))(*vert_codeblocks)
        n = self.lib.schro_hip_codeblock_layout(width, height, depth, hc, vc, stride, itemsize, None, 0)
        if n < 0:
            raise SchroHipError(self.lib.schro_hip_last_error().decode())
        tab = (_lib.Codeblock * n)()
        check(min(0, self.lib.schro_hip_codeblock_layout(width, height, depth, hc, vc, stride, itemsize, tab, n)))
        return tab

    def dequant_batch(self, jobs, arith=0):
        """jobs: (dst DevicePlane (s16 / s32), values device blob (DevicePlane of bytes, or an object
        with .ptr) or None, codeblocks -- a C table from codeblock_table / codeblock_layout or a list
        of (dst_offset, dst_stride, width, height, src_offset, src_bytes, quant_index) --,
        is_intra) per component."""
        n = len(jobs)
        arr = (_lib.DequantPlane * n)()
        keep = []
        for a, (dst, values, cbs, intra) in zip(arr, jobs):
            tab = cbs if isinstance(cbs, C.Array) else self.codeblock_table(cbs)
            keep.append(tab)
            a.dst, a.values = dst.ptr, values.ptr if values is not None else None
            a.codeblocks, a.ncodeblocks, a.is_intra = tab, len(cbs), 1 if intra else 0
        check(self.lib.schro_hip_dequant_batch(self.h, arr, n, jobs[0][0].dtype.itemsize, arith))

    def _dequant_planes(self, jobs):
        n = len(jobs)
        arr = (_lib.DequantPlane * n)()
        keep = []
        for a, (dst, values, cbs, intra) in zip(arr, jobs):
            tab = cbs if isinstance(cbs, C.Array) else self.codeblock_table(cbs)
            keep.append(tab)
            a.dst, a.values = dst.ptr, values.ptr if values is not None else None
            a.codeblocks, a.ncodeblocks, a.is_intra = tab, len(cbs), 1 if intra else 0
        return arr, keep

    def dequant_plan(self, jobs, arith=0):
        """schro_hip_dequant_plan_new over `jobs` (as dequant_batch takes them): the geometry of their codeblock
        records, resident on the device.  plan.run(jobs) dequantises a batch with the same geometry."""
        return DequantPlan(self, jobs, arith)

    def dc_predict_batch(self, planes):
        """In-place DC prediction of LL bands given as DevicePlanes (s16 / s32)."""
        n = len(planes)
        arr = (_lib.DcPlane * n)()
        for a, p in zip(arr, planes):
            a.data, a.stride, a.width, a.height = p.ptr, p.stride, p.width, p.height
        check(self.lib.schro_hip_dc_predict_batch(self.h, arr, n, planes[0].dtype.itemsize))

    def convert_u8_batch(self, pairs):
        n = len(pairs)
        arr = (_lib.ConvertPlane * n)()
        bpp = pairs[0][0].dtype.itemsize
        for k, (s, d) in enumerate(pairs):
            arr[k] = _lib.ConvertPlane(s.ptr, s.stride, d.ptr, d.stride, d.width, d.height)
        check(self.lib.schro_hip_convert_u8_batch(self.h, arr, n, bpp))

    def add_batch(self, pairs):
        """pairs: [(dst s16 DevicePlane, src s16 | u8 DevicePlane)]: dst += src over their common size
        (schro_frame_add / schro_gpuframe_add on planes)."""
        n = len(pairs)
        arr = (_lib.ConvertPlane * n)()
        for k, (d, s) in enumerate(pairs):
            assert d.dtype == np.int16 and s.dtype.itemsize == pairs[0][1].dtype.itemsize
            arr[k] = _lib.ConvertPlane(s.ptr, s.stride, d.ptr, d.stride, min(d.width, s.width), min(d.height, s.height))
        check(self.lib.schro_hip_add_batch(self.h, arr, n, pairs[0][1].dtype.itemsize))

    def upsample_batch(self, pairs):
        """pairs: [(src u8 plane h x w, dst HpPlane)] or [((src U, src V), dst pair HpPlane)]."""
        n = len(pairs)
        arr = (_lib.UpsamplePlane * n)()
        for k, (s, d) in enumerate(pairs):
            sv = None
            if isinstance(s, (tuple, list)):
                s, sv = s
                assert getattr(d, "pair", False) and (sv.height, sv.width) == (s.height, s.width)
            else:
                assert not getattr(d, "pair", False)
            assert d.height == 2 * s.height and d.width == 2 * s.width
            arr[k] = _lib.UpsamplePlane(s.ptr, s.stride, d.ptr, d.stride, s.width, s.height,
                                        sv.ptr if sv is not None else None, sv.stride if sv is not None else 0)
        check(self.lib.schro_hip_upsample_batch(self.h, arr, n))

    def obmc_batch(self, planes):
        n = len(planes)
        arr = (_lib.ObmcPlane * n)(*planes)
        check(self.lib.schro_hip_obmc_batch(self.h, arr, n))


def obmc_plane(mvs, params, component, ref1, ref2, residual, out, prediction_only=False):
    """Fill a SchroHipObmcPlane.  params: dict with the SchroParams motion fields
    plus chroma_h_shift / chroma_v_shift; mvs: DevicePlane holding the records.
    prediction_only (residual None): `out` receives the prediction for iiwt_batch's combine form."""
    p = _lib.ObmcPlane()
    p.mvs = mvs.ptr
    for name in ("x_num_blocks", "y_num_blocks", "xblen_luma", "yblen_luma", "xbsep_luma",
                 "ybsep_luma", "mv_precision", "picture_weight_bits", "picture_weight_1",
                 "picture_weight_2", "chroma_h_shift", "chroma_v_shift"):
        setattr(p, name, int(params[name]))
    p.component = component
    p.ref1, p.ref1_stride = ref1.ptr, ref1.stride
    if ref2 is not None:
        p.ref2, p.ref2_stride = ref2.ptr, ref2.stride
    if residual is not None:        # None: nothing to add (a zero_residual picture)
        p.residual, p.residual_stride = residual.ptr, residual.stride
        p.residual_bpp = residual.dtype.itemsize
    p.out, p.out_stride = out.ptr, out.stride
    p.width, p.height = out.width, out.height
    # (prediction_only 2, r06: `out` is an s16 plane that receives the prediction - 128)
    p.prediction_only = int(prediction_only)
    p.ref_pair = 1 if getattr(ref1, "pair", False) else 0      # (U, V) pair images (HpPlane (pair=True))
    assert ref2 is None or bool(getattr(ref2, "pair", False)) == bool(p.ref_pair)
    return p


class DequantPlan:
    def __init__(self, ctx, jobs, arith=0):
        self.ctx = ctx
        arr, keep = ctx._dequant_planes(jobs)
        self.h = ctx.lib.schro_hip_dequant_plan_new(ctx.h, arr, len(jobs), jobs[0][0].dtype.itemsize, arith)
        if not self.h:
            raise SchroHipError(ctx.lib.schro_hip_last_error().decode())

    def planes(self, jobs):
        """The C array of a batch (dst, values, C table, is_intra per component): build it once per frame pool."""
        return self.ctx._dequant_planes(jobs)

    def run(self, jobs=None, planes=None):
        if planes is None:
            planes = self.planes(jobs)
        check(self.ctx.lib.schro_hip_dequant_plan_run(self.h, planes[0], len(planes[0])))

    def free(self):
        if self.h:
            self.ctx.lib.schro_hip_dequant_plan_free(self.h)
            self.h = None


class Scheduler:
    """schro_hip_scheduler_*: one exec-domain thread and context per device; pictures follow
    their references (include/schro_hip.h).  func(ctx, device_index) is the picture's pixel
    path; ctx is a Context of that device (None on virtual devices)."""

    def __init__(self, n_devices=0, virtual=False, devices=None):
        self.lib = _lib.load()
        if devices is not None:         # an explicit list; a device may repeat (two contexts on one GPU)
            arr = (C.c_int * len(devices))(*devices)
            self.h = self.lib.schro_hip_scheduler_new_on(arr, len(devices))
        else:
            self.h = (self.lib.schro_hip_scheduler_new_virtual if virtual
                      else self.lib.schro_hip_scheduler_new)(n_devices)
        if not self.h:
            raise SchroHipError(self.lib.schro_hip_last_error().decode())
        self.n_devices = self.lib.schro_hip_scheduler_n_devices(self.h)
        self.contexts = []
        for k in range(self.n_devices):
            hctx = self.lib.schro_hip_scheduler_context(self.h, k)
            c = None
            if hctx:
                c = Context.__new__(Context)
                c.lib, c.h, c.device = self.lib, hctx, k
            self.contexts.append(c)
        self._keep = []

    def submit(self, number, refs, is_ref, func):
        """Returns (device index, foreign reference or -1)."""
        def thunk(hctx, index, priv):
            try:
                return int(func(self.contexts[index], index) or 0)
            except Exception:       # an exception must not unwind into the C thread
                import traceback
                traceback.print_exc()
                return -99
        cb = _lib.PICTURE_FUNC(thunk)
        self._keep.append(cb)
        arr = (C.c_int * max(len(refs), 1))(*refs)
        foreign = C.c_int(-1)
        dev = self.lib.schro_hip_scheduler_submit(self.h, number, arr, len(refs), 1 if is_ref else 0, cb, None,
                                                  C.byref(foreign))
        if dev < 0:
            raise SchroHipError(self.lib.schro_hip_last_error().decode())
        return dev, foreign.value

    def retire(self, number):
        check(self.lib.schro_hip_scheduler_retire(self.h, number))

    def publish_reference(self, device_index, frame_ptr):
        """Called by a reference picture's function: the device frame (SchroHipFrame pointer; any
        non-zero token on virtual devices) its dependents read."""
        check(self.lib.schro_hip_scheduler_publish_reference(self.h, device_index, frame_ptr))

    def reference_frame(self, device_index, number):
        """Called by a picture's function: the frame of its reference `number` on this device."""
        return self.lib.schro_hip_scheduler_reference_frame(self.h, device_index, number)

    def moves(self):
        return self.lib.schro_hip_scheduler_moves(self.h)

    def skipped(self):
        """Pictures that did not run because a reference of theirs had failed (or had been skipped)."""
        return self.lib.schro_hip_scheduler_skipped(self.h)

    def refs_in_flight_max(self):
        """Most reference pictures of one device whose device work was still running at once."""
        return self.lib.schro_hip_scheduler_refs_in_flight_max(self.h)

    def wait(self):
        r = self.lib.schro_hip_scheduler_wait(self.h)
        self._keep = []
        return r

    def close(self):
        if self.h:
            self.lib.schro_hip_scheduler_free(self.h)
            self.h = None
