"""GPU parity: half-pel upsample and s16/s32 -> u8 convert vs the CPU oracle.

Upsample follows testsuite/upsample.c's design (sizes 1..20 squared, all
planes) with the oracle's exact restatement of schroframe.c as the reference.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import schroedinger_amd as sa
import synth

pytestmark = pytest.mark.gpu


def gpu_upsample(ctx, pic):
    h, w = pic.shape
    src = ctx.upload(pic)
    dst = ctx.hp_plane(h, w).fill(0xa5)
    ctx.upsample_batch([(src, dst)])
    hp = dst.download()
    src.free()
    dst.free()
    return hp


def check_planes(hp, up):
    assert np.array_equal(hp[0::2, 0::2], up.plane(0))
    assert np.array_equal(hp[0::2, 1::2], up.plane(1))
    assert np.array_equal(hp[1::2, 0::2], up.plane(2))
    assert np.array_equal(hp[1::2, 1::2], up.plane(3))


def test_upsample_small_sizes(ctx):
    for h in list(range(1, 21)) + [33, 40]:
        for w in (1, 2, 3, 7, 8, 9, 16, 20, 63, 64, 65, 96):
            pic = synth.picture_u8(h, w, seed=h * 97 + w, blur=False)
            check_planes(gpu_upsample(ctx, pic), O.UpComp(pic))


@pytest.mark.parametrize("h,w", [(240, 320), (240, 160), (1080, 1920), (2160, 3840)])
def test_upsample_picture_sizes(ctx, h, w):
    pic = synth.picture_u8(h, w, seed=4)
    check_planes(gpu_upsample(ctx, pic), O.UpComp(pic))


def test_upsample_extremes(ctx):
    # saturating inputs: checkerboards and steps drive the 8-tap sum past 0..255
    h, w = 24, 40
    yy, xx = np.mgrid[0:h, 0:w]
    for pic in (((yy + xx) & 1) * 255, (xx > 20) * 255, (yy > 11) * 255, np.full((h, w), 255)):
        pic = pic.astype(np.uint8)
        check_planes(gpu_upsample(ctx, pic), O.UpComp(pic))


def test_upsample_unaligned_sources(ctx):
    # The kernel stages 16-byte source chunks; a component that is a window of a larger
    # plane (any byte offset, the parent's pitch) takes the byte-by-byte gather instead.
    import schroedinger_amd as sa
    big = synth.picture_u8(70, 200, seed=11, blur=False)
    parent = ctx.upload(big)
    for (y0, x0, h, w) in [(0, 1, 20, 40), (3, 5, 33, 130), (7, 16, 16, 128), (1, 67, 9, 131), (2, 3, 1, 1)]:
        pic = np.ascontiguousarray(big[y0:y0 + h, x0:x0 + w])
        src = sa.SubPlane(parent, y0, x0, h, w)
        dst = ctx.hp_plane(h, w).fill(0x5a)
        ctx.upsample_batch([(src, dst)])
        check_planes(dst.download(), O.UpComp(pic))
        dst.free()
    parent.free()


@pytest.mark.parametrize("dtype", [np.int16, np.int32])
def test_convert_crop_and_saturate(ctx, dtype):
    # iwt-padded 64x48 source, 61x45 picture: crop as schrovirtframe.c:1853 does
    src_np = synth.full_range(48, 64, dtype, seed=8)
    src_np[0, :8] = [-129, -128, -127, 0, 126, 127, 128, 32767 if dtype == np.int16 else 2**31 - 1]
    src = ctx.upload(src_np)
    dst = ctx.plane(45, 61, np.uint8).fill(7)
    ctx.convert_u8_batch([(src, dst)])
    assert np.array_equal(dst.download(), O.convert_u8(src_np, 61, 45))
    src.free()
    dst.free()


def test_convert_picture_size(ctx):
    src_np = (synth.image_s(1088, 1920, np.int16, seed=3).astype(np.int32) * 3).astype(np.int16)
    src = ctx.upload(src_np)
    dst = ctx.plane(1080, 1920, np.uint8)
    ctx.convert_u8_batch([(src, dst)])
    assert np.array_equal(dst.download(), O.convert_u8(src_np, 1920, 1080))
    src.free()
    dst.free()


@pytest.mark.parametrize("h,w", [(1, 1), (5, 7), (16, 16), (33, 47), (64, 130), (135, 240), (270, 481)])
def test_half_pel_planes_every_home_and_the_aprons(ctx, h, w):
    """The r03 half-pel buffer byte by byte (include/schro_hip.h): every column of every plane sits in
    TWO chunks (byte xp & 15 of chunk xp >> 4 and byte 16 + (xp & 15) of the chunk before), and 32
    columns either side plus everything up to the end of the last chunk repeat the edge sample of plane 0
    (planes 0, 1) / plane 2 (planes 2, 3) -- schro_frame_mc_edgeextend_horiz's sources in
    schro_upsampled_frame_upsample (schroframe.c:2012-2029).  schro_hip_upsampled_download reads one home
    only; the OBMC kernels read both and the aprons."""
    pic = synth.picture_u8(h, w, seed=h * 31 + w)
    src, dst = ctx.upload(pic), ctx.hp_plane(h, w).fill(0xa5)
    ctx.upsample_batch([(src, dst)])
    up = O.UpComp(pic)
    planes = [up.plane(i) for i in range(4)]
    raw = np.empty((1, dst.nbytes), np.uint8)
    sa.check(ctx.lib.schro_hip_download_2d(ctx.h, raw.ctypes.data_as(C.c_void_p), dst.nbytes, dst.ptr, dst.nbytes,
                                          dst.nbytes, 1))
    raw = raw.reshape(-1)
    stride = dst.stride
    nch = stride // 512
    assert nch == (w + 64 + 15) // 16 + 1
    for p in range(4):
        edge_src = planes[0] if p < 2 else planes[2]
        for y in range(h):
            # expected padded row: 32 apron columns, the plane's row, aprons to the end of the last chunk
            cols = 16 * nch + 16
            want = np.empty(cols, np.uint8)
            want[:32] = edge_src[y, 0]
            want[32:32 + w] = planes[p][y]
            want[32 + w:] = edge_src[y, w - 1]
            base = (y >> 2) * stride + p * 128 + (y & 3) * 32
            for c in range(nch):
                got = raw[base + c * 512: base + c * 512 + 32]
                assert np.array_equal(got, want[16 * c:16 * c + 32]), (p, y, c)


# ---- r04: (U, V) pair images -------------------------------------------------------------------------

def gpu_upsample_pair(ctx, pu, pv):
    h, w = pu.shape
    su, sv = ctx.upload(pu), ctx.upload(pv)
    dst = ctx.hp_plane(h, w, pair=True).fill(0xa5)
    ctx.upsample_batch([((su, sv), dst)])
    hu, hv = dst.download()
    for p in (su, sv, dst):
        p.free()
    return hu, hv


def test_upsample_pair_small_sizes(ctx):
    for h in list(range(1, 21)) + [33, 40]:
        for w in (1, 2, 3, 7, 8, 9, 16, 20, 63, 64, 65, 96, 130):
            pu = synth.picture_u8(h, w, seed=h * 97 + w, blur=False)
            pv = synth.picture_u8(h, w, seed=h * 89 + w + 1, blur=False)
            hu, hv = gpu_upsample_pair(ctx, pu, pv)
            check_planes(hu, O.UpComp(pu))
            check_planes(hv, O.UpComp(pv))


@pytest.mark.parametrize("h,w", [(120, 160), (240, 160), (540, 960), (1080, 1920)])
def test_upsample_pair_picture_sizes(ctx, h, w):
    pu, pv = synth.picture_u8(h, w, seed=4), synth.picture_u8(h, w, seed=5)
    hu, hv = gpu_upsample_pair(ctx, pu, pv)
    check_planes(hu, O.UpComp(pu))
    check_planes(hv, O.UpComp(pv))


def test_upsample_pair_beside_single_planes(ctx):
    # one launch with a luma plane and a chroma pair (what the frame layer sends), unaligned sources
    big = synth.picture_u8(80, 200, seed=21, blur=False)
    parent = ctx.upload(big)
    y = synth.picture_u8(66, 140, seed=22)
    sy, dy = ctx.upload(y), ctx.hp_plane(66, 140).fill(1)
    su, sv = sa.SubPlane(parent, 1, 3, 33, 70), sa.SubPlane(parent, 40, 101, 33, 70)
    dp = ctx.hp_plane(33, 70, pair=True).fill(2)
    ctx.upsample_batch([(sy, dy), ((su, sv), dp)])
    check_planes(dy.download(), O.UpComp(y))
    hu, hv = dp.download()
    check_planes(hu, O.UpComp(np.ascontiguousarray(big[1:34, 3:73])))
    check_planes(hv, O.UpComp(np.ascontiguousarray(big[40:73, 101:171])))
    for p in (parent, sy, dy, dp):
        p.free()


@pytest.mark.parametrize("h,w", [(1, 1), (5, 7), (16, 16), (33, 47), (64, 130), (135, 240), (270, 481)])
def test_pair_images_every_home_and_the_aprons(ctx, h, w):
    """The pair image byte by byte: the layout of test_half_pel_planes_every_home_and_the_aprons over samples of
    two bytes (U, V) -- byte column 2 * (column + 32) + c, chunks of 32 bytes that advance by 16 byte columns,
    aprons of 64 bytes either side (and to the end of the last chunk) that repeat the edge SAMPLE."""
    pu, pv = synth.picture_u8(h, w, seed=h * 31 + w), synth.picture_u8(h, w, seed=h * 29 + w + 7)
    su, sv, dst = ctx.upload(pu), ctx.upload(pv), ctx.hp_plane(h, w, pair=True).fill(0xa5)
    ctx.upsample_batch([((su, sv), dst)])
    ups = (O.UpComp(pu), O.UpComp(pv))
    planes = [[up.plane(i) for i in range(4)] for up in ups]
    raw = np.empty((1, dst.nbytes), np.uint8)
    sa.check(ctx.lib.schro_hip_download_2d(ctx.h, raw.ctypes.data_as(C.c_void_p), dst.nbytes, dst.ptr, dst.nbytes,
                                          dst.nbytes, 1))
    raw = raw.reshape(-1)
    stride = dst.stride
    nch = stride // 512
    assert nch == (2 * w + 128 + 15) // 16 + 1
    for p in range(4):
        for y in range(h):
            cols = 16 * nch + 16
            want = np.empty(cols, np.uint8)
            for c in range(2):
                edge_src = planes[c][0] if p < 2 else planes[c][2]
                want[c:64:2] = edge_src[y, 0]
                want[64 + c:64 + 2 * w:2] = planes[c][p][y]
                want[64 + 2 * w + c::2] = edge_src[y, w - 1]
            base = (y >> 2) * stride + p * 128 + (y & 3) * 32
            for ch in range(nch):
                got = raw[base + ch * 512: base + ch * 512 + 32]
                assert np.array_equal(got, want[16 * ch:16 * ch + 32]), (p, y, ch)
    for q in (su, sv, dst):
        q.free()
