"""GPU parity: OBMC + residual add + clamp (HIP gather kernel, through the C ABI)
vs the CPU oracle's restatement of schro_motion_render_u8 (block scatter, edge /
interior split, aprons).  The reference asserts nothing about OBMC output
(testsuite/motion.c prints cycles only), so the case matrix follows SURVEY.md
Appendix C: chroma formats x block sets x mv_precision x weights x MV range.
"""
import numpy as np
import pytest

import oracle_lib as O
import schroedinger_amd as sa
import synth

pytestmark = pytest.mark.gpu

BLOCK_SETS = [(8, 4), (12, 8), (16, 12), (24, 16),      # schroparams.c:192-198
              (16, 8), (24, 12), (32, 16)]              # full overlap: what the reference's ENCODER makes by default (schroengine.c:411-453)


def comp_size(w, h, k, chroma):
    if k == 0:
        return w, h
    return -(-w // (1 << chroma[0])), -(-h // (1 << chroma[1]))


def run_case(ctx, w, h, xblen, xbsep, prec, weights, chroma, mv_range, seed, res_dtype=np.int16,
             modes=(0.05, 0.45, 0.15, 0.35), edit_mv=None, pair=False, only=None, yblen=None, ybsep=None):
    """pair: the chroma references are (U, V) PAIR images (include/schro_hip.h, r04) -- sub-pel precisions
    of horizontally subsampled chroma only; only: the components whose planes are rendered (default all);
    yblen, ybsep: blocks that are not square (default: as wide as high)."""
    P = synth.motion_params(w, h, xblen, xbsep, prec, weights, chroma, yblen=yblen, ybsep=ybsep)
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], mv_range, seed, modes)
    if edit_mv is not None:
        edit_mv(mv, P)
    op = O.MotionParams(**P)
    d_mv = ctx.upload_bytes(mv)
    pair = pair and prec > 0 and chroma[0] == 1
    jobs, keep, want = [], [], []
    refs_np = [[synth.picture_u8(*comp_size(w, h, k, chroma)[::-1], seed=seed + 10 * (r + 1) + k) for k in range(3)]
               for r in range(2)]
    pair_hp = []
    if pair:
        cw, ch = comp_size(w, h, 1, chroma)
        for r in range(2):
            pu, pv, g = ctx.upload(refs_np[r][1]), ctx.upload(refs_np[r][2]), ctx.hp_plane(ch, cw, pair=True)
            ctx.upsample_batch([((pu, pv), g)])
            pair_hp.append(g)
            keep += [pu, pv, g]
    for k in range(3):
        cw, ch = comp_size(w, h, k, chroma)
        r1, r2 = refs_np[0][k], refs_np[1][k]
        res = (synth.image_s(ch + 8, cw + 16, res_dtype, seed=seed + 30 + k).astype(np.int64) * 2
               ).astype(res_dtype)       # residual lives in the iwt-padded frame
        if only is not None and k not in only:
            continue
        u1, u2 = O.UpComp(r1, upsample=prec > 0), O.UpComp(r2, upsample=prec > 0)
        want.append(O.motion_render(mv, op, k, u1, u2, res, cw, ch))
        if prec == 0:
            g1, g2 = ctx.upload(r1), ctx.upload(r2)
            keep += [g1, g2]
        elif pair and k:
            g1, g2 = pair_hp
        else:
            p1, p2 = ctx.upload(r1), ctx.upload(r2)
            g1, g2 = ctx.hp_plane(ch, cw), ctx.hp_plane(ch, cw)
            ctx.upsample_batch([(p1, g1), (p2, g2)])
            keep += [p1, p2, g1, g2]
        d_res = ctx.upload(res)
        out = ctx.plane(ch, cw, np.uint8).fill(0x33)
        jobs.append(sa.obmc_plane(d_mv, P, k, g1, g2, d_res, out))
        keep += [d_res, out]
        want[-1] = (want[-1], out, k)
    ctx.obmc_batch(jobs)
    for ref, out, k in want:
        got = out.download()
        if not np.array_equal(got, ref):
            bad = np.argwhere(got != ref)
            raise AssertionError("component %d: %d mismatches, first at (y,x)=%s got %d want %d" % (
                k, len(bad), tuple(bad[0]), got[tuple(bad[0])], ref[tuple(bad[0])]))
    for p in keep + [d_mv]:
        p.free()


@pytest.mark.parametrize("chroma", [(0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("prec", [0, 1, 2, 3])
@pytest.mark.parametrize("blk", BLOCK_SETS)
def test_default_weights(ctx, blk, prec, chroma):
    for mv_range, seed in ((3 << prec, 2), (80 << prec, 3)):     # near, and far beyond the apron
        run_case(ctx, 96, 64, blk[0], blk[1], prec, (1, 1, 1), chroma, mv_range, seed)


@pytest.mark.parametrize("chroma", [(0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("prec", [0, 1, 2, 3])
@pytest.mark.parametrize("blk", [(20, 12), (20, 16), (28, 16), (28, 24)])
def test_block_lengths_between_the_presets(ctx, blk, prec, chroma):
    """r06: rows of 20 and 28 bytes (lengths the syntax allows, schroparams.c:255-270, and no preset uses) as two segments of 10 / 14 in
    the 12- / 16-byte segment kernels -- a segment's weights beyond its length are zero; with pair images their 10- / 14-sample chroma
    rows the same way; a size with interior tiles and tile edges inside blocks."""
    run_case(ctx, 96, 64, blk[0], blk[1], prec, (1, 1, 1), chroma, 80 << prec, 3)
    run_case(ctx, 300, 150, blk[0], blk[1], prec, (1, 1, 1), chroma, 24 << prec, 5, pair=True, yblen=blk[0] - 4, ybsep=blk[1] - 4)


@pytest.mark.parametrize("weights", [(3, 5, 3), (1, 2, 2), (2, 3, 1), (3, -1, 1), (5, 3, 2), (1, 3, 2), (7, 1, 3), (0, 1, 0), (21, 43, 6)])
@pytest.mark.parametrize("prec", [0, 2, 3])
def test_weighted_prediction(ctx, weights, prec):
    # (r06: non-negative weights that add up to 1 << bits -- (3, 5, 3), (1, 3, 2), (7, 1, 3), (0, 1, 0), (21, 43, 6): the fades --
    # run the row kernels' weighted blend; the others obmc.hip's general kernel)
    # includes gain > 1 and a negative weight: the edge (u8) and interior (s16) block
    # arithmetic of the reference differ there and both must be reproduced
    for blk in ((12, 8), (16, 12)):
        run_case(ctx, 96, 64, blk[0], blk[1], prec, weights, (1, 1), 24 << prec, 5)


def test_picture_weight_bits_zero(ctx):
    # bits 0: weights that sum to 1 go through (no rounding term anywhere); any other sum would
    # need the reference's 1 << (bits - 1) with bits 0 (schromotion8.c:391-397) and is refused
    for prec in (0, 2):
        run_case(ctx, 96, 64, 12, 8, prec, (1, 0, 0), (1, 1), 24 << prec, 11)
        run_case(ctx, 96, 64, 12, 8, prec, (2, -1, 0), (1, 1), 24 << prec, 12)
    with pytest.raises(sa.SchroHipError, match="picture_weight_bits 0"):
        run_case(ctx, 96, 64, 12, 8, 0, (2, 1, 0), (1, 1), 24, 13)


def test_dc_values_outside_8_bits(ctx):
    # A DC value outside [-128, 127] makes the reference's s16 accumulator wrap (interior
    # blocks multiply the 16-bit dc + 128; edge blocks store it into a uint8_t).  No legal
    # stream has one, the arithmetic is still the reference's.  The kernel packs two rows
    # per accumulator word and must not let such a sum carry into the neighbour row: wide
    # blocks early (first chunk of a tile) and late (chroma tiles span several chunks).
    def widen(where):
        def edit(mv, P):
            nbx, nby = P["x_num_blocks"], P["y_num_blocks"]
            dc_blocks = np.flatnonzero((mv["flags"] & 3) == 0)
            pick = dc_blocks[::5] if where == "all" else dc_blocks[dc_blocks >= (nby - 3) * nbx][::2]
            vals = np.array([300, -400, 1000, -3000, 32767, -32768, 127 + 129, -129], np.int16)
            for n, b in enumerate(pick):
                mv["v"][b, :3] = vals[(n + np.arange(3)) % len(vals)]
        return edit
    for where in ("all", "last-rows"):
        for prec in (0, 2):
            run_case(ctx, 320, 128, 12, 8, prec, (1, 1, 1), (1, 1), 8 << prec, 7,
                     modes=(0.3, 0.3, 0.1, 0.3), edit_mv=widen(where))
    run_case(ctx, 96, 64, 8, 4, 1, (1, 1, 1), (0, 0), 6, 9, modes=(0.5, 0.2, 0.1, 0.2), edit_mv=widen("all"))


def test_ragged_sizes(ctx):
    # picture sizes that are not multiples of the block separation, and a tile edge
    for (w, h) in [(100, 70), (97, 61), (64, 36), (130, 18), (72, 132)]:
        for prec in (0, 2):
            run_case(ctx, w, h, 12, 8, prec, (1, 1, 1), (1, 1), 40, 7)


def test_large_and_odd_block_geometries(ctx):
    # legal block sizes (schro_params_verify_block_params: multiples of 4, sep <= len <=
    # 2 sep) outside the four standard sets: up to SCHRO_LIMIT_BLOCK_SIZE (64), blocks whose
    # (rows x segments) exceed the item kernel's weight table (exact rim path for all of
    # them), chroma widths that are not a multiple of 4, len == sep (no overlap), len == 2 sep
    for (blen, bsep) in [(32, 24), (48, 32), (64, 32), (20, 12), (8, 8), (16, 8), (24, 12), (4, 4)]:
        for prec in (0, 1, 2):
            run_case(ctx, 192, 128, blen, bsep, prec, (1, 1, 1), (1, 1), 20 << prec, 13)


def test_all_modes_and_s32_residual(ctx):
    for modes in ((1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)):
        run_case(ctx, 96, 64, 12, 8, 2, (1, 1, 1), (1, 1), 64, 11, modes=modes)
    run_case(ctx, 96, 64, 12, 8, 2, (1, 1, 1), (1, 0), 64, 12, res_dtype=np.int32)


def test_test_stream_geometry(ctx):
    # BASELINE config 2 geometry: 320x240 4:2:2, 12x12/8x8 blocks, full-pel, 40x32 blocks
    run_case(ctx, 320, 240, 12, 8, 0, (1, 1, 1), (1, 0), 16, 21)


def test_2160p_config(ctx):
    # BASELINE config 3: 3840x2160 4:2:0, 12x12/8x8, quarter-pel, MVs +-64 quarter-pels
    run_case(ctx, 3840, 2160, 12, 8, 2, (1, 1, 1), (1, 1), 64, 2)


def test_illegal_block_parameters_are_refused(ctx):
    # schro_params_verify_block_params (schroparams.c:241-272): the reference never renders
    # these; neither does the library (an error, not a silently different picture)
    for (blen, bsep) in [(10, 8), (12, 6), (8, 12), (40, 16)]:
        with pytest.raises(sa.SchroHipError):
            run_case(ctx, 96, 64, blen, bsep, 0, (1, 1, 1), (1, 1), 4, 3)


def test_rotating_references_in_a_batch(ctx):
    # A decoder's references move through a frame pool: the same batch shape comes back with
    # other reference pointers every GOP.  The tile-order table is keyed on which jobs SHARE
    # a reference, not on addresses: three launches of a two-picture batch with the
    # references rotated through three buffers, each checked against the oracle.
    w, h, prec = 320, 128, 2
    P = synth.motion_params(w, h, 12, 8, prec, (1, 1, 1), (1, 1))
    op = O.MotionParams(**P)
    dims = [comp_size(w, h, k, (1, 1)) for k in range(3)]
    pics = [[synth.picture_u8(ch, cw, seed=40 + 10 * r + k) for k, (cw, ch) in enumerate(dims)]
            for r in range(3)]
    ups = [[O.UpComp(p, upsample=True) for p in comps] for comps in pics]
    hp = []
    for comps in pics:
        row = []
        for p in comps:
            d, g = ctx.upload(p), ctx.hp_plane(*p.shape)
            ctx.upsample_batch([(d, g)])
            row.append(g)
        hp.append(row)
    mvs = [synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 40, seed=60 + f) for f in range(2)]
    d_mvs = [ctx.upload_bytes(m) for m in mvs]
    res = [[synth.image_s(ch, cw, np.int16, seed=70 + 3 * f + k) for k, (cw, ch) in enumerate(dims)]
           for f in range(2)]
    d_res = [[ctx.upload(r) for r in rf] for rf in res]
    for rot in range(3):
        jobs, outs = [], []
        for f in range(2):
            a, b = (rot + f) % 3, (rot + f + 1) % 3
            for k, (cw, ch) in enumerate(dims):
                out = ctx.plane(ch, cw, np.uint8).fill(0x55)
                jobs.append(sa.obmc_plane(d_mvs[f], P, k, hp[a][k], hp[b][k], d_res[f][k], out))
                outs.append((out, O.motion_render(mvs[f], op, k, ups[a][k], ups[b][k], res[f][k], cw, ch)))
        ctx.obmc_batch(jobs)
        for n, (out, want) in enumerate(outs):
            assert np.array_equal(out.download(), want), "rotation %d plane %d" % (rot, n)


# ---- r04: chroma references as (U, V) pair images ---------------------------------------------------

@pytest.mark.parametrize("chroma", [(1, 0), (1, 1)])
@pytest.mark.parametrize("prec", [1, 2, 3])
@pytest.mark.parametrize("blk", BLOCK_SETS)
def test_pair_images_default_weights(ctx, blk, prec, chroma):
    # half / quarter pel with chroma rows of up to 8 samples: the U and V planes as ONE job of the row kernel's
    # UV form (one fetch per tap for both); 8/4 blocks (too many per tile), 24/16 (rows of 12 samples)
    # and eighth pel: the item kernel reads each component out of the pair images
    for mv_range, seed in ((3 << prec, 2), (80 << prec, 3)):
        run_case(ctx, 96, 64, blk[0], blk[1], prec, (1, 1, 1), chroma, mv_range, seed, pair=True)


@pytest.mark.parametrize("weights", [(3, 5, 3), (2, 3, 1), (3, -1, 1)])
def test_pair_images_weighted_prediction(ctx, weights):
    # other picture weights: the per-pixel kernel, per component out of the pair images
    for prec in (2, 3):
        run_case(ctx, 96, 64, 12, 8, prec, weights, (1, 1), 24 << prec, 5, pair=True)


def test_pair_images_one_component_alone(ctx):
    # a U or a V plane on its own (no partner next to it in the batch): read out of the pair images
    for only in ((1,), (2,), (0, 2)):
        for prec in (1, 2):
            run_case(ctx, 160, 96, 12, 8, prec, (1, 1, 1), (1, 1), 24 << prec, 17, pair=True, only=only)


def test_pair_images_edges_and_odd_cases(ctx):
    # ragged picture sizes (tiles cut by the picture's right / bottom edge, chroma widths that are not a
    # multiple of 8), DC values outside 8 bits (exact adds, per-sample rim path of the UV form), every single
    # prediction mode, an s32 residual (plain finish), block geometries outside the standard sets
    for (w, h) in [(100, 70), (97, 61), (64, 36), (130, 18), (72, 132), (260, 66)]:
        run_case(ctx, w, h, 12, 8, 2, (1, 1, 1), (1, 1), 40, 7, pair=True)
        run_case(ctx, w, h, 16, 12, 1, (1, 1, 1), (1, 0), 20, 8, pair=True)

    def widen(mv, P):
        dc_blocks = np.flatnonzero((mv["flags"] & 3) == 0)
        vals = np.array([300, -400, 1000, -3000, 32767, -32768, 127 + 129, -129], np.int16)
        for n, b in enumerate(dc_blocks[::5]):
            mv["v"][b, :3] = vals[(n + np.arange(3)) % len(vals)]
    for prec in (1, 2):
        run_case(ctx, 320, 128, 12, 8, prec, (1, 1, 1), (1, 1), 8 << prec, 7, modes=(0.3, 0.3, 0.1, 0.3),
                 edit_mv=widen, pair=True)
    for modes in ((1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)):
        run_case(ctx, 96, 64, 12, 8, 2, (1, 1, 1), (1, 1), 64, 11, modes=modes, pair=True)
    run_case(ctx, 96, 64, 12, 8, 2, (1, 1, 1), (1, 0), 64, 12, res_dtype=np.int32, pair=True)
    for (blen, bsep) in [(32, 24), (20, 12), (8, 8), (16, 8), (24, 12), (4, 4), (16, 16)]:
        for prec in (1, 2):
            run_case(ctx, 192, 128, blen, bsep, prec, (1, 1, 1), (1, 1), 20 << prec, 13, pair=True)


def test_pair_images_2160p_config(ctx):
    # BASELINE config 3 as bench.py runs it: chroma from pair images
    run_case(ctx, 3840, 2160, 12, 8, 2, (1, 1, 1), (1, 1), 64, 2, pair=True)


def test_references_of_another_layout_are_refused(ctx):
    # a plain plane or a one-component image where a pair image is declared (and the reverse) would be
    # read out of bounds: the strides tell them apart
    P = synth.motion_params(96, 64, 12, 8, 2, (1, 1, 1), (1, 1))
    d_mv = ctx.upload_bytes(synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 8, 1))
    res, out = ctx.plane(32, 48, np.int16), ctx.plane(32, 48, np.uint8)
    single, pair_img, plain = ctx.hp_plane(32, 48), ctx.hp_plane(32, 48, pair=True), ctx.plane(32, 48, np.uint8)
    job = sa.obmc_plane(d_mv, P, 1, single, single, res, out)
    job.ref_pair = 1
    with pytest.raises(sa.SchroHipError, match="not a half-pel image"):
        ctx.obmc_batch([job])
    with pytest.raises(sa.SchroHipError, match="not a half-pel image"):
        ctx.obmc_batch([sa.obmc_plane(d_mv, P, 1, plain, plain, res, out)])
    for p in (d_mv, res, out, single, pair_img, plain):
        p.free()


def test_more_than_1024_blocks_meet_a_tile(ctx):
    """r04 regression (found by the fuzz campaign): 8 x 8 luma blocks every 4 pixels are 4 x 4 chroma blocks every 2 -- a
    128 x 32 tile of a 4:2:0 chroma plane meets 62 x 18 = 1116 of them, and the item kernel's 16-bit block-row division was
    exact only below 2^16 / 62 = 1057: the tile's last block was decoded a row down and a column left of the grid (one
    pixel of the plane's last column wrong).  Plain and half-pel references, both kernels' routes."""
    for prec in (0, 1, 2):
        run_case(ctx, 244, 150, 8, 4, prec, (1, 1, 1), (1, 1), 20 << prec, 47753)
        run_case(ctx, 260, 136, 8, 4, prec, (1, 1, 1), (1, 0), 20 << prec, 11, pair=prec > 0)
    run_case(ctx, 250, 140, 4, 4, 0, (1, 1, 1), (1, 1), 16, 3)


def test_more_block_geometries_in_a_call_than_a_table_slot_holds(ctx):
    """r05: the row kernels copy their weight table in from a table the host makes per block geometry (one slot of 64 KB per
    launch: 35 tables of 16-pixel rows).  48 luma planes of 48 different geometries in ONE call go out in two launches."""
    w, h, prec = 72, 56, 2
    geos = [(16, xbsep, yblen, ybsep) for xbsep in (8, 12, 16) for ybsep in (4, 8, 12, 16, 20, 24, 28, 32)
            for yblen in range(ybsep, min(2 * ybsep, 32) + 1, 4)][:48]
    assert len(set(geos)) == 48
    r1, r2 = synth.picture_u8(h, w, seed=3), synth.picture_u8(h, w, seed=4)
    u1, u2 = O.UpComp(r1, upsample=True), O.UpComp(r2, upsample=True)
    p1, p2 = ctx.upload(r1), ctx.upload(r2)
    g1, g2 = ctx.hp_plane(h, w), ctx.hp_plane(h, w)
    ctx.upsample_batch([(p1, g1), (p2, g2)])
    jobs, want, keep = [], [], [p1, p2, g1, g2]
    for n, (xblen, xbsep, yblen, ybsep) in enumerate(geos):
        P = synth.motion_params(w, h, xblen, xbsep, prec, (1, 1, 1), (1, 1), yblen=yblen, ybsep=ybsep)
        mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 24, 100 + n)
        res = synth.image_s(h, w, np.int16, seed=200 + n)
        d_mv, d_res, out = ctx.upload_bytes(mv), ctx.upload(res), ctx.plane(h, w, np.uint8).fill(0x44)
        jobs.append(sa.obmc_plane(d_mv, P, 0, g1, g2, d_res, out))
        want.append((O.motion_render(mv, O.MotionParams(**P), 0, u1, u2, res, w, h), out, (xblen, yblen, ybsep)))
        keep += [d_mv, d_res, out]
    ctx.profile_enable(True)
    ctx.profile_reset()
    ctx.obmc_batch(jobs)
    ctx.synchronize()
    launches = ctx.profile_read()["obmc"][1]
    ctx.profile_enable(False)
    # (38 of the geometries take the 16-pixel row kernel -- 35 + 3 --, the ten with the most rows per tile the item kernel)
    assert launches >= 3, launches
    for ref, out, geo in want:
        assert np.array_equal(out.download(), ref), geo
    for p in keep:
        p.free()


@pytest.mark.parametrize("prec", [0, 2, 3])
@pytest.mark.parametrize("weights", [(1, 1, 1), (3, -1, 1), (5, 3, 2)])
def test_prediction_into_an_s16_plane(ctx, prec, weights):
    """r06, prediction_only = 2 -- schro_motion_render (add = FALSE) / schro_motion_render_cuda's dest: the s16 plane
    receives (acc - 8160) >> 6 (orc_rrshift6_s16_ip_2d: the prediction - 128) for ANY weights and DC values, the
    reference's 16-bit wrap included; checked against the oracle's accumulator through the reference's own compiled
    kernel (oracle_lib.rrshift6_s16)."""
    w, h, chroma = 136, 72, (1, 1)
    P = synth.motion_params(w, h, 12, 8, prec, weights, chroma)
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 24 << prec, 31, (0.2, 0.3, 0.2, 0.3))
    dc_blocks = np.flatnonzero((mv["flags"] & 3) == 0)
    mv["v"][dc_blocks[::3], :3] = np.array([300, -400, 32767], np.int16)      # DC values outside 8 bits wrap as the reference's
    d_mv = ctx.upload_bytes(mv)
    jobs, want, keep = [], [], [d_mv]
    for k in range(3):
        cw, ch = comp_size(w, h, k, chroma)
        r1, r2 = (synth.picture_u8(ch, cw, seed=50 + 10 * r + k) for r in range(2))
        _, acc = O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(r1, upsample=prec > 0), O.UpComp(r2, upsample=prec > 0),
                                 np.zeros((ch, cw), np.int16), cw, ch, return_acc=True)
        want.append(O.rrshift6_s16(acc))
        if prec == 0:
            g1, g2 = ctx.upload(r1), ctx.upload(r2)
        else:
            p1, p2 = ctx.upload(r1), ctx.upload(r2)
            g1, g2 = ctx.hp_plane(ch, cw), ctx.hp_plane(ch, cw)
            ctx.upsample_batch([(p1, g1), (p2, g2)])
            keep += [p1, p2]
        out = ctx.plane(ch + 3, cw + 5, np.int16).fill(0x11)      # (a larger plane: the rest stays as it was)
        out.width, out.height = cw, ch
        jobs.append(sa.obmc_plane(d_mv, P, k, g1, g2, None, out, prediction_only=2))
        keep += [g1, g2, out]
    ctx.obmc_batch(jobs)
    outs = [x for x in keep if isinstance(x, sa.DevicePlane) and x.dtype == np.int16]
    for k, out in enumerate(outs):
        cw, ch = comp_size(w, h, k, chroma)
        out.width, out.height = cw + 5, ch + 3
        got = out.download()
        assert np.array_equal(got[:ch, :cw], want[k]), k
        assert (got[ch:, :] == 0x1111).all() and (got[:, cw:] == 0x1111).all()
    for p in keep:
        p.free()


def test_add_batch(ctx):
    """r06: dst (s16) += src (s16 | u8), schro_frame_add's two cases, 16-bit wrap, ragged sizes and unaligned rows."""
    rng = np.random.default_rng(11)
    for (h, w) in ((48, 64), (37, 53), (5, 7), (270, 1920)):
        for sdt in (np.int16, np.uint8):
            d = rng.integers(-32768, 32768, (h, w)).astype(np.int16)
            s = (rng.integers(-32768, 32768, (h + 2, w + 3)).astype(np.int16) if sdt == np.int16
                 else rng.integers(0, 256, (h + 2, w + 3)).astype(np.uint8))
            gd, gs = ctx.upload(d), ctx.upload(s)
            ctx.add_batch([(gd, gs)])
            assert np.array_equal(gd.download(), O.frame_add(d, s)), (h, w, sdt)
            gd.free()
            gs.free()


def test_fades_on_every_row_form(ctx):
    """r06: normalised non-negative picture weights (a fade) through every form of the row kernels -- pair images and
    plain planes, every precision, every preset and every default of the reference's encoder (8 / 4 ... 32 / 16) and the lengths
    between them (20, 28), edge-class blocks (vectors far outside), the residual form and (tests/test_gpu_combine.py covers it
    too) prediction-only."""
    for weights in ((3, 5, 3), (1, 3, 2), (63, 1, 6)):
        for blk in ((8, 4), (12, 8), (16, 12), (24, 16), (16, 8), (24, 12), (32, 16), (20, 12), (28, 16)):
            for prec in (0, 1, 2, 3):
                for chroma in ((1, 1), (0, 0)):
                    run_case(ctx, 136, 72, blk[0], blk[1], prec, weights, chroma, 40 << prec, 17, pair=True)


def test_the_reference_encoders_default_block_sets_at_size(ctx):
    """r06: the reference's encoder picks the block separation by picture size and, by default, FULL overlap
    (schroengine.c:411-453): 16 / 8 below 960 x 540, 24 / 12 below 1080p, 32 / 16 from 1080p on -- with mv_precision 0.  Pictures
    large enough for whole interior tiles, every precision, 4:2:0 (pair images where there are half-pel images) and 4:4:4."""
    for (w, h, blk) in ((416, 240, (16, 8)), (704, 200, (24, 12)), (640, 168, (32, 16))):
        for prec in (0, 2, 3):
            for chroma in ((1, 1), (0, 0)):
                run_case(ctx, w, h, blk[0], blk[1], prec, (1, 1, 1), chroma, 20 << prec, 23, pair=True)


def test_dc_values_outside_8_bits_in_two_segment_blocks(ctx):
    """r06: a block of the 24 / 16, 24 / 12 or 32 / 16 set is two records in the row kernels (its segments); one whose DC value
    does not fit 8 bits takes the rim path as a WHOLE block from its first segment's record, with the exact 16-bit
    accumulation for the tile -- pair images and plain planes, blocks at the picture's rim and inside."""
    def widen(mv, P):
        dc_blocks = np.flatnonzero((mv["flags"] & 3) == 0)
        vals = np.array([300, -400, 1000, -3000, 32767, -32768, 256, -129], np.int16)
        for n, b in enumerate(dc_blocks[::3]):
            mv["v"][b, :3] = vals[(n + np.arange(3)) % len(vals)]
    for blk in ((24, 16), (24, 12), (32, 16)):
        for prec in (0, 2):
            run_case(ctx, 416, 160, blk[0], blk[1], prec, (1, 1, 1), (1, 1), 12 << prec, 29,
                     modes=(0.3, 0.3, 0.1, 0.3), edit_mv=widen, pair=True)
