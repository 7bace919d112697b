"""A compiled C program (tests/c/stage_replay.c) calls the frame layer the way a patched
schrodecoder.c would -- SchroMemoryDomain-shaped HIP domain, SchroFrame / SchroParams /
SchroMotion-shaped structs, stages in the decoder's order -- and its decoded picture must equal
the oracle's.  (the boundary needs a C caller, not only ctypes.)"""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "c", "_build", "stage_replay")


def build_harness():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "c")])
    return EXE


@pytest.mark.parametrize("w,h,chroma,prec,blk,filt,depth", [
    (176, 144, (1, 1), 2, (12, 8), 0, 3),       # CIF-ish 4:2:0, quarter-pel, DD(9,7)
    (320, 240, (1, 0), 0, (12, 8), 1, 4),       # the test stream's shape: 4:2:2, full-pel, LeGall depth 4
    (200, 120, (0, 0), 3, (16, 12), 3, 2),      # 4:4:4, eighth-pel, Haar
])
def test_c_caller_decodes_an_inter_picture(tmp_path, w, h, chroma, prec, blk, filt, depth):
    exe = build_harness()
    hs, vs = chroma
    P = synth.motion_params(w, h, blk[0], blk[1], prec, (1, 1, 1), chroma)
    cw, ch = -(-w >> hs), -(-h >> vs)
    up = lambda v: -(-v // (1 << depth)) * (1 << depth)
    iw = [(up(h), up(w)), (up(ch), up(cw)), (up(ch), up(cw))]
    pd = [(h, w), (ch, cw), (ch, cw)]
    resid = [synth.image_s(ih, iwd, np.int16, seed=3 + k) for k, (ih, iwd) in enumerate(iw)]
    coeffs = [O.forward_iwt(r, depth, filt) for r in resid]
    refs = [[synth.picture_u8(ph, pw, seed=11 + 10 * r + k) for k, (ph, pw) in enumerate(pd)] for r in range(2)]
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 20 << prec, seed=5)
    d = tmp_path
    (d / "params.txt").write_text(" ".join(str(v) for v in [
        w, h, hs, vs, iw[0][1], iw[0][0], iw[1][1], iw[1][0], depth, filt, prec, blk[0], blk[0], blk[1], blk[1],
        P["x_num_blocks"], P["y_num_blocks"]]))
    np.concatenate([c.ravel() for c in coeffs]).tofile(d / "coeffs.bin")
    for r in range(2):
        np.concatenate([p.ravel() for p in refs[r]]).tofile(d / ("ref%d.bin" % r))
    mv.tofile(d / "mvs.bin")
    p = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    out = np.fromfile(d / "out.bin", np.uint8)
    res = np.fromfile(d / "residual.bin", np.int16)
    o = r0 = 0
    for k, ((ph, pw), (ih, iwd)) in enumerate(zip(pd, iw)):
        want_res = O.inverse_iwt(coeffs[k], depth, filt)
        got_res = res[r0:r0 + ih * iwd].reshape(ih, iwd)
        r0 += ih * iwd
        assert np.array_equal(got_res, want_res), "component %d residual" % k
        want = O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(refs[0][k], upsample=prec > 0),
                               O.UpComp(refs[1][k], upsample=prec > 0), want_res, pw, ph)
        got = out[o:o + ph * pw].reshape(ph, pw)
        o += ph * pw
        assert np.array_equal(got, want), "component %d picture" % k


def test_c_caller_decodes_a_sequence_both_ways(tmp_path):
    """tests/c/stage_loop.c: a sequence of inter pictures through the frame layer (i) under the reference's
    contract -- one picture at a time, every stage call complete on return, host frames and vectors --,
    (ii) pipelined as INTEGRATION.md 3a describes (stage completion off, pinned host frames, copy queues,
    five pictures in flight, marks), and (iii, r05) pipelined with the QUANTISED hand-over: codeblock records + a blob of
    quantised values per picture through schro_hipframe_dequantise (schrodecoder.c:3525-3640's data-parallel half on the
    device).  The dense coefficient frames of (i) and (ii) are the quantised sets dequantised by the library: they must equal
    the oracle's dequantisation of the dumped records and values.  Every picture of the three passes equals the
    oracle's.  (bench.py runs the same program at 2160p for its frame_layer_2160p figures.)"""
    import json
    import schroedinger_amd as sa
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "c")])
    exe = os.path.join(ROOT, "tests", "c", "_build", "stage_loop")
    w, h, npic, depth, filt = 320, 192, 19, 3, 0
    p = subprocess.run([exe, str(tmp_path), str(w), str(h), str(npic), "1"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["passes_agree"] and line["pictures"] == npic and line["pictures_in_flight"] == 5
    P = synth.motion_params(w, h, 12, 8, 2, (1, 1, 1), (1, 1))
    dims = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]

    def planes(name, dtype):
        a = np.fromfile(tmp_path / name, dtype)
        out, o = [], 0
        for (ph, pw) in dims:
            out.append(a[o:o + ph * pw].reshape(ph, pw))
            o += ph * pw
        assert o == a.size
        return out
    coeffs = [planes("coeffs%d.bin" % s, np.int16) for s in range(4)]
    # the quantised sets -> the oracle's dequantisation (oracle_dequant.c: schrodecoder.c:3060-3083, the arithmetic-coded
    # path's C int arithmetic; inter pictures: schro_table_offset_3_8) must give exactly those coefficient frames
    CB = np.dtype([("dst_offset", "<i4"), ("dst_stride", "<i4"), ("width", "<i4"), ("height", "<i4"), ("src_offset", "<i4"),
                   ("src_bytes", "u1"), ("quant_index", "u1"), ("pad", "u1", 2)])
    assert CB.itemsize == 24
    n_zero = n_all = 0
    for s in range(4):
        blob = np.fromfile(tmp_path / ("qblob%d.bin" % s), np.uint8)
        off = 0
        for c, (ph, pw) in enumerate(dims):
            recs = np.fromfile(tmp_path / ("qrec%d_%d.bin" % (s, c)), CB)
            assert len(recs) == 8 * 8 * (1 + 3 * depth)
            stride = (pw * 2 + 15) & ~15                # the C caller's frames: ROUND_UP_16 (width * 2), schroframe.c:60-191
            plane = np.zeros((ph, stride // 2), np.int16)
            # the component's values start at the next multiple of 256 of the blob (the C caller's layout)
            off = (off + 255) & ~255
            used = 0
            for r in recs:
                n_all += 1
                if r["width"] == 0 or r["height"] == 0:
                    continue
                y0, x0 = divmod(int(r["dst_offset"]), stride)
                x0 //= 2
                step = int(r["dst_stride"]) // stride
                dst = plane[y0:y0 + step * int(r["height"]):step, x0:x0 + int(r["width"])]
                q = None
                if r["src_offset"] >= 0:
                    dt = {1: np.int8, 2: np.int16}[int(r["src_bytes"])]
                    nb = int(r["width"]) * int(r["height"]) * int(r["src_bytes"])
                    a = off + int(r["src_offset"])
                    q = blob[a:a + nb].view(dt).reshape(int(r["height"]), int(r["width"]))
                    used = max(used, int(r["src_offset"]) + nb)
                else:
                    n_zero += 1
                O.dequant_codeblock(dst, q, int(r["quant_index"]), False, 0)
            off += used
            assert np.array_equal(plane[:, :pw], coeffs[s][c]), ("dequantised coefficients", s, c)
    assert 0.2 < n_zero / n_all < 0.7, (n_zero, n_all)
    mvs = [np.fromfile(tmp_path / ("mvs%d.bin" % s), sa.MV_DTYPE) for s in range(4)]
    assert mvs[0].size == P["x_num_blocks"] * P["y_num_blocks"]
    res = [[O.inverse_iwt(c, depth, filt) for c in cs] for cs in coeffs]
    ngroups = -(-npic // 8)
    ups = [[O.UpComp(pl) for pl in planes("ref%d.bin" % r, np.uint8)] for r in range(2 * ngroups)]
    for k in range(npic):
        g, s = k // 8, k % 4
        want = [O.motion_render(mvs[s], O.MotionParams(**P), c, ups[2 * g][c], ups[2 * g + 1][c], res[s][c], dims[c][1], dims[c][0])
                for c in range(3)]
        for mode in ("contract", "pipelined", "quantised"):
            got = planes("out_%s%d.bin" % (mode, k), np.uint8)
            for c in range(3):
                assert np.array_equal(got[c], want[c]), (mode, k, c)
