"""Test helper: the reference's test stream through the oracle (oracle/dirac_stream.py for the
bitstream, the C oracle for the pixel path).  TEST INFRASTRUCTURE."""
import json
import os
import sys

import numpy as np

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dirac_stream as D  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
PARAM_KEYS = sorted(["x_num_blocks", "y_num_blocks", "xblen_luma", "yblen_luma", "xbsep_luma", "ybsep_luma",
                     "mv_precision", "picture_weight_bits", "picture_weight_1", "picture_weight_2",
                     "chroma_h_shift", "chroma_v_shift"])


def load_tables():
    return json.load(open(os.path.join(GOLDEN, "arith_lut.json")))


def load_stream():
    return open(os.path.join(GOLDEN, "test_stream.drc"), "rb").read()


def decode_stream(data, tables, limit=None, quantised=False):
    """Yields per picture (coded order) a dict with everything decoded.  quantised: also
    "quant" (int32 planes of quantised values) and "codeblocks" (per component, see
    dirac_stream.Decoder.decode_coefficients): the hand-over point of device-side
    dequantisation.  The compatibility heuristic of old streams carries state from sub-band to
    sub-band, so the two decodes of a picture run on copies of the decoder."""
    qf, qo12 = O.quant_tables()
    dec = D.Decoder(tables["arith_lut"], qf, qo12, tables["schro_table_offset_3_8"])
    refs, n = {}, 0
    for code, payload in D.parse_units(data):
        if code == 0x00:
            dec.fmt = D.parse_sequence_header(payload)
        elif code & 8:
            pic = dec.parse_picture(code, payload)
            fmt = dec.fmt
            dims = [(pic.height, pic.width), (pic.ch, pic.cw), (pic.ch, pic.cw)]
            if pic.zero_residual:
                res = None
                coeffs = None
            else:
                quant = cbs = None
                if quantised:
                    import copy
                    dq = copy.copy(dec)
                    quant = dq.decode_coefficients(pic, quantised=True)
                    cbs = dq.codeblocks
                coeffs = dec.decode_coefficients(pic)
                res = [O.inverse_iwt(c, pic.depth, pic.wavelet) for c in coeffs]
            rec = dict(number=pic.number, num_refs=pic.num_refs, refs=list(pic.refs), is_ref=pic.is_ref,
                       coeffs=coeffs, zero_residual=pic.zero_residual)
            if quantised and not pic.zero_residual:
                rec.update(quant=quant, codeblocks=cbs)
            if pic.num_refs == 0:
                out = [O.convert_u8(res[k], dims[k][1], dims[k][0]) for k in range(3)]
            else:
                mv = dec.decode_block_data(pic)
                P = dict(x_num_blocks=pic.x_num_blocks, y_num_blocks=pic.y_num_blocks, xblen_luma=pic.xblen,
                         yblen_luma=pic.yblen, xbsep_luma=pic.xbsep, ybsep_luma=pic.ybsep,
                         mv_precision=pic.mv_precision, picture_weight_bits=pic.weight_bits,
                         picture_weight_1=pic.weight1, picture_weight_2=pic.weight2,
                         chroma_h_shift=fmt["h_shift"], chroma_v_shift=fmt["v_shift"])
                op = O.MotionParams(**P)
                out = []
                for k in range(3):
                    u = [O.UpComp(refs[r][k], upsample=pic.mv_precision > 0) for r in pic.refs]
                    h, w = dims[k]
                    residual = res[k] if res is not None else np.zeros((h, w), np.int16)
                    out.append(O.motion_render(mv, op, k, u[0], u[1] if len(u) > 1 else None, residual, w, h))
                rec.update(mv=mv, params=P)
            if not pic.zero_residual:
                rec.update(depth=pic.depth, wavelet=pic.wavelet)
            rec["out"] = out
            rec["md5"] = D.frame_md5(out)
            if pic.is_ref:
                refs[pic.number] = out
            yield rec
            n += 1
            if limit and n >= limit:
                return


