"""Synthetic inputs shared by tests/ and bench.py.

LCG x <- x*1103515245 + 12345 (mod 2^32), value = x >> 16.
"""
import numpy as np

MV_DTYPE = np.dtype([("flags", "<u4"), ("metric", "<u4"), ("chroma_metric", "<u4"),
                     ("v", "<i2", (4,))])

_A = np.uint32(1103515245)
_C = np.uint32(12345)


def lcg(n, seed=1):
    """n successive LCG outputs (x >> 16, 16 bits each) as uint32."""
    with np.errstate(over="ignore"):
        apow = np.cumprod(np.full(n, _A, dtype=np.uint32), dtype=np.uint32)      # a^1..a^n
        geo = np.concatenate(([np.uint32(1)], apow[:-1]))                        # a^0..a^(n-1)
        s = np.cumsum(geo, dtype=np.uint32)                                      # sum_{k<n} a^k
        x = apow * np.uint32(seed) + _C * s
    return (x >> np.uint32(16)).astype(np.uint32)


def image_s(h, w, dtype=np.int16, seed=1):
    """(v & 0xff) - 128 per sample: legal 8-bit residual-like picture."""
    v = lcg(h * w, seed)
    return ((v & 0xff).astype(np.int32) - 128).astype(dtype).reshape(h, w)


def full_range(h, w, dtype=np.int16, seed=3):
    """Uniform full-range samples: pins the wrap semantics (parity only)."""
    v = lcg(2 * h * w, seed).astype(np.uint64)
    if np.dtype(dtype) == np.int16:
        return v[: h * w].astype(np.uint16).view(np.int16).reshape(h, w)
    lo, hi = v[0::2], v[1::2]
    return ((hi << np.uint64(16)) | lo).astype(np.uint32).view(np.int32).reshape(h, w)


def picture_u8(h, w, seed=5, blur=True):
    """LCG u8 noise, optionally 3x3 box-blurred (reference-frame-like)."""
    p = (lcg(h * w, seed) & 0xff).astype(np.int32).reshape(h, w)
    if blur:
        q = np.pad(p, 1, mode="edge")
        p = sum(q[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)) // 9
    return p.astype(np.uint8)


def num_blocks(width, height, xbsep, ybsep):
    """schro_params_calculate_mc_sizes, schroparams.c:165-185."""
    return 4 * -(-width // (4 * xbsep)), 4 * -(-height // (4 * ybsep))


def motion_field(nbx, nby, mv_range=64, seed=2, modes=(0.05, 0.45, 0.15, 0.35), global_bits=False):
    """pred_mode drawn from `modes`, dx/dy uniform in [-mv_range, mv_range],
    dc uniform in [-128, 127]; split = 2."""
    n = nbx * nby
    r = lcg(6 * n, seed).astype(np.int64).reshape(6, n)
    u = (r[0] % 1000) / 1000.0
    cum = np.cumsum(modes)
    mode = np.searchsorted(cum, u, side="right").clip(0, 3).astype(np.uint32)
    mv = np.zeros(n, MV_DTYPE)
    mv["flags"] = mode | (2 << 3)
    span = 2 * mv_range + 1
    vec = np.stack([r[1] % span - mv_range, r[2] % span - mv_range,
                    r[3] % span - mv_range, r[4] % span - mv_range], axis=1).astype(np.int16)
    dc = np.stack([r[1] % 256 - 128, r[2] % 256 - 128, r[3] % 256 - 128,
                   np.zeros(n, np.int64)], axis=1).astype(np.int16)
    mv["v"] = np.where((mode == 0)[:, None], dc, vec)
    mv["metric"] = r[5].astype(np.uint32)
    return mv


def motion_params(width, height, xblen, xbsep, prec, weights=(1, 1, 1), chroma=(1, 1),
                  yblen=None, ybsep=None):
    yblen = xblen if yblen is None else yblen
    ybsep = xbsep if ybsep is None else ybsep
    nbx, nby = num_blocks(width, height, xbsep, ybsep)
    return dict(x_num_blocks=nbx, y_num_blocks=nby, xblen_luma=xblen, yblen_luma=yblen,
                xbsep_luma=xbsep, ybsep_luma=ybsep, mv_precision=prec,
                picture_weight_bits=weights[2], picture_weight_1=weights[0],
                picture_weight_2=weights[1], chroma_h_shift=chroma[0], chroma_v_shift=chroma[1])


# ---- VC-2 low-delay pictures ------------------------------------------------------

# schro_tables_lowdelay_quants is the host's business (the ABI takes the matrix); these are
# plausible matrices of the right length
def lowdelay_params(width, height, chroma, depth, slice_w, slice_h, slice_bytes_num, slice_bytes_denom=1,
                    quant_matrix=None):
    """Parameter dict of a low-delay picture: iwt sizes padded to 2^depth
    (schro_params_calculate_iwt_sizes, schroparams.c:121-140), slices of slice_w x slice_h
    luma samples (counts rounded up)."""
    cw, ch = -(-width // (1 << chroma[0])), -(-height // (1 << chroma[1]))
    rnd = lambda v: -(-v // (1 << depth)) * (1 << depth)
    if quant_matrix is None:
        quant_matrix = [0] + [q for lvl in range(depth) for q in (min(4 + 2 * lvl, 12),) * 2 + (min(6 + 2 * lvl, 14),)]
    return dict(transform_depth=depth, iwt_luma_width=rnd(width), iwt_luma_height=rnd(height),
                iwt_chroma_width=rnd(cw), iwt_chroma_height=rnd(ch),
                n_horiz_slices=max(1, rnd(width) // slice_w), n_vert_slices=max(1, rnd(height) // slice_h),
                slice_bytes_num=slice_bytes_num, slice_bytes_denom=slice_bytes_denom,
                quant_matrix=list(quant_matrix))


def quantised_planes(P, seed=1, scale=3.0, big_every=0, big_range=1 << 31):
    """Laplacian-ish quantised coefficients (mostly 0 / +-1, as in a real stream) as int32
    planes in the frame layout; big_every > 0 sprinkles values below big_range that need
    the long exp-Golomb path / wrap in 16 bits."""
    out = []
    for k in range(3):
        w = P["iwt_chroma_width"] if k else P["iwt_luma_width"]
        h = P["iwt_chroma_height"] if k else P["iwt_luma_height"]
        r = lcg(2 * h * w, seed + 7 * k).astype(np.int64)
        u = (r[: h * w] + 1) / 65537.0
        mag = np.floor(-scale * np.log(u)).astype(np.int64)
        v = np.where(r[h * w:] & 1, -mag, mag)
        if big_every:
            idx = np.arange(0, h * w, big_every)
            big = ((r[idx] * 2654435761) >> 3) % big_range
            v[idx] = np.where(r[idx] & 2, -big, big)
        out.append(v.astype(np.int32).reshape(h, w))
    return out


def lowdelay_base_index(P, seed=1, lo=0, hi=40):
    n = P["n_horiz_slices"] * P["n_vert_slices"]
    return (lo + lcg(n, seed + 99) % (hi - lo + 1)).astype(np.uint8)


# ---- VC-2 low-delay slices without the oracle (bench.py's lowdelay_8k key) ------------------------
# The slice syntax as the reference READS it (schro_decoder_decode_slice_slow(_s32),
# schrolowdelay.c:110-270; schro_unpack_decode_uint / _sint, schrounpack.c:214-245): 7 bits base
# index, ilog2up (8 * slice_bytes) bits luma length, the luma codes of sub-bands 0 .. 3 * depth (each
# the slice's rectangle, row-major), then the U / V codes interleaved sample by sample; a code is the
# interleaved exp-Golomb form of |v| (value + 1 in binary: a 0 and the next bit per bit below the
# leading one, then a 1) followed by a sign bit when v != 0.

def _ilog2up(x):
    n = 0
    while (1 << n) < x:
        n += 1
    return n


def _sint_bits(v):
    m = abs(int(v)) + 1
    bits = []
    for b in range(m.bit_length() - 2, -1, -1):
        bits += [0, (m >> b) & 1]
    bits.append(1)
    if v:
        bits.append(1 if v < 0 else 0)
    return bits


def subband_geometry(w, h, depth, index):
    """(position, column 0, row 0, row step, width, height) of sub-band `index` inside the
    interleaved coefficient plane (schro_subband_get_frame_data, schroparams.c:319-352)."""
    position = 0 if index == 0 else (((index - 1) // 3) << 2) | ((index - 1) % 3 + 1)
    shift = depth - (position >> 2)
    bw, bh = w >> shift, h >> shift
    return position, (bw if position & 1 else 0), ((1 << shift) >> 1 if position & 2 else 0), 1 << shift, bw, bh


def lowdelay_slice_values(P, seed, scale=0.9):
    """Quantised values of ONE slice: (luma, u, v), each a list of 1 + 3 * depth row-major arrays."""
    rng = np.random.default_rng(seed)
    depth = P["transform_depth"]
    out = []
    for k in range(3):
        w = (P["iwt_chroma_width"] if k else P["iwt_luma_width"]) // P["n_horiz_slices"]
        h = (P["iwt_chroma_height"] if k else P["iwt_luma_height"]) // P["n_vert_slices"]
        bands = []
        for index in range(1 + 3 * depth):
            _, _, _, _, bw, bh = subband_geometry(w, h, depth, index)
            mag = np.floor(rng.exponential(scale, (bh, bw))).astype(np.int64)
            bands.append(np.where(rng.integers(0, 2, (bh, bw)) == 1, -mag, mag))
        out.append(bands)
    return out


def lowdelay_write_slice(P, base_index, values):
    """One slice of slice_bytes_num bytes (slice_bytes_denom must be 1): bytes, or None when the
    codes do not fit."""
    assert P["slice_bytes_denom"] == 1
    nbytes = P["slice_bytes_num"]
    bits = [(base_index >> b) & 1 for b in range(6, -1, -1)]
    luma = [b for band in values[0] for v in band.reshape(-1) for b in _sint_bits(v)]
    nlen = _ilog2up(8 * nbytes)
    bits += [(len(luma) >> b) & 1 for b in range(nlen - 1, -1, -1)] + luma
    for bu, bv in zip(values[1], values[2]):
        for a, b in zip(bu.reshape(-1), bv.reshape(-1)):
            bits += _sint_bits(a) + _sint_bits(b)
    if len(bits) > 8 * nbytes:
        return None
    bits += [1] * (8 * nbytes - len(bits))         # (the reader sees ones past the end: value 0 codes)
    return np.packbits(np.array(bits, np.uint8))


def lowdelay_picture(P, seed=1, kinds=16):
    """The slice bytes of a whole picture from `kinds` different slices dealt out pseudo-randomly,
    and what went into them: (bytes, kind of every slice [ny, nx], [(base_index, values)] per kind)."""
    made = []
    s = seed
    while len(made) < kinds:
        base = 4 + (s * 7) % 25
        vals = lowdelay_slice_values(P, s)
        data = lowdelay_write_slice(P, base, vals)
        s += 1
        if data is not None:
            made.append((base, vals, data))
    ny, nx = P["n_vert_slices"], P["n_horiz_slices"]
    kind = (lcg(ny * nx, seed + 5) % kinds).reshape(ny, nx)
    table = np.stack([m[2] for m in made])
    return table[kind.reshape(-1)].reshape(-1), kind, [(m[0], m[1]) for m in made]


def lowdelay_dequantise(q, quant_index, tables):
    """schro_dequantise (schroutils.c:180-189) with schro_table_quant / schro_table_offset_1_2."""
    f, o = tables["schro_table_quant"][quant_index], tables["schro_table_offset_1_2"][quant_index]
    a = (np.abs(q) * f + o + 2) >> 2
    return np.where(q == 0, 0, np.where(q < 0, -a, a)).astype(np.int64)


def lowdelay_expected_band(P, comp, index, base_index, values, tables):
    """Dequantised coefficients of sub-band `index` of one slice (the slice's rectangle of the band)."""
    qi = min(max(base_index - P["quant_matrix"][index], 0), 60)
    return lowdelay_dequantise(values[comp][index], qi, tables)
