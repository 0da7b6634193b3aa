"""Minimal OpenEXR scanline reader / writer (no OpenEXR / imageio / cv2 in the image).

Covers what the environment-map tooling of the reference needs -- `imageio.imread(pano.exr)` in scripts/pano2cube.py:46 and
`imageio.imwrite(...pano.exr)` in modules/integral_equirect.py:363-371: single-part scan-line files with HALF / FLOAT / UINT
channels, compression NONE, RLE, ZIPS, ZIP, and (reading only) DWAA / DWAB -- the codec of every panorama under the reference's
backgrounds/ -- for channels its classifier sends to the lossy DCT path (R, G, B, Y, RY, BY as HALF or FLOAT; other channels
of a DWA file raise).  PIZ / PXR24 / B44 raise with the codec's name; tiled, deep and multi-part files raise as well.
File layout: OpenEXR "File Layout" document (magic 20000630, version, attribute list, line offset table, chunks of <y, size,
data>).  DWA chunk layout and decoding steps: see _dwa_chunk (a restatement of the published format of OpenEXR's
DwaCompressor: 11 sizes, channel rules, zlib'd DC plane, Huffman- or zlib-coded run-length AC stream, 8 x 8 inverse DCT,
Y'CbCr -> R'G'B', perceptual -> linear through half precision); the library itself is not in the image, so the decoder is
checked on the reference's own panoramas by invariants (tests/test_exr_cpu.py), not against OpenEXR's output bit for bit.
"""
import struct
import zlib

import numpy as np

MAGIC = 20000630
_COMPRESSION = {0: "NONE", 1: "RLE", 2: "ZIPS", 3: "ZIP", 4: "PIZ", 5: "PXR24", 6: "B44", 7: "B44A", 8: "DWAA", 9: "DWAB"}
_LINES = {0: 1, 1: 1, 2: 1, 3: 16, 8: 32, 9: 256}
_PIXEL = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}


class ExrError(ValueError):
    pass


def _cstr(buf, p):
    e = buf.index(b"\0", p)
    return buf[p:e].decode("latin-1"), e + 1


def read_header(buf):
    """-> (attributes {name: (type, raw bytes)}, offset of the line offset table)"""
    magic, version = struct.unpack_from("<iI", buf, 0)
    if magic != MAGIC:
        raise ExrError("not an OpenEXR file (bad magic number)")
    if version & 0x200:
        raise ExrError("tiled OpenEXR files are not supported")
    if version & (0x800 | 0x1000):
        raise ExrError("deep / multi-part OpenEXR files are not supported")
    p, attrs = 8, {}
    while True:
        name, p = _cstr(buf, p)
        if name == "":
            break
        typ, p = _cstr(buf, p)
        (size,) = struct.unpack_from("<i", buf, p)
        p += 4
        attrs[name] = (typ, bytes(buf[p:p + size]))
        p += size
    return attrs, p


def _channels(raw):
    """chlist -> [(name, pixel type 0/1/2, xSampling, ySampling)] in file (alphabetical) order"""
    out, p = [], 0
    while raw[p] != 0:
        name, p = _cstr(raw, p)
        ptype, _plin, xs, ys = struct.unpack_from("<iB3xii", raw, p)
        p += 16
        out.append((name, ptype, xs, ys))
    return out


def _unpredict(data):
    """inverse of the ZIP / RLE pre-filter: delta decoding (t[i] = t[i-1] + d[i] - 128 mod 256), then the two half
    buffers are interleaved back (even bytes | odd bytes)"""
    d = np.frombuffer(data, dtype=np.uint8).astype(np.int64)
    n = d.shape[0]
    if n == 0:
        return b""
    t = d.copy()
    t[1:] -= 128
    t = (np.cumsum(t) & 255).astype(np.uint8)
    out = np.empty(n, dtype=np.uint8)
    half = (n + 1) // 2
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _predict(raw):
    r = np.frombuffer(raw, dtype=np.uint8)
    n = r.shape[0]
    t = np.concatenate([r[0::2], r[1::2]]).astype(np.int64)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 128 + 256) & 255
    return d.astype(np.uint8).tobytes() if n else b""


def _rle_decode(data, expect):
    out, p, n = bytearray(), 0, len(data)
    while p < n:
        c = data[p] - 256 if data[p] > 127 else data[p]
        p += 1
        if c < 0:
            out += data[p:p - c]
            p += -c
        else:
            out += bytes([data[p]]) * (c + 1)
            p += 1
    if len(out) != expect:
        raise ExrError("corrupt RLE chunk")
    return bytes(out)


# ---- DWAA / DWAB ------------------------------------------------------------------------------------------------------
_ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14,
                    21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60,
                    61, 54, 47, 55, 62, 63])
# orthonormal 8-point DCT-II: _DCT[k, n] = c_k cos((2 n + 1) k pi / 16), c_0 = sqrt(1/8), c_k = 1/2
_DCT = np.array([[(np.sqrt(0.125) if k == 0 else 0.5) * np.cos((2 * n + 1) * k * np.pi / 16) for n in range(8)] for k in range(8)],
                dtype=np.float32)


def _huf_uncompress(data, n_raw):
    """OpenEXR's Huffman coder over 16-bit symbols (shared with PIZ): header {first symbol, last symbol = run-length code,
    table bytes, data bits, reserved}, code lengths packed in 6 bits with zero runs, canonical codes (longest first), data
    MSB first; the run-length code is followed by an 8-bit repeat count of the previous symbol."""
    out = np.zeros(n_raw, dtype=np.uint16)
    if len(data) == 0:
        if n_raw:
            raise ExrError("corrupt DWA chunk (empty AC stream)")
        return out
    im, iM, table_len, n_bits, _ = struct.unpack_from("<5I", data, 0)
    if im > 65536 or iM > 65536:
        raise ExrError("corrupt Huffman table")
    # ---- code lengths
    lens = np.zeros(65537, dtype=np.int64)
    acc, lc, p = 0, 0, 20

    def bits(n):
        nonlocal acc, lc, p
        while lc < n:
            acc = (acc << 8) | data[p]
            p += 1
            lc += 8
        lc -= n
        v = (acc >> lc) & ((1 << n) - 1)
        acc &= (1 << lc) - 1
        return v

    i = im
    while i <= iM:
        v = bits(6)
        if v == 63:
            i += bits(8) + 6
        elif v >= 59:
            i += v - 59 + 2
        else:
            lens[i] = v
            i += 1
    # ---- canonical codes
    count = np.bincount(lens, minlength=59)
    base, c = [0] * 59, 0
    for l in range(58, 0, -1):
        nc = (c + int(count[l])) >> 1
        base[l] = c
        c = nc
    tab_sym, tab_len, long_codes = [0] * 65536, [0] * 65536, {}
    for sym in np.nonzero(lens)[0].tolist():
        l = int(lens[sym])
        code = base[l]
        base[l] += 1
        if l <= 16:
            lo = code << (16 - l)
            for k in range(lo, lo + (1 << (16 - l))):
                tab_sym[k], tab_len[k] = sym, l
        else:
            long_codes[(l, code)] = sym
    # ---- data
    d = data[20 + table_len:]
    acc, lc, p, left, o, nd = 0, 0, 0, int(n_bits), 0, len(d)
    while o < n_raw:
        while lc < 64 and p < nd:
            acc = (acc << 8) | d[p]
            p += 1
            lc += 8
        pre = (acc >> (lc - 16)) & 0xffff if lc >= 16 else (acc << (16 - lc)) & 0xffff
        l = tab_len[pre]
        if l:
            sym = tab_sym[pre]
        else:
            sym = None
            for l in range(17, 59):
                if l > lc:
                    break
                sym = long_codes.get((l, (acc >> (lc - l)) & ((1 << l) - 1)))
                if sym is not None:
                    break
            if sym is None:
                raise ExrError("corrupt Huffman stream")
        if l > lc or l > left:
            raise ExrError("corrupt Huffman stream (out of bits)")
        lc -= l
        left -= l
        if sym == iM:                      # run-length code
            if lc < 8 or o == 0:
                raise ExrError("corrupt Huffman stream (run)")
            n = (acc >> (lc - 8)) & 0xff
            lc -= 8
            left -= 8
            if o + n > n_raw:
                raise ExrError("corrupt Huffman stream (run past the end)")
            out[o:o + n] = out[o - 1]
            o += n
        else:
            out[o] = sym
            o += 1
        acc &= (1 << lc) - 1
    return out


def _dwa_to_linear(h):
    """the codec stores LOSSY_DCT channels perceptually: x <= 1: x^(1/2.2), above: log(x) / 2.2 + 1 (sign kept).  This is the
    inverse, applied to half values and rounded back to half as the codec's lookup table does."""
    x = h.astype(np.float32)
    a = np.abs(x)
    with np.errstate(over="ignore", invalid="ignore"):
        lin = np.where(a <= 1.0, np.power(a, np.float32(2.2)), np.exp(np.float32(2.2) * (a - 1.0)))
    lin = np.where(np.isfinite(x), np.sign(x) * lin, 0.0)
    with np.errstate(over="ignore"):
        return lin.astype(np.float16)


def _dwa_planes(ac, dc, W, rows, ncomp, ac_pos, dc_pos):
    """one lossy-DCT decoder (ncomp = 3: an R, G, B set stored as Y' Cb Cr; 1: a single channel) over the run-length AC stream
    and the DC plane.  -> (half planes [ncomp][rows][W] in linear light, AC symbols used, DC values used)"""
    bx, by = (W + 7) // 8, (rows + 7) // 8
    nblk = bx * by
    coef = np.zeros((nblk, ncomp, 64), dtype=np.uint16)        # half bit patterns in zig-zag order
    coef[:, :, 0] = dc[dc_pos:dc_pos + ncomp * nblk].reshape(ncomp, nblk).T
    acl = ac.tolist()
    q, n_ac = ac_pos, len(acl)
    flat = coef.reshape(nblk * ncomp, 64)
    for b in range(nblk * ncomp):
        k, row = 1, None
        while k < 64:
            if q >= n_ac:
                raise ExrError("corrupt DWA chunk (AC stream too short)")
            t = acl[q]
            q += 1
            if t == 0xff00:
                break
            if t >> 8 == 0xff:
                k += t & 0xff
            else:
                if row is None:
                    row = flat[b]
                row[k] = t
                k += 1
    blocks = np.empty((nblk, ncomp, 64), dtype=np.float32)
    blocks[:, :, _ZIGZAG] = coef.view(np.float16).astype(np.float32)
    blocks = blocks.reshape(nblk, ncomp, 8, 8)
    pix = np.einsum("ki,bckl,lj->bcij", _DCT, blocks, _DCT, optimize=True).astype(np.float32)
    if ncomp == 3:                                                # Rec. 709 Y' Cb Cr -> R' G' B'
        y, cb, cr = pix[:, 0], pix[:, 1], pix[:, 2]
        pix = np.stack([y + np.float32(1.5747) * cr, y - np.float32(0.1873) * cb - np.float32(0.4682) * cr,
                        y + np.float32(1.8556) * cb], axis=1)
    with np.errstate(over="ignore"):
        half = _dwa_to_linear(pix.astype(np.float16))
    img = half.reshape(by, bx, ncomp, 8, 8).transpose(2, 0, 3, 1, 4).reshape(ncomp, by * 8, bx * 8)
    return img[:, :rows, :W], q - ac_pos, ncomp * nblk


def _dwa_scheme(name, ptype):
    """the codec's default channel rules: what goes through the lossy DCT (and in which colour set)"""
    suffix = name.rsplit(".", 1)[-1]
    if ptype in (1, 2) and suffix in ("R", "G", "B", "Y", "RY", "BY"):
        return "dct"
    return "other"


def _dwa_chunk(data, W, rows, chans):
    """-> {channel name: float32 [rows, W]}.  Chunk = 11 little-endian uint64 {version, unknown raw / packed size, AC packed size,
    DC packed size, RLE packed / raw / unpacked size, AC count, DC count, AC coding (0 Huffman, 1 zlib)}, the channel rules
    (version 2), then the packed unknown | AC | DC | RLE sections."""
    if len(data) < 88:
        raise ExrError("corrupt DWA chunk")
    (version, _unk_raw, unk_size, ac_size, dc_size, rle_size, _rle_raw, _rle_un, ac_count, dc_count, ac_coding) = \
        struct.unpack_from("<11Q", data, 0)
    p = 88
    if version >= 2:
        (rule_size,) = struct.unpack_from("<H", data, p)
        p += rule_size
    if any(_dwa_scheme(n, t) != "dct" for n, t, _, _ in chans) or unk_size or rle_size:
        raise ExrError("DWA: only files whose channels all take the lossy DCT path (R, G, B, Y, RY, BY as HALF / FLOAT) are read")
    ac_buf = data[p + unk_size:p + unk_size + ac_size]
    dc_buf = data[p + unk_size + ac_size:p + unk_size + ac_size + dc_size]
    if ac_count == 0:
        ac = np.zeros(0, dtype=np.uint16)
    elif ac_coding == 0:
        ac = _huf_uncompress(ac_buf, int(ac_count))
    else:
        ac = np.frombuffer(zlib.decompress(ac_buf), dtype="<u2")
    dc = np.frombuffer(_unpredict(zlib.decompress(dc_buf)), dtype="<u2") if dc_count else np.zeros(0, dtype=np.uint16)
    if ac.shape[0] != ac_count or dc.shape[0] != dc_count:
        raise ExrError("corrupt DWA chunk (coefficient counts)")
    # colour sets first (R, G, B of one layer), then the remaining lossy channels, each in file order
    names = [n for n, _, _, _ in chans]
    out, used, ac_pos, dc_pos = {}, set(), 0, 0
    for n in names:
        prefix, _, suf = n.rpartition(".")
        if suf != "R":
            continue
        trio = [(prefix + "." if prefix else "") + s for s in ("R", "G", "B")]
        if all(t in names for t in trio):
            planes, na, nd = _dwa_planes(ac, dc, W, rows, 3, ac_pos, dc_pos)
            ac_pos, dc_pos = ac_pos + na, dc_pos + nd
            for t, pl in zip(trio, planes):
                out[t] = pl.astype(np.float32)
            used.update(trio)
    for n in names:
        if n not in used:
            planes, na, nd = _dwa_planes(ac, dc, W, rows, 1, ac_pos, dc_pos)
            ac_pos, dc_pos = ac_pos + na, dc_pos + nd
            out[n] = planes[0].astype(np.float32)
    if ac_pos != ac_count or dc_pos != dc_count:
        raise ExrError("corrupt DWA chunk (coefficients left over)")
    return out


def imread(path):
    """-> float32 array [H, W, C] with the channels in R, G, B(, A) order when present (otherwise file order)"""
    buf = open(path, "rb").read()
    attrs, p = read_header(buf)
    comp = attrs["compression"][1][0]
    if comp not in _LINES:
        raise ExrError(f"OpenEXR compression {_COMPRESSION.get(comp, comp)} is not supported (NONE, RLE, ZIPS, ZIP, DWAA, DWAB "
                       "are); re-save the panorama with e.g. `oiiotool in.exr --compression zip -o out.exr`")
    xmin, ymin, xmax, ymax = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = xmax - xmin + 1, ymax - ymin + 1
    chans = _channels(attrs["channels"][1])
    if any(xs != 1 or ys != 1 for _, _, xs, ys in chans):
        raise ExrError("sub-sampled channels are not supported")
    lines = _LINES[comp]
    n_chunks = (H + lines - 1) // lines
    offsets = struct.unpack_from(f"<{n_chunks}Q", buf, p)
    bpp = [_PIXEL[t].itemsize for _, t, _, _ in chans]
    line_bytes = W * sum(bpp)
    planes = {name: np.empty((H, W), dtype=np.float32) for name, _, _, _ in chans}
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        data = buf[off + 8:off + 8 + size]
        rows = min(lines, ymax - y + 1)
        expect = rows * line_bytes
        if comp in (8, 9):
            for name, pl in _dwa_chunk(data, W, rows, chans).items():
                planes[name][y - ymin:y - ymin + rows] = pl
            continue
        if comp == 0 or size == expect:                  # blocks that do not shrink are stored raw
            raw = data
        elif comp == 1:
            raw = _unpredict(_rle_decode(data, expect))
        else:
            raw = _unpredict(zlib.decompress(data))
        if len(raw) != expect:
            raise ExrError("corrupt chunk (size mismatch)")
        q = 0
        for r in range(rows):
            for (name, ptype, _, _), nb in zip(chans, bpp):
                planes[name][y - ymin + r] = np.frombuffer(raw, dtype=_PIXEL[ptype], count=W, offset=q).astype(np.float32)
                q += W * nb
    names = [c[0] for c in chans]
    order = [n for n in ("R", "G", "B", "A") if n in names] or names
    if set(("R", "G", "B")) - set(names):
        order = names
    return np.stack([planes[n] for n in order], axis=-1)


def imwrite(path, im, compression="ZIP"):
    """im [H, W, 3|4|1] float -> FLOAT channels R, G, B(, A) (Y for one channel), scan-line, increasing Y"""
    im = np.asarray(im, dtype=np.float32)
    if im.ndim == 2:
        im = im[..., None]
    H, W, C = im.shape
    names = {1: ["Y"], 3: ["B", "G", "R"], 4: ["A", "B", "G", "R"]}[C]         # channel list is sorted by name
    src = {"R": 0, "G": 1, "B": 2, "A": 3, "Y": 0}
    comp = {"NONE": 0, "ZIPS": 2, "ZIP": 3}[compression]
    lines = _LINES[comp]

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val

    chl = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)
    hdr = struct.pack("<iI", MAGIC, 2)
    hdr += attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp]))
    hdr += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box)
    hdr += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    hdr += b"\0"
    chunks = []
    for y0 in range(0, H, lines):
        rows = im[y0:y0 + lines]
        raw = b"".join(np.ascontiguousarray(rows[r, :, src[n]], dtype="<f4").tobytes() for r in range(rows.shape[0]) for n in names)
        data = raw
        if comp:
            z = zlib.compress(_predict(raw))
            if len(z) < len(raw):
                data = z
        chunks.append(struct.pack("<ii", y0, len(data)) + data)
    table_at = len(hdr)
    pos = table_at + 8 * len(chunks)
    offs = []
    for c in chunks:
        offs.append(pos)
        pos += len(c)
    with open(path, "wb") as f:
        f.write(hdr + struct.pack(f"<{len(offs)}Q", *offs) + b"".join(chunks))
