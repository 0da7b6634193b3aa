"""Minimal OpenEXR scanline reader / writer (no OpenEXR / imageio / cv2 in the image).

Covers what the environment-map tooling of the reference needs -- `imageio.imread(pano.exr)` in scripts/pano2cube.py:46 and
`imageio.imwrite(...pano.exr)` in modules/integral_equirect.py:363-371: single-part scan-line files with HALF / FLOAT / UINT
channels, compression NONE, RLE, ZIPS, ZIP.  PIZ / PXR24 / B44 / DWAA / DWAB (lossy wavelet / DCT codecs) are not implemented
and raise with the codec's name; tiled, deep and multi-part files raise as well.  File layout: OpenEXR "File Layout" document
(magic 20000630, version, attribute list, line offset table, chunks of <y, size, data>).
"""
import struct
import zlib

import numpy as np

MAGIC = 20000630
_COMPRESSION = {0: "NONE", 1: "RLE", 2: "ZIPS", 3: "ZIP", 4: "PIZ", 5: "PXR24", 6: "B44", 7: "B44A", 8: "DWAA", 9: "DWAB"}
_LINES = {0: 1, 1: 1, 2: 1, 3: 16}
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


def imread(path):
    """-> float32 array [H, W, C] with the channels in R, G, B(, A) order when present (otherwise file order)"""
    buf = open(path, "rb").read()
    attrs, p = read_header(buf)
    comp = attrs["compression"][1][0]
    if comp not in _LINES:
        raise ExrError(f"OpenEXR compression {_COMPRESSION.get(comp, comp)} is not supported (NONE, RLE, ZIPS, ZIP are); "
                       "re-save the panorama with e.g. `oiiotool in.exr --compression zip -o out.exr`")
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
