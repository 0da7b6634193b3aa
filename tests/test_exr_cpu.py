"""nmf_amd/exr.py: the minimal OpenEXR reader / writer behind IntegralEquirect.save and the panorama import
(modules/integral_equirect.py:363-371, scripts/pano2cube.py:46 of the reference use imageio for this)."""
import struct
import zlib

import numpy as np
import pytest

from nmf_amd import exr


@pytest.mark.parametrize("comp", ["NONE", "ZIPS", "ZIP"])
@pytest.mark.parametrize("shape", [(5, 7, 3), (33, 20, 4), (16, 16, 1)])
def test_round_trip(tmp_path, comp, shape):
    rng = np.random.default_rng(3)
    im = rng.standard_normal(shape).astype(np.float32) * 10
    im[0, 0] = 0.0
    p = str(tmp_path / "a.exr")
    exr.imwrite(p, im, compression=comp)
    out = exr.imread(p)
    assert out.dtype == np.float32 and out.shape == shape and np.array_equal(out, im)


def test_zip_shrinks_smooth_images_and_predictor_is_an_involution(tmp_path):
    yy, xx = np.mgrid[0:64, 0:128].astype(np.float32)
    im = np.stack([np.sin(xx / 20), np.cos(yy / 15), xx * 0 + 0.5], -1).astype(np.float32)
    exr.imwrite(str(tmp_path / "z.exr"), im, "ZIP")
    exr.imwrite(str(tmp_path / "n.exr"), im, "NONE")
    assert (tmp_path / "z.exr").stat().st_size < 0.8 * (tmp_path / "n.exr").stat().st_size
    raw = bytes(range(256)) * 3 + b"\x07"
    assert exr._unpredict(exr._predict(raw)) == raw


def test_half_and_rle_files_written_by_hand(tmp_path):
    """a file as another writer would produce it: HALF channels, RLE compression, non-zero data window origin"""
    H, W = 3, 4
    data = (np.arange(H * W * 3, dtype=np.float32).reshape(H, W, 3) / 8).astype(np.float16)

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val

    chl = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", 1, 0, 1, 1) for n in ("B", "G", "R")) + b"\0"
    box = struct.pack("<4i", 10, 20, 10 + W - 1, 20 + H - 1)
    hdr = struct.pack("<iI", exr.MAGIC, 2) + attr("channels", "chlist", chl) + attr("compression", "compression", b"\x01")
    hdr += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + b"\0"

    def rle(raw):                                   # literal runs only (a valid, if lazy, RLE stream)
        out = b""
        for i in range(0, len(raw), 100):
            c = raw[i:i + 100]
            out += struct.pack("b", -len(c)) + c
        return out

    chunks = []
    for y in range(H):
        raw = b"".join(data[y, :, c].tobytes() for c in (2, 1, 0))          # B, G, R planes of the line
        chunks.append(struct.pack("<ii", 20 + y, len(rle(exr._predict(raw)))) + rle(exr._predict(raw)))
    pos = len(hdr) + 8 * H
    offs = []
    for c in chunks:
        offs.append(pos)
        pos += len(c)
    p = tmp_path / "h.exr"
    p.write_bytes(hdr + struct.pack(f"<{H}Q", *offs) + b"".join(chunks))
    out = exr.imread(str(p))
    assert out.shape == (H, W, 3) and np.array_equal(out, data.astype(np.float32))


def test_unsupported_codecs_are_named(tmp_path):
    im = np.zeros((2, 2, 3), np.float32)
    p = tmp_path / "d.exr"
    exr.imwrite(str(p), im, "NONE")
    b = bytearray(p.read_bytes())
    i = b.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    b[i] = 4
    p.write_bytes(bytes(b))
    with pytest.raises(exr.ExrError, match="PIZ"):
        exr.imread(str(p))
    with pytest.raises(exr.ExrError, match="magic"):
        (tmp_path / "x.exr").write_bytes(b"\0" * 64)
        exr.imread(str(tmp_path / "x.exr"))


# ---- DWAA / DWAB (the codec of the reference's backgrounds/*.exr) --------------------------------------------------------
def _huf_compress(symbols):
    """test-side encoder of OpenEXR's Huffman container (the format _huf_uncompress reads): code lengths from a plain Huffman
    tree, canonical codes assigned as the format prescribes, no run-length codes in the data (the run symbol is declared)."""
    import heapq
    freq = np.bincount(np.asarray(symbols, dtype=np.int64), minlength=65537).astype(np.int64)
    im = int(np.nonzero(freq)[0].min())
    iM = int(np.nonzero(freq)[0].max()) + 1                       # the run-length code: one past the last real symbol
    freq[iM] = 1
    heap = [(int(f), i, (i,)) for i, f in enumerate(freq) if f]
    heapq.heapify(heap)
    lens = np.zeros(65537, dtype=np.int64)
    if len(heap) == 1:
        lens[heap[0][1]] = 1
    while len(heap) > 1:
        a, b = heapq.heappop(heap), heapq.heappop(heap)
        for s in a[2] + b[2]:
            lens[s] += 1
        heapq.heappush(heap, (a[0] + b[0], min(a[1], b[1]), a[2] + b[2]))
    count = np.bincount(lens, minlength=59)
    base, c = [0] * 59, 0
    for l in range(58, 0, -1):
        nc = (c + int(count[l])) >> 1
        base[l] = c
        c = nc
    code = {}
    for s in np.nonzero(lens)[0].tolist():
        code[s] = (int(lens[s]), base[int(lens[s])])
        base[int(lens[s])] += 1

    def pack(pairs):                                              # [(n bits, value)] -> bytes, MSB first
        acc = n = 0
        for k, v in pairs:
            acc, n = (acc << k) | v, n + k
        pad = (-n) % 8
        return (acc << pad).to_bytes((n + pad) // 8, "big"), n
    table, i = [], im
    while i <= iM:                                                # 6-bit lengths; zero runs as 59..62 (2..5) or 63 + 8 bits (6..261)
        if lens[i] == 0:
            j = i
            while j <= iM and lens[j] == 0 and j - i < 261:
                j += 1
            run = j - i
            if run >= 6:
                table += [(6, 63), (8, run - 6)]
            elif run >= 2:
                table += [(6, 59 + run - 2)]
            else:
                table += [(6, 0)]
            i = j
        else:
            table.append((6, int(lens[i])))
            i += 1
    tbytes, _ = pack(table)
    dbytes, nbits = pack([code[int(s)] for s in symbols])
    return struct.pack("<5I", im, iM, len(tbytes), nbits, 0) + tbytes + dbytes


def test_huffman_container_round_trip():
    rng = np.random.default_rng(5)
    sym = np.concatenate([rng.integers(0, 40, 3000), rng.integers(0x3c00, 0x3c20, 500), [0xff00] * 700, [65535, 0, 7]])
    rng.shuffle(sym)
    out = exr._huf_uncompress(_huf_compress(sym), sym.shape[0])
    assert np.array_equal(out, sym.astype(np.uint16))
    one = exr._huf_uncompress(_huf_compress(np.full(17, 12345)), 17)
    assert np.array_equal(one, np.full(17, 12345, dtype=np.uint16))


def _dwa_file(path, coef, W, H, ac_coding, comp=9):
    """writes a DWAB file with FLOAT channels B, G, R from DCT coefficients coef [blocks_y, blocks_x, 3 (Y' Cb Cr), 64] given as
    float16 in NATURAL order: zig-zag scan, zero runs / end-of-block tokens, zlib'd DC plane -- the layout _dwa_chunk parses"""
    by, bx = coef.shape[:2]
    zz = coef.reshape(by * bx, 3, 64)[:, :, exr._ZIGZAG].view(np.uint16)
    dc = np.ascontiguousarray(zz[:, :, 0].T).reshape(-1)
    ac = []
    for b in range(by * bx):
        for c in range(3):
            k = 1
            while k < 64:
                if zz[b, c, k] != 0:
                    ac.append(int(zz[b, c, k]))
                    k += 1
                    continue
                run = 1
                while k + run < 64 and zz[b, c, k + run] == 0:
                    run += 1
                ac.append(0 if run == 1 else (0xff00 if k + run == 64 else 0xff00 | run))
                k += run
    ac = np.asarray(ac, dtype="<u2")
    acz = zlib.compress(ac.tobytes()) if ac_coding == 1 else _huf_compress(ac)
    dcz = zlib.compress(exr._predict(dc.astype("<u2").tobytes()))
    rules = struct.pack("<H", 2)
    chunk = struct.pack("<11Q", 2, 0, 0, len(acz), len(dcz), 0, 0, 0, ac.shape[0], dc.shape[0], ac_coding) + rules + acz + dcz

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val
    chl = b"".join(n + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)
    hdr = struct.pack("<iI", exr.MAGIC, 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp]))
    hdr += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + b"\0"
    with open(path, "wb") as f:
        f.write(hdr + struct.pack("<Q", len(hdr) + 8) + struct.pack("<ii", 0, len(chunk)) + chunk)


@pytest.mark.parametrize("ac_coding", [0, 1])
def test_dwab_chunk_decoding_against_a_direct_evaluation(tmp_path, ac_coding):
    """container, Huffman / zlib AC stream, run-length tokens, zig-zag order, DC plane, partial edge blocks: a file assembled from
    known coefficients must decode to the inverse DCT -> Y'CbCr->RGB -> perceptual->linear of exactly those coefficients"""
    rng = np.random.default_rng(11)
    W, H = 21, 13                                   # 3 x 2 blocks, both edges partial
    coef = np.zeros((2, 3, 3, 64), dtype=np.float16)
    coef[..., 0] = rng.uniform(2.0, 6.0, (2, 3, 3))            # DC: luma-like level (x 1/8 per pixel)
    coef[:, :, 1:, 0] = rng.uniform(-0.5, 0.5, (2, 3, 2))      # chroma
    coef[0, 0, :, 1:] = rng.normal(0, 0.2, (3, 63))            # a dense block
    coef[1, 2, 0, [1, 8, 9, 63]] = [0.5, -0.25, 0.125, 0.06]   # sparse blocks: long zero runs, a coefficient in the last slot
    coef[0, 1, 2, [5, 6]] = [0.3, 0.2]
    # block (1, 0) stays DC-only, block (0, 2) has a lone zero between two coefficients
    coef[0, 2, 1, [1, 3]] = [0.4, -0.4]
    p = str(tmp_path / "d.exr")
    _dwa_file(p, coef, W, H, ac_coding)
    out = exr.imread(p)
    assert out.shape == (H, W, 3) and out.dtype == np.float32
    D = exr._DCT.astype(np.float64)
    pix = np.einsum("ki,yxckl,lj->yxcij", D, coef.astype(np.float64).reshape(2, 3, 3, 8, 8), D)
    y, cb, cr = pix[:, :, 0], pix[:, :, 1], pix[:, :, 2]
    rgb = np.stack([y + 1.5747 * cr, y - 0.1873 * cb - 0.4682 * cr, y + 1.8556 * cb], 2)
    rgb = rgb.transpose(0, 3, 1, 4, 2).reshape(16, 24, 3)[:H, :W]
    nl = rgb.astype(np.float16).astype(np.float64)
    lin = np.sign(nl) * np.where(np.abs(nl) <= 1, np.abs(nl) ** 2.2, np.exp(2.2 * (np.abs(nl) - 1)))
    assert np.allclose(out, lin, rtol=3e-3, atol=1e-4), float(np.abs(out - lin).max())      # half precision twice


def test_dwab_panorama_of_the_reference_tree():
    """backgrounds/studio.exr of the reference (CC0, Greg Zaal / HDRI Haven; DWAB, FLOAT R G B, 1024 x 512) committed as a data
    fixture: every chunk must parse to the last coefficient (the decoder raises otherwise) and the picture must be a picture --
    finite, mostly positive, smooth (neighbouring pixels strongly correlated, no 8 x 8 block seams), bright lamps on a dark room."""
    import os
    im = exr.imread(os.path.join(os.path.dirname(__file__), "golden", "studio_dwab.exr"))
    assert im.shape == (512, 1024, 3) and np.isfinite(im).all()
    assert float(im.min()) > -0.01 and 50.0 < float(im.max()) < 500.0 and 0.1 < float(im.mean()) < 0.5
    lum = np.log(np.clip(im.mean(-1), 1e-4, None))
    cx = np.corrcoef(lum[:, :-1].ravel(), lum[:, 1:].ravel())[0, 1]
    cy = np.corrcoef(lum[:-1].ravel(), lum[1:].ravel())[0, 1]
    assert cx > 0.98 and cy > 0.98, (cx, cy)
    # block seams: the mean step across 8-pixel boundaries against the steps inside blocks.  The codec is lossy (level 300): 1.34
    # on this dark picture (1.06 - 1.09 on the daylight panoramas); a transposed zig-zag scan gives 4.7, a mirrored DCT basis 3.0
    dx = np.abs(np.diff(lum, axis=1))
    seam, inside = dx[:, 7::8].mean(), np.delete(dx, np.s_[7::8], axis=1).mean()
    assert seam < 1.8 * inside, (seam, inside)
    assert float(np.median(im)) < 0.1                  # a dark studio ...
    assert float((im.mean(-1) > 20).mean()) < 0.01     # ... with a few small lamps
