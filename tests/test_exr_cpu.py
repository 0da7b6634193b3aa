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
    b[i] = 9
    p.write_bytes(bytes(b))
    with pytest.raises(exr.ExrError, match="DWAB"):
        exr.imread(str(p))
    with pytest.raises(exr.ExrError, match="magic"):
        (tmp_path / "x.exr").write_bytes(b"\0" * 64)
        exr.imread(str(tmp_path / "x.exr"))
