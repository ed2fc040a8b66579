"""OpenEXR input for Scene.add_envmap (render.py:150-151 reads an EXR with imageio, which is not shipped here):
the reader against files made by this package's own writer AND against a hand-assembled file whose bytes follow the
OpenEXR layout document field by field (so that reader and writer cannot agree on a common mistake)."""
import struct
import zlib

import numpy as np
import pytest

from zdr_amd import exr


@pytest.mark.parametrize("compression", [exr.NO_COMPRESSION, exr.RLE, exr.ZIPS, exr.ZIP])
@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("shape", [(32, 64, 3), (37, 21, 4), (1, 1, 3)])
def test_round_trip(tmp_path, compression, half, shape):
    rng = np.random.default_rng(shape[0])
    img = rng.uniform(0.0, 50.0, shape).astype(np.float32)
    if half: img = img.astype(np.float16).astype(np.float32)
    p = str(tmp_path / "a.exr")
    exr.write_exr(p, img, compression=compression, half=half)
    got = exr.read_exr(p)
    assert got.shape == shape and got.dtype == np.float32
    assert np.array_equal(got, img)


def test_hand_assembled_file(tmp_path):
    # 3 x 2 image, channels B, G, R stored alphabetically as HALF, HALF, FLOAT; ZIPS; data window offset from the origin
    W, H = 3, 2
    R = np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]], np.float32)
    G = R * 0.5; B = R * 0.25
    def cstr(s): return s.encode() + b"\0"
    chl = b"".join(cstr(n) + struct.pack("<i", t) + bytes([0, 0, 0, 0]) + struct.pack("<ii", 1, 1) for n, t in (("B", 1), ("G", 1), ("R", 2))) + b"\0"
    def attr(n, t, v): return cstr(n) + cstr(t) + struct.pack("<i", len(v)) + v
    head = bytes([0x76, 0x2F, 0x31, 0x01]) + struct.pack("<i", 2)
    head += attr("channels", "chlist", chl) + attr("compression", "compression", bytes([2]))
    head += attr("dataWindow", "box2i", struct.pack("<4i", 10, 20, 10 + W - 1, 20 + H - 1)) + attr("displayWindow", "box2i", struct.pack("<4i", 0, 0, 63, 63))
    head += attr("lineOrder", "lineOrder", bytes([0])) + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    chunks = []
    for y in range(H):
        raw = B[y].astype("<f2").tobytes() + G[y].astype("<f2").tobytes() + R[y].astype("<f4").tobytes()
        # OpenEXR ZIP: split into even / odd bytes, byte-delta predictor (+128), deflate
        a = bytearray(raw); t = bytes(a[0::2]) + bytes(a[1::2])
        d = bytearray(len(t)); d[0] = t[0]
        for i in range(1, len(t)): d[i] = (t[i] - t[i - 1] + 128) & 0xFF
        comp = zlib.compress(bytes(d))
        data = comp if len(comp) < len(raw) else raw
        chunks.append(struct.pack("<ii", 20 + y, len(data)) + data)
    offs, at = [], len(head) + 8 * H
    for c in chunks:
        offs.append(at); at += len(c)
    p = tmp_path / "hand.exr"
    p.write_bytes(head + struct.pack(f"<{H}Q", *offs) + b"".join(chunks))
    got = exr.read_exr(str(p))
    assert got.shape == (H, W, 3)
    np.testing.assert_array_equal(got[..., 0], R); np.testing.assert_array_equal(got[..., 1], G); np.testing.assert_array_equal(got[..., 2], B)


def test_unsupported_compression_is_named(tmp_path):
    p = str(tmp_path / "a.exr")
    exr.write_exr(p, np.ones((4, 4, 3), np.float32), compression=exr.NO_COMPRESSION)
    b = bytearray(open(p, "rb").read())
    i = b.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    b[i] = exr.PIZ
    open(p, "wb").write(bytes(b))
    with pytest.raises(NotImplementedError, match="PIZ"):
        exr.read_exr(p)


def test_envmap_preparation_accepts_an_exr(tmp_path):
    from zdr_amd import envmap
    img = np.random.default_rng(0).uniform(0.05, 2.0, (16, 32, 3)).astype(np.float32)
    p = str(tmp_path / "sky.exr")
    exr.write_exr(p, img)
    a = envmap.prepare_image(envmap.load_image(p))
    b = envmap.prepare_image(img)
    assert np.array_equal(a, b)


def test_rle_runs_and_literals_decode_as_the_format_document_says():
    # count byte n >= 0: next byte repeated n + 1 times; n < 0: -n literal bytes
    coded = bytes([2, 7, 0xFD, 1, 2, 3, 0, 9])
    assert exr._unrle(coded) == bytes([7, 7, 7, 1, 2, 3, 9])
    raw = bytes([5] * 300 + list(range(40)) + [8, 8, 8, 8, 1])
    assert exr._unrle(exr._rle(raw)) == raw


def test_constant_image_round_trips_through_rle(tmp_path):
    p = str(tmp_path / "c.exr")
    exr.write_exr(p, np.full((8, 64, 3), 2.5, np.float32), compression=exr.RLE)
    assert np.array_equal(exr.read_exr(p), np.full((8, 64, 3), 2.5, np.float32))


def test_truncated_and_corrupt_files_raise_a_clean_error(tmp_path):
    """A damaged file must end in ValueError (or NotImplementedError when the damage reads as an unsupported feature) —
    never in an IndexError / struct.error from the middle of the parser, never in a huge allocation."""
    img = np.random.default_rng(0).uniform(0, 2, (9, 13, 3)).astype(np.float32)
    p = str(tmp_path / "a.exr")
    rng = np.random.default_rng(1)
    for comp in (exr.NO_COMPRESSION, exr.RLE, exr.ZIPS, exr.ZIP):
        exr.write_exr(p, img, compression=comp)
        good = open(p, "rb").read()
        for trial in range(150):
            bad = bytearray(good)
            if trial % 3 == 0:
                bad = bad[:rng.integers(1, len(bad))]
            else:
                for _ in range(rng.integers(1, 4)):
                    bad[rng.integers(0, len(bad))] = rng.integers(0, 256)
            open(p, "wb").write(bytes(bad))
            try:
                out = exr.read_exr(p)
                assert out.ndim == 3 and out.shape[0] * out.shape[1] <= (1 << 28)
            except (ValueError, NotImplementedError):
                pass
