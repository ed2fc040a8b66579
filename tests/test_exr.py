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
    b[i] = exr.DWAA
    open(p, "wb").write(bytes(b))
    with pytest.raises(NotImplementedError, match="DWAA"):
        exr.read_exr(p)


# ---- PIZ: a hand-written ENCODER that follows OpenEXR's ImfPizCompressor / ImfWav / ImfHuf step by step (scalar loops, the forward
# direction of every stage), so that the vectorised decoder of zdr_amd/exr.py is checked against the format's other half rather than
# against itself.  (No real PIZ file exists in this offline image: the check is against the format as documented, not against OpenEXR's
# own output — stated in DESIGN.md.)
def _piz_wenc14(a, b):
    s16 = lambda x: ((x + 0x8000) & 0xFFFF) - 0x8000
    a, b = s16(a), s16(b)
    m, d = (a + b) >> 1, a - b
    return m & 0xFFFF, d & 0xFFFF


def _piz_wenc16(a, b):
    ao = (a + 0x8000) & 0xFFFF
    m, d = (ao + b) >> 1, ao - b
    if d < 0:
        m = (m + 0x8000) & 0xFFFF
    return m, d & 0xFFFF


def _piz_wav2_encode(a, nx, ox, ny, oy, mx, base):
    """ImfWav.cpp wav2Encode on the flat list `a` starting at `base`, pixel stride ox, line stride oy."""
    enc = _piz_wenc14 if mx < (1 << 14) else _piz_wenc16
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        py, ey = base, base + oy * (ny - p2)
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        while py <= ey:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01, p10 = px + ox1, px + oy1
                p11 = p10 + ox1
                i00, i01 = enc(a[px], a[p01]); i10, i11 = enc(a[p10], a[p11])
                a[px], a[p10] = enc(i00, i10); a[p01], a[p11] = enc(i01, i11)
                px += ox2
            if nx & p:
                p10 = px + oy1
                a[px], a[p10] = enc(a[px], a[p10])
            py += oy2
        if ny & p:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01 = px + ox1
                a[px], a[p01] = enc(a[px], a[p01])
                px += ox2
        p, p2 = p2, p2 << 1


class _BitWriter:
    def __init__(self): self.out, self.c, self.lc = bytearray(), 0, 0
    def put(self, nbits, bits):
        self.c = (self.c << nbits) | bits; self.lc += nbits
        while self.lc >= 8:
            self.lc -= 8; self.out.append((self.c >> self.lc) & 0xFF)
        self.c &= (1 << self.lc) - 1
    def flush(self):
        if self.lc: self.out.append((self.c << (8 - self.lc)) & 0xFF)
        n = len(self.out) * 8 - ((8 - self.lc) % 8 if self.lc else 0); self.c = self.lc = 0
        return n


def _piz_huf_compress(words):
    import heapq
    freq = {}
    for w in words: freq[w] = freq.get(w, 0) + 1
    im, iM = min(freq), max(freq) + 1
    freq[iM] = 1                                              # the run-length escape (hufBuildEncTable)
    heap = [(f, s, (s,)) for s, f in freq.items()]; heapq.heapify(heap)
    length = {s: 0 for s in freq}
    while len(heap) > 1:
        f1, k1, m1 = heapq.heappop(heap); f2, k2, m2 = heapq.heappop(heap)
        for s in m1 + m2: length[s] += 1
        heapq.heappush(heap, (f1 + f2, min(k1, k2), m1 + m2))
    if len(freq) == 1: length[iM] = 1
    assert max(length.values()) <= 58
    n = [0] * 59                                              # hufCanonicalCodeTable
    for l in length.values(): n[l] += 1
    c = 0
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1; n[i] = c; c = nc
    code = {}
    for s in sorted(length):
        code[s] = n[length[s]]; n[length[s]] += 1
    tab = _BitWriter()                                        # hufPackEncTable
    s = im
    while s <= iM:
        l = length.get(s, 0)
        if l == 0:
            zerun = 1
            while s < iM and zerun < 255 + 6 and length.get(s + 1, 0) == 0:
                s += 1; zerun += 1
            if zerun >= 2:
                if zerun >= 6: tab.put(6, 63); tab.put(8, zerun - 6)
                else: tab.put(6, 59 + zerun - 2)
                s += 1
                continue
        tab.put(6, l)
        s += 1
    tab.flush()
    bw = _BitWriter()                                         # hufEncode with sendCode's run-length choice
    def send(sym, run):
        if length[sym] + length[iM] + 8 < length[sym] * run:
            bw.put(length[sym], code[sym]); bw.put(length[iM], code[iM]); bw.put(8, run)
        else:
            for _ in range(run + 1): bw.put(length[sym], code[sym])
    cur, cs = words[0], 0
    for w in words[1:]:
        if w == cur and cs < 255: cs += 1
        else: send(cur, cs); cs = 0
        cur = w
    send(cur, cs)
    nbits = len(bw.out) * 8 + bw.lc
    bw.flush()
    return struct.pack("<5I", im, iM, len(tab.out), nbits, 0) + bytes(tab.out) + bytes(bw.out)


def _piz_block(lines, W):
    """lines: list over scan lines of lists over channels (alphabetical) of numpy arrays (HALF or FLOAT) -> PIZ block bytes."""
    rows = len(lines)
    sizes = [ch.dtype.itemsize // 2 for ch in lines[0]]
    planes = [[] for _ in sizes]
    for line in lines:
        for k, ch in enumerate(line):
            planes[k] += list(np.frombuffer(ch.astype(ch.dtype.newbyteorder("<")).tobytes(), "<u2").astype(int))
    words = [w for pl in planes for w in pl]
    present = sorted(set(words) | {0})
    nonzero = [v for v in present if v]
    bitmap = bytearray(8192)
    for v in nonzero: bitmap[v >> 3] |= 1 << (v & 7)
    used = [i for i in range(8192) if bitmap[i]]
    mn, mx = (min(used), max(used)) if used else (8191, 0)
    fwd = {v: k for k, v in enumerate(present)}               # forwardLutFromBitmap: rank among the occurring values
    words = [fwd[w] for w in words]
    max_value = len(present) - 1
    at = 0
    for size in sizes:
        for j in range(size):
            _piz_wav2_encode(words, W, size, rows, W * size, max_value, at + j)
        at += rows * W * size
    huf = _piz_huf_compress(words)
    return struct.pack("<HH", mn, mx) + (bytes(bitmap[mn:mx + 1]) if mn <= mx else b"") + struct.pack("<i", len(huf)) + huf


def _piz_file(path, chans, W, H):
    """chans: {name: (H, W) array of float16 / float32} -> a PIZ-compressed scan-line EXR assembled by hand."""
    def cstr(s): return s.encode() + b"\0"
    def attr(n, t, v): return cstr(n) + cstr(t) + struct.pack("<i", len(v)) + v
    names = sorted(chans)
    chl = b"".join(cstr(n) + struct.pack("<i", 1 if chans[n].dtype == np.float16 else 2) + bytes(4) + struct.pack("<ii", 1, 1) for n in names) + b"\0"
    head = bytes([0x76, 0x2F, 0x31, 0x01]) + struct.pack("<i", 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([4]))
    head += attr("dataWindow", "box2i", struct.pack("<4i", 0, 0, W - 1, H - 1)) + attr("displayWindow", "box2i", struct.pack("<4i", 0, 0, W - 1, H - 1))
    head += attr("lineOrder", "lineOrder", bytes([0])) + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    chunks = []
    for y in range(0, H, 32):
        rows = min(32, H - y)
        data = _piz_block([[chans[n][y + r] for n in names] for r in range(rows)], W)
        raw_size = rows * W * sum(chans[n].dtype.itemsize for n in names)
        if len(data) >= raw_size:                             # a block the coder cannot shrink is stored raw, in the uncompressed line layout
            data = b"".join(chans[n][y + r].astype(chans[n].dtype.newbyteorder("<")).tobytes() for r in range(rows) for n in names)
        chunks.append(struct.pack("<ii", y, len(data)) + data)
    offs, at = [], len(head) + 8 * len(chunks)
    for c in chunks:
        offs.append(at); at += len(c)
    open(path, "wb").write(head + struct.pack(f"<{len(offs)}Q", *offs) + b"".join(chunks))


@pytest.mark.parametrize("case", ["half_smooth_w14", "half_noisy_w16", "float_mixed", "tiny_1x1", "constant"])
def test_piz_decoding_against_a_hand_written_encoder(tmp_path, case):
    rng = np.random.default_rng(len(case))
    if case == "half_smooth_w14":                              # few distinct values: the 14-bit wavelet (wdec14), long runs
        H, W = 37, 21
        y, x = np.mgrid[0:H, 0:W]
        base = np.round((np.sin(x / 5.0) + np.cos(y / 7.0) + 2.5) * 8) / 8
        chans = {"R": base.astype(np.float16), "G": (base * 0.5).astype(np.float16), "B": (base * 0.25).astype(np.float16)}
    elif case == "half_noisy_w16":                             # more than 2^14 distinct values in a block: wdec16
        H, W = 70, 131
        chans = {n: rng.uniform(0.0, 4000.0, (H, W)).astype(np.float16) for n in "RGB"}
    elif case == "float_mixed":                                # FLOAT channels (two words per pixel) next to a HALF alpha; odd sizes
        H, W = 33, 17
        chans = {"R": (np.round(rng.uniform(0, 3, (H, W)) * 16) / 16).astype(np.float32), "G": (np.round(rng.uniform(0, 3, (H, W)) * 4) / 4).astype(np.float32),
                 "B": np.full((H, W), 0.75, np.float32), "A": np.ones((H, W), np.float16)}
    elif case == "tiny_1x1":
        H, W = 1, 1
        chans = {"R": np.array([[1.5]], np.float16), "G": np.array([[2.5]], np.float16), "B": np.array([[0.0]], np.float16), "A": np.array([[1.0]], np.float16)}
    else:
        H, W = 40, 9
        chans = {n: np.full((H, W), 3.25, np.float16) for n in "RGB"}
    p = str(tmp_path / "piz.exr")
    _piz_file(p, chans, W, H)
    got = exr.read_exr(p)
    order = [c for c in "RGBA" if c in chans]
    assert got.shape == (H, W, len(order))
    for k, n in enumerate(order):
        np.testing.assert_array_equal(got[..., k], chans[n].astype(np.float32), err_msg=f"{case} channel {n}")


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
    for comp in (exr.NO_COMPRESSION, exr.RLE, exr.ZIPS, exr.ZIP, exr.PIZ):
        if comp == exr.PIZ:
            _piz_file(p, {n: np.round(img[..., k] * 8).astype(np.float16) for k, n in enumerate("RGB")}, 13, 9)
        else:
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
