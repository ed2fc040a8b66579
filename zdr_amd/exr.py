"""Minimal OpenEXR reader / writer for environment maps (Scene.add_envmap; the reference reads its EXR files with
imageio, /root/reference/envmap.py:117-121, which this environment does not ship).

Scope: single-part scan-line files, channels of type HALF / FLOAT / UINT with sampling 1, compression NONE, RLE, ZIPS, ZIP
(what Blender and OpenEXR's own tools write by default or on request) or PIZ (OpenEXR's historical default, common in HDRI
libraries; read only — wavelet + Huffman decoding below).  PXR24, B44 and DWA files are refused with a message naming the
compression.  PIZ CAVEAT: no file written by OpenEXR itself exists in this offline image, so the PIZ decoder is checked only
against the encoder in tests/test_exr.py, written from the same reading of ImfPizCompressor / ImfHuf / ImfWav — a shared misreading
of the format would pass.  Its Huffman stage is a Python loop per 16-bit word (about 1.3 M words/s: a 4096 x 2048 half-float map
takes ~20 s, during which add_envmap looks hung).  Out of the hot path's scope; frozen as it is.  Layout per the OpenEXR file-layout document: magic 0x01312f76,
version word, attribute list, chunk-offset table, chunks {y, size, data}; inside a chunk the scan lines follow one
another, each holding its channels in alphabetical order; ZIP data is deflate over a byte-delta predictor applied to the
even/odd byte split of the chunk.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

MAGIC = 20000630
NO_COMPRESSION, RLE, ZIPS, ZIP, PIZ, PXR24, B44, B44A, DWAA, DWAB = range(10)
_NAMES = {PIZ: "PIZ", PXR24: "PXR24", B44: "B44", B44A: "B44A", DWAA: "DWAA", DWAB: "DWAB"}
_LINES = {NO_COMPRESSION: 1, RLE: 1, ZIPS: 1, ZIP: 16, PIZ: 32}
_DTYPES = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}      # UINT, HALF, FLOAT


def _cstr(buf: bytes, pos: int):
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin-1"), end + 1


def _unrle(data: bytes) -> bytes:
    """OpenEXR run-length coding: a count byte n >= 0 is followed by one byte repeated n + 1 times, n < 0 by -n literal bytes."""
    out = bytearray()
    i = 0
    while i < len(data):
        n = data[i] - 256 if data[i] > 127 else data[i]
        i += 1
        if n < 0:
            out += data[i:i - n]; i -= n
        else:
            out += bytes([data[i]]) * (n + 1); i += 1
    return bytes(out)


def _rle(raw: bytes) -> bytes:
    out = bytearray()
    i, n = 0, len(raw)
    while i < n:
        j = i + 1
        while j < n and raw[j] == raw[i] and j - i < 128:
            j += 1
        if j - i >= 3:
            out += bytes([j - i - 1, raw[i]]); i = j
        else:
            j = i
            while j < n and j - i < 127 and not (j + 2 < n and raw[j] == raw[j + 1] == raw[j + 2]):
                j += 1
            out += bytes([256 - (j - i)]) + raw[i:j]; i = j
    return bytes(out)


def _unpredict(t: np.ndarray, size: int) -> bytes:
    # predictor: t[i] = t[i-1] + t[i] - 128 (mod 256), then the halves hold the even and the odd bytes
    t = np.cumsum(t.astype(np.int64) - 128, dtype=np.int64)
    t = ((t + 128) & 0xFF).astype(np.uint8)
    half = (size + 1) // 2
    out = np.empty(size, np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _predict(raw: bytes) -> np.ndarray:
    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int64)
    d = np.empty_like(t)
    d[0] = t[0]
    d[1:] = t[1:] - t[:-1] + 128
    return (d & 0xFF).astype(np.uint8)


def _unzip(data: bytes, size: int, rle: bool = False) -> bytes:
    if len(data) == size:                       # stored raw when the coder did not help
        return data
    t = np.frombuffer(_unrle(data) if rle else zlib.decompress(data), np.uint8)
    if t.size != size:
        raise ValueError("EXR: corrupt compressed chunk")
    return _unpredict(t, size)


def _zip(raw: bytes, rle: bool = False) -> bytes:
    d = _predict(raw).tobytes()
    comp = _rle(d) if rle else zlib.compress(d)
    return comp if len(comp) < len(raw) else raw


# ------------------------------------------------------------------------------------------ PIZ
# A PIZ block (32 scan lines) holds, per OpenEXR's ImfPizCompressor / ImfHuf / ImfWav:
#   u16 minNonZero, u16 maxNonZero, bitmap[minNonZero .. maxNonZero]   which 16-bit values occur (bit v & 7 of byte v >> 3; 0 always does)
#   i32 length, Huffman-coded u16 stream                               the block as u16 words, channel after channel (each channel
#                                                                      ny rows of nx pixels of 1 (HALF) or 2 (FLOAT, UINT) words), values
#                                                                      replaced by their rank among the occurring ones, then Haar-wavelet
#                                                                      transformed in place, per channel and per word of the pixel
# Huffman stream: u32 im, iM, tableLength, nBits, reserved; code lengths of symbols im..iM packed 6 bits each (59..62: a run of 2..5
# zero lengths, 63 + 8 bits: a run of 6..261), padded to a byte; then nBits of codes, MSB first.  Codes are canonical (the longest get the
# smallest values); symbol iM is the run-length escape: it is followed by an 8-bit count of repeats of the previous word.
_HUF_DECBITS = 14
_WINDOW = (1 << 96) - 1        # bit window of the readers below: wide enough for a 58-bit code behind 32 buffered bits


class _Bits:
    """MSB-first bit reader over bytes (ImfHuf getBits / getChar)."""

    def __init__(self, data: bytes, pos: int = 0):
        self.d, self.p, self.c, self.lc = data, pos, 0, 0

    def get(self, n: int) -> int:
        while self.lc < n:
            self.c = ((self.c << 8) | self.d[self.p]) & _WINDOW; self.p += 1; self.lc += 8
        self.lc -= n
        return (self.c >> self.lc) & ((1 << n) - 1)


def _huf_canonical(lengths):
    """hufCanonicalCodeTable: code of every symbol from the code lengths (0 = symbol absent).  Returns {symbol: (length, code)}."""
    n = [0] * 59
    for l in lengths.values():
        n[l] += 1
    c = 0
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1
        n[i] = c
        c = nc
    out = {}
    for sym in sorted(lengths):
        l = lengths[sym]
        if l > 0:
            out[sym] = (l, n[l]); n[l] += 1
    return out


def _huf_uncompress(data: bytes, n_raw: int) -> np.ndarray:
    """hufUncompress: -> n_raw uint16 words."""
    if n_raw == 0:
        return np.zeros(0, np.uint16)
    im, iM, _tl, nbits, _r = struct.unpack_from("<5I", data, 0)
    if im > 65536 or iM > 65536 or im > iM:
        raise ValueError("EXR/PIZ: corrupt Huffman header")
    br = _Bits(data, 20)
    lengths, sym = {}, im
    while sym <= iM:                                         # hufUnpackEncTable
        l = br.get(6)
        if l == 63:
            sym += br.get(8) + 6
        elif l >= 59:
            sym += l - 59 + 2
        else:
            if l:
                lengths[sym] = l
            sym += 1
    codes = _huf_canonical(lengths)
    # decoding table over the first 14 bits for the short codes; longer codes (rare) are searched by length
    tab_sym = [-1] * (1 << _HUF_DECBITS); tab_len = [0] * (1 << _HUF_DECBITS)
    long_codes = {}
    for symb, (l, code) in codes.items():
        if l <= _HUF_DECBITS:
            lo = code << (_HUF_DECBITS - l)
            for k in range(lo, lo + (1 << (_HUF_DECBITS - l))):
                tab_sym[k] = symb; tab_len[k] = l
        else:
            long_codes.setdefault(l, {})[code] = symb
    long_lengths = sorted(long_codes)
    out = np.empty(n_raw, np.uint16)
    d, p, c, lc, no = data, br.p, 0, 0, 0                     # the table is padded to a byte: the codes start at br.p
    end_bit = nbits                                           # bits of code left to consume
    mask14 = (1 << _HUF_DECBITS) - 1
    total = (nbits + 7) // 8
    if p + total > len(d):
        raise ValueError("EXR/PIZ: truncated Huffman data")
    stop = p + total
    while end_bit > 0:
        while lc < 32 and p < stop:                           # keep >= 14 + 8 bits in the window when the stream has them
            c = ((c << 8) | d[p]) & _WINDOW; p += 1; lc += 8
        avail = min(lc, end_bit)
        if avail <= 0:
            break
        idx = ((c >> (lc - _HUF_DECBITS)) if lc >= _HUF_DECBITS else (c << (_HUF_DECBITS - lc))) & mask14
        l = tab_len[idx]
        if l and l <= avail:
            symb = tab_sym[idx]
        else:
            symb = -1
            for l in long_lengths:
                while lc < l and p < stop:
                    c = ((c << 8) | d[p]) & _WINDOW; p += 1; lc += 8
                if l <= min(lc, end_bit):
                    symb = long_codes[l].get((c >> (lc - l)) & ((1 << l) - 1), -1)
                    if symb >= 0:
                        break
            if symb < 0:
                raise ValueError("EXR/PIZ: invalid Huffman code")
        lc -= l; end_bit -= l
        if symb == iM:                                        # run-length escape: repeat the previous word
            while lc < 8 and p < stop:
                c = ((c << 8) | d[p]) & _WINDOW; p += 1; lc += 8
            lc -= 8; end_bit -= 8
            run = (c >> lc) & 0xFF
            if no == 0 or no + run > n_raw:
                raise ValueError("EXR/PIZ: corrupt run")
            out[no:no + run] = out[no - 1]; no += run
        else:
            if no >= n_raw:
                raise ValueError("EXR/PIZ: too many words")
            out[no] = symb; no += 1
    if no != n_raw:
        raise ValueError("EXR/PIZ: Huffman stream ends early")
    return out


def _wdec14(l, h):
    """ImfWav wdec14 on arrays of signed 16-bit values (held as int32)."""
    a = l + (h & 1) + (h >> 1)
    return _s16(a), _s16(a - h)


def _wdec16(l, h):
    bb = (l - (h >> 1)) & 0xFFFF
    aa = (h + bb - 0x8000) & 0xFFFF
    return aa, bb


def _s16(x):
    return ((x + 0x8000) & 0xFFFF) - 0x8000


def _wav2_decode(a: np.ndarray, mx: int) -> None:
    """Inverse 2-D Haar wavelet of ImfWav.cpp (wav2Decode) on the (ny, nx) int32 plane `a`, in place; every level is one
    vectorised step over the strided sub-grids the C loops visit."""
    ny, nx = a.shape
    w14 = mx < (1 << 14)
    if w14:
        a[...] = _s16(a)
        dec = _wdec14
    else:
        dec = _wdec16
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        ys = np.arange(0, ny - p2 + 1, p2) if ny >= p2 else np.zeros(0, int)
        xs = np.arange(0, nx - p2 + 1, p2) if nx >= p2 else np.zeros(0, int)
        if ys.size and xs.size:
            Y, X = np.meshgrid(ys, xs, indexing="ij")
            i00, i10 = dec(a[Y, X], a[Y + p, X])
            i01, i11 = dec(a[Y, X + p], a[Y + p, X + p])
            a[Y, X], a[Y, X + p] = dec(i00, i01)
            a[Y + p, X], a[Y + p, X + p] = dec(i10, i11)
        if (nx & p) and ys.size:                              # odd column: 1-D in y at the first x the 2-D loop did not reach
            x = xs[-1] + p2 if xs.size else 0
            lo, hi = dec(a[ys, x], a[ys + p, x])
            a[ys, x], a[ys + p, x] = lo, hi
        if (ny & p) and xs.size:                              # odd line: 1-D in x
            y = ys[-1] + p2 if ys.size else 0
            lo, hi = dec(a[y, xs], a[y, xs + p])
            a[y, xs], a[y, xs + p] = lo, hi
        p2 = p
        p >>= 1
    if w14:
        a[...] = a & 0xFFFF


def _unpiz(data: bytes, rows: int, W: int, channels) -> bytes:
    """One PIZ block -> the block's bytes in the uncompressed scan-line layout (line after line, channels alphabetically)."""
    sizes = [dt.itemsize // 2 for _, dt in channels]          # u16 words per pixel
    n_words = rows * W * sum(sizes)
    mn, mx_nz = struct.unpack_from("<HH", data, 0)
    pos = 4
    bitmap = np.zeros(8192, np.uint8)
    if mx_nz >= 8192:
        raise ValueError("EXR/PIZ: corrupt bitmap range")
    if mn <= mx_nz:
        bitmap[mn:mx_nz + 1] = np.frombuffer(data, np.uint8, mx_nz - mn + 1, pos); pos += mx_nz - mn + 1
    present = np.unpackbits(bitmap, bitorder="little").astype(bool)
    present[0] = True                                          # zero is never stored in the bitmap and always present
    lut = np.zeros(65536, np.uint16)
    vals = np.nonzero(present)[0]
    lut[:vals.size] = vals                                     # reverseLutFromBitmap
    max_value = vals.size - 1
    length, = struct.unpack_from("<i", data, pos); pos += 4
    if length < 0 or pos + length > len(data):
        raise ValueError("EXR/PIZ: corrupt block length")
    words = _huf_uncompress(data[pos:pos + length], n_words).astype(np.int32)
    out = np.empty((rows, sum(sizes) * W), np.uint16)
    at, col = 0, 0
    for size in sizes:
        plane = words[at:at + rows * W * size].reshape(rows, W, size)
        for j in range(size):
            sub = np.ascontiguousarray(plane[:, :, j])
            _wav2_decode(sub, max_value)
            plane[:, :, j] = sub
        out[:, col:col + W * size] = lut[plane.reshape(rows, W * size) & 0xFFFF]
        at += rows * W * size; col += W * size
    return out.astype("<u2").tobytes()


def read_exr(path: str) -> np.ndarray:
    """-> (H, W, C) float32; channels ordered R, G, B(, A) when the file has them, else alphabetically.
    A file that is truncated or corrupt raises ValueError (an unsupported feature NotImplementedError)."""
    import zlib
    try:
        return _read_exr(path)
    except (struct.error, KeyError, IndexError, zlib.error, OverflowError, MemoryError) as e:
        raise ValueError(f"{path}: truncated or corrupt OpenEXR file ({type(e).__name__}: {e})") from e


def _read_exr(path: str) -> np.ndarray:
    buf = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != MAGIC:
        raise ValueError(f"{path}: not an OpenEXR file")
    if version & 0x1A00:                        # tiled, deep data, multi-part
        raise NotImplementedError(f"{path}: only single-part scan-line OpenEXR files are supported")
    pos, attrs = 8, {}
    while buf[pos] != 0:
        name, pos = _cstr(buf, pos)
        typ, pos = _cstr(buf, pos)
        size, = struct.unpack_from("<i", buf, pos)
        attrs[name] = (typ, buf[pos + 4:pos + 4 + size])
        pos += 4 + size
    pos += 1
    channels, p, cl = [], 0, attrs["channels"][1]
    while cl[p] != 0:
        name, p = _cstr(cl, p)
        ptype, _plin, xs, ys = struct.unpack_from("<iB3xii", cl, p)
        p += 16
        if xs != 1 or ys != 1:
            raise NotImplementedError(f"{path}: sub-sampled channel {name}")
        channels.append((name, _DTYPES[ptype]))
    comp = attrs["compression"][1][0]
    if comp not in _LINES:
        raise NotImplementedError(f"{path}: {_NAMES.get(comp, comp)} compression is not supported (NONE, RLE, ZIPS, ZIP and PIZ are); re-save the file, e.g. `oiiotool in.exr --compression zip -o out.exr`")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    if W < 1 or H < 1 or W * H > (1 << 28) or not channels:
        raise ValueError(f"{path}: implausible data window {W} x {H} or no channels")
    nlines = _LINES[comp]
    nchunks = (H + nlines - 1) // nlines
    offsets = struct.unpack_from(f"<{nchunks}Q", buf, pos)
    line_bytes = sum(dt.itemsize for _, dt in channels) * W
    planes = {name: np.zeros((H, W), np.float32) for name, _ in channels}
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        rows = min(nlines, y1 - y + 1)
        if y < y0 or rows < 1 or size < 0 or off + 8 + size > len(buf):
            raise ValueError(f"{path}: corrupt scan-line block at offset {off}")
        data = buf[off + 8:off + 8 + size]
        if comp == NO_COMPRESSION or len(data) == rows * line_bytes:      # a block the coder could not shrink is stored raw
            raw = data
        elif comp == PIZ:
            raw = _unpiz(data, rows, W, channels)
        else:
            raw = _unzip(data, rows * line_bytes, rle=(comp == RLE))
        q = 0
        for r in range(rows):
            for name, dt in channels:           # the file lists (and stores) channels alphabetically
                n = W * dt.itemsize
                planes[name][y - y0 + r] = np.frombuffer(raw, dt, W, q).astype(np.float32)
                q += n
    order = [c for c in ("R", "G", "B", "A") if c in planes] or sorted(planes)
    if not {"R", "G", "B"} <= set(planes):
        order = sorted(planes)
    return np.stack([planes[c] for c in order], axis=-1)


def write_exr(path: str, image: np.ndarray, compression: int = ZIP, half: bool = False) -> None:
    """image: (H, W, 3|4) -> channels B, G, R(, A) as FLOAT or HALF."""
    img = np.asarray(image, np.float32)
    H, W, C = img.shape
    names = ["R", "G", "B", "A"][:C]
    dt = np.dtype("<f2") if half else np.dtype("<f4")
    if compression not in _LINES:
        raise NotImplementedError("write_exr: NONE, RLE, ZIPS or ZIP")
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", 1 if half else 2, 0, 1, 1) for n in sorted(names)) + b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val
    head = struct.pack("<ii", MAGIC, 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression])) + \
        attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + \
        attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + \
        attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    nlines = _LINES[compression]
    chunks = []
    for y in range(0, H, nlines):
        rows = min(nlines, H - y)
        raw = b"".join(img[y + r, :, names.index(n)].astype(dt).tobytes() for r in range(rows) for n in sorted(names))
        data = raw if compression == NO_COMPRESSION else _zip(raw, rle=(compression == RLE))
        chunks.append(struct.pack("<ii", y, len(data)) + data)
    table_at = len(head)
    offs, at = [], table_at + 8 * len(chunks)
    for c in chunks:
        offs.append(at); at += len(c)
    with open(path, "wb") as f:
        f.write(head + struct.pack(f"<{len(offs)}Q", *offs) + b"".join(chunks))
