"""Minimal OpenEXR reader / writer for environment maps (Scene.add_envmap; the reference reads its EXR files with
imageio, /root/reference/envmap.py:117-121, which this environment does not ship).

Scope: single-part scan-line files, channels of type HALF / FLOAT / UINT with sampling 1, compression NONE, RLE, ZIPS or ZIP
(what Blender, OpenEXR's own tools and most HDRI libraries write by default or on request).  PIZ, PXR24, B44 and DWA
files are refused with a message naming the compression.  Layout per the OpenEXR file-layout document: magic 0x01312f76,
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
_LINES = {NO_COMPRESSION: 1, RLE: 1, ZIPS: 1, ZIP: 16}
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
        raise NotImplementedError(f"{path}: {_NAMES.get(comp, comp)} compression is not supported (NONE, RLE, ZIPS and ZIP are); re-save the file, e.g. `oiiotool in.exr --compression zip -o out.exr`")
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
        raw = data if comp == NO_COMPRESSION else _unzip(data, rows * line_bytes, rle=(comp == RLE))
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
