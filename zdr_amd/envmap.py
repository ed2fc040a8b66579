"""Environment lighting, host half (/root/reference/envmap.py:17-57,116-203): importance-sampling tables
for a lat-long environment map.  The device half (sample_envmap, env_sampled_light_pdf, the texture
lookup) lives in csrc/scene.h; the C-ABI receives the texture and the finished tables.

Unpinned third-party behaviour: the reference samples the map through LuisaCompute's
``heap.texture2d_sample`` with "default filter & address mode" (envmap.py:130).  Here a lookup is
bilinear between texel centres ((i + 0.5) / N) with clamp-to-edge addressing, in the weight map
below, in the oracle and in the kernels alike.
"""
from __future__ import annotations

import math

import numpy as np

SAMPLE_MAP_W, SAMPLE_MAP_H = 512, 256          # envmap.py:114


def load_image(path: str) -> np.ndarray:
    """envmap.py:117-121 reads the file with imageio; here: OpenEXR (zdr_amd/exr.py) or a .npy array."""
    if path.lower().endswith(".exr"):
        from .exr import read_exr
        return read_exr(path)
    if path.lower().endswith(".npy"):
        return np.load(path)
    raise NotImplementedError(f"{path}: environment maps are read from .exr or .npy files (or passed as arrays)")


def prepare_image(img) -> np.ndarray:
    """(H, W, 3|4) -> float32 RGBA, made square like load_envmap (envmap.py:122-128, render.py:151-154)."""
    img = np.asarray(img, np.float32)
    if img.ndim != 3 or img.shape[2] not in (3, 4):
        raise ValueError("envmap must be (H, W, 3) or (H, W, 4)")
    if img.shape[2] == 3:
        img = np.concatenate([img, np.ones_like(img[..., :1])], axis=-1)
    if img.shape[0] != img.shape[1]:
        if img.shape[1] == img.shape[0] * 2:
            img = img.repeat(2, axis=0)
        else:
            raise RuntimeError("envmap must be strictly 1:2 or 1:1")
    return np.ascontiguousarray(img, np.float32)


def texture_sample(img: np.ndarray, u: np.ndarray, v: np.ndarray) -> np.ndarray:
    """bilinear, texel centres at (i + 0.5) / N, clamp to edge; u, v broadcastable float32 arrays."""
    H, W = img.shape[:2]
    x = u.astype(np.float32) * np.float32(W) - np.float32(0.5)
    y = v.astype(np.float32) * np.float32(H) - np.float32(0.5)
    x0f, y0f = np.floor(x), np.floor(y)
    fx, fy = (x - x0f).astype(np.float32), (y - y0f).astype(np.float32)
    x0 = np.clip(x0f.astype(np.int64), 0, W - 1); x1 = np.clip(x0f.astype(np.int64) + 1, 0, W - 1)
    y0 = np.clip(y0f.astype(np.int64), 0, H - 1); y1 = np.clip(y0f.astype(np.int64) + 1, 0, H - 1)
    c00, c10, c01, c11 = img[y0, x0], img[y0, x1], img[y1, x0], img[y1, x1]
    top = c00 + (c10 - c00) * fx[..., None]
    bot = c01 + (c11 - c01) * fx[..., None]
    return (top + (bot - top) * fy[..., None]).astype(np.float32)


def weight_map(img: np.ndarray) -> np.ndarray:
    """generate_weight_map_kernel (envmap.py:136-159): Gaussian-filtered luminance * sin(theta), 17 x 17 taps."""
    W, H = SAMPLE_MAP_W, SAMPLE_MAP_H
    cx = (np.arange(W, dtype=np.float32) + np.float32(0.5))[None, :]
    cy = (np.arange(H, dtype=np.float32) + np.float32(0.5))[:, None]
    n = int(math.ceil(1.0 / 0.125))
    sum_w = np.float32(0.0); sum_s = np.zeros((H, W), np.float32)
    for dy in range(-n, n + 1):
        for dx in range(-n, n + 1):
            ox, oy = np.float32(dx * 0.125), np.float32(dy * 0.125)
            u = np.broadcast_to((cx + ox) / np.float32(W), (H, W))
            v = np.broadcast_to((cy + oy) / np.float32(H), (H, W))
            rgb = texture_sample(img, u, v)
            scale = np.float32(0.212671) * rgb[..., 0] + np.float32(0.715160) * rgb[..., 1] + np.float32(0.072169) * rgb[..., 2]
            w = np.float32(math.exp(-4.0 * (float(ox) ** 2 + float(oy) ** 2)))
            sum_s += w * np.minimum(scale * np.sin(v * np.float32(math.pi)), np.float32(1e8))
            sum_w += w
    return (sum_s / sum_w).astype(np.float32)


def create_alias_table(values):
    """Vose alias table as the reference builds it (envmap.py:17-57): returns (prob, alias, pdf)."""
    values = [float(v) for v in values]
    n = len(values)
    total = sum(abs(v) for v in values)
    pdf = [1.0 / n] * n if total == 0.0 else [abs(v) / total for v in values]
    ratio = n / total if total > 0.0 else 1.0
    prob = [v * ratio for v in values]
    alias = list(range(n))
    over = [i for i, p in enumerate(prob) if p > 1.0]
    under = [i for i, p in enumerate(prob) if p < 1.0]
    while over and under:
        o, u = over.pop(), under.pop()
        prob[o] -= 1.0 - prob[u]
        alias[u] = o
        if prob[o] > 1.0:
            over.append(o)
        elif prob[o] < 1.0:
            under.append(o)
    for i in over + under:
        prob[i], alias[i] = 1.0, i
    return prob, alias, pdf


def build_tables(img: np.ndarray, compensate_mis: bool = True):
    """load_envmap (envmap.py:133-200): returns (alias_prob float32[H + H*W], alias_idx int32[same], pdf float32[H*W])
    with the marginal p(y) table first, then the H conditional p(x|y) tables."""
    W, H = SAMPLE_MAP_W, SAMPLE_MAP_H
    scale_map = weight_map(img).reshape(-1).copy()
    if compensate_mis:                                              # envmap.py:167-175
        row_weight = [math.sin((y + 0.5) / H * math.pi) for y in range(H)]
        average_scale = scale_map.mean()
        weight_average = float(np.mean(row_weight))
        for y in range(H):
            scale_map[y * W:(y + 1) * W] -= average_scale * row_weight[y] / weight_average
        scale_map = np.maximum(scale_map, 0.0)
    probs, aliases, pdfs, row_avg = [], [], [], []
    for y in range(H):
        row = scale_map[y * W:(y + 1) * W]
        row_avg.append(row.mean())
        p, a, d = create_alias_table(row)
        probs.extend(p); aliases.extend(a); pdfs.extend(d)
    mp, ma, mpdf = create_alias_table(row_avg)
    pdf = np.asarray(pdfs, np.float64).reshape(H, W) * (np.asarray(mpdf, np.float64)[:, None] * (W * H))
    return (np.asarray(mp + probs, np.float32), np.asarray(ma + aliases, np.int32), np.ascontiguousarray(pdf.reshape(-1), np.float32))
