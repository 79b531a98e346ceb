"""Deterministic synthetic inputs for the tracker / matcher hot path.

Everything here is pure numpy integer/float64 arithmetic (no libm transcendental in the image
generator except one cos/sin pair for the global motion), so both hosts produce identical bytes.

* images : 8-bit gray procedural value noise, three octaves (cell 24 / 9 / 4 px).
* motion : cur(q) = ref(T^-1 q) with T = similarity (scale, rotation about the image centre,
           translation); the noise function is continuous, so the warp is exact (no resampling).
* pyramid: host-side truncating 2x2 box mean (the repo's normative CreateImagePyramid).
* features: uniform fractional positions away from the border, plus 1 % near the border to
           exercise the validity paths.
* descriptors: BRIEF-like 256-bit strings, cur = permuted ref with 20 bit flips.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "value_noise", "make_image_pair", "build_pyramid", "make_features", "make_descriptors",
    "pack_bits", "CONFIGS",
]

# BASELINE.json configs (N, W, H, levels, half patch, model, method)
CONFIGS = {
    "config1": dict(n=200, width=640, height=480, levels=3, half=5, model="basic", method="inverse"),
    "config2": dict(n=2000, width=640, height=480, levels=4, half=10, model="basic", method="inverse"),
    "config3": dict(n=5000, width=1280, height=720, levels=5, half=6, model="affine", method="inverse"),
    "config4": dict(n=10000, width=640, height=480, levels=4, half=6, model="lssd", method="fast"),
    "config5_shard": dict(n=25000, width=1920, height=1080, levels=4, half=6, model="basic", method="inverse"),
}


def _hash01(ix: np.ndarray, iy: np.ndarray, seed: int) -> np.ndarray:
    m = np.uint64(0xFFFFFFFF)
    h = (ix.astype(np.int64).astype(np.uint64) & m) * np.uint64(0x9E3779B1)
    h = (h + (iy.astype(np.int64).astype(np.uint64) & m) * np.uint64(0x85EBCA77)) & m
    h = (h + np.uint64(seed) * np.uint64(0xC2B2AE3D)) & m
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & m
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & m
    h ^= h >> np.uint64(15)
    return h.astype(np.float64) / 4294967296.0


def _lattice_noise(x: np.ndarray, y: np.ndarray, cell: float, seed: int) -> np.ndarray:
    gx = x / cell
    gy = y / cell
    ix = np.floor(gx)
    iy = np.floor(gy)
    fx = gx - ix
    fy = gy - iy
    sx = fx * fx * (3.0 - 2.0 * fx)
    sy = fy * fy * (3.0 - 2.0 * fy)
    ix = ix.astype(np.int64)
    iy = iy.astype(np.int64)
    n00 = _hash01(ix, iy, seed)
    n10 = _hash01(ix + 1, iy, seed)
    n01 = _hash01(ix, iy + 1, seed)
    n11 = _hash01(ix + 1, iy + 1, seed)
    top = n00 + (n10 - n00) * sx
    bot = n01 + (n11 - n01) * sx
    return top + (bot - top) * sy


def value_noise(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Continuous intensity in [0, 255] at float coordinates (x = col, y = row)."""
    v = 0.5 * _lattice_noise(x, y, 24.0, 1) + 0.3 * _lattice_noise(x, y, 9.0, 2) + 0.2 * _lattice_noise(x, y, 4.0, 3)
    return np.clip(255.0 * v, 0.0, 255.0)


def make_image_pair(width: int, height: int, translation=(3.3, -2.1), rotation_deg: float = 0.0, scale: float = 1.0):
    """Returns (ref, cur) uint8 images of shape (height, width); a ref point p appears in cur at
    T(p) = scale * R(rotation) (p - c) + c + translation."""
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    ref = np.floor(value_noise(xs, ys) + 0.5).astype(np.uint8)
    cx, cy = (width - 1) * 0.5, (height - 1) * 0.5
    th = np.deg2rad(rotation_deg)
    c, s = np.cos(th), np.sin(th)
    # inverse map: p = R^T (q - c - t) / scale + c
    qx = xs - cx - translation[0]
    qy = ys - cy - translation[1]
    px = (c * qx + s * qy) / scale + cx
    py = (-s * qx + c * qy) / scale + cy
    cur = np.floor(value_noise(px, py) + 0.5).astype(np.uint8)
    return np.ascontiguousarray(ref), np.ascontiguousarray(cur)


def build_pyramid(image: np.ndarray, levels: int):
    """Host pyramid: level 0 is the image itself, level i+1 the truncating 2x2 box mean of level i."""
    out = [np.ascontiguousarray(image)]
    for _ in range(1, levels):
        src = out[-1]
        r, c = src.shape[0] // 2, src.shape[1] // 2
        s = src[: 2 * r, : 2 * c].astype(np.uint16)
        dst = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2]) >> 2
        out.append(np.ascontiguousarray(dst.astype(np.uint8)))
    return out


def make_features(n: int, width: int, height: int, seed: int = 12345, margin: float = 40.0, border_fraction: float = 0.01, half: int = 6):
    """(n, 2) float32 array of (u, v) = (col, row); the last ~1 % sit within `half` px of the border."""
    rs = np.random.RandomState(seed)
    uv = np.empty((n, 2), dtype=np.float64)
    uv[:, 0] = margin + rs.random_sample(n) * (width - 2.0 * margin)
    uv[:, 1] = margin + rs.random_sample(n) * (height - 2.0 * margin)
    n_border = int(round(n * border_fraction))
    if n_border > 0:
        side = rs.randint(0, 4, size=n_border)
        off = rs.random_sample(n_border) * half
        along_w = rs.random_sample(n_border) * (width - 1)
        along_h = rs.random_sample(n_border) * (height - 1)
        bu = np.where(side == 0, off, np.where(side == 1, (width - 1) - off, along_w))
        bv = np.where(side == 2, off, np.where(side == 3, (height - 1) - off, along_h))
        uv[n - n_border:, 0] = bu
        uv[n - n_border:, 1] = bv
    return uv.astype(np.float32)


def make_descriptors(n_ref: int, n_cur: int | None = None, n_bits: int = 256, flips: int = 20, seed: int = 7):
    """Per-bit descriptors (uint8 0/1, shape (n, n_bits)): cur[j] = ref[(7919 j) mod n_ref] with `flips` flipped bits."""
    n_cur = n_ref if n_cur is None else n_cur
    rs = np.random.RandomState(seed)
    ref = rs.randint(0, 2, size=(n_ref, n_bits)).astype(np.uint8)
    perm = (7919 * np.arange(n_cur, dtype=np.int64)) % n_ref
    cur = ref[perm].copy()
    for j in range(n_cur):
        idx = rs.choice(n_bits, size=flips, replace=False)
        cur[j, idx] ^= 1
    return ref, cur, perm.astype(np.int32)


def make_float_descriptors(n_ref: int, n_cur: int | None = None, dim: int = 256, noise: float = 0.25, seed: int = 11, normalize: bool = True):
    """Float descriptors shaped like SuperPoint (dim 256) / DISK (dim 128) outputs: ref rows are random unit
    vectors, cur[j] = ref[(7919 j) mod n_ref] + noise * gaussian (re-normalised), so the true match has cosine
    distance ~ noise^2 / 4 and every other pair ~ 0.5.  Returns (ref, cur, perm)."""
    n_cur = n_ref if n_cur is None else n_cur
    rs = np.random.RandomState(seed)
    ref = rs.standard_normal((n_ref, dim)).astype(np.float32)
    perm = (7919 * np.arange(n_cur, dtype=np.int64)) % max(n_ref, 1)
    cur = ref[perm] / np.sqrt(dim, dtype=np.float32) * np.float32(np.sqrt(dim)) if n_ref else np.zeros((n_cur, dim), np.float32)
    cur = (cur + np.float32(noise) * rs.standard_normal((n_cur, dim)).astype(np.float32)).astype(np.float32)
    if normalize:
        ref = (ref / np.linalg.norm(ref, axis=1, keepdims=True)).astype(np.float32) if n_ref else ref
        cur = (cur / np.linalg.norm(cur, axis=1, keepdims=True)).astype(np.float32) if n_cur else cur
    return np.ascontiguousarray(ref), np.ascontiguousarray(cur), perm.astype(np.int32)


def pack_bits(bits: np.ndarray) -> np.ndarray:
    """(n, n_bits) 0/1 bytes -> (n, ceil(n_bits/32)) uint32 words, bit i of the descriptor in bit (i % 32) of word i // 32."""
    n, n_bits = bits.shape
    words = (n_bits + 31) // 32
    padded = np.zeros((n, words * 32), dtype=np.uint8)
    padded[:, :n_bits] = bits & 1
    packed = np.packbits(padded.reshape(n, words, 32), axis=-1, bitorder="little")  # (n, words, 4) bytes
    return np.ascontiguousarray(packed).view("<u4").reshape(n, words)
