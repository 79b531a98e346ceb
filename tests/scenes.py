"""Shared synthetic scenes for the parity tests (small enough for the CPU oracle to finish in seconds)."""
from __future__ import annotations

import functools

import numpy as np

from feature_tracker_amd import synth


@functools.lru_cache(maxsize=None)
def scene(width, height, levels, motion="easy", kind="translation"):
    """Returns (ref_levels, cur_levels) host pyramids."""
    t = (3.3, -2.1) if motion == "easy" else (11.7, -8.4)
    if kind == "translation":
        ref, cur = synth.make_image_pair(width, height, t)
    elif kind == "similarity":
        ref, cur = synth.make_image_pair(width, height, t, rotation_deg=1.5, scale=1.02)
    elif kind == "flat":
        ref = np.full((height, width), 117, dtype=np.uint8)
        cur = ref.copy()
    else:
        raise ValueError(kind)
    return tuple(synth.build_pyramid(ref, levels)), tuple(synth.build_pyramid(cur, levels))


def features(n, width, height, half, seed=12345, border_fraction=0.05):
    return synth.make_features(n, width, height, seed=seed, margin=min(40.0, width / 8.0), border_fraction=border_fraction, half=half)
