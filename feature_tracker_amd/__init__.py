"""feature_tracker_amd — MI355X-native pyramidal KLT trackers and descriptor matchers (BRIEF Hamming, float cosine).

Product layout: ``csrc/`` (hand-written HIP kernels + the C ABI of ``include/ftk.h``),
``host/`` (C++ classes with the reference's names on top of the ABI), ``tracker.py`` (the same
interface for Python callers), ``dist.py`` (feature sharding across the GPUs of a node) and
``synth.py`` (deterministic synthetic workloads).  Importing the package does not load the
native library; the first compute call does, and fails loudly if it has not been built.
"""
from . import synth  # noqa: F401
from .tracker import (  # noqa: F401
    BriefDescriptor, BriefMatcher, CosineMatcher, DirectMethod, DirectMethodOptions, DiskMatcher, SuperpointMatcher, Context, FeaturePointHarrisDetector, DescriptorMatcherOptions, ImagePyramid, OpticalFlow, OpticalFlowAffineKlt, OpticalFlowBasicKlt,
    OpticalFlowLssdKlt, OpticalFlowOptions, default_context, pack_brief, unpack_brief, refresh_env_switches,
    NOT_TRACKED, TRACKED, LARGE_RESIDUAL, OUTSIDE, NUMERIC_ERROR,
)

__all__ = [
    "BriefDescriptor", "BriefMatcher", "CosineMatcher", "DirectMethod", "DirectMethodOptions", "DiskMatcher", "SuperpointMatcher", "Context", "FeaturePointHarrisDetector", "DescriptorMatcherOptions", "ImagePyramid", "OpticalFlow", "OpticalFlowAffineKlt", "OpticalFlowBasicKlt",
    "OpticalFlowLssdKlt", "OpticalFlowOptions", "default_context", "pack_brief", "unpack_brief", "refresh_env_switches", "synth",
    "NOT_TRACKED", "TRACKED", "LARGE_RESIDUAL", "OUTSIDE", "NUMERIC_ERROR",
]
