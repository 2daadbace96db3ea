"""MI355X (gfx950) implementation of the Lowcash/SfM per-frame hot path.

The compute lives in sfm_amd/lib/libsfm_hip.so (hand-written HIP kernels behind
the C ABI declared in include/sfm_hip.h).  This package is the Python host side:
`capi` binds the C ABI with ctypes, `feature_processing` and `reconstruction`
mirror the reference's wrapper classes (DescriptorMatcher, OptFlow,
Reconstruction), `allpairs` shards windowed matching over ranks.
There is no CPU fallback: without the library or without a GPU every compute
call raises.
"""
from . import capi  # noqa: F401
from .feature_processing import DescriptorMatcher, FeatureDetector, OptFlow  # noqa: F401
from .reconstruction import CameraParameters, Reconstruction  # noqa: F401
from .tracking import RecoveryPose, Tracking  # noqa: F401

__all__ = ["capi", "DescriptorMatcher", "FeatureDetector", "OptFlow", "Reconstruction", "CameraParameters", "RecoveryPose", "Tracking"]
