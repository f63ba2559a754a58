"""plvs_amd — MI355X-native hot path of PLVS (ORB/LBD front end, Hamming
matching, TSDF integrate) behind PLVS's own operator surface.

Only the thin host-side mirror of the reference interface lives in Python; all
compute is in plvs_amd/lib/libplvs_hip.so (hand-written HIP for gfx950).
"""
from . import _lib  # noqa: F401  (raises if the HIP library is missing)
from .matcher import BinaryDescriptorMatcher, BFMatcherHamming, DMatch  # noqa: F401
from .lines import LineExtractor, LSDOptions  # noqa: F401
from .orb import ORBextractor  # noqa: F401
from .tsdf import PointCloudMapChisel, TsdfChisel  # noqa: F401

__all__ = ["BinaryDescriptorMatcher", "BFMatcherHamming", "DMatch", "ORBextractor", "LineExtractor", "LSDOptions",
           "PointCloudMapChisel", "TsdfChisel"]
