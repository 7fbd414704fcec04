"""itr_amd -- MI355X-native encode -> score -> hinge-loss / Recall@K hot path of
WangFei-2019/Image-text-Retrieval, behind the reference's own Python seams.

Arithmetic lives in hand-written gfx950 HIP kernels (csrc/) reached through the C ABI of
include/itr_hip.h; this package only mirrors the reference's call signatures on top of it.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "ops"]
