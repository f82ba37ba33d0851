"""renderbaby_amd -- MI355X (gfx950) backend for RenderBaby's path-tracing hot path.

``csrc/`` holds the HIP kernels and the C-ABI runtime (include/rb_abi.h);
``engine`` mirrors the reference's Renderer / FrameIterator interface over that
ABI; ``scenes`` generates the seeded synthetic configurations of BASELINE.json.
"""
from . import abi  # noqa: F401
from .engine import Change, Engine, Frame, FrameIterator, RenderConfig, RenderConfigBuilder, RenderError  # noqa: F401

__all__ = ["abi", "Change", "Engine", "Frame", "FrameIterator", "RenderConfig", "RenderConfigBuilder", "RenderError"]
