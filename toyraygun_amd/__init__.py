"""toyraygun_amd -- MI355X-native path tracer behind ToyRaygun's Renderer/Scene plugin surface.

Python reaches the renderer only through the C ABI of include/trg.h (toyraygun_amd.capi); the
host-side C++ mirror of the reference's Engine/Renderer/Scene classes lives in libtoyraygun.so
(toyraygun_amd.host).  There is no CPU fallback: importing works anywhere, creating a context
requires a gfx950 device and the built HIP library.
"""
from . import capi  # noqa: F401
from .capi import Context, Stats, TrgError, Uniforms  # noqa: F401

__all__ = ["capi", "Context", "Stats", "TrgError", "Uniforms"]
