"""toyrenderer_amd -- MI355X-native (gfx950, HIP) implementation of ToyRenderer's GPU-driven
meshlet visibility path (two-phase instance + meshlet frustum / HZB-occlusion / cone cull, LOD
select, ordered visible-list compaction) behind the reference's RenderGraph / IRenderer /
Graphic::AddComputePass API.  See DESIGN.md.

Sub-modules:
  interop  numpy mirrors of the wire formats (ShaderInterop.h)
  synth    synthetic scene / camera / depth generators (bench + tests)
  rhi      ctypes binding of the C-ABI device layer (include/trhip.h -> lib/libtrhip.so)
  host     ctypes binding of the C++ host mirror (RenderGraph, Graphic, BasePassRenderer)
The HIP library is loaded lazily by rhi/host and loading FAILS LOUDLY if it is missing: there is
no CPU fallback in the product.
"""
__all__ = ["interop", "synth", "rhi", "host"]
