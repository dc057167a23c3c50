"""numpy mirrors of the wire formats on the meshlet-visibility path.

Layouts follow the reference's shared C++/HLSL header source/shaders/ShaderInterop.h
(BasePassInstanceConstants :70-77, MeshLODData :174-180, MeshData :182-189, MeshletData :191-198,
MeshletAmplificationData :207-212, DispatchIndirectArguments :117-122, GPUCullingPassConstants
:131-144, BasePassConstants :49-68, MinMaxDownsampleConsts :214-218, NodeLocalTransform :220-227).
The C++ mirror with static_asserts is toyrenderer_amd/csrc/ShaderInterop.h.
"""
import numpy as np

kNumThreadsPerWave = 32
kMaxThreadGroupsPerDimension = 65535
kCullingFlagFrustumCullingEnable = 1
kCullingFlagOcclusionCullingEnable = 2
kCullingFlagMeshletConeCullingEnable = 4
kMaxNumMeshLODs = 8
kInvalidMeshLOD = 0xFF

BasePassInstanceConstants = np.dtype([
    ("m_WorldMatrix", np.float32, (4, 4)), ("m_PrevWorldMatrix", np.float32, (4, 4)),
    ("m_MeshDataIdx", np.uint32), ("m_MaterialDataIdx", np.uint32), ("PAD0", np.float32, (2,))])
MeshLODData = np.dtype([
    ("m_MeshletDataBufferIdx", np.uint32), ("m_NumMeshlets", np.uint32), ("m_Error", np.float32), ("PAD0", np.uint32)])
MeshData = np.dtype([
    ("m_BoundingSphere", np.float32, (4,)), ("m_MeshLODDatas", MeshLODData, (kMaxNumMeshLODs,)),
    ("m_NumLODs", np.uint32), ("m_GlobalVertexBufferIdx", np.uint32), ("m_GlobalIndexBufferIdx", np.uint32)])
MeshletData = np.dtype([
    ("m_BoundingSphere", np.float32, (4,)), ("m_ConeAxisAndCutoff", np.uint32),
    ("m_MeshletVertexIDsBufferIdx", np.uint32), ("m_MeshletIndexIDsBufferIdx", np.uint32),
    ("m_VertexAndTriangleCount", np.uint32)])
MeshletAmplificationData = np.dtype([
    ("m_InstanceConstIdx", np.uint32), ("m_MeshLOD", np.uint32), ("m_MeshletGroupOffset", np.uint32)])
DispatchIndirectArguments = np.dtype([
    ("m_ThreadGroupCountX", np.uint32), ("m_ThreadGroupCountY", np.uint32), ("m_ThreadGroupCountZ", np.uint32)])
GPUCullingPassConstants = np.dtype([
    ("m_NbInstances", np.uint32), ("m_CullingFlags", np.uint32), ("m_HZBDimensions", np.uint32, (2,)),
    ("m_Frustum", np.float32, (4,)), ("m_WorldToView", np.float32, (4, 4)), ("m_PrevWorldToView", np.float32, (4, 4)),
    ("m_NearPlane", np.float32), ("m_P00", np.float32), ("m_P11", np.float32), ("m_ForcedMeshLOD", np.uint32),
    ("m_MeshLODTarget", np.float32)])
BasePassConstants = np.dtype([
    ("m_WorldToClip", np.float32, (4, 4)), ("m_PrevWorldToClip", np.float32, (4, 4)), ("m_WorldToView", np.float32, (4, 4)),
    ("m_Frustum", np.float32, (4,)), ("m_HZBDimensions", np.uint32, (2,)), ("m_P00", np.float32), ("m_P11", np.float32),
    ("m_NearPlane", np.float32), ("m_CullingFlags", np.uint32), ("m_DebugMode", np.uint32), ("PAD0", np.uint32),
    ("m_OutputResolution", np.uint32, (2,)), ("m_bVisualizeMinMipTilesOnAlbedoOutput", np.uint32),
    ("m_bWriteSamplerFeedback", np.uint32)])
MinMaxDownsampleConsts = np.dtype([("m_OutputDimensions", np.uint32, (2,)), ("m_bDownsampleMax", np.uint32)])
DrawIndexedIndirectArguments = np.dtype([("m_IndexCount", np.uint32), ("m_InstanceCount", np.uint32), ("m_StartIndexLocation", np.uint32),
                                         ("m_BaseVertexLocation", np.int32), ("m_StartInstanceLocation", np.uint32)])     # ShaderInterop.h:108-115
GIProbeVisualizationUpdateConsts = np.dtype([                                                                              # ShaderInterop.h:249-261
    ("m_NumProbes", np.uint32), ("m_CameraOrigin", np.float32, (3,)), ("m_Frustum", np.float32, (4,)), ("m_WorldToView", np.float32, (4, 4)),
    ("m_HZBDimensions", np.uint32, (2,)), ("m_P00", np.float32), ("m_P11", np.float32), ("m_NearPlane", np.float32), ("m_ProbeRadius", np.float32),
    ("m_bHideInactiveProbes", np.uint32)])
NodeLocalTransform = np.dtype([
    ("m_ParentNodeIdx", np.uint32), ("m_Position", np.float32, (3,)), ("m_Rotation", np.float32, (4,)),
    ("m_Scale", np.float32, (3,)), ("PAD0", np.uint32)])
UpdateInstanceConstsPassConstants = np.dtype([("m_NumInstances", np.uint32)])

SIZES = {
    "BasePassInstanceConstants": 144, "MeshLODData": 16, "MeshData": 156, "MeshletData": 32,
    "MeshletAmplificationData": 12, "DispatchIndirectArguments": 12, "GPUCullingPassConstants": 180,
    "BasePassConstants": 256, "MinMaxDownsampleConsts": 12, "NodeLocalTransform": 48,
}
for _n, _s in SIZES.items():
    assert globals()[_n].itemsize == _s, (_n, globals()[_n].itemsize, _s)


RawVertexFormat = np.dtype([("m_Position", np.float32, (3,)), ("m_PackedNormal", np.uint32), ("m_TexCoord", np.uint16, (2,))])   # ShaderInterop.h:278-283
assert RawVertexFormat.itemsize == 20


def world_to_clip(world_to_view, view_to_clip) -> np.ndarray:
    """m_WorldToClip = WorldToView * ViewToClip in float32, summed left to right without fused multiply-add: the
    one definition used by every host side of this repo (Python and csrc/host/MathUtilities), so that the oracle
    and the GPU are handed the same 16 numbers."""
    a = np.asarray(world_to_view, np.float32).reshape(4, 4)
    b = np.asarray(view_to_clip, np.float32).reshape(4, 4)
    out = np.zeros((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            acc = np.float32(a[i, 0] * b[0, j])
            for k_ in (1, 2, 3):
                acc = np.float32(acc + np.float32(a[i, k_] * b[k_, j]))
            out[i, j] = acc
    return out


def get_next_pow2(x: int) -> int:
    """MathUtilities.h:47-61"""
    if x == 0:
        return 1
    x -= 1
    for s in (1, 2, 4, 8, 16):
        x |= x >> s
    return (x + 1) & 0xFFFFFFFF


def compute_nb_mips(w: int, h: int) -> int:
    """Graphic.h:227-231 (std::bit_width of the larger dimension)"""
    return int(max(w, h)).bit_length()


def hzb_dims(render_w: int, render_h: int):
    """BasePassRenderers.cpp:601-602"""
    return get_next_pow2(render_w) >> 1, get_next_pow2(render_h) >> 1


def hzb_layout(w: int, h: int):
    """Linear R16F mip chain used on the HIP side and by the oracle: mip k = max(w>>k,1) x
    max(h>>k,1) texels, row-major, mips packed back to back.  Returns (mips, offsets, total)."""
    mips = compute_nb_mips(w, h)
    offs, off = [], 0
    for k in range(mips):
        offs.append(off)
        off += max(w >> k, 1) * max(h >> k, 1)
    return mips, offs, off


def fmaf(a, b, c) -> np.float32:
    """Exact scalar float32 fused multiply-add (std::fmaf): the product is exact in float64, the
    float64 sum is rounded to odd, then rounded once to float32."""
    a, b, c = np.float64(np.float32(a)), np.float64(np.float32(b)), np.float64(np.float32(c))
    p = a * b
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)
    if err != 0 and np.isfinite(s) and (np.float64(s).view(np.int64) & 1) == 0:
        s = np.nextafter(s, np.inf if err > 0 else -np.inf)
    return np.float32(s)
