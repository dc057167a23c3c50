// ShaderInterop.h -- wire formats of the meshlet-visibility path, shared by the C++ host mirror
// and the HIP kernels.  Restates the layouts of the reference's shared C++/HLSL header
// source/shaders/ShaderInterop.h (only the structs on the hot path); every
// size/offset the HLSL side relies on is pinned by a static_assert.
#pragma once

#include <cstddef>
#include <cstdint>

namespace interop
{

// ShaderInterop.h:6-7
static constexpr uint32_t kNumThreadsPerWave = 32;              // reference group width (D3D wave32); a gfx950 wave64 runs two groups
static constexpr uint32_t kMaxThreadGroupsPerDimension = 65535;

// ShaderInterop.h:15-17
static constexpr uint32_t kCullingFlagFrustumCullingEnable = (1u << 0);
static constexpr uint32_t kCullingFlagOcclusionCullingEnable = (1u << 1);
static constexpr uint32_t kCullingFlagMeshletConeCullingEnable = (1u << 2);

// ShaderInterop.h:19-24
static constexpr uint32_t kMaxMeshletVertices = 64;
static constexpr uint32_t kMaxMeshletTriangles = 96;
static constexpr uint32_t kMaxNumMeshLODs = 8;
static constexpr uint32_t kInvalidMeshLOD = 0xFF;

struct Vector2U { uint32_t x, y; };
struct Vector3U { uint32_t x, y, z; };
struct Vector4 { float x, y, z, w; };
struct Matrix { float m[4][4]; }; // row-major, row vectors (compileallshaders.bat:73 --matrixRowMajor)

// ShaderInterop.h:49-68
struct BasePassConstants
{
    Matrix m_WorldToClip;
    Matrix m_PrevWorldToClip;
    Matrix m_WorldToView;
    Vector4 m_Frustum;
    Vector2U m_HZBDimensions;
    float m_P00;
    float m_P11;
    float m_NearPlane;
    uint32_t m_CullingFlags;
    uint32_t m_DebugMode;
    uint32_t PAD0;
    Vector2U m_OutputResolution;
    uint32_t m_bVisualizeMinMipTilesOnAlbedoOutput;
    uint32_t m_bWriteSamplerFeedback;
};

// ShaderInterop.h:70-77
struct BasePassInstanceConstants
{
    Matrix m_WorldMatrix;
    Matrix m_PrevWorldMatrix;
    uint32_t m_MeshDataIdx;
    uint32_t m_MaterialDataIdx;
    float PAD0[2];
};

// ShaderInterop.h:117-122
struct DispatchIndirectArguments
{
    uint32_t m_ThreadGroupCountX;
    uint32_t m_ThreadGroupCountY;
    uint32_t m_ThreadGroupCountZ;
};

// ShaderInterop.h:131-144
struct GPUCullingPassConstants
{
    uint32_t m_NbInstances;
    uint32_t m_CullingFlags;
    Vector2U m_HZBDimensions;
    Vector4 m_Frustum;
    Matrix m_WorldToView;
    Matrix m_PrevWorldToView;
    float m_NearPlane;
    float m_P00;
    float m_P11;
    uint32_t m_ForcedMeshLOD;
    float m_MeshLODTarget;
};

// ShaderInterop.h:174-180
struct MeshLODData
{
    uint32_t m_MeshletDataBufferIdx;
    uint32_t m_NumMeshlets;
    float m_Error;
    uint32_t PAD0;
};

// ShaderInterop.h:182-189
struct MeshData
{
    Vector4 m_BoundingSphere;
    MeshLODData m_MeshLODDatas[kMaxNumMeshLODs];
    uint32_t m_NumLODs;
    uint32_t m_GlobalVertexBufferIdx;
    uint32_t m_GlobalIndexBufferIdx;
};

// ShaderInterop.h:191-198
struct MeshletData
{
    Vector4 m_BoundingSphere;
    uint32_t m_ConeAxisAndCutoff; // 4x u8: axis xyz mapped [0,255] -> [-1,1], cutoff /255
    uint32_t m_MeshletVertexIDsBufferIdx;
    uint32_t m_MeshletIndexIDsBufferIdx;
    uint32_t m_VertexAndTriangleCount;
};

// ShaderInterop.h:200-205 (Q9: 64 slots, at most 32 written)
struct MeshletPayload
{
    uint32_t m_MeshletIndices[64];
    uint32_t m_InstanceConstIdx;
    uint32_t m_MeshLOD;
};

// ShaderInterop.h:207-212
struct MeshletAmplificationData
{
    uint32_t m_InstanceConstIdx;
    uint32_t m_MeshLOD;
    uint32_t m_MeshletGroupOffset;
};

// ShaderInterop.h:108-115
struct DrawIndexedIndirectArguments
{
    uint32_t m_IndexCount;
    uint32_t m_InstanceCount;
    uint32_t m_StartIndexLocation;
    int32_t  m_BaseVertexLocation;
    uint32_t m_StartInstanceLocation;
};
static_assert(sizeof(DrawIndexedIndirectArguments) == 20, "DrawIndexedIndirectArguments");

// ShaderInterop.h:249-261
struct GIProbeVisualizationUpdateConsts
{
    uint32_t m_NumProbes;
    float m_CameraOrigin[3];
    Vector4 m_Frustum;
    Matrix m_WorldToView;
    Vector2U m_HZBDimensions;
    float m_P00;
    float m_P11;
    float m_NearPlane;
    float m_ProbeRadius;
    uint32_t m_bHideInactiveProbes;
};
static_assert(sizeof(GIProbeVisualizationUpdateConsts) == 124 && offsetof(GIProbeVisualizationUpdateConsts, m_WorldToView) == 32, "GIProbeVisualizationUpdateConsts");

// ShaderInterop.h:214-218
struct MinMaxDownsampleConsts
{
    Vector2U m_OutputDimensions;
    uint32_t m_bDownsampleMax;
};

// ShaderInterop.h:220-227
struct NodeLocalTransform
{
    uint32_t m_ParentNodeIdx;
    float m_Position[3];
    float m_Rotation[4];
    float m_Scale[3];
    uint32_t PAD0;
};

// ShaderInterop.h:317-320
struct UpdateInstanceConstsPassConstants
{
    uint32_t m_NumInstances;
};
// this build's extension (not in the reference): the same pass over the instance range [m_FirstInstance, + m_NumInstances) --
// a rank of a sharded scene updates the instances it culls (csrc/host/BasePassRenderers.cpp, Scene::m_InstanceUpdate*)
struct UpdateInstanceConstsShardConstants
{
    uint32_t m_NumInstances;
    uint32_t m_FirstInstance;
};

// FFXHelpers.cpp:15-23 (push constants of the SPD pass)
struct SPDConstants
{
    uint32_t mips;
    uint32_t numWorkGroups;
    uint32_t workGroupOffset[2];
    float invInputSize[2]; // only used for linear sampling mode
    float padding[2];
};

// ---- extension of this build (documented in DESIGN.md) --------------------------------------
// The meshlet-dispatch argument buffer may be 16 bytes: the 4th word receives the number of
// leading amplification records that are defined (Q2: everything from the first dropped
// instance on is undefined in the reference).  A 12-byte buffer keeps the reference layout.
struct DispatchIndirectArgumentsEx
{
    DispatchIndirectArguments m_Args;
    uint32_t m_ValidRecords;
};

static_assert(sizeof(Matrix) == 64);
static_assert(sizeof(BasePassConstants) == 256);
static_assert(offsetof(BasePassConstants, m_WorldToView) == 128);
static_assert(offsetof(BasePassConstants, m_Frustum) == 192);
static_assert(offsetof(BasePassConstants, m_HZBDimensions) == 208);
static_assert(offsetof(BasePassConstants, m_NearPlane) == 224);
static_assert(offsetof(BasePassConstants, m_CullingFlags) == 228);
static_assert(sizeof(BasePassInstanceConstants) == 144);
static_assert(offsetof(BasePassInstanceConstants, m_MeshDataIdx) == 128);
static_assert(sizeof(DispatchIndirectArguments) == 12);
static_assert(sizeof(GPUCullingPassConstants) == 180);
static_assert(offsetof(GPUCullingPassConstants, m_Frustum) == 16);
static_assert(offsetof(GPUCullingPassConstants, m_WorldToView) == 32);
static_assert(offsetof(GPUCullingPassConstants, m_PrevWorldToView) == 96);
static_assert(offsetof(GPUCullingPassConstants, m_NearPlane) == 160);
static_assert(offsetof(GPUCullingPassConstants, m_MeshLODTarget) == 176);
static_assert(sizeof(MeshLODData) == 16);
static_assert(sizeof(MeshData) == 156);
static_assert(offsetof(MeshData, m_MeshLODDatas) == 16);
static_assert(offsetof(MeshData, m_NumLODs) == 144);
static_assert(sizeof(MeshletData) == 32);
static_assert(offsetof(MeshletData, m_ConeAxisAndCutoff) == 16);
static_assert(sizeof(MeshletPayload) == 264);
static_assert(sizeof(MeshletAmplificationData) == 12);
static_assert(sizeof(MinMaxDownsampleConsts) == 12);
static_assert(sizeof(NodeLocalTransform) == 48);
static_assert(sizeof(SPDConstants) == 32);
static_assert(sizeof(DispatchIndirectArgumentsEx) == 16);

} // namespace interop
