// Scene.cpp -- see Scene.h.  Cites are to the reference's source/Scene.cpp.
#include "Scene.h"

#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "HostProfile.h"

#include <cstring>

#include "Graphic.h"
#include "GraphicConstants.h"
#include "../ShaderInterop.h"

extern IRenderer* g_UpdateInstanceConstsRenderer;
extern IRenderer* g_GBufferRenderer;
extern IRenderer* g_GIDebugRenderer;

void View::Update()
{
    // Scene.cpp:116-118: keep last frame's matrices
    m_PrevWorldToView = m_WorldToView;
    m_PrevViewToClip = m_ViewToClip;
    if (m_bHasPending) m_WorldToView = m_PendingWorldToView;                  // :121-122 (camera from the application)

    if (!m_bUseExplicitProjection) {                                          // :124-125
        const float kKindaBigNumber = 1e10f;
        m_ViewToClip = CreatePerspectiveFieldOfView(m_FOV, m_AspectRatio, m_ZNearP, kKindaBigNumber);
        ModifyPerspectiveMatrix(m_ViewToClip, m_ZNearP, kKindaBigNumber, GraphicConstants::kInversedDepthBuffer, GraphicConstants::kInfiniteDepthBuffer);
    }
    // TAA jitter (:127-131) is out of scope (no TAA renderer)

    if (!g_Scene->m_bFreezeCullingCamera) {                                   // :139-144
        m_CullingPrevWorldToView = m_PrevWorldToView;
        m_CullingWorldToView = m_WorldToView;
    }
}

void Scene::Initialize()
{
    m_RenderGraph = std::make_shared<RenderGraph>();
    m_RenderGraph->Initialize();
    m_View.m_AspectRatio = (float)g_Graphic.m_RenderResolution.x / (float)g_Graphic.m_RenderResolution.y;
}

void Scene::LoadFromArrays(const void* instances, uint32_t numInstances, const void* meshData, uint32_t numMeshes,
                           const void* meshlets, uint64_t numMeshlets, const uint32_t* opaqueIds, uint32_t numOpaque,
                           const uint32_t* alphaMaskIds, uint32_t numAlphaMask)
{
    nvrhi::DeviceHandle device = g_Graphic.m_NVRHIDevice;
    auto make = [&](const char* name, uint64_t bytes, uint32_t stride, bool uav) {
        nvrhi::BufferDesc d;
        d.byteSize = bytes ? bytes : stride;
        d.structStride = stride;
        d.debugName = name;
        d.canHaveUAVs = uav;
        d.initialState = nvrhi::ResourceStates::ShaderResource;
        return device->createBuffer(d);
    };
    auto upload = [&](nvrhi::BufferHandle b, const void* src, uint64_t bytes) {
        if (bytes) nvrhi::throwIfFailed(trhip_buffer_upload(b->native(), 0, src, bytes), "Scene upload");
    };
    m_NumPrimitives = numInstances;
    // UpdateInstanceConstsRenderer::CreateInstanceConstsBuffer (BasePassRenderers.cpp:23-62)
    m_InstanceConstsBuffer = make("Instance Consts Buffer", (uint64_t)numInstances * sizeof(interop::BasePassInstanceConstants), sizeof(interop::BasePassInstanceConstants), true);
    upload(m_InstanceConstsBuffer, instances, (uint64_t)numInstances * sizeof(interop::BasePassInstanceConstants));
    // UploadGlobalMeshBuffers (SceneLoading.cpp:1016-1088)
    g_Graphic.m_GlobalMeshDataBuffer = make("GlobalMeshDataBuffer", (uint64_t)numMeshes * sizeof(interop::MeshData), sizeof(interop::MeshData), false);
    upload(g_Graphic.m_GlobalMeshDataBuffer, meshData, (uint64_t)numMeshes * sizeof(interop::MeshData));
    g_Graphic.m_GlobalMeshletDataBuffer = make("GlobalMeshletDataBuffer", numMeshlets * sizeof(interop::MeshletData), sizeof(interop::MeshletData), false);
    if (meshlets) upload(g_Graphic.m_GlobalMeshletDataBuffer, meshlets, numMeshlets * sizeof(interop::MeshletData));   // else: streamed in by the caller
    // Scene::UpdateInstanceIDsBuffers (Scene.cpp:282-362)
    m_OpaquePrimitiveIDs.assign(opaqueIds, opaqueIds + numOpaque);
    m_AlphaMaskPrimitiveIDs.assign(alphaMaskIds, alphaMaskIds + numAlphaMask);
    m_OpaqueInstanceIDsBuffer = make("OpaqueInstanceIDsBuffer", (uint64_t)numOpaque * 4, 4, false);
    upload(m_OpaqueInstanceIDsBuffer, opaqueIds, (uint64_t)numOpaque * 4);
    m_AlphaMaskInstanceIDsBuffer = make("AlphaMaskInstanceIDsBuffer", (uint64_t)numAlphaMask * 4, 4, false);
    upload(m_AlphaMaskInstanceIDsBuffer, alphaMaskIds, (uint64_t)numAlphaMask * 4);
}

void Scene::LoadCachedData(const char* path, const void* instances, uint32_t numInstances, const uint32_t* opaqueIds, uint32_t numOpaque,
                           const uint32_t* alphaMaskIds, uint32_t numAlphaMask)
{
    // SceneLoading.cpp:57-79
    struct Header { uint32_t m_Version, m_MeshOptVersion, m_NumVertices, m_NumIndices, m_NumMeshes, m_NumMeshletVertexIdxOffsets, m_NumMeshletIndices, m_NumMeshletDatas; };
    static_assert(sizeof(Header) == 32, "CachedData::Header");
    constexpr uint32_t kCurrentVersion = 3;
    constexpr size_t kRawVertexFormatBytes = 20, kMeshSpecificDataBytes = 32;
    struct File { FILE* f; ~File() { if (f) fclose(f); } } file{ fopen(path, "rb") };
    if (!file.f) throw std::runtime_error(std::string("cached data: cannot open ") + path);
    Header h{};
    if (fread(&h, sizeof h, 1, file.f) != 1) throw std::runtime_error("cached data: no header");
    if (h.m_Version != kCurrentVersion) throw std::runtime_error("cached data: version " + std::to_string(h.m_Version) + ", this reader handles 3");
    auto take = [&](std::vector<uint8_t>& dst, size_t elemBytes, uint64_t count, const char* what) {     // :728-748, same order
        dst.resize((size_t)(elemBytes * count));
        if (count && fread(dst.data(), elemBytes, (size_t)count, file.f) != count) throw std::runtime_error(std::string("cached data: truncated in ") + what);
    };
    std::vector<uint8_t> vertices, indices, meshData, vertexIds, triangles, meshlets, meshSpecific;
    take(vertices, kRawVertexFormatBytes, h.m_NumVertices, "vertices");
    take(indices, sizeof(uint32_t), h.m_NumIndices, "indices");
    take(meshData, sizeof(interop::MeshData), h.m_NumMeshes, "mesh data");
    take(vertexIds, sizeof(uint32_t), h.m_NumMeshletVertexIdxOffsets, "meshlet vertex ids");
    take(triangles, sizeof(uint32_t), h.m_NumMeshletIndices, "meshlet triangles");
    take(meshlets, sizeof(interop::MeshletData), h.m_NumMeshletDatas, "meshlets");
    take(meshSpecific, kMeshSpecificDataBytes, h.m_NumMeshes, "mesh specific data");
    // the ranges the cull and the mesh shader follow blindly
    const interop::MeshData* md = (const interop::MeshData*)meshData.data();
    for (uint32_t i = 0; i < h.m_NumMeshes; ++i) {
        if (md[i].m_NumLODs < 1 || md[i].m_NumLODs > interop::kMaxNumMeshLODs) throw std::runtime_error("cached data: mesh " + std::to_string(i) + " has a bad LOD count");
        for (uint32_t l = 0; l < md[i].m_NumLODs; ++l)
            if ((uint64_t)md[i].m_MeshLODDatas[l].m_MeshletDataBufferIdx + md[i].m_MeshLODDatas[l].m_NumMeshlets > h.m_NumMeshletDatas)
                throw std::runtime_error("cached data: mesh " + std::to_string(i) + " points past the meshlet buffer");
    }
    const interop::MeshletData* ml = (const interop::MeshletData*)meshlets.data();
    for (uint32_t i = 0; i < h.m_NumMeshletDatas; ++i) {
        const uint32_t nv = ml[i].m_VertexAndTriangleCount & 0xFFu, nt = (ml[i].m_VertexAndTriangleCount >> 8) & 0xFFu;
        if ((uint64_t)ml[i].m_MeshletVertexIDsBufferIdx + nv > h.m_NumMeshletVertexIdxOffsets || (uint64_t)ml[i].m_MeshletIndexIDsBufferIdx + nt > h.m_NumMeshletIndices)
            throw std::runtime_error("cached data: meshlet " + std::to_string(i) + " points past its index buffers");
    }
    LoadFromArrays(instances, numInstances, meshData.data(), h.m_NumMeshes, meshlets.data(), h.m_NumMeshletDatas, opaqueIds, numOpaque, alphaMaskIds, numAlphaMask);
    LoadGeometry(vertices.data(), h.m_NumVertices, (const uint32_t*)vertexIds.data(), h.m_NumMeshletVertexIdxOffsets,
                 (const uint32_t*)triangles.data(), h.m_NumMeshletIndices);
}

void Scene::LoadGeometry(const void* vertices, uint64_t numVertices, const uint32_t* meshletVertexIds, uint64_t numVertexIds,
                         const uint32_t* meshletTriangles, uint64_t numTriangles)
{
    // UploadGlobalMeshBuffers (SceneLoading.cpp:1016-1088): the vertex / meshlet-vertex-id / meshlet-triangle buffers
    nvrhi::DeviceHandle device = g_Graphic.m_NVRHIDevice;
    auto make = [&](const char* name, const void* src, uint64_t bytes, uint32_t stride) {
        nvrhi::BufferDesc d;
        d.byteSize = bytes ? bytes : stride;
        d.structStride = stride;
        d.debugName = name;
        d.initialState = nvrhi::ResourceStates::ShaderResource;
        nvrhi::BufferHandle b = device->createBuffer(d);
        if (bytes) nvrhi::throwIfFailed(trhip_buffer_upload(b->native(), 0, src, bytes), "Scene geometry upload");
        return b;
    };
    g_Graphic.m_GlobalVertexBuffer = make("GlobalVertexBuffer", vertices, numVertices * 20u, 20u);                       // RawVertexFormat (ShaderInterop.h:278-283)
    g_Graphic.m_GlobalMeshletVertexOffsetsBuffer = make("GlobalMeshletVertexOffsetsBuffer", meshletVertexIds, numVertexIds * 4u, 4u);
    g_Graphic.m_GlobalMeshletIndicesBuffer = make("GlobalMeshletIndicesBuffer", meshletTriangles, numTriangles * 4u, 4u);
}

void Scene::LoadNodes(const void* nodes, uint32_t numNodes, const uint32_t* primitiveToNode)
{
    // UpdateInstanceConstsRenderer::CreateNodeTransformsBuffer (BasePassRenderers.cpp:64-104)
    nvrhi::DeviceHandle device = g_Graphic.m_NVRHIDevice;
    m_NumNodes = numNodes;
    m_NodeLocalTransforms.assign((const uint8_t*)nodes, (const uint8_t*)nodes + (size_t)numNodes * sizeof(interop::NodeLocalTransform));
    nvrhi::BufferDesc d;
    d.byteSize = (uint64_t)numNodes * sizeof(interop::NodeLocalTransform);
    d.structStride = sizeof(interop::NodeLocalTransform);
    d.debugName = "Node Transforms Buffer";
    m_NodeLocalTransformsBuffer = device->createBuffer(d);
    nvrhi::BufferDesc p;
    p.byteSize = (uint64_t)m_NumPrimitives * 4;
    p.structStride = 4;
    p.debugName = "PrimitiveIDToNodeID Buffer";
    m_PrimitiveIDToNodeIDBuffer = device->createBuffer(p);
    nvrhi::throwIfFailed(trhip_buffer_upload(m_PrimitiveIDToNodeIDBuffer->native(), 0, primitiveToNode, p.byteSize), "Scene upload");
    m_bUpdateInstanceTransforms = true;
    m_bNodeLocalTransformsDirty = true;
}

void Scene::LoadGIProbes(const float* positions, const float* states, uint32_t numProbes, float probeRadius, bool hideInactive)
{
    nvrhi::DeviceHandle device = g_Graphic.m_NVRHIDevice;
    m_GIProbePositionsBuffer = m_GIProbeStatesBuffer = nullptr;
    m_NumGIProbes = numProbes;
    m_GIProbeRadius = probeRadius;
    m_bHideInactiveGIProbes = hideInactive;
    m_bShowGIProbes = numProbes != 0;
    if (!numProbes) return;
    nvrhi::BufferDesc d;
    d.byteSize = 12ull * numProbes; d.structStride = 12; d.debugName = "GI Probe World Positions";
    m_GIProbePositionsBuffer = device->createBuffer(d);
    nvrhi::throwIfFailed(trhip_buffer_upload(m_GIProbePositionsBuffer->native(), 0, positions, d.byteSize), "Scene::LoadGIProbes");
    d.byteSize = 4ull * numProbes; d.structStride = 4; d.debugName = "GI Probe States";
    m_GIProbeStatesBuffer = device->createBuffer(d);
    nvrhi::throwIfFailed(trhip_buffer_upload(m_GIProbeStatesBuffer->native(), 0, states, d.byteSize), "Scene::LoadGIProbes");
}

void Scene::PostSceneLoad() {}

void Scene::Update()
{
    { HOST_PROFILE_SCOPE("Scene::Update view"); m_View.Update(); }           // Scene.cpp:475

    tf::Taskflow tf;
    { HOST_PROFILE_SCOPE("RenderGraph::InitializeForFrame"); m_RenderGraph->InitializeForFrame(tf); }   // :487
    // pass schedule (:491-512): only the passes of the visibility path exist here
    {
        HOST_PROFILE_SCOPE("RenderGraph::AddRenderer x2 (Setup)");
        m_RenderGraph->AddRenderer(g_UpdateInstanceConstsRenderer);
        m_RenderGraph->AddRenderer(g_GBufferRenderer);
        m_RenderGraph->AddRenderer(g_GIDebugRenderer);                        // :509 (after the base pass: it reads this frame's HZB)
    }
    { HOST_PROFILE_SCOPE("RenderGraph::Compile"); m_RenderGraph->Compile(); }                            // :515
    { HOST_PROFILE_SCOPE("Executor::corun (Render)"); m_Executor.corun(tf); }                            // :518
}

void Scene::Shutdown()
{
    m_RenderGraph->Shutdown();
    m_RenderGraph.reset();
    m_InstanceConstsBuffer = nullptr;
    m_OpaqueInstanceIDsBuffer = m_AlphaMaskInstanceIDsBuffer = nullptr;
    m_NodeLocalTransformsBuffer = m_PrimitiveIDToNodeIDBuffer = nullptr;
    m_HZB = nullptr;
    m_SyntheticDepth = nullptr;
    m_GIProbePositionsBuffer = m_GIProbeStatesBuffer = nullptr;
    m_NumGIProbes = 0; m_bShowGIProbes = false;
}
