#include <atomic>
// BasePassRenderers.cpp -- the renderers of the GPU-driven visibility path re-authored against the
// RenderGraph / IRenderer / Graphic::AddComputePass API over the HIP back end.
//
// Reference: source/BasePassRenderers.cpp (UpdateInstanceConstsRenderer :18-165, BasePassRenderer
// :167-589, GBufferRenderer :591-722).  Same pass structure, same transient buffers, same constants,
// same dispatch order.  Differences, all forced by "compute only, no rasteriser":
//   * RenderInstances (:406-503) sets a meshlet PSO and calls dispatchMeshIndirect; here the
//     amplification shader's cull half runs as the compute shader "basepass_AS_Main LATE_CULL=N"
//     dispatched INDIRECTLY on the same MeshletDispatchArgumentsBuffer, and its per-group
//     DispatchMesh payload becomes three buffers (visibility mask per group, ordered visible list,
//     draw arguments).  Mesh / pixel shading is out of scope.
//   * the reference reuses ONE amplification / argument buffer for its four cull passes because the
//     rasteriser consumes each result immediately; the results are the product here, so every pass
//     slot (early/late x opaque/alpha-mask) owns its buffers.
//   * the depth buffer GenerateHZB reads is copied from Scene::m_SyntheticDepth (stand-in for the
//     rasteriser's output).
#include <string>

#include "CommonResources.h"
#include "FFXHelpers.h"
#include "Graphic.h"
#include "HostProfile.h"
#include "GraphicConstants.h"
#include "RenderGraph.h"
#include "Scene.h"
#include "VisibilityOutputs.h"
#include "../ShaderInterop.h"

using namespace interop;

RenderGraph::ResourceHandle g_DepthStencilBufferRDGTextureHandle;            // BasePassRenderers.cpp:15

// Multi-GPU hook (not in the reference): see trhost.h, trhost_set_shard_late_exchange.
namespace
{
struct { ShardLateFn fn = nullptr; void* user = nullptr; uint32_t presence = 0; ShardDepthFn depthFn = nullptr; void* depthUser = nullptr; } g_ShardLateExchange;
struct ShardDepthCall { void* words = nullptr; uint64_t count = 0; };
ShardDepthCall g_ShardDepthCall;
std::atomic<bool> g_ShardDepthFailed{ false };      // set on the queue's callback thread, read (and cleared) by the recording thread

void ShardDepthTrampoline(void* user, void* hipStream)
{
    const ShardDepthCall* c = (const ShardDepthCall*)user;
    if (g_ShardLateExchange.depthFn && g_ShardLateExchange.depthFn(g_ShardLateExchange.depthUser, c->words, c->count, hipStream) != 0)
        g_ShardDepthFailed.store(true, std::memory_order_release);
}

// Does SOME rank hold ids of this bucket (0 opaque, 1 alpha mask)?  Without an exchange: this rank's own list.
bool BucketPresentAnywhere(bool bAlphaMaskPrimitives)
{
    const bool local = bAlphaMaskPrimitives ? !g_Scene->m_AlphaMaskPrimitiveIDs.empty() : !g_Scene->m_OpaquePrimitiveIDs.empty();
    if (!g_ShardLateExchange.fn || g_ShardLateExchange.presence == 0) return local;
    return local || ((g_ShardLateExchange.presence >> (bAlphaMaskPrimitives ? 1 : 0)) & 1u);
}
struct ShardLateCall { nvrhi::BufferHandle info; void* lateCount = nullptr; void* infoPtr = nullptr; int bucket = 0; };
ShardLateCall g_ShardLateCalls[2];                                           // opaque, alpha mask

template <int PHASE>
void ShardLateTrampoline(void* user, void* hipStream)
{
    const ShardLateCall* c = (const ShardLateCall*)user;
    if (g_ShardLateExchange.fn) g_ShardLateExchange.fn(g_ShardLateExchange.user, hipStream, c->lateCount, c->infoPtr, c->bucket, PHASE);
}

ShardLateCall& PrepareShardLateCall(bool bAlphaMaskPrimitives, nvrhi::IBuffer* lateCullInstanceCountBuffer)
{
    ShardLateCall& call = g_ShardLateCalls[bAlphaMaskPrimitives ? 1 : 0];
    if (!call.info) {
        nvrhi::BufferDesc desc;
        desc.byteSize = 2 * sizeof(uint32_t);
        desc.structStride = sizeof(uint32_t);
        desc.canHaveUAVs = true;
        desc.debugName = bAlphaMaskPrimitives ? "ShardLateInfoAlphaMask" : "ShardLateInfoOpaque";
        call.info = g_Graphic.m_NVRHIDevice->createBuffer(desc);
    }
    call.lateCount = trhip_buffer_device_ptr(lateCullInstanceCountBuffer->native());
    call.infoPtr = trhip_buffer_device_ptr(call.info->native());
    call.bucket = bAlphaMaskPrimitives ? 1 : 0;
    return call;
}
}

void SetShardLateExchange(ShardLateFn fn, void* user, uint32_t listPresenceMask, ShardDepthFn depthFn, void* depthUser)
{
    g_ShardLateExchange.fn = fn;
    g_ShardLateExchange.user = user;
    g_ShardLateExchange.presence = fn ? listPresenceMask : 0;
    g_ShardLateExchange.depthFn = fn ? depthFn : nullptr;
    g_ShardLateExchange.depthUser = fn ? depthUser : nullptr;
}

// ---------------------------------------------------------------------------------------------------
class UpdateInstanceConstsRenderer : public IRenderer
{
public:
    UpdateInstanceConstsRenderer() : IRenderer{ "UpdateInstanceConstsRenderer" } {}

    bool Setup(RenderGraph& renderGraph) override
    {
        // :115-123 (returns false without primitives); static scenes skip the pass as well
        if (g_Scene->m_NumPrimitives == 0 || !g_Scene->m_bUpdateInstanceTransforms) return false;
        renderGraph.AddExternalWriteDependency(g_Scene->m_InstanceConstsBuffer.Get());   // async compute: see RenderGraph::Compile
        return true;
    }

    void Render(nvrhi::CommandListHandle commandList, const RenderGraph&) override
    {
        if (g_Scene->m_bNodeLocalTransformsDirty) {
            // :127-130 uploads the whole array every frame; here only when the application changed it (375 MB on the
            // 1 B-meshlet config): the update dispatch below runs every frame either way (Prev = World, World = new)
            PROFILE_GPU_SCOPED(commandList, "Upload Node Transforms");
            commandList->writeBuffer(g_Scene->m_NodeLocalTransformsBuffer, g_Scene->m_NodeLocalTransforms.data(), g_Scene->m_NodeLocalTransforms.size());
            g_Scene->m_bNodeLocalTransformsDirty = false;
        }
        // (multi-GPU: a rank updates the instances it culls -- Scene::m_InstanceUpdateFirst / Count, the whole table by default)
        const uint32_t first = std::min(g_Scene->m_InstanceUpdateFirst, g_Scene->m_NumPrimitives);
        const uint32_t numPrimitives = std::min(g_Scene->m_InstanceUpdateCount, g_Scene->m_NumPrimitives - first);
        if (numPrimitives == 0) return;
        UpdateInstanceConstsShardConstants passConstants;
        passConstants.m_NumInstances = numPrimitives;
        passConstants.m_FirstInstance = first;

        nvrhi::BindingSetDesc bindingSetDesc;                                 // :137-145 (u1 = TLAS descriptors: ray tracing, out of scope)
        bindingSetDesc.bindings = {
            nvrhi::BindingSetItem::PushConstants(0, sizeof(passConstants)),
            nvrhi::BindingSetItem::StructuredBuffer_SRV(0, g_Scene->m_NodeLocalTransformsBuffer),
            nvrhi::BindingSetItem::StructuredBuffer_SRV(1, g_Scene->m_PrimitiveIDToNodeIDBuffer),
            nvrhi::BindingSetItem::StructuredBuffer_UAV(0, g_Scene->m_InstanceConstsBuffer),
        };
        Graphic::ComputePassParams computePassParams;                         // :147-155
        computePassParams.m_CommandList = commandList;
        computePassParams.m_ShaderName = "updateinstanceconsts_CS_UpdateInstanceConstsAndBuildTLAS";
        computePassParams.m_BindingSetDesc = bindingSetDesc;
        computePassParams.m_DispatchGroupSize = ComputeShaderUtils::GetGroupCount(passConstants.m_NumInstances, kNumThreadsPerWave);
        computePassParams.m_PushConstantsData = &passConstants;
        computePassParams.m_PushConstantsBytes = sizeof(passConstants);
        g_Graphic.AddComputePass(computePassParams);
    }
};
DEFINE_RENDERER(UpdateInstanceConstsRenderer);

// ---------------------------------------------------------------------------------------------------
class BasePassRenderer : public IRenderer
{
public:
    enum PassSlot { kEarlyOpaque = 0, kLateOpaque, kEarlyAlphaMask, kLateAlphaMask, kNumPassSlots };

    // results of the last recorded frame, for whoever consumes the visibility (rasteriser, gather, tests)
    struct PassOutputs
    {
        bool m_bRan = false;
        nvrhi::BufferHandle m_MeshletAmplificationDataBuffer, m_MeshletDispatchArgumentsBuffer;
        nvrhi::BufferHandle m_MeshletVisibilityMaskBuffer, m_VisibleMeshletListBuffer, m_VisibleMeshletDrawArgsBuffer;
    };
    PassOutputs m_Outputs[kNumPassSlots];
    nvrhi::BufferHandle m_LastLateCullInstanceCountBuffer, m_LastLateCullDispatchIndirectArgsBuffer;
    nvrhi::TextureHandle m_CurrentDepthBuffer, m_LastDepthBuffer;     // depth attachment of the pass being recorded / of the last frame (read-back)

protected:
    RenderGraph::ResourceHandle m_LateCullDispatchIndirectArgsRDGBufferHandle;
    RenderGraph::ResourceHandle m_LateCullInstanceCountBufferRDGBufferHandle;
    RenderGraph::ResourceHandle m_LateCullInstanceIDsBufferRDGBufferHandle;

    FFXHelpers::SPD m_SPDHelper;

    RenderGraph::ResourceHandle m_MeshletAmplificationDataBufferRDGBufferHandle[kNumPassSlots];
    RenderGraph::ResourceHandle m_MeshletDispatchArgumentsBufferRDGBufferHandle[kNumPassSlots];
    RenderGraph::ResourceHandle m_MeshletVisibilityMaskBufferRDGBufferHandle[kNumPassSlots];
    RenderGraph::ResourceHandle m_VisibleMeshletListBufferRDGBufferHandle[kNumPassSlots];
    RenderGraph::ResourceHandle m_VisibleMeshletDrawArgsBufferRDGBufferHandle[kNumPassSlots];

    bool m_DoFrustumCulling = true;
    bool m_bDoOcclusionCulling = true;
    bool m_bDoMeshletConeCulling = true;
    uint32_t m_CullingFlags = 0;
    Vector2U m_HZBDimensions = Vector2U{ 1, 1 };
    Vector4 m_CullingFrustum = Vector4{ 0.0f, 0.0f, 0.0f, 0.0f };
    uint32_t m_NumSlotsThisFrame = 0;

public:
    struct RenderBasePassParams
    {
        nvrhi::TextureHandle m_DepthBuffer;        // the reference carries a FramebufferDesc (:195); only its depth attachment matters here
    };

    BasePassRenderer(const char* rendererName) : IRenderer(rendererName) {}

    bool Setup(RenderGraph& renderGraph) override
    {
        const uint32_t nbInstances = g_Scene->m_NumPrimitives;                // :225-229
        if (nbInstances == 0) return true;
        // resources outside the graph this pass touches (for the cross-queue waits of RenderGraph::Compile)
        renderGraph.AddExternalReadDependency(g_Scene->m_InstanceConstsBuffer.Get());
        renderGraph.AddExternalWriteDependency(g_Scene->m_HZB.Get());

        m_DoFrustumCulling = g_Scene->m_bEnableFrustumCulling;                // :231-233
        m_bDoOcclusionCulling = g_Scene->m_bEnableOcclusionCulling;
        m_bDoMeshletConeCulling = g_Scene->m_bEnableMeshletConeCulling;

        const uint32_t maxGroups = g_Graphic.m_MaxMeshletGroups;              // kMaxThreadGroupsPerDimension in the reference (:237)
        // multi-GPU: a rank without alpha-mask ids still walks the alpha-mask passes when another rank has some (it posts
        // the in-frame collectives with a zero count, see GPUCulling)
        m_NumSlotsThisFrame = BucketPresentAnywhere(true) ? (uint32_t)kNumPassSlots : 2u;
        // The pass's transient buffers (BasePassRenderers.cpp:235-293), one line each: {handle, name, element, count, kind}.
        // Arguments buffers are indirect-argument resources, everything else a shader resource; all are UAV-capable.
        enum Kind { kData, kArgs };
        auto transient = [&renderGraph](RenderGraph::ResourceHandle& handle, const char* name, uint32_t elementBytes, uint64_t elements, Kind kind) {
            nvrhi::BufferDesc desc;
            desc.debugName = name;
            desc.structStride = elementBytes;
            desc.byteSize = elementBytes * elements;
            desc.canHaveUAVs = true;
            desc.isDrawIndirectArgs = kind == kArgs;
            desc.initialState = kind == kArgs ? nvrhi::ResourceStates::IndirectArgument : nvrhi::ResourceStates::ShaderResource;
            renderGraph.CreateTransientResource(handle, desc);
        };
        for (uint32_t slot = 0; slot < m_NumSlotsThisFrame; ++slot) {
            transient(m_MeshletAmplificationDataBufferRDGBufferHandle[slot], "MeshletAmplificationDataBuffer", sizeof(MeshletAmplificationData), maxGroups, kData);   // :235-243
            // :245-254, + a 4th word: the valid records (ShaderInterop.h DispatchIndirectArgumentsEx)
            transient(m_MeshletDispatchArgumentsBufferRDGBufferHandle[slot], "MeshletDispatchArgumentsBuffer", sizeof(DispatchIndirectArgumentsEx), 1, kArgs);
            // what replaces MeshletPayload (ShaderInterop.h:200-205) and DispatchMesh(numVisible,1,1) (basepass.hlsl:120-121)
            transient(m_MeshletVisibilityMaskBufferRDGBufferHandle[slot], "MeshletVisibilityMaskBuffer", sizeof(uint32_t), maxGroups, kData);
            transient(m_VisibleMeshletListBufferRDGBufferHandle[slot], "VisibleMeshletListBuffer", sizeof(uint32_t), (uint64_t)maxGroups * kNumThreadsPerWave, kData);
            transient(m_VisibleMeshletDrawArgsBufferRDGBufferHandle[slot], "VisibleMeshletDrawArgsBuffer", sizeof(DispatchIndirectArguments), 1, kArgs);
        }

        if (m_bDoOcclusionCulling) {                                          // :256-293
            m_SPDHelper.CreateTransientResources(renderGraph);
            transient(m_LateCullDispatchIndirectArgsRDGBufferHandle, "LateCullDispatchIndirectArgs", sizeof(DispatchIndirectArguments), 1, kArgs);
            transient(m_LateCullInstanceCountBufferRDGBufferHandle, "LateCullInstanceCountBuffer", sizeof(uint32_t), 1, kData);
            transient(m_LateCullInstanceIDsBufferRDGBufferHandle, "LateCullInstanceIDsBuffer", sizeof(uint32_t), nbInstances, kData);
        }
        return true;
    }

    void GPUCulling(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph, PassSlot slot, bool bLateCull, bool bAlphaMaskPrimitives)
    {
        HOST_PROFILE_SCOPE("BasePassRenderer::GPUCulling");
        PROFILE_GPU_SCOPED(commandList, "GPU Culling");                       // :306

        const uint32_t nbInstances = (uint32_t)(bAlphaMaskPrimitives ? g_Scene->m_AlphaMaskPrimitiveIDs.size() : g_Scene->m_OpaquePrimitiveIDs.size());
        if (nbInstances == 0) {                                               // :310-314
            // multi-GPU: whether a rank joins the in-frame late-count collective must not depend on rank-local state.  A
            // rank whose list of this bucket is empty while another rank's is not posts both hook phases with a late
            // count of 0 and skips only the dispatches.
            if (g_ShardLateExchange.fn && m_bDoOcclusionCulling && BucketPresentAnywhere(bAlphaMaskPrimitives)) {
                nvrhi::BufferHandle lateCount = renderGraph.GetBuffer(m_LateCullInstanceCountBufferRDGBufferHandle);
                if (!bLateCull) {
                    commandList->clearBufferUInt(lateCount, 0);
                    commandList->hostCallback(&ShardLateTrampoline<0>, &PrepareShardLateCall(bAlphaMaskPrimitives, lateCount.Get()));
                } else {
                    commandList->hostCallback(&ShardLateTrampoline<1>, &PrepareShardLateCall(bAlphaMaskPrimitives, lateCount.Get()));
                }
            }
            return;
        }

        nvrhi::BufferHandle meshletAmplificationDataBuffer = renderGraph.GetBuffer(m_MeshletAmplificationDataBufferRDGBufferHandle[slot]);
        nvrhi::BufferHandle meshletDispatchArgumentsBuffer = renderGraph.GetBuffer(m_MeshletDispatchArgumentsBufferRDGBufferHandle[slot]);
        nvrhi::BufferHandle lateCullDispatchIndirectArgsBuffer = m_bDoOcclusionCulling ? renderGraph.GetBuffer(m_LateCullDispatchIndirectArgsRDGBufferHandle) : g_CommonResources.DummyUIntStructuredBuffer;
        nvrhi::BufferHandle lateCullInstanceCountBuffer = m_bDoOcclusionCulling ? renderGraph.GetBuffer(m_LateCullInstanceCountBufferRDGBufferHandle) : g_CommonResources.DummyUIntStructuredBuffer;
        nvrhi::BufferHandle lateCullInstanceIDsBuffer = m_bDoOcclusionCulling ? renderGraph.GetBuffer(m_LateCullInstanceIDsBufferRDGBufferHandle) : g_CommonResources.DummyUIntStructuredBuffer;

        {
            PROFILE_GPU_SCOPED(commandList, "Clear Buffers");                  // :322-332
            commandList->clearBufferUInt(meshletDispatchArgumentsBuffer, 0);
            if (!bLateCull && m_bDoOcclusionCulling) {
                commandList->clearBufferUInt(lateCullInstanceCountBuffer, 0);
                commandList->clearBufferUInt(lateCullInstanceIDsBuffer, 0);
            }
        }

        using Item = nvrhi::BindingSetItem;
        const View& view = g_Scene->m_View;
        GPUCullingPassConstants k{};                                          // :334-347
        k.m_NbInstances = nbInstances;
        k.m_CullingFlags = m_CullingFlags;
        k.m_Frustum = m_CullingFrustum;
        k.m_HZBDimensions = m_HZBDimensions;
        k.m_WorldToView = view.m_CullingWorldToView;
        k.m_PrevWorldToView = view.m_CullingPrevWorldToView;
        k.m_NearPlane = view.m_ZNearP;
        k.m_P00 = view.m_ViewToClip.m[0][0];
        k.m_P11 = view.m_ViewToClip.m[1][1];
        k.m_ForcedMeshLOD = g_Scene->m_ForceMeshLOD >= 0 ? (uint32_t)g_Scene->m_ForceMeshLOD : kInvalidMeshLOD;
        k.m_MeshLODTarget = (2.0f / k.m_P11) * (1.0f / (float)g_Graphic.m_RenderResolution.y);

        // one pass description for the (direct) early and the (indirect) late dispatch (:349-375, :392-402)
        Graphic::ComputePassParams cull;
        cull.m_CommandList = commandList;
        cull.m_ShaderName = bLateCull ? "gpuculling_CS_GPUCulling LATE_CULL=1" : "gpuculling_CS_GPUCulling LATE_CULL=0";
        cull.m_BindingSetDesc.bindings = {
            Item::ConstantBuffer(0, g_Graphic.CreateConstantBuffer(commandList, k)),
            Item::Sampler(0, g_CommonResources.LinearClampMinReductionSampler),
            Item::StructuredBuffer_SRV(0, g_Scene->m_InstanceConstsBuffer),
            Item::StructuredBuffer_SRV(1, bAlphaMaskPrimitives ? g_Scene->m_AlphaMaskInstanceIDsBuffer : g_Scene->m_OpaqueInstanceIDsBuffer),
            Item::StructuredBuffer_SRV(2, g_Graphic.m_GlobalMeshDataBuffer),
            Item::Texture_SRV(3, m_bDoOcclusionCulling ? g_Scene->m_HZB : g_CommonResources.BlackTexture),
            Item::StructuredBuffer_UAV(0, meshletAmplificationDataBuffer),
            Item::StructuredBuffer_UAV(1, meshletDispatchArgumentsBuffer),
            Item::StructuredBuffer_UAV(2, lateCullInstanceCountBuffer),
            Item::StructuredBuffer_UAV(3, lateCullInstanceIDsBuffer),
        };

        if (!bLateCull) {
            cull.m_DispatchGroupSize = ComputeShaderUtils::GetGroupCount(nbInstances, kNumThreadsPerWave);
            g_Graphic.AddComputePass(cull);

            if (m_bDoOcclusionCulling) {                                      // :377-389: the late dispatch's size from the late count
                Graphic::ComputePassParams lateArgs;
                lateArgs.m_CommandList = commandList;
                lateArgs.m_ShaderName = "gpuculling_CS_BuildLateCullIndirectArgs";
                lateArgs.m_DispatchGroupSize = Vector3U{ 1, 1, 1 };
                lateArgs.m_BindingSetDesc.bindings = { Item::StructuredBuffer_SRV(0, lateCullInstanceCountBuffer), Item::StructuredBuffer_UAV(0, lateCullDispatchIndirectArgsBuffer) };
                g_Graphic.AddComputePass(lateArgs);
                // multi-GPU: the late-list length is final from here on -- the exchange starts now and has the whole
                // early meshlet cull to complete (trhost.h)
                if (g_ShardLateExchange.fn)
                    commandList->hostCallback(&ShardLateTrampoline<0>, &PrepareShardLateCall(bAlphaMaskPrimitives, lateCullInstanceCountBuffer.Get()));
            }
        } else if (m_bDoOcclusionCulling) {
            if (g_ShardLateExchange.fn) {
                // multi-GPU: the late dispatch size rule sees the whole scene's late list (trhost.h)
                ShardLateCall& call = PrepareShardLateCall(bAlphaMaskPrimitives, lateCullInstanceCountBuffer.Get());
                commandList->hostCallback(&ShardLateTrampoline<1>, &call);
                cull.m_BindingSetDesc.bindings.push_back(Item::StructuredBuffer_SRV(4, call.info));
            }
            cull.m_IndirectArgsBuffer = lateCullDispatchIndirectArgsBuffer;
            g_Graphic.AddComputePass(cull);
        }
        m_LastLateCullInstanceCountBuffer = lateCullInstanceCountBuffer;
        m_LastLateCullDispatchIndirectArgsBuffer = lateCullDispatchIndirectArgsBuffer;
    }

    void RenderInstances(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph, PassSlot slot, bool bIsLateCull, bool bAlphaMaskPrimitives)
    {
        HOST_PROFILE_SCOPE("BasePassRenderer::RenderInstances");
        PROFILE_GPU_SCOPED(commandList, "Render Instances");                  // :414

        const uint32_t nbInstances = (uint32_t)(bAlphaMaskPrimitives ? g_Scene->m_AlphaMaskPrimitiveIDs.size() : g_Scene->m_OpaquePrimitiveIDs.size());
        if (nbInstances == 0) return;                                         // :420-423
        if (bIsLateCull && !m_bDoOcclusionCulling) return;

        nvrhi::BufferHandle meshletAmplificationDataBuffer = renderGraph.GetBuffer(m_MeshletAmplificationDataBufferRDGBufferHandle[slot]);
        nvrhi::BufferHandle meshletDispatchArgumentsBuffer = renderGraph.GetBuffer(m_MeshletDispatchArgumentsBufferRDGBufferHandle[slot]);
        nvrhi::BufferHandle visMaskBuffer = renderGraph.GetBuffer(m_MeshletVisibilityMaskBufferRDGBufferHandle[slot]);
        nvrhi::BufferHandle visibleListBuffer = renderGraph.GetBuffer(m_VisibleMeshletListBufferRDGBufferHandle[slot]);
        nvrhi::BufferHandle visibleDrawArgsBuffer = renderGraph.GetBuffer(m_VisibleMeshletDrawArgsBufferRDGBufferHandle[slot]);

        using Item = nvrhi::BindingSetItem;
        const View& view = g_Scene->m_View;
        BasePassConstants basePassConstants{};                                // :436-458 (the culling-relevant members)
        basePassConstants.m_WorldToView = view.m_CullingWorldToView;
        basePassConstants.m_Frustum = m_CullingFrustum;
        // alpha-mask primitives count as double-sided: no cone culling for them (:436-442, Q8)
        basePassConstants.m_CullingFlags = bAlphaMaskPrimitives ? (m_CullingFlags & ~kCullingFlagMeshletConeCullingEnable) : m_CullingFlags;
        basePassConstants.m_HZBDimensions = m_HZBDimensions;
        basePassConstants.m_P00 = view.m_ViewToClip.m[0][0];
        basePassConstants.m_P11 = view.m_ViewToClip.m[1][1];
        basePassConstants.m_NearPlane = view.m_ZNearP;
        basePassConstants.m_OutputResolution = g_Graphic.m_RenderResolution;

        // :460-502: PSODesc.AS = "basepass_AS_Main LATE_CULL=%d", dispatchMeshIndirect(0) on meshletDispatchArgumentsBuffer.
        // Of the reference's binding set the amplification stage reads b0, t0, t2, t4, t7, t8, s4 (t1, t3, t5, t6 feed the
        // mesh / pixel stages); u0..u2 are this build's outputs.
        Graphic::ComputePassParams amplification;
        amplification.m_CommandList = commandList;
        amplification.m_ShaderName = bIsLateCull ? "basepass_AS_Main LATE_CULL=1" : "basepass_AS_Main LATE_CULL=0";
        amplification.m_IndirectArgsBuffer = meshletDispatchArgumentsBuffer;
        amplification.m_BindingSetDesc.bindings = {
            Item::ConstantBuffer(0, g_Graphic.CreateConstantBuffer(commandList, basePassConstants)),
            Item::Sampler(4, g_CommonResources.LinearClampMinReductionSampler),
            Item::StructuredBuffer_SRV(0, g_Scene->m_InstanceConstsBuffer),
            Item::StructuredBuffer_SRV(2, g_Graphic.m_GlobalMeshDataBuffer),
            Item::StructuredBuffer_SRV(4, g_Graphic.m_GlobalMeshletDataBuffer),
            Item::StructuredBuffer_SRV(7, meshletAmplificationDataBuffer),
            Item::Texture_SRV(8, m_bDoOcclusionCulling ? g_Scene->m_HZB : g_CommonResources.BlackTexture),
            Item::StructuredBuffer_UAV(0, visMaskBuffer),
            Item::StructuredBuffer_UAV(1, visibleListBuffer),
            Item::StructuredBuffer_UAV(2, visibleDrawArgsBuffer),
        };
        g_Graphic.AddComputePass(amplification);

        if (g_Scene->m_bRasterDepth) {
            // The mesh + depth stages of the same DispatchMeshIndirect (basepass.hlsl:124-188, PSO :481-495), depth only:
            // one wave per visible meshlet, dispatched on the draw arguments the cull just wrote.
            check(g_Graphic.m_GlobalVertexBuffer && m_CurrentDepthBuffer);
            basePassConstants.m_WorldToClip = MultiplyNoFMA(g_Scene->m_View.m_CullingWorldToView, g_Scene->m_View.m_ViewToClip);   // :447
            nvrhi::BufferHandle rasterConstants = g_Graphic.CreateConstantBuffer(commandList, basePassConstants);
            nvrhi::BindingSetDesc rasterBindings;
            rasterBindings.bindings = {
                nvrhi::BindingSetItem::ConstantBuffer(0, rasterConstants),
                nvrhi::BindingSetItem::StructuredBuffer_SRV(0, g_Scene->m_InstanceConstsBuffer),
                nvrhi::BindingSetItem::StructuredBuffer_SRV(1, g_Graphic.m_GlobalVertexBuffer),
                nvrhi::BindingSetItem::StructuredBuffer_SRV(2, g_Graphic.m_GlobalMeshDataBuffer),
                nvrhi::BindingSetItem::StructuredBuffer_SRV(4, g_Graphic.m_GlobalMeshletDataBuffer),
                nvrhi::BindingSetItem::StructuredBuffer_SRV(5, g_Graphic.m_GlobalMeshletVertexOffsetsBuffer),
                nvrhi::BindingSetItem::StructuredBuffer_SRV(6, g_Graphic.m_GlobalMeshletIndicesBuffer),
                nvrhi::BindingSetItem::StructuredBuffer_SRV(7, meshletAmplificationDataBuffer),
                nvrhi::BindingSetItem::StructuredBuffer_SRV(9, visibleListBuffer),
                nvrhi::BindingSetItem::Texture_UAV(0, m_CurrentDepthBuffer),
            };
            Graphic::ComputePassParams rasterPass;
            rasterPass.m_CommandList = commandList;
            rasterPass.m_ShaderName = "basepass_MS_Main_depth";
            rasterPass.m_BindingSetDesc = rasterBindings;
            rasterPass.m_IndirectArgsBuffer = visibleDrawArgsBuffer;
            g_Graphic.AddComputePass(rasterPass);
        }

        PassOutputs& out = m_Outputs[slot];
        out.m_bRan = true;
        out.m_MeshletAmplificationDataBuffer = meshletAmplificationDataBuffer;
        out.m_MeshletDispatchArgumentsBuffer = meshletDispatchArgumentsBuffer;
        out.m_MeshletVisibilityMaskBuffer = visMaskBuffer;
        out.m_VisibleMeshletListBuffer = visibleListBuffer;
        out.m_VisibleMeshletDrawArgsBuffer = visibleDrawArgsBuffer;
    }

    void GenerateHZB(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph, const RenderBasePassParams& params)
    {
        if (g_Scene->m_bFreezeCullingCamera) return;                          // :507-510

        PROFILE_GPU_SCOPED(commandList, "Generate HZB");                      // :513

        MinMaxDownsampleConsts passParameters;                                // :515-517
        passParameters.m_OutputDimensions = m_HZBDimensions;
        passParameters.m_bDownsampleMax = !GraphicConstants::kInversedDepthBuffer;

        nvrhi::TextureHandle depthStencilBuffer = params.m_DepthBuffer;       // :519

        if (g_Scene->m_bRasterDepth && g_ShardLateExchange.fn) {
            // multi-GPU with self-rasterised depth: each rank drew only its shard's visible meshlets; the HZB every rank
            // builds must come from the whole scene's depth = element-wise MAX (reverse-Z) over the ranks.
            if (g_ShardDepthFailed.exchange(false, std::memory_order_acq_rel)) throw nvrhi::Error("depth all-reduce (MAX) across ranks failed");
            if (!g_ShardLateExchange.depthFn)
                throw nvrhi::Error("raster depth + shard exchange needs trhost_exchange_desc.depth_allreduce_max: per-rank depth buffers would give per-rank HZBs");
            g_ShardDepthCall.words = trhip_texture_device_ptr(depthStencilBuffer->native());
            g_ShardDepthCall.count = (uint64_t)depthStencilBuffer->getDesc().width * depthStencilBuffer->getDesc().height;
            commandList->hostCallback(&ShardDepthTrampoline, &g_ShardDepthCall);
        }

        nvrhi::BindingSetDesc bindingSetDesc;                                 // :521-527
        bindingSetDesc.bindings = {
            nvrhi::BindingSetItem::PushConstants(0, sizeof(passParameters)),
            nvrhi::BindingSetItem::Texture_SRV(0, depthStencilBuffer),
            nvrhi::BindingSetItem::Texture_UAV(0, g_Scene->m_HZB),
            nvrhi::BindingSetItem::Sampler(0, g_CommonResources.PointClampSampler)
        };

        Graphic::ComputePassParams computePassParams;                         // :529-537
        computePassParams.m_CommandList = commandList;
        computePassParams.m_ShaderName = "minmaxdownsample_CS_Main";
        computePassParams.m_BindingSetDesc = bindingSetDesc;
        computePassParams.m_DispatchGroupSize = ComputeShaderUtils::GetGroupCount(m_HZBDimensions, 8);
        computePassParams.m_PushConstantsData = &passParameters;
        computePassParams.m_PushConstantsBytes = sizeof(passParameters);
        g_Graphic.AddComputePass(computePassParams);

        // generate HZB mip chain (:539-541)
        const nvrhi::SamplerReductionType reductionType = GraphicConstants::kInversedDepthBuffer ? nvrhi::SamplerReductionType::Minimum : nvrhi::SamplerReductionType::Maximum;
        m_SPDHelper.Execute(commandList, renderGraph, depthStencilBuffer, g_Scene->m_HZB, reductionType);
    }

    void RenderBasePass(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph, const RenderBasePassParams& params)
    {
        for (PassOutputs& o : m_Outputs) o = PassOutputs{};

        m_CullingFlags = m_DoFrustumCulling ? kCullingFlagFrustumCullingEnable : 0;                 // :551-553
        m_CullingFlags |= m_bDoOcclusionCulling ? kCullingFlagOcclusionCullingEnable : 0;
        m_CullingFlags |= m_bDoMeshletConeCulling ? kCullingFlagMeshletConeCullingEnable : 0;

        m_HZBDimensions = m_bDoOcclusionCulling ? Vector2U{ g_Scene->m_HZB->getDesc().width, g_Scene->m_HZB->getDesc().height } : Vector2U{ 1, 1 };   // :555

        m_CullingFrustum = CullingFrustumOf(g_Scene->m_View.m_ViewToClip);                           // :557-563

        // Where GenerateHZB finds the frame's depth: the depth attachment the rasteriser wrote -- or, when nothing rasterises
        // (the stand-in of the culling benchmarks and tests), the uploaded depth image itself.  (Round 1 copied the image
        // into the attachment every frame: 33 MB and a launch for nothing.)
        RenderBasePassParams hzbParams = params;
        if (g_Scene->m_SyntheticDepth && !g_Scene->m_bRasterDepth) hzbParams.m_DepthBuffer = g_Scene->m_SyntheticDepth;
        m_CurrentDepthBuffer = params.m_DepthBuffer;
        m_LastDepthBuffer = hzbParams.m_DepthBuffer;
        if (g_Scene->m_bRasterDepth)                                                                 // the base pass starts from a cleared depth buffer
            commandList->clearTextureFloat(params.m_DepthBuffer, nvrhi::AllSubresources, nvrhi::Color{ GraphicConstants::kFarDepth });

        GPUCulling(commandList, renderGraph, kEarlyOpaque, false /* bLateCull */, false /* bAlphaMaskPrimitives */);          // :565-566
        RenderInstances(commandList, renderGraph, kEarlyOpaque, false, false);

        if (m_bDoOcclusionCulling) {                                                                 // :568-581
            GenerateHZB(commandList, renderGraph, hzbParams);

            GPUCulling(commandList, renderGraph, kLateOpaque, true, false);
            RenderInstances(commandList, renderGraph, kLateOpaque, true, false);

            if (m_NumSlotsThisFrame == kNumPassSlots) {
                GPUCulling(commandList, renderGraph, kEarlyAlphaMask, false, true);
                RenderInstances(commandList, renderGraph, kEarlyAlphaMask, false, true);
                GPUCulling(commandList, renderGraph, kLateAlphaMask, true, true);
                RenderInstances(commandList, renderGraph, kLateAlphaMask, true, true);
            }
            GenerateHZB(commandList, renderGraph, hzbParams);
        } else if (m_NumSlotsThisFrame == kNumPassSlots) {
            // cull & render for alpha mask primitives, but no occlusion culling (:583-587)
            GPUCulling(commandList, renderGraph, kEarlyAlphaMask, false, true);
            RenderInstances(commandList, renderGraph, kEarlyAlphaMask, false, true);
        }
    }
};

// ---------------------------------------------------------------------------------------------------
class GBufferRenderer : public BasePassRenderer
{
public:
    GBufferRenderer() : BasePassRenderer("GBufferRenderer") {}

    void Initialize() override
    {
        nvrhi::TextureDesc desc;                                              // :600-610
        desc.width = GetNextPow2(g_Graphic.m_RenderResolution.x) >> 1;
        desc.height = GetNextPow2(g_Graphic.m_RenderResolution.y) >> 1;
        desc.format = GraphicConstants::kHZBFormat;
        desc.isUAV = true;
        desc.debugName = "HZB";
        desc.mipLevels = ComputeNbMips(desc.width, desc.height);
        desc.useClearValue = false;
        desc.initialState = nvrhi::ResourceStates::ShaderResource;
        g_Scene->m_HZB = g_Graphic.m_NVRHIDevice->createTexture(desc);

        nvrhi::TextureDesc depth;                                             // stand-in for the rasteriser's depth output
        depth.width = g_Graphic.m_RenderResolution.x;
        depth.height = g_Graphic.m_RenderResolution.y;
        depth.format = GraphicConstants::kDepthStencilFormat;
        depth.debugName = "Synthetic Depth Source";
        g_Scene->m_SyntheticDepth = g_Graphic.m_NVRHIDevice->createTexture(depth);

        nvrhi::CommandListHandle commandList = g_Graphic.AllocateCommandList();   // :612-615
        SCOPED_COMMAND_LIST_AUTO_QUEUE(commandList, "GBufferRenderer::Initialize");
        commandList->clearTextureFloat(g_Scene->m_HZB, nvrhi::AllSubresources, nvrhi::Color{ GraphicConstants::kFarDepth });
        commandList->clearTextureFloat(g_Scene->m_SyntheticDepth, nvrhi::AllSubresources, nvrhi::Color{ GraphicConstants::kFarDepth });
    }

    bool Setup(RenderGraph& renderGraph) override
    {
        BasePassRenderer::Setup(renderGraph);                                 // :620
        {
            nvrhi::TextureDesc desc;                                          // :646-655 (G-buffer colour targets :622-644 are pixel work, out of scope)
            desc.width = g_Graphic.m_RenderResolution.x;
            desc.height = g_Graphic.m_RenderResolution.y;
            desc.format = GraphicConstants::kDepthStencilFormat;
            desc.debugName = "Depth Buffer";
            desc.isRenderTarget = true;
            desc.isUAV = true;                     // this build: the compute rasteriser writes depth through a UAV
            desc.setClearValue(nvrhi::Color{ GraphicConstants::kFarDepth });
            desc.initialState = nvrhi::ResourceStates::DepthRead;
            renderGraph.CreateTransientResource(g_DepthStencilBufferRDGTextureHandle, desc);
        }
        return true;
    }

    void Render(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph) override
    {
        if (g_Scene->m_NumPrimitives == 0) return;                            // :668-671
        nvrhi::TextureHandle depthStencilBuffer = renderGraph.GetTexture(g_DepthStencilBufferRDGTextureHandle);
        RenderBasePassParams params;                                          // :691-695
        params.m_DepthBuffer = depthStencilBuffer;
        RenderBasePass(commandList, renderGraph, params);
    }
};
DEFINE_RENDERER(GBufferRenderer);

// accessor for consumers of the visibility results (the reference has none: nothing reads them back)
bool GetVisibilityPassBuffers(uint32_t slot, VisibilityPassBuffers* out)
{
    GBufferRenderer* r = static_cast<GBufferRenderer*>(g_GBufferRenderer);
    if (slot >= BasePassRenderer::kNumPassSlots || !out) return false;
    const BasePassRenderer::PassOutputs& o = r->m_Outputs[slot];
    out->m_bRan = o.m_bRan;
    out->m_MeshletAmplificationDataBuffer = o.m_MeshletAmplificationDataBuffer;
    out->m_MeshletDispatchArgumentsBuffer = o.m_MeshletDispatchArgumentsBuffer;
    out->m_MeshletVisibilityMaskBuffer = o.m_MeshletVisibilityMaskBuffer;
    out->m_VisibleMeshletListBuffer = o.m_VisibleMeshletListBuffer;
    out->m_VisibleMeshletDrawArgsBuffer = o.m_VisibleMeshletDrawArgsBuffer;
    out->m_LateCullInstanceCountBuffer = r->m_LastLateCullInstanceCountBuffer;
    out->m_LateCullDispatchIndirectArgsBuffer = r->m_LastLateCullDispatchIndirectArgsBuffer;
    return true;
}

nvrhi::TextureHandle GetLastDepthBuffer() { return static_cast<GBufferRenderer*>(g_GBufferRenderer)->m_LastDepthBuffer; }

void ReleaseVisibilityPassBuffers()
{
    GBufferRenderer* r = static_cast<GBufferRenderer*>(g_GBufferRenderer);
    for (auto& o : r->m_Outputs) o = BasePassRenderer::PassOutputs{};
    r->m_LastLateCullInstanceCountBuffer = nullptr;
    r->m_LastLateCullDispatchIndirectArgsBuffer = nullptr;
    r->m_CurrentDepthBuffer = nullptr; r->m_LastDepthBuffer = nullptr;
    for (ShardLateCall& c : g_ShardLateCalls) c = ShardLateCall{};
    SetShardLateExchange(nullptr, nullptr);
}
