// GIRenderer.cpp -- the one pass of the reference's GI debug renderer that belongs to the visibility path: the probe
// culling dispatch of GIDebugRenderer (source/GIRenderer.cpp:598-808; Setup :612-657, RenderDDGIDebug :672-735), the
// second consumer of FrustumCull / OcclusionCull and of compaction into indirect draw arguments.
//
// The DDGI volume itself (probe tracing / blending, RTXGI SDK) is out of scope: the probes' world positions and states
// arrive as two buffers (Scene::LoadGIProbes) where the reference binds the volume descriptors (t10) and the probe-data
// texture array (u10).  The draw of the unit spheres that consumes the indirect arguments (:737-806) is pixel work.
#include "CommonResources.h"
#include "Graphic.h"
#include "RenderGraph.h"
#include "Scene.h"
#include "VisibilityOutputs.h"
#include "../ShaderInterop.h"

using namespace interop;

class GIDebugRenderer : public IRenderer
{
    RenderGraph::ResourceHandle m_ProbePositionsRDGBufferHandle;
    RenderGraph::ResourceHandle m_ProbeDrawIndirectArgsRDGBufferHandle;
    RenderGraph::ResourceHandle m_InstanceIDToProbeIndexRDGBufferHandle;

public:
    nvrhi::BufferHandle m_LastProbePositions, m_LastProbeDrawIndirectArgs, m_LastInstanceIDToProbeIndex;

    GIDebugRenderer() : IRenderer("GIDebugRenderer") {}

    bool Setup(RenderGraph& renderGraph) override
    {
        if (!g_Scene->m_bShowGIProbes || g_Scene->m_NumGIProbes == 0) return false;   // :661-670: only with DDGI and the debug view on
        const uint32_t numProbes = g_Scene->m_NumGIProbes;
        renderGraph.AddExternalReadDependency(g_Scene->m_HZB.Get());          // :652 AddReadDependency(volume descs) in the reference
        {
            nvrhi::BufferDesc desc;                                           // :618-626
            desc.byteSize = sizeof(float) * 3ull * numProbes;
            desc.structStride = sizeof(float) * 3;
            desc.canHaveUAVs = true;
            desc.debugName = "Probe Positions";
            desc.initialState = nvrhi::ResourceStates::ShaderResource;
            renderGraph.CreateTransientResource(m_ProbePositionsRDGBufferHandle, desc);
        }
        {
            nvrhi::BufferDesc desc;                                           // :628-639
            desc.byteSize = sizeof(DrawIndexedIndirectArguments);
            desc.structStride = sizeof(DrawIndexedIndirectArguments);
            desc.canHaveUAVs = true;
            desc.isDrawIndirectArgs = true;
            desc.debugName = "Probe Draw Indirect Args";
            desc.initialState = nvrhi::ResourceStates::IndirectArgument;
            renderGraph.CreateTransientResource(m_ProbeDrawIndirectArgsRDGBufferHandle, desc);
        }
        {
            nvrhi::BufferDesc desc;                                           // :641-650
            desc.byteSize = sizeof(uint32_t) * (uint64_t)numProbes;
            desc.structStride = sizeof(uint32_t);
            desc.canHaveUAVs = true;
            desc.debugName = "Instance ID to Probe Index";
            desc.initialState = nvrhi::ResourceStates::ShaderResource;
            renderGraph.CreateTransientResource(m_InstanceIDToProbeIndexRDGBufferHandle, desc);
        }
        return true;
    }

    void Render(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph) override
    {
        nvrhi::BufferHandle probePositionsBuffer = renderGraph.GetBuffer(m_ProbePositionsRDGBufferHandle);              // :678-681
        nvrhi::BufferHandle probeDrawIndirectArgsBuffer = renderGraph.GetBuffer(m_ProbeDrawIndirectArgsRDGBufferHandle);
        nvrhi::BufferHandle instanceIDToProbeIndexBuffer = renderGraph.GetBuffer(m_InstanceIDToProbeIndexRDGBufferHandle);

        DrawIndexedIndirectArguments indirectArgs{};                          // :683-685
        indirectArgs.m_IndexCount = g_Scene->m_GIProbeSphereIndexCount;       // g_CommonResources.UnitSphere.m_NumIndices in the reference
        commandList->writeBuffer(probeDrawIndirectArgsBuffer, &indirectArgs, sizeof(indirectArgs));

        const uint32_t numProbes = g_Scene->m_NumGIProbes;                    // :687

        Matrix projectionT = Transpose(g_Scene->m_View.m_ViewToClip);         // :691-695
        Vector4 frustumX = Vector4{ projectionT.m[3][0] + projectionT.m[0][0], projectionT.m[3][1] + projectionT.m[0][1], projectionT.m[3][2] + projectionT.m[0][2], projectionT.m[3][3] + projectionT.m[0][3] };
        Vector4 frustumY = Vector4{ projectionT.m[3][0] + projectionT.m[1][0], projectionT.m[3][1] + projectionT.m[1][1], projectionT.m[3][2] + projectionT.m[1][2], projectionT.m[3][3] + projectionT.m[1][3] };
        frustumX = Normalize(frustumX);
        frustumY = Normalize(frustumY);

        GIProbeVisualizationUpdateConsts passParameters{};                    // :697-708
        passParameters.m_NumProbes = numProbes;
        passParameters.m_Frustum = Vector4{ frustumX.x, frustumX.z, frustumY.y, frustumY.z };
        passParameters.m_WorldToView = g_Scene->m_View.m_WorldToView;
        passParameters.m_HZBDimensions = Vector2U{ g_Scene->m_HZB->getDesc().width, g_Scene->m_HZB->getDesc().height };
        passParameters.m_P00 = g_Scene->m_View.m_ViewToClip.m[0][0];
        passParameters.m_P11 = g_Scene->m_View.m_ViewToClip.m[1][1];
        passParameters.m_NearPlane = g_Scene->m_View.m_ZNearP;
        passParameters.m_ProbeRadius = g_Scene->m_GIProbeRadius;
        passParameters.m_bHideInactiveProbes = g_Scene->m_bHideInactiveGIProbes ? 1u : 0u;

        nvrhi::BufferHandle passParametersBuffer = g_Graphic.CreateConstantBuffer(commandList, passParameters);   // :710

        nvrhi::BindingSetDesc bindingSetDesc;                                 // :712-723 (t10 / u10: probe inputs instead of the DDGI volume)
        bindingSetDesc.bindings = {
            nvrhi::BindingSetItem::ConstantBuffer(0, passParametersBuffer),
            nvrhi::BindingSetItem::Texture_SRV(0, g_Scene->m_HZB),
            nvrhi::BindingSetItem::StructuredBuffer_SRV(10, g_Scene->m_GIProbePositionsBuffer),
            nvrhi::BindingSetItem::StructuredBuffer_SRV(11, g_Scene->m_GIProbeStatesBuffer),
            nvrhi::BindingSetItem::StructuredBuffer_UAV(0, probePositionsBuffer),
            nvrhi::BindingSetItem::StructuredBuffer_UAV(1, probeDrawIndirectArgsBuffer),
            nvrhi::BindingSetItem::StructuredBuffer_UAV(2, instanceIDToProbeIndexBuffer),
            nvrhi::BindingSetItem::Sampler(0, g_CommonResources.LinearClampMinReductionSampler),
        };

        Graphic::ComputePassParams computePassParams;                         // :725-731
        computePassParams.m_CommandList = commandList;
        computePassParams.m_ShaderName = "giprobevisualization_CS_VisualizeGIProbesCulling";
        computePassParams.m_BindingSetDesc = bindingSetDesc;
        computePassParams.m_DispatchGroupSize = ComputeShaderUtils::GetGroupCount(numProbes, kNumThreadsPerWave);
        g_Graphic.AddComputePass(computePassParams);

        m_LastProbePositions = probePositionsBuffer;
        m_LastProbeDrawIndirectArgs = probeDrawIndirectArgsBuffer;
        m_LastInstanceIDToProbeIndex = instanceIDToProbeIndexBuffer;
    }
};
DEFINE_RENDERER(GIDebugRenderer);

bool GetGIProbeCullBuffers(nvrhi::BufferHandle* positions, nvrhi::BufferHandle* drawArgs, nvrhi::BufferHandle* instanceToProbe)
{
    GIDebugRenderer* r = static_cast<GIDebugRenderer*>(g_GIDebugRenderer);
    if (!r->m_LastProbeDrawIndirectArgs) return false;
    *positions = r->m_LastProbePositions; *drawArgs = r->m_LastProbeDrawIndirectArgs; *instanceToProbe = r->m_LastInstanceIDToProbeIndex;
    return true;
}

void ReleaseGIProbeCullBuffers()
{
    GIDebugRenderer* r = static_cast<GIDebugRenderer*>(g_GIDebugRenderer);
    r->m_LastProbePositions = nullptr; r->m_LastProbeDrawIndirectArgs = nullptr; r->m_LastInstanceIDToProbeIndex = nullptr;
}
