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
    // the pass's three transient outputs (GIRenderer.cpp:618-650), described once
    struct Output
    {
        const char* debugName;
        uint32_t stride;
        bool perProbe;                 // one element per probe, else a single element
        bool indirectArgs;
        RenderGraph::ResourceHandle handle;
        nvrhi::BufferHandle last;      // what the last frame wrote (trhost_gi_probe_buffers)
    };
    enum { kPositions, kDrawArgs, kInstanceToProbe, kNumOutputs };
    Output m_Outputs[kNumOutputs] = {
        { "Probe Positions", sizeof(float) * 3, true, false, {}, nullptr },
        { "Probe Draw Indirect Args", sizeof(DrawIndexedIndirectArguments), false, true, {}, nullptr },
        { "Instance ID to Probe Index", sizeof(uint32_t), true, false, {}, nullptr },
    };

    static GIProbeVisualizationUpdateConsts ConstantsOf(const Scene& scene)     // :687-708
    {
        const View& view = scene.m_View;
        GIProbeVisualizationUpdateConsts k{};
        k.m_NumProbes = scene.m_NumGIProbes;
        // :701 m_CameraOrigin = m_View.m_Eye.  This mirror takes the camera as a world-to-view matrix (Scene.h, View::SetCamera):
        // the eye is the point that maps to the view-space origin, -t * R^T for the rigid transform [R | t] a camera is.
        // (The culling shader does not read the field; it is filled so that the uploaded block equals the reference's.)
        for (int j = 0; j < 3; ++j)
            k.m_CameraOrigin[j] = -(view.m_WorldToView.m[3][0] * view.m_WorldToView.m[j][0] + view.m_WorldToView.m[3][1] * view.m_WorldToView.m[j][1] +
                                    view.m_WorldToView.m[3][2] * view.m_WorldToView.m[j][2]);
        k.m_Frustum = CullingFrustumOf(view.m_ViewToClip);
        k.m_WorldToView = view.m_WorldToView;
        k.m_HZBDimensions = Vector2U{ scene.m_HZB->getDesc().width, scene.m_HZB->getDesc().height };
        k.m_P00 = view.m_ViewToClip.m[0][0];
        k.m_P11 = view.m_ViewToClip.m[1][1];
        k.m_NearPlane = view.m_ZNearP;
        k.m_ProbeRadius = scene.m_GIProbeRadius;
        k.m_bHideInactiveProbes = scene.m_bHideInactiveGIProbes;
        return k;
    }

public:
    GIDebugRenderer() : IRenderer("GIDebugRenderer") {}

    nvrhi::BufferHandle LastOutput(int which) const { return m_Outputs[which].last; }
    void DropLastOutputs() { for (Output& o : m_Outputs) o.last = nullptr; }

    bool Setup(RenderGraph& renderGraph) override
    {
        const uint32_t numProbes = g_Scene->m_NumGIProbes;
        if (numProbes == 0 || !g_Scene->m_bShowGIProbes) return false;          // :661-670: only with DDGI and the debug view on
        renderGraph.AddExternalReadDependency(g_Scene->m_HZB.Get());            // where the reference declares the volume descriptors (:652)
        for (Output& o : m_Outputs) {
            nvrhi::BufferDesc desc;
            desc.debugName = o.debugName;
            desc.structStride = o.stride;
            desc.byteSize = (uint64_t)o.stride * (o.perProbe ? numProbes : 1u);
            desc.canHaveUAVs = true;
            desc.isDrawIndirectArgs = o.indirectArgs;
            desc.initialState = o.indirectArgs ? nvrhi::ResourceStates::IndirectArgument : nvrhi::ResourceStates::ShaderResource;
            renderGraph.CreateTransientResource(o.handle, desc);
        }
        return true;
    }

    void Render(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph) override
    {
        nvrhi::BufferHandle out[kNumOutputs];
        for (int i = 0; i < kNumOutputs; ++i) out[i] = renderGraph.GetBuffer(m_Outputs[i].handle);      // :678-681

        // the draw the culling feeds: unit-sphere indices, instance count 0 until the shader has counted (:683-685)
        DrawIndexedIndirectArguments sphereDraw{};
        sphereDraw.m_IndexCount = g_Scene->m_GIProbeSphereIndexCount;           // g_CommonResources.UnitSphere.m_NumIndices in the reference
        commandList->writeBuffer(out[kDrawArgs], &sphereDraw, sizeof sphereDraw);

        const GIProbeVisualizationUpdateConsts consts = ConstantsOf(*g_Scene);
        using Item = nvrhi::BindingSetItem;
        Graphic::ComputePassParams pass;                                        // :710-733
        pass.m_CommandList = commandList;
        pass.m_ShaderName = "giprobevisualization_CS_VisualizeGIProbesCulling";
        pass.m_DispatchGroupSize = ComputeShaderUtils::GetGroupCount(consts.m_NumProbes, kNumThreadsPerWave);
        pass.m_BindingSetDesc.bindings = {
            Item::ConstantBuffer(0, g_Graphic.CreateConstantBuffer(commandList, consts)),
            Item::Texture_SRV(0, g_Scene->m_HZB),
            Item::Sampler(0, g_CommonResources.LinearClampMinReductionSampler),
            // t10 / t11: probe positions and states, where the reference binds the DDGI volume descriptors (t10) and
            // the probe-data texture array (u10)
            Item::StructuredBuffer_SRV(10, g_Scene->m_GIProbePositionsBuffer),
            Item::StructuredBuffer_SRV(11, g_Scene->m_GIProbeStatesBuffer),
            Item::StructuredBuffer_UAV(0, out[kPositions]),
            Item::StructuredBuffer_UAV(1, out[kDrawArgs]),
            Item::StructuredBuffer_UAV(2, out[kInstanceToProbe]),
        };
        g_Graphic.AddComputePass(pass);

        for (int i = 0; i < kNumOutputs; ++i) m_Outputs[i].last = out[i];
    }

    enum { Positions = kPositions, DrawArgs = kDrawArgs, InstanceToProbe = kInstanceToProbe };
};
DEFINE_RENDERER(GIDebugRenderer);

bool GetGIProbeCullBuffers(nvrhi::BufferHandle* positions, nvrhi::BufferHandle* drawArgs, nvrhi::BufferHandle* instanceToProbe)
{
    const GIDebugRenderer* r = static_cast<const GIDebugRenderer*>(g_GIDebugRenderer);
    if (!r->LastOutput(GIDebugRenderer::DrawArgs)) return false;
    *positions = r->LastOutput(GIDebugRenderer::Positions);
    *drawArgs = r->LastOutput(GIDebugRenderer::DrawArgs);
    *instanceToProbe = r->LastOutput(GIDebugRenderer::InstanceToProbe);
    return true;
}

void ReleaseGIProbeCullBuffers() { static_cast<GIDebugRenderer*>(g_GIDebugRenderer)->DropLastOutputs(); }
