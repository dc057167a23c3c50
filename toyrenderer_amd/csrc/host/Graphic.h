// Graphic.h -- the compute-dispatch half of the reference's Graphic layer (source/Graphic.h:39-231)
// over the HIP back end: AddComputePass / ComputePassParams, CreateConstantBuffer, the command-list
// pool with ordered submission, IRenderer + DEFINE_RENDERER, ComputeShaderUtils, ComputeNbMips.
// Graphics/meshlet PSOs, full-screen passes, swap chain, shader hot reload and RenderDoc are out of
// scope.
#pragma once

#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <typeindex>
#include <vector>

#include "MathUtilities.h"
#include "nvrhi_lite.h"

class RenderGraph;
class Scene;
class CommonResources;

// Graphic.h:21-37 (interface) / GraphicRHI.cpp:53-318: device + queue creation.  The D3D12 adapter /
// direct-queue / validation-layer set-up becomes "one HIP device, one stream" behind the C ABI.
class GraphicRHI
{
public:
    static nvrhi::DeviceHandle CreateDevice(int deviceIndex, void* externalHipStream = nullptr);
};

class Graphic
{
public:
    static Graphic& GetInstance();

    void Initialize(int deviceIndex, Vector2U renderResolution, void* externalHipStream = nullptr);   // Graphic.cpp:608-647
    void PostSceneLoad();                                                                              // Graphic.cpp:649-662
    void Shutdown();
    void Update();                                                                                     // Graphic.cpp:702-784 (frame)

    // Graphic.h:66-72
    template <typename T>
    [[nodiscard]] nvrhi::BufferHandle CreateConstantBuffer(nvrhi::CommandListHandle commandList, const T& srcData)
    {
        nvrhi::BufferHandle buffer = m_NVRHIDevice->createBuffer(
            nvrhi::utils::CreateVolatileConstantBufferDesc(sizeof(T), std::type_index{ typeid(T) }.name(), 1));
        commandList->writeBuffer(buffer, &srcData, sizeof(T));
        return buffer;
    }

    // Graphic.h:74-80, Graphic.cpp:520-606,786-830
    [[nodiscard]] nvrhi::CommandListHandle AllocateCommandList(nvrhi::CommandQueue queueType = nvrhi::CommandQueue::Graphics);
    void FreeCommandList(nvrhi::CommandListHandle cmdList);
    void BeginCommandList(nvrhi::CommandListHandle cmdList, std::string_view name);
    void EndCommandList(nvrhi::CommandListHandle cmdList, bool bQueueCmdlist, bool bImmediateExecute);
    void ExecuteAllCommandLists();
    // waitFor (this build, RenderGraph.cpp:251 "TODO: compute queue"): lists queued EARLIER whose work this list's queue
    // must wait for when they run on another queue
    void QueueCommandList(nvrhi::CommandListHandle commandList, std::vector<const nvrhi::ICommandList*> waitFor = {})
    {
        std::lock_guard<std::mutex> lock(m_PendingCommandListsLock);
        m_PendingCommandLists.push_back(PendingCommandList{ commandList, std::move(waitFor) });
    }
    uint64_t m_NumCrossQueueWaits = 0;                                       // stats: queueWaitForCommandList calls the graph asked for

    // Graphic.h:83-111
    struct AddPassParamsCommon
    {
        nvrhi::CommandListHandle m_CommandList;
        std::string m_ShaderName;
        nvrhi::BindingSetDesc m_BindingSetDesc;
        const void* m_PushConstantsData = nullptr;
        size_t m_PushConstantsBytes = 0;
    };
    struct ComputePassParams : public AddPassParamsCommon
    {
        Vector3U m_DispatchGroupSize = Vector3U{ 0, 0, 0 };
        nvrhi::BufferHandle m_IndirectArgsBuffer;
        uint32_t m_IndirectArgsBufferOffsetBytes = 0;
    };
    void AddComputePass(const ComputePassParams& computePassParams);     // Graphic.cpp:893-947

    bool HasShader(std::string_view shaderBinName) const;                 // Graphic.cpp:270-278 (GetShader)

    nvrhi::DeviceHandle m_NVRHIDevice;
    std::shared_ptr<Scene> m_Scene;
    std::shared_ptr<CommonResources> m_CommonResources;

    // Graphic.h:137-143 (only the buffers on the path)
    nvrhi::BufferHandle m_GlobalMeshDataBuffer;
    nvrhi::BufferHandle m_GlobalMeshletDataBuffer;
    // what the mesh shader reads (basepass.hlsl t1, t5, t6): only needed when the frame rasterises its own depth
    nvrhi::BufferHandle m_GlobalVertexBuffer, m_GlobalMeshletVertexOffsetsBuffer, m_GlobalMeshletIndicesBuffer;

    Vector2U m_RenderResolution{ 0, 0 };
    uint32_t m_FrameCounter = 0;
    int m_DeviceIndex = 0;
    bool m_bEnableGPUTimers = true;                                          // per-renderer timer queries (RenderGraph.cpp:262-281); instrumentation only
    float m_LastRecordMs = 0.f, m_LastSubmitMs = 0.f;                      // host time of the last Update(): recording, submission
    // this build: capacity of the amplification-record buffer.  The reference hard-codes
    // kMaxThreadGroupsPerDimension = 65535 (BasePassRenderers.cpp:237, Q2); large scenes raise it.
    uint32_t m_MaxMeshletGroups = 65535;

private:
    std::vector<nvrhi::CommandListHandle> m_AllCommandLists;
    std::deque<nvrhi::CommandListHandle> m_FreeCommandLists[(size_t)nvrhi::CommandQueue::Count];
    std::mutex m_FreeCommandListsLock;
    std::mutex m_PendingCommandListsLock;
    struct PendingCommandList { nvrhi::CommandListHandle m_CommandList; std::vector<const nvrhi::ICommandList*> m_WaitFor; };
    std::vector<PendingCommandList> m_PendingCommandLists;
};
#define g_Graphic Graphic::GetInstance()

// Graphic.h:164-191
class IRenderer
{
public:
    IRenderer(const char* rendererName) : m_Name(rendererName) { ms_AllRenderers.push_back(this); }
    virtual ~IRenderer() = default;
    virtual void Initialize() {}
    virtual void PostSceneLoad() {}
    virtual bool HasImguiControls() const { return false; }
    virtual void UpdateImgui() {}
    // return false if the renderer is not going to be used
    virtual bool Setup(RenderGraph& renderGraph) { (void)renderGraph; return true; }
    virtual void Render(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph) = 0;

    const std::string m_Name;
    // Which queue the pass records for (this build; the reference has the graphics queue only, RenderGraph.cpp:251).
    // Compute: the pass runs on the second stream, concurrently with graphics-queue passes it shares no resource with.
    nvrhi::CommandQueue m_Queue = nvrhi::CommandQueue::Graphics;
    float m_CPUFrameTime = 0.0f;
    float m_GPUFrameTime = 0.0f;
    nvrhi::TimerQueryHandle m_FrameTimerQuery[2];

    inline static std::vector<IRenderer*> ms_AllRenderers;
};

// Graphic.h:193-195
#define DEFINE_RENDERER(name) \
    static name gs_##name;    \
    IRenderer* g_##name = &gs_##name;

// Graphic.h:197-217
struct ScopedCommandList
{
    ScopedCommandList(nvrhi::CommandListHandle cmdList, std::string_view name, bool bAutoQueue, bool bImmediateExecute)
        : m_CommandList(cmdList), m_bAutoQueue(bAutoQueue), m_bImmediateExecute(bImmediateExecute)
    {
        check(!(m_bAutoQueue && m_bImmediateExecute));
        g_Graphic.BeginCommandList(cmdList, name);
    }
    ~ScopedCommandList() { g_Graphic.EndCommandList(m_CommandList, m_bAutoQueue, m_bImmediateExecute); }
    nvrhi::CommandListHandle m_CommandList;
    const bool m_bAutoQueue;
    const bool m_bImmediateExecute;
};

// Graphic.h:219-225
namespace ComputeShaderUtils
{
constexpr Vector3U GetGroupCount(uint32_t threadCount, uint32_t groupSize) { return Vector3U{ DivideAndRoundUp(threadCount, groupSize), 1, 1 }; }
constexpr Vector3U GetGroupCount(Vector2U threadCount, uint32_t groupSize) { return Vector3U{ DivideAndRoundUp(threadCount.x, groupSize), DivideAndRoundUp(threadCount.y, groupSize), 1 }; }
} // namespace ComputeShaderUtils

// Graphic.h:227-231 (std::bit_width of the larger dimension)
constexpr uint32_t ComputeNbMips(uint32_t width, uint32_t height)
{
    uint32_t resolution = width > height ? width : height, n = 0;
    while (resolution) { ++n; resolution >>= 1; }
    return n;
}

#define TR_CONCAT_(a, b) a##b
#define TR_CONCAT(a, b) TR_CONCAT_(a, b)
// Graphic.h:233-236: GPU scope = command-list marker (microprofile is out of scope; per-shader GPU
// times come from the back end's profile, trhip_profile_*)
#define PROFILE_GPU_SCOPED(cmdList, NAME) nvrhi::utils::ScopedMarker TR_CONCAT(scopedMarker_, __LINE__){ cmdList, NAME }
#define PROFILE_FUNCTION()
#define SCOPED_COMMAND_LIST(commandList, NAME) \
    ScopedCommandList TR_CONCAT(scopedCommandList_, __LINE__){ commandList, NAME, false, false }
#define SCOPED_COMMAND_LIST_AUTO_QUEUE(commandList, NAME) \
    ScopedCommandList TR_CONCAT(scopedCommandList_, __LINE__){ commandList, NAME, true, false }
