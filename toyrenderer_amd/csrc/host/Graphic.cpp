// Graphic.cpp -- see Graphic.h.  Re-authored against the HIP back end; cites are to the reference's
// source/Graphic.cpp and source/GraphicRHI.cpp.
#include <unordered_map>

#include "Graphic.h"

#include "HostProfile.h"

#include <chrono>

#include "CommonResources.h"
#include "GraphicConstants.h"
#include "RenderGraph.h"
#include "Scene.h"
#include "../ShaderInterop.h"

// Graphic.cpp:14-20: layouts shared with the kernels must agree
static_assert(sizeof(interop::DispatchIndirectArguments) == 12);
static_assert(GraphicConstants::kMaxThreadGroupsPerDimension == interop::kMaxThreadGroupsPerDimension);

nvrhi::DeviceHandle GraphicRHI::CreateDevice(int deviceIndex, void* externalHipStream)
{
    // GraphicRHI.cpp:56-200: adapter pick + ONE direct queue (:152-177)  ->  one HIP device + one stream.
    trhip_device dev = nullptr;
    if (externalHipStream)
        nvrhi::throwIfFailed(trhip_device_create_on_stream(deviceIndex, externalHipStream, &dev), "GraphicRHI::CreateDevice");
    else
        nvrhi::throwIfFailed(trhip_device_create(deviceIndex, &dev), "GraphicRHI::CreateDevice");
    // Graphic.cpp:84-94 asserts waveLaneCount == kNumThreadsPerWave (32).  gfx950 is wave64-only: one
    // wave runs two reference groups, so the requirement here is 2 * kNumThreadsPerWave.
    uint32_t cus = 0, wave = 0;
    uint64_t mem = 0;
    nvrhi::throwIfFailed(trhip_device_info(dev, &cus, &wave, &mem), "trhip_device_info");
    check(wave == 2 * interop::kNumThreadsPerWave);
    nvrhi::DeviceHandle device(new nvrhi::IDevice(dev));
    device->setDeviceIndex(deviceIndex);
    return device;
}

Graphic& Graphic::GetInstance()
{
    static Graphic s_Instance;
    return s_Instance;
}

void Graphic::Initialize(int deviceIndex, Vector2U renderResolution, void* externalHipStream)
{
    m_RenderResolution = renderResolution;                                   // Graphic.cpp:612
    m_DeviceIndex = deviceIndex;
    m_NVRHIDevice = GraphicRHI::CreateDevice(deviceIndex, externalHipStream); // InitDevice
    // InitShaders (Graphic.cpp:103-251) loads DXIL blobs into a name->shader map; here the kernels are
    // linked into the back end and registered under the same names.
    check(HasShader("gpuculling_CS_GPUCulling LATE_CULL=0"));
    m_CommonResources = std::make_shared<CommonResources>();
    m_CommonResources->Initialize();
    m_Scene = std::make_shared<Scene>();
    m_Scene->Initialize();
    for (IRenderer* renderer : IRenderer::ms_AllRenderers)                    // Graphic.cpp:630-638
        renderer->Initialize();
    ExecuteAllCommandLists();
}

void Graphic::PostSceneLoad()
{
    m_Scene->PostSceneLoad();                                                 // Graphic.cpp:649-662
    for (IRenderer* renderer : IRenderer::ms_AllRenderers)
        renderer->PostSceneLoad();
    ExecuteAllCommandLists();
}

void Graphic::Shutdown()
{
    if (!m_NVRHIDevice) return;
    m_NVRHIDevice->waitForIdle();
    for (IRenderer* renderer : IRenderer::ms_AllRenderers) {
        renderer->m_FrameTimerQuery[0] = nullptr;
        renderer->m_FrameTimerQuery[1] = nullptr;
        renderer->m_Queue = nvrhi::CommandQueue::Graphics;
    }
    m_Scene->Shutdown();
    m_Scene.reset();
    m_CommonResources.reset();
    m_GlobalMeshDataBuffer = nullptr;
    m_GlobalMeshletDataBuffer = nullptr;
    m_GlobalVertexBuffer = nullptr; m_GlobalMeshletVertexOffsetsBuffer = nullptr; m_GlobalMeshletIndicesBuffer = nullptr;
    m_PendingCommandLists.clear();
    for (auto& pool : m_FreeCommandLists) pool.clear();
    m_AllCommandLists.clear();
    m_NVRHIDevice->destroyQueues();
    m_NVRHIDevice = nullptr;
    m_NumCrossQueueWaits = 0;
    m_FrameCounter = 0;
}

void Graphic::Update()
{
    ++m_FrameCounter;                                                         // Graphic.cpp:706
    const auto t0 = std::chrono::steady_clock::now();
    m_Scene->Update();                                                        // records every pass
    const auto t1 = std::chrono::steady_clock::now();
    ExecuteAllCommandLists();                                                 // the CPU->GPU boundary
    m_NVRHIDevice->runGarbageCollection();                                    // Graphic.cpp:761-765
    const auto t2 = std::chrono::steady_clock::now();
    m_LastRecordMs = std::chrono::duration<float, std::milli>(t1 - t0).count();
    m_LastSubmitMs = std::chrono::duration<float, std::milli>(t2 - t1).count();
}

bool Graphic::HasShader(std::string_view shaderBinName) const
{
    return trhip_shader_exists(std::string(shaderBinName).c_str()) != 0;
}

nvrhi::CommandListHandle Graphic::AllocateCommandList(nvrhi::CommandQueue queueType)
{
    std::lock_guard<std::mutex> lock(m_FreeCommandListsLock);                 // Graphic.cpp:520-555 (one pool per queue type)
    std::deque<nvrhi::CommandListHandle>& pool = m_FreeCommandLists[(size_t)queueType];
    if (!pool.empty()) {
        nvrhi::CommandListHandle cl = pool.front();
        pool.pop_front();
        return cl;
    }
    nvrhi::CommandListHandle cl = m_NVRHIDevice->createCommandList(queueType);
    m_AllCommandLists.push_back(cl);
    return cl;
}

void Graphic::FreeCommandList(nvrhi::CommandListHandle cmdList)
{
    std::lock_guard<std::mutex> lock(m_FreeCommandListsLock);
    m_FreeCommandLists[(size_t)cmdList->m_Queue].push_back(cmdList);
}

void Graphic::BeginCommandList(nvrhi::CommandListHandle cmdList, std::string_view name)
{
    cmdList->open();                                                          // Graphic.cpp:564-583
    cmdList->beginMarker(std::string(name).c_str());
}

void Graphic::EndCommandList(nvrhi::CommandListHandle cmdList, bool bQueueCmdlist, bool bImmediateExecute)
{
    cmdList->endMarker();                                                     // Graphic.cpp:585-606
    cmdList->close();
    if (bQueueCmdlist) QueueCommandList(cmdList);
    if (bImmediateExecute) {
        m_NVRHIDevice->executeCommandList(cmdList);
        FreeCommandList(cmdList);
    }
}

void Graphic::ExecuteAllCommandLists()
{
    std::vector<PendingCommandList> lists;
    {
        std::lock_guard<std::mutex> lock(m_PendingCommandListsLock);
        lists.swap(m_PendingCommandLists);
    }
    if (lists.empty()) return;
    // Graphic.cpp:790 does waitForIdle before every submit (upload-manager versioning).  The streams are
    // in-order and recorded lists own their staging copies, so the wait is not needed for
    // correctness here and would only serialise CPU recording with GPU execution.
    //
    // Lists are submitted in queue order (= pass order); runs of lists of one queue type go down in one call
    // (Graphic.cpp:816).  A list that depends on a list of the OTHER queue (RenderGraph::Compile) first makes its queue
    // wait for that one (nvrhi queueWaitForCommandList).  Frames are joined: whatever the compute queue still runs from
    // the previous submission is awaited by this one's first graphics list and the other way round -- the graph does not
    // see hazards across frames.
    using Q = nvrhi::CommandQueue;
    bool anyCompute = false;
    for (const PendingCommandList& p : lists) anyCompute |= p.m_CommandList->m_Queue == Q::Compute;
    if (anyCompute || m_NVRHIDevice->hasComputeQueue()) {
        if (m_NVRHIDevice->hasComputeQueue() && m_NVRHIDevice->lastInstance(Q::Compute))
            m_NVRHIDevice->queueWaitForCommandList(Q::Graphics, Q::Compute, m_NVRHIDevice->lastInstance(Q::Compute));
        if (anyCompute && m_NVRHIDevice->lastInstance(Q::Graphics))
            m_NVRHIDevice->queueWaitForCommandList(Q::Compute, Q::Graphics, m_NVRHIDevice->lastInstance(Q::Graphics));
    }
    std::unordered_map<const nvrhi::ICommandList*, std::pair<Q, uint64_t>> submitted;     // list -> (queue, instance)
    size_t i = 0;
    while (i < lists.size()) {
        const Q queue = lists[i].m_CommandList->m_Queue;
        for (const nvrhi::ICommandList* dep : lists[i].m_WaitFor) {
            auto it = submitted.find(dep);
            check(it != submitted.end());                                     // dependencies point backwards in queue order
            if (it->second.first != queue) {
                // the producer's queue is in order, so waiting for its LATEST submission covers the one meant
                m_NVRHIDevice->queueWaitForCommandList(queue, it->second.first, m_NVRHIDevice->lastInstance(it->second.first));
                ++m_NumCrossQueueWaits;
            }
        }
        std::vector<nvrhi::ICommandList*> raw{ lists[i].m_CommandList.Get() };
        size_t j = i + 1;
        while (j < lists.size() && lists[j].m_CommandList->m_Queue == queue && lists[j].m_WaitFor.empty()) raw.push_back(lists[j++].m_CommandList.Get());
        const uint64_t instance = m_NVRHIDevice->executeCommandLists(raw.data(), raw.size(), queue);   // Graphic.cpp:816
        for (nvrhi::ICommandList* cl : raw) submitted[cl] = { queue, instance };
        i = j;
    }
    for (auto& p : lists) FreeCommandList(p.m_CommandList);
}

void Graphic::AddComputePass(const ComputePassParams& p)
{
    HOST_PROFILE_SCOPE("Graphic::AddComputePass");
    check(p.m_CommandList);                                                   // Graphic.cpp:895-896
    check(!p.m_ShaderName.empty());
    PROFILE_GPU_SCOPED(p.m_CommandList, p.m_ShaderName.c_str());              // :899

    nvrhi::ComputeState computeState;                                         // :901-921
    check(HasShader(p.m_ShaderName));                                         // GetShader() asserts on unknown names (:276)
    computeState.pipeline = p.m_ShaderName;
    computeState.bindings = p.m_BindingSetDesc;

    if (p.m_IndirectArgsBuffer) {
        // indirect dispatch does not need group size (:923-928)
        check(p.m_DispatchGroupSize.x == 0 && p.m_DispatchGroupSize.y == 0 && p.m_DispatchGroupSize.z == 0);
        computeState.indirectParams = p.m_IndirectArgsBuffer;
    }
    p.m_CommandList->setComputeState(computeState);                           // :930

    if (p.m_PushConstantsData) {                                              // :932-936
        check(p.m_PushConstantsBytes > 0);
        p.m_CommandList->setPushConstants(p.m_PushConstantsData, p.m_PushConstantsBytes);
    }
    if (p.m_IndirectArgsBuffer) {                                             // :938-946
        p.m_CommandList->dispatchIndirect(p.m_IndirectArgsBufferOffsetBytes);
    } else {
        check(p.m_DispatchGroupSize.x != 0 && p.m_DispatchGroupSize.y != 0 && p.m_DispatchGroupSize.z != 0);
        p.m_CommandList->dispatch(p.m_DispatchGroupSize.x, p.m_DispatchGroupSize.y, p.m_DispatchGroupSize.z);
    }
}
