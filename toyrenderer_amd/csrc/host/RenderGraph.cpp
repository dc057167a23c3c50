// RenderGraph.cpp -- see RenderGraph.h.  Behaviour follows the reference's source/RenderGraph.cpp
// (cited per function); the code is new and sits on the HIP back end.
#include "RenderGraph.h"

#include <algorithm>
#include <chrono>
#include <functional>

#include "Graphic.h"

// RenderGraph.cpp:10: the pass a worker thread is currently executing (GetBuffer/GetTexture checks)
thread_local RenderGraph::PassID tl_CurrentThreadPassID = RenderGraph::kInvalidPassID;

static const uint64_t kDefaultHeapBlockSize = 16ull << 20;   // RenderGraph.cpp:13
static const uint32_t kHeapAlignment = 64u << 10;             // :15 (D3D12 placed-resource alignment, kept)
static const uint32_t kMaxTransientResourceAge = 2;           // :16
uint64_t RenderGraph::ms_MaxHeapBlockSize = 1ull << 30;       // :14 (1 GB in the reference; a knob here)

namespace
{
inline void HashCombine(std::size_t& seed, std::size_t v) { seed ^= v + 0x9e3779b9 + (seed << 6) + (seed >> 2); }
template <typename T> inline void HashValue(std::size_t& seed, const T& v) { HashCombine(seed, std::hash<T>{}(v)); }

// RenderGraph.cpp:18-37: which TextureDesc fields force a re-allocation
std::size_t HashResourceDesc(const nvrhi::TextureDesc& d)
{
    std::size_t s = 0;
    HashValue(s, d.width); HashValue(s, d.height); HashValue(s, d.depth); HashValue(s, d.arraySize);
    HashValue(s, d.mipLevels); HashValue(s, d.sampleCount); HashValue(s, d.sampleQuality);
    HashValue(s, (uint32_t)d.format); HashValue(s, (uint32_t)d.dimension);
    HashValue(s, d.isRenderTarget); HashValue(s, d.isUAV); HashValue(s, d.isTypeless); HashValue(s, d.isShadingRateSurface);
    HashValue(s, d.clearValue.r); HashValue(s, d.clearValue.g); HashValue(s, d.clearValue.b); HashValue(s, d.clearValue.a);
    HashValue(s, d.useClearValue);
    return s;
}

// RenderGraph.cpp:39-56
std::size_t HashResourceDesc(const nvrhi::BufferDesc& d)
{
    std::size_t s = 0;
    HashValue(s, d.byteSize); HashValue(s, d.structStride); HashValue(s, (uint32_t)d.format);
    HashValue(s, d.canHaveUAVs); HashValue(s, d.canHaveTypedViews); HashValue(s, d.canHaveRawViews);
    HashValue(s, d.isVertexBuffer); HashValue(s, d.isIndexBuffer); HashValue(s, d.isConstantBuffer);
    HashValue(s, d.isDrawIndirectArgs); HashValue(s, d.isAccelStructBuildInput); HashValue(s, d.isAccelStructStorage);
    HashValue(s, d.isShaderBindingTable);
    return s;
}

inline uint64_t AlignUp64(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }
} // namespace

void RenderGraph::Initialize()
{
    CreateNewHeap(kDefaultHeapBlockSize);                                     // RenderGraph.cpp:58-61
}

void RenderGraph::InitializeForFrame(tf::Taskflow& taskFlow)
{
    m_TaskFlow = &taskFlow;                                                   // :63-74
    m_CommandListQueueTasks.clear();
    m_Passes.clear();
    m_CurrentPhase = Phase::Setup;
}

void RenderGraph::Shutdown()
{
    for (ResourceHandle* h : m_ResourceHandles) {                             // :76-82
        h->m_Resource = nullptr;
        h->m_HeapIdx = UINT32_MAX;
        h->m_HeapOffset = UINT64_MAX;
        h->m_AllocatedFrameIdx = UINT32_MAX;
        h->m_DescIdx = UINT32_MAX;
        h->m_FirstAccess = h->m_LastAccess = kInvalidPassID;
    }
    m_ResourceHandles.clear();
    m_ResourceDescs.clear();
    m_ResourcesToAlloc.clear();
    m_HeapsToFree.clear();
    m_Passes.clear();
    m_CommandListQueueTasks.clear();
    m_Heaps.clear();
}

void RenderGraph::Compile()
{
    m_CurrentPhase = Phase::Execute;                                          // :88

    // command lists are queued in pass (registration) order (:91-94)
    for (size_t i = 1; i < m_CommandListQueueTasks.size(); ++i)
        m_CommandListQueueTasks[i].succeed(m_CommandListQueueTasks[i - 1]);

    // first / last access per resource; the first access must be a write (:97-121)
    for (size_t i = 0; i < m_Passes.size(); ++i) {
        const PassID passID = (PassID)i;
        for (const ResourceAccess& access : m_Passes[i].m_ResourceAccesses) {
            ResourceHandle& r = *access.m_ResourceHandle;
            if (r.m_FirstAccess == kInvalidPassID) {
                check(access.m_AccessType == ResourceHandle::AccessType::Write);
                r.m_FirstAccess = passID;
            }
            r.m_LastAccess = passID;
        }
    }

    // ---- this build: cross-queue waits (async compute).  Pass i waits for an earlier pass j on another queue when they
    // touch a common resource and at least one of the two writes it (read-after-write, write-after-write,
    // write-after-read).  Queues are in order, so the LATEST such j per foreign queue is enough.
    m_FrameStats = FrameStats{};
    for (size_t i = 0; i < m_Passes.size(); ++i) {
        Pass& pi = m_Passes[i];
        pi.m_WaitForPasses.clear();
        const nvrhi::CommandQueue qi = pi.m_CommandList->m_Queue;
        if (qi == nvrhi::CommandQueue::Compute) ++m_FrameStats.m_NumComputeQueuePasses;
        for (size_t j = i; j-- > 0;) {
            const Pass& pj = m_Passes[j];
            if (pj.m_CommandList->m_Queue == qi) continue;
            bool hazard = false;
            for (const ResourceAccess& a : pi.m_ResourceAccesses)
                for (const ResourceAccess& b : pj.m_ResourceAccesses)
                    hazard |= a.m_ResourceHandle == b.m_ResourceHandle &&
                              (a.m_AccessType == ResourceHandle::AccessType::Write || b.m_AccessType == ResourceHandle::AccessType::Write);
            for (const ExternalAccess& a : pi.m_ExternalAccesses)
                for (const ExternalAccess& b : pj.m_ExternalAccesses)
                    hazard |= a.m_Resource == b.m_Resource &&
                              (a.m_AccessType == ResourceHandle::AccessType::Write || b.m_AccessType == ResourceHandle::AccessType::Write);
            if (hazard) { pi.m_WaitForPasses.push_back((PassID)j); ++m_FrameStats.m_NumCrossQueueWaits; break; }
        }
    }

    // age out transient resources nobody asked for during the last frames (:123-135)
    for (ResourceHandle* h : m_ResourceHandles) {
        check(h->m_AllocatedFrameIdx != UINT32_MAX);
        const int32_t age = (int32_t)(g_Graphic.m_FrameCounter - h->m_AllocatedFrameIdx);
        check(age >= 0);
        if (h->m_Resource && (uint32_t)age > kMaxTransientResourceAge) FreeResource(*h);
    }

    // create + place the resources that were (re)requested this frame (:139-208)
    nvrhi::DeviceHandle device = g_Graphic.m_NVRHIDevice;
    for (ResourceHandle* r : m_ResourcesToAlloc) {
        check(r->m_DescIdx != UINT32_MAX);
        uint64_t memReq = 0;
        if (r->m_Type == ResourceHandle::Type::Texture) {
            r->m_Resource = device->createTexture(m_ResourceDescs[r->m_DescIdx].m_TextureDesc).Get();
            memReq = device->getTextureMemoryRequirements((nvrhi::ITexture*)r->m_Resource.Get()).size;
        } else {
            r->m_Resource = device->createBuffer(m_ResourceDescs[r->m_DescIdx].m_BufferDesc).Get();
            memReq = device->getBufferMemoryRequirements((nvrhi::IBuffer*)r->m_Resource.Get()).size;
        }
        memReq = AlignUp64(memReq, kHeapAlignment);
        check(memReq != 0);
        check(memReq <= ms_MaxHeapBlockSize);

        uint32_t heapIdx = UINT32_MAX;
        uint64_t heapOffset = UINT64_MAX;
        for (uint32_t i = 0; i < m_Heaps.size(); ++i) {
            if (m_Heaps[i].m_Heap->getDesc().capacity < memReq) continue;
            heapOffset = m_Heaps[i].Allocate(memReq);
            if (heapOffset != UINT64_MAX) { heapIdx = i; break; }
        }
        if (heapIdx == UINT32_MAX) {                                          // :176-182
            CreateNewHeap(std::max(memReq, kDefaultHeapBlockSize));
            heapIdx = (uint32_t)m_Heaps.size() - 1;
            heapOffset = m_Heaps.back().Allocate(memReq);
        }
        check(heapIdx != UINT32_MAX && heapOffset != UINT64_MAX);
        r->m_HeapIdx = heapIdx;
        r->m_HeapOffset = heapOffset;
        if (r->m_Type == ResourceHandle::Type::Texture)
            check(device->bindTextureMemory((nvrhi::ITexture*)r->m_Resource.Get(), m_Heaps[heapIdx].m_Heap, heapOffset));
        else
            check(device->bindBufferMemory((nvrhi::IBuffer*)r->m_Resource.Get(), m_Heaps[heapIdx].m_Heap, heapOffset));
    }
    m_ResourcesToAlloc.clear();

    // heap ranges released this frame become free only now (:211-220)
    for (const HeapToFree& e : m_HeapsToFree) m_Heaps.at(e.m_Idx).Free(e.m_Offset);
    m_HeapsToFree.clear();

    // ---- this build: what lifetime aliasing would save (statistics only).  Peak over the passes of the bytes of the
    // resources live at that pass = the optimum for interval-shaped lifetimes.
    std::vector<uint64_t> liveAtPass(m_Passes.size() + 1, 0);
    for (ResourceHandle* h : m_ResourceHandles) {
        if (!h->m_Resource || h->m_FirstAccess == kInvalidPassID) continue;
        uint64_t bytes = 0;
        if (h->m_Type == ResourceHandle::Type::Texture) bytes = device->getTextureMemoryRequirements((nvrhi::ITexture*)h->m_Resource.Get()).size;
        else bytes = device->getBufferMemoryRequirements((nvrhi::IBuffer*)h->m_Resource.Get()).size;
        bytes = AlignUp64(bytes, kHeapAlignment);
        m_FrameStats.m_TransientBytes += bytes;
        for (size_t p = h->m_FirstAccess; p <= h->m_LastAccess && p < m_Passes.size(); ++p) liveAtPass[p] += bytes;
    }
    for (uint64_t b : liveAtPass) m_FrameStats.m_AliasedBytes = std::max(m_FrameStats.m_AliasedBytes, b);
}

tf::Task RenderGraph::AddRenderer(IRenderer* renderer)
{
    check(renderer);                                                          // :223-230 (single-threaded Setup phase)
    check(m_CurrentPhase == Phase::Setup);
    check(m_Passes.size() < kInvalidPassID);
    const PassID passIdx = (PassID)m_Passes.size();
    m_Passes.emplace_back();

    if (!renderer->Setup(*this)) {                                            // :237-248
        // a renderer that opts out must not have registered any access
        check(m_Passes.back().m_ResourceAccesses.empty() && m_Passes.back().m_ExternalAccesses.empty());
        m_Passes.pop_back();
        renderer->m_CPUFrameTime = 0.0f;
        renderer->m_GPUFrameTime = 0.0f;
        return m_TaskFlow->placeholder();
    }

    m_Passes.back().m_Renderer = renderer;
    m_Passes.back().m_CommandList = g_Graphic.AllocateCommandList(renderer->m_Queue);   // :251 "TODO: compute queue": the renderer's choice

    tf::Task renderTask = m_TaskFlow->emplace([this, passIdx] {               // :254-288
        tl_CurrentThreadPassID = passIdx;
        Pass& pass = m_Passes.at(passIdx);
        IRenderer* r = pass.m_Renderer;
        check(r && pass.m_CommandList);
        const auto t0 = std::chrono::steady_clock::now();
        {
            SCOPED_COMMAND_LIST(pass.m_CommandList, r->m_Name.c_str());
            if (g_Graphic.m_bEnableGPUTimers) {
                nvrhi::TimerQueryHandle& query = r->m_FrameTimerQuery[g_Graphic.m_FrameCounter % 2];
                if (!query) query = g_Graphic.m_NVRHIDevice->createTimerQuery();
                r->m_GPUFrameTime = 1e3f * g_Graphic.m_NVRHIDevice->getTimerQueryTime(query);   // result of 2 frames ago
                g_Graphic.m_NVRHIDevice->resetTimerQuery(query);
                pass.m_CommandList->beginTimerQuery(query);
                r->Render(pass.m_CommandList, *this);
                pass.m_CommandList->endTimerQuery(query);
            } else {
                r->m_GPUFrameTime = 0.0f;
                r->Render(pass.m_CommandList, *this);
            }
        }
        r->m_CPUFrameTime = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        tl_CurrentThreadPassID = kInvalidPassID;
    });

    tf::Task queueTask = m_TaskFlow->emplace([this, passIdx] {                // :291-299
        Pass& pass = m_Passes.at(passIdx);
        check(pass.m_CommandList);
        std::vector<const nvrhi::ICommandList*> waitFor;                      // cross-queue hazards found by Compile()
        for (PassID j : pass.m_WaitForPasses) waitFor.push_back(m_Passes.at(j).m_CommandList.Get());
        g_Graphic.QueueCommandList(pass.m_CommandList, std::move(waitFor));
    });
    queueTask.succeed(renderTask);
    m_CommandListQueueTasks.push_back(queueTask);
    return renderTask;
}

template <typename ResourceDescT>
void RenderGraph::CreateTransientResource(ResourceHandle& h, const ResourceDescT& inputDesc)
{
    check(m_CurrentPhase == Phase::Setup);                                    // :307
    constexpr ResourceHandle::Type type = std::is_same_v<ResourceDescT, nvrhi::TextureDesc> ? ResourceHandle::Type::Texture : ResourceHandle::Type::Buffer;

    if (h.m_AllocatedFrameIdx == UINT32_MAX) {                                // first registration (:311-320)
        m_ResourceHandles.push_back(&h);
        h.m_DescIdx = (uint32_t)m_ResourceDescs.size();
        m_ResourceDescs.emplace_back();
    }

    bool realloc = type != h.m_Type;                                          // :322-333
    realloc |= (g_Graphic.m_FrameCounter - h.m_AllocatedFrameIdx) > kMaxTransientResourceAge;
    if constexpr (type == ResourceHandle::Type::Texture)
        realloc |= HashResourceDesc(m_ResourceDescs[h.m_DescIdx].m_TextureDesc) != HashResourceDesc(inputDesc);
    else
        realloc |= HashResourceDesc(m_ResourceDescs[h.m_DescIdx].m_BufferDesc) != HashResourceDesc(inputDesc);
    realloc |= !h.m_Resource;

    if (realloc) {                                                            // :335-339
        FreeResource(h);
        if (std::find(m_ResourcesToAlloc.begin(), m_ResourcesToAlloc.end(), &h) == m_ResourcesToAlloc.end())
            m_ResourcesToAlloc.push_back(&h);
    }
    h.m_AllocatedFrameIdx = g_Graphic.m_FrameCounter;
    h.m_Type = type;

    if constexpr (type == ResourceHandle::Type::Texture) {                    // :344-355
        nvrhi::TextureDesc& d = m_ResourceDescs[h.m_DescIdx].m_TextureDesc;
        d = inputDesc;
        d.isVirtual = true;
    } else {
        nvrhi::BufferDesc& d = m_ResourceDescs[h.m_DescIdx].m_BufferDesc;
        d = inputDesc;
        d.isVirtual = true;
    }
    AddWriteDependency(h);                                                    // the creator writes it (:358)
}
template void RenderGraph::CreateTransientResource(ResourceHandle&, const nvrhi::TextureDesc&);
template void RenderGraph::CreateTransientResource(ResourceHandle&, const nvrhi::BufferDesc&);

void RenderGraph::AddDependencyInternal(ResourceHandle& h, ResourceHandle::AccessType accessType)
{
    check(m_CurrentPhase == Phase::Setup);                                    // :363-378
    check(!m_Passes.empty());
    std::vector<ResourceAccess>& accesses = m_Passes.back().m_ResourceAccesses;
    for (const ResourceAccess& a : accesses) check(a.m_ResourceHandle != &h);  // one dependency per pass and resource
    accesses.push_back(ResourceAccess{ &h, accessType });
}

void RenderGraph::AddExternalDependencyInternal(const nvrhi::IResource* resource, ResourceHandle::AccessType accessType)
{
    check(m_CurrentPhase == Phase::Setup);
    check(!m_Passes.empty());
    if (!resource) return;
    std::vector<ExternalAccess>& accesses = m_Passes.back().m_ExternalAccesses;
    for (ExternalAccess& a : accesses)
        if (a.m_Resource == resource) { if (accessType == ResourceHandle::AccessType::Write) a.m_AccessType = accessType; return; }
    accesses.push_back(ExternalAccess{ resource, accessType });
}

nvrhi::IResource* RenderGraph::GetResourceInternal(const ResourceHandle& h, ResourceHandle::Type type) const
{
    check(m_CurrentPhase == Phase::Execute);                                  // :380-399
    check(h.m_AllocatedFrameIdx != UINT32_MAX);                               // never registered
    check(h.m_AllocatedFrameIdx == g_Graphic.m_FrameCounter);                 // not requested this frame
    check(tl_CurrentThreadPassID != kInvalidPassID);
    const std::vector<ResourceAccess>& accesses = m_Passes.at(tl_CurrentThreadPassID).m_ResourceAccesses;
    check(std::any_of(accesses.begin(), accesses.end(), [&h](const ResourceAccess& a) { return a.m_ResourceHandle == &h; }));
    check(h.m_Resource);
    check(h.m_Type == type);
    return h.m_Resource.Get();
}

void RenderGraph::FreeResource(ResourceHandle& h)
{
    h.m_Resource = nullptr;                                                   // :401-422
    h.m_FirstAccess = h.m_LastAccess = kInvalidPassID;
    if (h.m_HeapIdx != UINT32_MAX) {
        check(h.m_HeapOffset != UINT64_MAX);
        m_HeapsToFree.push_back({ h.m_HeapIdx, h.m_HeapOffset });
    }
    h.m_HeapIdx = UINT32_MAX;
    h.m_HeapOffset = UINT64_MAX;
}

const char* RenderGraph::GetResourceName(const ResourceHandle& h) const
{
    return h.m_Type == ResourceHandle::Type::Texture ? m_ResourceDescs.at(h.m_DescIdx).m_TextureDesc.debugName.c_str()
                                                      : m_ResourceDescs.at(h.m_DescIdx).m_BufferDesc.debugName.c_str();
}

void RenderGraph::CreateNewHeap(uint64_t size)
{
    Heap& heap = m_Heaps.emplace_back();                                      // :431-441
    heap.m_Blocks.push_back({ size, false });
    heap.m_Heap = g_Graphic.m_NVRHIDevice->createHeap(nvrhi::HeapDesc{ size, nvrhi::HeapType::DeviceLocal, "RDG Heap" });
}

// ---- free-list allocator (RenderGraph.cpp:443-580) ---------------------------------------------------
uint64_t RenderGraph::Heap::Allocate(uint64_t size)
{
    check(!m_Blocks.empty());
    check(size % kHeapAlignment == 0);
    uint32_t idx = UINT32_MAX;
    uint64_t offset = 0;
    FindBest(size, idx, offset);
    if (idx == UINT32_MAX) return UINT64_MAX;
    check(!m_Blocks[idx].m_Allocated);
    const uint64_t remaining = m_Blocks[idx].m_Size - size;
    if (remaining > 0) m_Blocks.insert(m_Blocks.begin() + idx + 1, Block{ remaining, false });   // split
    m_Blocks[idx].m_Size = size;
    m_Blocks[idx].m_Allocated = true;
    m_Used += size;
    m_Peak = std::max(m_Peak, m_Used);
    return offset;     // offsets, not indices: Free() merges blocks and would invalidate indices
}

void RenderGraph::Heap::Free(uint64_t heapOffset)
{
    check(heapOffset != UINT64_MAX && heapOffset % kHeapAlignment == 0);
    uint32_t idx = 0;
    for (uint64_t off = 0; idx < m_Blocks.size(); off += m_Blocks[idx].m_Size, ++idx)
        if (off == heapOffset) break;
    check(idx < m_Blocks.size());
    check(m_Blocks[idx].m_Allocated);
    m_Used -= m_Blocks[idx].m_Size;
    m_Blocks[idx].m_Allocated = false;
    if (idx + 1 < m_Blocks.size() && !m_Blocks[idx + 1].m_Allocated) {       // merge with the next block
        m_Blocks[idx].m_Size += m_Blocks[idx + 1].m_Size;
        m_Blocks.erase(m_Blocks.begin() + idx + 1);
    }
    if (idx > 0 && !m_Blocks[idx - 1].m_Allocated) {                          // merge with the previous block
        m_Blocks[idx - 1].m_Size += m_Blocks[idx].m_Size;
        m_Blocks.erase(m_Blocks.begin() + idx);
    }
    check(!m_Blocks.empty());
}

void RenderGraph::Heap::FindBest(uint64_t size, uint32_t& foundIdx, uint64_t& heapOffset)
{
    // Best fit = smallest left-over.  Like the reference (:533-557) a block whose left-over would be
    // kDefaultHeapBlockSize or more is not considered, so small requests never carve up a big heap.
    uint64_t smallest = kDefaultHeapBlockSize;
    uint64_t off = 0;
    for (uint32_t i = 0; i < m_Blocks.size(); off += m_Blocks[i].m_Size, ++i) {
        if (m_Blocks[i].m_Allocated || m_Blocks[i].m_Size < size) continue;
        const uint64_t remaining = m_Blocks[i].m_Size - size;
        if (remaining < smallest) { foundIdx = i; heapOffset = off; smallest = remaining; }
    }
}

void RenderGraph::Heap::FindFirst(uint64_t size, uint32_t& foundIdx, uint64_t& heapOffset)
{
    uint64_t off = 0;                                                         // :559-580
    for (uint32_t i = 0; i < m_Blocks.size(); off += m_Blocks[i].m_Size, ++i)
        if (!m_Blocks[i].m_Allocated && m_Blocks[i].m_Size >= size) { foundIdx = i; heapOffset = off; return; }
}
