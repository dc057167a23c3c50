// RenderGraph.h -- pass / transient-resource registration API of the reference
// (source/RenderGraph.h:8-121), kept call-compatible so that renderers written against it
// (BasePassRenderers.cpp) compile unchanged in structure: ResourceHandle, CreateTransientResource,
// AddRead/WriteDependency, GetBuffer/GetTexture, AddRenderer, Compile, placed resources from pooled
// heaps.  Underneath, heaps are hipMalloc blocks and resources are placed buffers of the C ABI.
#pragma once

#include <limits>
#include <vector>

#include "nvrhi_lite.h"
#include "tf_lite.h"

class IRenderer;

class RenderGraph
{
public:
    using PassID = uint8_t;
    static const PassID kInvalidPassID = std::numeric_limits<PassID>::max();

    enum class Phase { Setup, Execute };

    // RenderGraph.h:17-34.  Caller-owned, persistent across frames; its ADDRESS is its identity
    // (the graph stores ResourceHandle*, RenderGraph.cpp:315).
    struct ResourceHandle
    {
        enum class Type : uint8_t { Texture, Buffer };
        enum class AccessType : uint8_t { Read, Write };

        nvrhi::ResourceHandle m_Resource;
        uint64_t m_HeapOffset = UINT64_MAX;
        uint32_t m_HeapIdx = UINT32_MAX;
        uint32_t m_AllocatedFrameIdx = UINT32_MAX;
        uint32_t m_DescIdx = UINT32_MAX;
        Type m_Type = Type::Buffer;
        PassID m_FirstAccess = kInvalidPassID;
        PassID m_LastAccess = kInvalidPassID;
    };

    struct ResourceDesc { nvrhi::TextureDesc m_TextureDesc; nvrhi::BufferDesc m_BufferDesc; };
    struct ResourceAccess { ResourceHandle* m_ResourceHandle; ResourceHandle::AccessType m_AccessType; };
    // m_ExternalAccesses / m_WaitForPasses: this build (async compute queue, RenderGraph.cpp:251 "TODO: compute queue")
    struct ExternalAccess { const nvrhi::IResource* m_Resource; ResourceHandle::AccessType m_AccessType; };
    struct Pass
    {
        IRenderer* m_Renderer = nullptr;
        std::vector<ResourceAccess> m_ResourceAccesses;
        nvrhi::CommandListHandle m_CommandList;
        std::vector<ExternalAccess> m_ExternalAccesses;
        std::vector<PassID> m_WaitForPasses;          // earlier passes on ANOTHER queue this pass has a hazard with
    };

    // RenderGraph.h:56-74: free-list allocator over one device heap
    struct Heap
    {
        uint64_t Allocate(uint64_t size);
        void Free(uint64_t heapOffset);
        void FindBest(uint64_t size, uint32_t& foundIdx, uint64_t& heapOffset);
        void FindFirst(uint64_t size, uint32_t& foundIdx, uint64_t& heapOffset);

        nvrhi::HeapHandle m_Heap;
        struct Block { uint64_t m_Size; bool m_Allocated; };
        std::vector<Block> m_Blocks;
        uint64_t m_Used = 0;
        uint64_t m_Peak = 0;
    };

    void Initialize();
    void InitializeForFrame(tf::Taskflow& taskFlow);
    void Shutdown();
    void Compile();
    tf::Task AddRenderer(IRenderer* renderer);

    // Setup-phase functions
    template <typename ResourceDescT>
    void CreateTransientResource(ResourceHandle& resourceHandle, const ResourceDescT& resourceDesc);
    void AddReadDependency(ResourceHandle& resourceHandle) { AddDependencyInternal(resourceHandle, ResourceHandle::AccessType::Read); }
    void AddWriteDependency(ResourceHandle& resourceHandle) { AddDependencyInternal(resourceHandle, ResourceHandle::AccessType::Write); }
    // this build: accesses to resources the graph does not own (scene buffers, the HZB).  On one queue the queue order
    // covers them, as in the reference; with passes on two queues the graph needs them to place the cross-queue waits.
    void AddExternalReadDependency(const nvrhi::IResource* resource) { AddExternalDependencyInternal(resource, ResourceHandle::AccessType::Read); }
    void AddExternalWriteDependency(const nvrhi::IResource* resource) { AddExternalDependencyInternal(resource, ResourceHandle::AccessType::Write); }

    // Execute-phase functions
    [[nodiscard]] nvrhi::TextureHandle GetTexture(const ResourceHandle& resourceHandle) const { return (nvrhi::ITexture*)GetResourceInternal(resourceHandle, ResourceHandle::Type::Texture); }
    [[nodiscard]] nvrhi::BufferHandle GetBuffer(const ResourceHandle& resourceHandle) const { return (nvrhi::IBuffer*)GetResourceInternal(resourceHandle, ResourceHandle::Type::Buffer); }

    // introspection for tests / stats (the reference shows these in an ImGui table, Scene.cpp:530-562)
    const std::vector<Heap>& GetHeaps() const { return m_Heaps; }
    size_t GetNumPasses() const { return m_Passes.size(); }
    // Of the last compiled frame: passes on the compute queue, cross-queue waits derived from the declared accesses,
    // bytes of the live transient resources, and the bytes a lifetime-aliasing allocator would need for them (resources
    // whose [first, last] pass ranges do not overlap sharing memory; the reference, like this build, places every
    // transient resource for the whole frame -- RenderGraph.cpp:139-208 -- and shows no such number).
    struct FrameStats { uint32_t m_NumComputeQueuePasses = 0, m_NumCrossQueueWaits = 0; uint64_t m_TransientBytes = 0, m_AliasedBytes = 0; };
    const FrameStats& GetFrameStats() const { return m_FrameStats; }
    // this build: upper bound of one transient resource (the reference asserts 1 GB, RenderGraph.cpp:158)
    static uint64_t ms_MaxHeapBlockSize;

private:
    void AddDependencyInternal(ResourceHandle& resourceHandle, ResourceHandle::AccessType accessType);
    void AddExternalDependencyInternal(const nvrhi::IResource* resource, ResourceHandle::AccessType accessType);
    FrameStats m_FrameStats;
    nvrhi::IResource* GetResourceInternal(const ResourceHandle& resourceHandle, ResourceHandle::Type resourceType) const;
    void FreeResource(ResourceHandle& resourceHandle);
    const char* GetResourceName(const ResourceHandle& resourceHandle) const;
    void CreateNewHeap(uint64_t size);

    tf::Taskflow* m_TaskFlow = nullptr;
    std::vector<tf::Task> m_CommandListQueueTasks;
    std::vector<Pass> m_Passes;
    std::vector<ResourceHandle*> m_ResourceHandles;
    std::vector<ResourceDesc> m_ResourceDescs;
    struct HeapToFree { uint32_t m_Idx; uint64_t m_Offset; };
    std::vector<HeapToFree> m_HeapsToFree;
    std::vector<ResourceHandle*> m_ResourcesToAlloc;
    Phase m_CurrentPhase = Phase::Setup;
    std::vector<Heap> m_Heaps;
};
