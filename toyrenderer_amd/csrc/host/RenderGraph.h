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
    struct Pass { IRenderer* m_Renderer = nullptr; std::vector<ResourceAccess> m_ResourceAccesses; nvrhi::CommandListHandle m_CommandList; };

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

    // Execute-phase functions
    [[nodiscard]] nvrhi::TextureHandle GetTexture(const ResourceHandle& resourceHandle) const { return (nvrhi::ITexture*)GetResourceInternal(resourceHandle, ResourceHandle::Type::Texture); }
    [[nodiscard]] nvrhi::BufferHandle GetBuffer(const ResourceHandle& resourceHandle) const { return (nvrhi::IBuffer*)GetResourceInternal(resourceHandle, ResourceHandle::Type::Buffer); }

    // introspection for tests / stats (the reference shows these in an ImGui table, Scene.cpp:530-562)
    const std::vector<Heap>& GetHeaps() const { return m_Heaps; }
    size_t GetNumPasses() const { return m_Passes.size(); }
    // this build: upper bound of one transient resource (the reference asserts 1 GB, RenderGraph.cpp:158)
    static uint64_t ms_MaxHeapBlockSize;

private:
    void AddDependencyInternal(ResourceHandle& resourceHandle, ResourceHandle::AccessType accessType);
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
