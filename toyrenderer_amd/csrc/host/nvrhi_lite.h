// nvrhi_lite.h -- the slice of the reference's RHI (extern/nvrhi, used as cited below) that the
// visibility path touches, re-typed over the C ABI in include/trhip.h.
//
// Same names, argument meaning and ownership model as the calls the reference makes
// (source/BasePassRenderers.cpp:223-616, source/Graphic.cpp:893-947, source/RenderGraph.cpp,
// source/FFXHelpers.cpp): intrusive ref-counted handles (nvrhi::RefCountPtr), descriptor structs
// with public fields, BindingSetItem factories, ICommandList recording, IDevice factories.  Error
// convention of the reference: no error returns, failures assert (PCH.h:42 `check`); here a failed
// C-ABI call throws nvrhi::Error carrying trhip_last_error() after logging it, and `check()`
// aborts like SDL_assert.
#pragma once

#include "HostProfile.h"

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/trhip.h"

#define check(expr)                                                                          \
    do {                                                                                     \
        if (!(expr)) {                                                                       \
            std::fprintf(stderr, "check failed: %s (%s:%d)\n", #expr, __FILE__, __LINE__);   \
            std::abort();                                                                    \
        }                                                                                    \
    } while (0)

namespace nvrhi
{

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

inline void throwIfFailed(int rc, const char* what)
{
    if (rc != TRHIP_OK) {
        std::string msg = std::string(what) + ": " + trhip_last_error();
        std::fprintf(stderr, "[nvrhi_lite] %s\n", msg.c_str());   // GraphicRHI.cpp:18-38 message callback
        throw Error(msg);
    }
}

// ---- ref counting (nvrhi::RefCountPtr / IResource) ---------------------------------------------
class IResource
{
public:
    unsigned long AddRef() { return ++m_RefCount; }
    unsigned long Release()
    {
        unsigned long r = --m_RefCount;
        if (r == 0) delete this;
        return r;
    }

protected:
    IResource() = default;
    virtual ~IResource() = default;

private:
    std::atomic<unsigned long> m_RefCount{0};
};

template <typename T>
class RefCountPtr
{
public:
    RefCountPtr() = default;
    RefCountPtr(std::nullptr_t) {}
    RefCountPtr(T* p) : m_Ptr(p) { if (m_Ptr) m_Ptr->AddRef(); }
    RefCountPtr(const RefCountPtr& o) : m_Ptr(o.m_Ptr) { if (m_Ptr) m_Ptr->AddRef(); }
    RefCountPtr(RefCountPtr&& o) noexcept : m_Ptr(o.m_Ptr) { o.m_Ptr = nullptr; }
    template <typename U> RefCountPtr(const RefCountPtr<U>& o) : m_Ptr(o.Get()) { if (m_Ptr) m_Ptr->AddRef(); }
    ~RefCountPtr() { if (m_Ptr) m_Ptr->Release(); }
    RefCountPtr& operator=(const RefCountPtr& o) { RefCountPtr(o).swap(*this); return *this; }
    RefCountPtr& operator=(RefCountPtr&& o) noexcept { RefCountPtr(std::move(o)).swap(*this); return *this; }
    RefCountPtr& operator=(std::nullptr_t) { RefCountPtr().swap(*this); return *this; }
    T* Get() const { return m_Ptr; }
    T* operator->() const { return m_Ptr; }
    operator T*() const { return m_Ptr; }
    explicit operator bool() const { return m_Ptr != nullptr; }
    void swap(RefCountPtr& o) { std::swap(m_Ptr, o.m_Ptr); }

private:
    T* m_Ptr = nullptr;
};

using ResourceHandle = RefCountPtr<IResource>;

// ---- enums / small structs -----------------------------------------------------------------------
enum class Format : uint8_t { UNKNOWN, R16_FLOAT, R32_FLOAT, D24S8 };        // GraphicConstants.h:26-28
enum class ResourceStates : uint32_t { Unknown = 0, ShaderResource, UnorderedAccess, IndirectArgument, DepthRead, DepthWrite, CopyDest };
enum class CommandQueue : uint8_t { Graphics = 0, Compute, Copy, Count };
enum class HeapType : uint8_t { DeviceLocal };
enum class SamplerReductionType : uint8_t { Standard, Comparison, Minimum, Maximum };
enum class SamplerAddressMode : uint8_t { Clamp, Wrap };

struct Color { float r = 0, g = 0, b = 0, a = 0; Color() = default; explicit Color(float c) : r(c), g(c), b(c), a(c) {} };

struct TextureSubresourceSet
{
    uint32_t baseMipLevel = 0, numMipLevels = 1, baseArraySlice = 0, numArraySlices = 1;
    TextureSubresourceSet() = default;
    TextureSubresourceSet(uint32_t m, uint32_t nm, uint32_t s, uint32_t ns) : baseMipLevel(m), numMipLevels(nm), baseArraySlice(s), numArraySlices(ns) {}
};
static const TextureSubresourceSet AllSubresources{0, ~0u, 0, ~0u};

struct MemoryRequirements { uint64_t size = 0, alignment = 0; };

// ---- descriptions (fields used on the path: BasePassRenderers.cpp:236-291,596-606) ----------------
struct BufferDesc
{
    uint64_t byteSize = 0;
    uint32_t structStride = 0;
    std::string debugName;
    Format format = Format::UNKNOWN;
    bool canHaveUAVs = false, canHaveTypedViews = false, canHaveRawViews = false;
    bool isVertexBuffer = false, isIndexBuffer = false, isConstantBuffer = false, isDrawIndirectArgs = false;
    bool isAccelStructBuildInput = false, isAccelStructStorage = false, isShaderBindingTable = false;
    bool isVolatile = false, isVirtual = false;
    ResourceStates initialState = ResourceStates::Unknown;
};

struct TextureDesc
{
    uint32_t width = 1, height = 1, depth = 1, arraySize = 1, mipLevels = 1, sampleCount = 1, sampleQuality = 0;
    Format format = Format::UNKNOWN;
    uint8_t dimension = 2;
    std::string debugName;
    bool isRenderTarget = false, isUAV = false, isTypeless = false, isShadingRateSurface = false, isVirtual = false;
    Color clearValue;
    bool useClearValue = false;
    ResourceStates initialState = ResourceStates::Unknown;
    TextureDesc& setClearValue(const Color& c) { clearValue = c; useClearValue = true; return *this; }
};

struct HeapDesc
{
    uint64_t capacity = 0;
    HeapType type = HeapType::DeviceLocal;
    std::string debugName;
};

struct SamplerDesc
{
    bool minFilter = true, magFilter = true, mipFilter = true;
    SamplerAddressMode addressU = SamplerAddressMode::Clamp, addressV = SamplerAddressMode::Clamp, addressW = SamplerAddressMode::Clamp;
    SamplerReductionType reductionType = SamplerReductionType::Standard;
};

namespace utils
{
// nvrhi::utils::CreateVolatileConstantBufferDesc (Graphic.h:69)
inline BufferDesc CreateVolatileConstantBufferDesc(uint32_t byteSize, const char* debugName, uint32_t /*maxVersions*/)
{
    BufferDesc d;
    d.byteSize = byteSize;
    d.debugName = debugName ? debugName : "";
    d.isConstantBuffer = true;
    d.isVolatile = true;
    return d;
}
} // namespace utils

// ---- resources -------------------------------------------------------------------------------------
class IHeap : public IResource
{
public:
    IHeap(trhip_heap h, HeapDesc d) : m_Native(h), m_Desc(std::move(d)) {}
    ~IHeap() override { trhip_heap_release(m_Native); }
    const HeapDesc& getDesc() const { return m_Desc; }
    trhip_heap native() const { return m_Native; }

private:
    trhip_heap m_Native;
    HeapDesc m_Desc;
};

class IBuffer : public IResource
{
public:
    IBuffer(trhip_buffer b, BufferDesc d) : m_Native(b), m_Desc(std::move(d)) {}
    ~IBuffer() override { trhip_buffer_release(m_Native); }
    const BufferDesc& getDesc() const { return m_Desc; }
    trhip_buffer native() const { return m_Native; }

private:
    trhip_buffer m_Native;
    BufferDesc m_Desc;
};

class ITexture : public IResource
{
public:
    ITexture(trhip_texture t, TextureDesc d) : m_Native(t), m_Desc(std::move(d)) {}
    ~ITexture() override { trhip_texture_release(m_Native); }
    const TextureDesc& getDesc() const { return m_Desc; }
    trhip_texture native() const { return m_Native; }

private:
    trhip_texture m_Native;
    TextureDesc m_Desc;
};

class ISampler : public IResource
{
public:
    explicit ISampler(SamplerDesc d) : m_Desc(d) {}
    const SamplerDesc& getDesc() const { return m_Desc; }

private:
    SamplerDesc m_Desc;
};

class ITimerQuery : public IResource
{
public:
    explicit ITimerQuery(trhip_timer t) : m_Native(t) {}
    ~ITimerQuery() override { trhip_timer_release(m_Native); }
    trhip_timer native() const { return m_Native; }
    bool m_Recorded = false;

private:
    trhip_timer m_Native;
};

using HeapHandle = RefCountPtr<IHeap>;
using BufferHandle = RefCountPtr<IBuffer>;
using TextureHandle = RefCountPtr<ITexture>;
using SamplerHandle = RefCountPtr<ISampler>;
using TimerQueryHandle = RefCountPtr<ITimerQuery>;

// ---- binding sets (Graphic.cpp:488-518; items used: BasePassRenderers.cpp:351-362,463-479,521-526) ---
struct BindingSetItem
{
    uint32_t type = 0;           // trhip_binding_type
    uint32_t slot = 0;
    IBuffer* buffer = nullptr;
    ITexture* texture = nullptr;
    ISampler* sampler = nullptr;
    uint32_t baseMip = 0;
    uint32_t pushBytes = 0;

    static BindingSetItem ConstantBuffer(uint32_t slot, IBuffer* b) { BindingSetItem i; i.type = TRHIP_BIND_CONSTANT_BUFFER; i.slot = slot; i.buffer = b; return i; }
    static BindingSetItem PushConstants(uint32_t slot, uint32_t bytes) { BindingSetItem i; i.type = TRHIP_BIND_PUSH_CONSTANTS; i.slot = slot; i.pushBytes = bytes; return i; }
    static BindingSetItem StructuredBuffer_SRV(uint32_t slot, IBuffer* b) { BindingSetItem i; i.type = TRHIP_BIND_STRUCTURED_SRV; i.slot = slot; i.buffer = b; return i; }
    static BindingSetItem StructuredBuffer_UAV(uint32_t slot, IBuffer* b) { BindingSetItem i; i.type = TRHIP_BIND_STRUCTURED_UAV; i.slot = slot; i.buffer = b; return i; }
    static BindingSetItem Texture_SRV(uint32_t slot, ITexture* t) { BindingSetItem i; i.type = TRHIP_BIND_TEXTURE_SRV; i.slot = slot; i.texture = t; return i; }
    static BindingSetItem Texture_UAV(uint32_t slot, ITexture* t, Format = Format::UNKNOWN, TextureSubresourceSet s = TextureSubresourceSet{})
    {
        BindingSetItem i; i.type = TRHIP_BIND_TEXTURE_UAV; i.slot = slot; i.texture = t; i.baseMip = s.baseMipLevel; return i;
    }
    static BindingSetItem Sampler(uint32_t slot, ISampler* s) { BindingSetItem i; i.type = TRHIP_BIND_SAMPLER; i.slot = slot; i.sampler = s; return i; }
};

struct BindingSetDesc
{
    std::vector<BindingSetItem> bindings;
};

// nvrhi::ComputeState: the "pipeline" of this back end is the shader-name key of the kernel registry
// (Graphic.cpp:270-278 GetShader + :452-473 PSO cache collapse into one string lookup).
struct ComputeState
{
    std::string pipeline;
    BindingSetDesc bindings;
    IBuffer* indirectParams = nullptr;
};

// ---- command list (nvrhi::ICommandList calls made on the path) ---------------------------------------
class ICommandList : public IResource
{
public:
    explicit ICommandList(trhip_cmdlist cl, CommandQueue queue = CommandQueue::Graphics) : m_Queue(queue), m_Native(cl) {}
    const CommandQueue m_Queue;                      // CommandListParameters::queueType
    ~ICommandList() override { trhip_cmd_release(m_Native); }
    trhip_cmdlist native() const { return m_Native; }

    void open() { m_Keep.clear(); throwIfFailed(trhip_cmd_open(m_Native), "ICommandList::open"); }
    void close() { throwIfFailed(trhip_cmd_close(m_Native), "ICommandList::close"); }
    void writeBuffer(IBuffer* b, const void* data, size_t bytes, uint64_t destOffset = 0)
    {
        keep(b);
        throwIfFailed(trhip_cmd_write_buffer(m_Native, b->native(), destOffset, data, bytes), "ICommandList::writeBuffer");
    }
    void clearBufferUInt(IBuffer* b, uint32_t value) { keep(b); throwIfFailed(trhip_cmd_clear_buffer_u32(m_Native, b->native(), value), "ICommandList::clearBufferUInt"); }
    void clearTextureFloat(ITexture* t, TextureSubresourceSet, const Color& c) { keep(t); throwIfFailed(trhip_cmd_clear_texture_f32(m_Native, t->native(), c.r), "ICommandList::clearTextureFloat"); }
    void copyBuffer(IBuffer* dst, uint64_t dstOff, IBuffer* src, uint64_t srcOff, uint64_t bytes)
    {
        keep(dst); keep(src);
        throwIfFailed(trhip_cmd_copy_buffer(m_Native, dst->native(), dstOff, src->native(), srcOff, bytes), "ICommandList::copyBuffer");
    }
    // Multi-GPU hook (trhip_cmd_host_callback): not part of NVRHI.
    void hostCallback(trhip_host_fn fn, void* user) { throwIfFailed(trhip_cmd_host_callback(m_Native, fn, user), "ICommandList::hostCallback"); }
    void copyTexture(ITexture* dst, ITexture* src) { keep(dst); keep(src); throwIfFailed(trhip_cmd_copy_texture(m_Native, dst->native(), src->native()), "ICommandList::copyTexture"); }

    void setComputeState(const ComputeState& s)
    {
        m_State = s;
        for (const BindingSetItem& i : s.bindings.bindings) { keep(i.buffer); keep(i.texture); }
        keep(s.indirectParams);
    }
    void setPushConstants(const void* data, size_t bytes) { m_Push.assign((const uint8_t*)data, (const uint8_t*)data + bytes); }
    void dispatch(uint32_t gx, uint32_t gy = 1, uint32_t gz = 1)
    {
        std::vector<trhip_binding> b = flatten();
        HOST_PROFILE_SCOPE("trhip_cmd_dispatch (record)");
        throwIfFailed(trhip_cmd_dispatch(m_Native, m_State.pipeline.c_str(), b.data(), (uint32_t)b.size(),
                                         m_Push.empty() ? nullptr : m_Push.data(), (uint32_t)m_Push.size(), gx, gy, gz), "ICommandList::dispatch");
        m_Push.clear();
    }
    void dispatchIndirect(uint32_t offsetBytes)
    {
        check(m_State.indirectParams);
        std::vector<trhip_binding> b = flatten();
        HOST_PROFILE_SCOPE("trhip_cmd_dispatch_indirect (record)");
        throwIfFailed(trhip_cmd_dispatch_indirect(m_Native, m_State.pipeline.c_str(), b.data(), (uint32_t)b.size(),
                                                  m_Push.empty() ? nullptr : m_Push.data(), (uint32_t)m_Push.size(),
                                                  m_State.indirectParams->native(), offsetBytes), "ICommandList::dispatchIndirect");
        m_Push.clear();
    }
    void beginTimerQuery(ITimerQuery* q) { keep(q); throwIfFailed(trhip_cmd_begin_timer(m_Native, q->native()), "ICommandList::beginTimerQuery"); }
    void endTimerQuery(ITimerQuery* q) { q->m_Recorded = true; throwIfFailed(trhip_cmd_end_timer(m_Native, q->native()), "ICommandList::endTimerQuery"); }
    void beginMarker(const char* name) { throwIfFailed(trhip_cmd_begin_marker(m_Native, name), "ICommandList::beginMarker"); }
    void endMarker() { throwIfFailed(trhip_cmd_end_marker(m_Native), "ICommandList::endMarker"); }

private:
    void keep(IResource* r) { if (r) m_Keep.emplace_back(r); }   // referenced resources live until the list is re-opened
    std::vector<trhip_binding> flatten() const
    {
        std::vector<trhip_binding> out;
        out.reserve(m_State.bindings.bindings.size());
        for (const BindingSetItem& i : m_State.bindings.bindings) {
            trhip_binding b{};
            b.type = i.type; b.slot = i.slot; b.baseMip = i.baseMip;
            if (i.buffer) b.resource = i.buffer->native();
            else if (i.texture) b.resource = i.texture->native();
            out.push_back(b);
        }
        return out;
    }

    trhip_cmdlist m_Native;
    ComputeState m_State;
    std::vector<uint8_t> m_Push;
    std::vector<ResourceHandle> m_Keep;
};
using CommandListHandle = RefCountPtr<ICommandList>;

namespace utils
{
struct ScopedMarker
{
    ScopedMarker(ICommandList* cl, const char* name) : m_Cl(cl) { m_Cl->beginMarker(name); }
    ~ScopedMarker() { m_Cl->endMarker(); }
    ICommandList* m_Cl;
};
} // namespace utils

// ---- device (nvrhi::IDevice calls made on the path) ----------------------------------------------------
class IDevice : public IResource
{
public:
    explicit IDevice(trhip_device d) : m_Native(d) {}
    ~IDevice() override { destroyQueues(); trhip_device_destroy(m_Native); }
    trhip_device native() const { return m_Native; }
    trhip_device queueDevice(CommandQueue queue)
    {
        if (queue != CommandQueue::Compute) return m_Native;
        if (!m_ComputeNative) {
            throwIfFailed(trhip_stream_create(m_DeviceIndex, &m_ComputeStream), "IDevice: compute-queue stream");
            throwIfFailed(trhip_device_create_on_stream(m_DeviceIndex, m_ComputeStream, &m_ComputeNative), "IDevice: compute-queue device");
        }
        return m_ComputeNative;
    }

    HeapHandle createHeap(const HeapDesc& d)
    {
        trhip_heap h = nullptr;
        throwIfFailed(trhip_heap_create(m_Native, d.capacity, &h), "IDevice::createHeap");
        return HeapHandle(new IHeap(h, d));
    }
    BufferHandle createBuffer(const BufferDesc& d)
    {
        trhip_buffer_desc n{};
        n.byteSize = d.byteSize; n.structStride = d.structStride; n.canHaveUAVs = d.canHaveUAVs; n.isDrawIndirectArgs = d.isDrawIndirectArgs;
        n.isVirtual = d.isVirtual; n.isVolatileConstant = d.isVolatile && d.isConstantBuffer; n.debugName = d.debugName.c_str();
        trhip_buffer b = nullptr;
        throwIfFailed(trhip_buffer_create(m_Native, &n, &b), "IDevice::createBuffer");
        return BufferHandle(new IBuffer(b, d));
    }
    TextureHandle createTexture(const TextureDesc& d)
    {
        trhip_texture_desc n{};
        n.width = d.width; n.height = d.height; n.mipLevels = d.mipLevels; n.isUAV = d.isUAV; n.isVirtual = d.isVirtual; n.debugName = d.debugName.c_str();
        // depth (D24S8 in the reference, GraphicConstants.h:26) is carried as 32-bit float depth here
        n.format = d.format == Format::R16_FLOAT ? TRHIP_FORMAT_R16_FLOAT : TRHIP_FORMAT_R32_FLOAT;
        trhip_texture t = nullptr;
        throwIfFailed(trhip_texture_create(m_Native, &n, &t), "IDevice::createTexture");
        return TextureHandle(new ITexture(t, d));
    }
    SamplerHandle createSampler(const SamplerDesc& d) { return SamplerHandle(new ISampler(d)); }
    MemoryRequirements getBufferMemoryRequirements(IBuffer* b)
    {
        MemoryRequirements r;
        throwIfFailed(trhip_buffer_memory_requirements(b->native(), &r.size, &r.alignment), "IDevice::getBufferMemoryRequirements");
        return r;
    }
    MemoryRequirements getTextureMemoryRequirements(ITexture* t)
    {
        MemoryRequirements r;
        throwIfFailed(trhip_texture_memory_requirements(t->native(), &r.size, &r.alignment), "IDevice::getTextureMemoryRequirements");
        return r;
    }
    bool bindBufferMemory(IBuffer* b, IHeap* h, uint64_t off) { throwIfFailed(trhip_buffer_bind_memory(b->native(), h->native(), off), "IDevice::bindBufferMemory"); return true; }
    bool bindTextureMemory(ITexture* t, IHeap* h, uint64_t off) { throwIfFailed(trhip_texture_bind_memory(t->native(), h->native(), off), "IDevice::bindTextureMemory"); return true; }
    // nvrhi::CommandListParameters::queueType.  The COMPUTE queue (the reference never creates one: GraphicRHI.cpp:152-177
    // has a single graphics queue, RenderGraph.cpp:251 says "TODO: compute queue") is a second back-end device on the
    // same GPU with a stream of its own, created on first use.
    CommandListHandle createCommandList(CommandQueue queue = CommandQueue::Graphics)
    {
        trhip_cmdlist cl = nullptr;
        throwIfFailed(trhip_cmd_create(queueDevice(queue), &cl), "IDevice::createCommandList");
        return CommandListHandle(new ICommandList(cl, queue));
    }
    // nvrhi executeCommandLists: returns the instance id of the submission on that queue
    uint64_t executeCommandLists(ICommandList* const* lists, size_t n, CommandQueue queue = CommandQueue::Graphics)
    {
        std::vector<trhip_cmdlist> v;
        for (size_t i = 0; i < n; ++i) { check(lists[i]->m_Queue == queue); v.push_back(lists[i]->native()); }
        throwIfFailed(trhip_queue_execute(queueDevice(queue), v.data(), (uint32_t)v.size()), "IDevice::executeCommandLists");
        return ++m_LastInstance[(size_t)queue];
    }
    uint64_t executeCommandList(ICommandList* cl) { return executeCommandLists(&cl, 1, cl->m_Queue); }
    // nvrhi queueWaitForCommandList: everything submitted to `waitQueue` from now on runs after submission `instance` of
    // `executionQueue` (which must be its latest: the event is recorded here, behind the producer's internal side stream)
    void queueWaitForCommandList(CommandQueue waitQueue, CommandQueue executionQueue, uint64_t instance)
    {
        if (waitQueue == executionQueue) return;
        check(instance == m_LastInstance[(size_t)executionQueue]);
        trhip_device producer = queueDevice(executionQueue);
        void*& ev = m_QueueEvents[(size_t)executionQueue][instance % kQueueEvents];
        if (!ev) throwIfFailed(trhip_event_create(m_DeviceIndex, &ev), "IDevice::queueWaitForCommandList: event");
        throwIfFailed(trhip_device_join_side_stream(producer), "IDevice::queueWaitForCommandList: join");
        throwIfFailed(trhip_event_record(ev, trhip_device_stream(producer)), "IDevice::queueWaitForCommandList: record");
        throwIfFailed(trhip_stream_wait_event(trhip_device_stream(queueDevice(waitQueue)), ev), "IDevice::queueWaitForCommandList: wait");
    }
    uint64_t lastInstance(CommandQueue queue) const { return m_LastInstance[(size_t)queue]; }
    bool hasComputeQueue() const { return m_ComputeNative != nullptr; }
    void setDeviceIndex(int index) { m_DeviceIndex = index; }
    void waitForIdle()
    {
        throwIfFailed(trhip_device_wait_idle(m_Native), "IDevice::waitForIdle");
        if (m_ComputeNative) throwIfFailed(trhip_device_wait_idle(m_ComputeNative), "IDevice::waitForIdle (compute queue)");
    }
    void destroyQueues()
    {
        for (auto& q : m_QueueEvents) for (void*& e : q) { if (e) trhip_event_destroy(e); e = nullptr; }
        if (m_ComputeNative) { trhip_device_destroy(m_ComputeNative); m_ComputeNative = nullptr; }
        if (m_ComputeStream) { trhip_stream_destroy(m_ComputeStream); m_ComputeStream = nullptr; }
    }
    void runGarbageCollection() {}
    TimerQueryHandle createTimerQuery()
    {
        trhip_timer t = nullptr;
        throwIfFailed(trhip_timer_create(m_Native, &t), "IDevice::createTimerQuery");
        return TimerQueryHandle(new ITimerQuery(t));
    }
    float getTimerQueryTime(ITimerQuery* q)   // seconds, like nvrhi
    {
        if (!q || !q->m_Recorded) return 0.0f;
        float ms = 0.0f;
        if (trhip_timer_get_ms(q->native(), &ms) != TRHIP_OK) return 0.0f;
        return ms * 1e-3f;
    }
    void resetTimerQuery(ITimerQuery* q) { if (q) q->m_Recorded = false; }

private:
    static constexpr size_t kQueueEvents = 16;
    trhip_device m_Native;
    trhip_device m_ComputeNative = nullptr;
    void* m_ComputeStream = nullptr;
    int m_DeviceIndex = 0;
    uint64_t m_LastInstance[(size_t)CommandQueue::Count] = {};
    void* m_QueueEvents[(size_t)CommandQueue::Count][kQueueEvents] = {};
};
using DeviceHandle = RefCountPtr<IDevice>;

} // namespace nvrhi
