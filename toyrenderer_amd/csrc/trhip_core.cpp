// trhip_core.cpp -- device, memory, command-list and queue half of the C ABI (include/trhip.h).
// Replaces GraphicRHI.cpp (device/queue) and the nvrhi::ICommandList / IDevice calls the
// reference makes on the visibility path.  Kernels live in k_*.hip and register themselves by
// the reference's shader-name strings.
#include "trhip_internal.h"

#include <chrono>
#include <cstdlib>

#include <algorithm>
#include <memory>

namespace trhip
{

static thread_local char tl_error[1024] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tl_error, sizeof tl_error, fmt, ap);
    va_end(ap);
    return code;
}

int hipfail(hipError_t e, const char* what)
{
    return fail(TRHIP_ERR_HIP, "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
}

struct ShaderEntry { std::string name; RecordFn fn; int variant; };
static std::vector<ShaderEntry>& shaders()
{
    static std::vector<ShaderEntry> s;
    return s;
}

void registerShader(const char* name, RecordFn fn, int variant)
{
    shaders().push_back({ name, fn, variant });
}

static const ShaderEntry* findShader(const char* name)
{
    for (const ShaderEntry& e : shaders())
        if (e.name == name) return &e;
    return nullptr;
}

trhip_buffer_t* DispatchCtx::buffer(uint32_t type, uint32_t slot) const
{
    for (uint32_t i = 0; i < numBindings; ++i)
        if (bindings[i].type == type && bindings[i].slot == slot)
            return (trhip_buffer_t*)bindings[i].resource;
    return nullptr;
}

trhip_texture_t* DispatchCtx::texture(uint32_t type, uint32_t slot, uint32_t* baseMip) const
{
    for (uint32_t i = 0; i < numBindings; ++i)
        if (bindings[i].type == type && bindings[i].slot == slot) {
            if (baseMip) *baseMip = bindings[i].baseMip;
            return (trhip_texture_t*)bindings[i].resource;
        }
    return nullptr;
}

const void* DispatchCtx::constants(uint32_t slot, size_t bytes) const
{
    for (uint32_t i = 0; i < numBindings; ++i) {
        if (bindings[i].slot != slot) continue;
        if (bindings[i].type == TRHIP_BIND_CONSTANT_BUFFER) {
            trhip_buffer_t* b = (trhip_buffer_t*)bindings[i].resource;
            if (!b || !b->isVolatileConstant || b->shadow.size() < bytes) return nullptr;
            return b->shadow.data();
        }
        if (bindings[i].type == TRHIP_BIND_PUSH_CONSTANTS) {
            if (!push || pushBytes < bytes) return nullptr;
            return push;
        }
    }
    return nullptr;
}

void DispatchCtx::emit(const char* kernelName, std::function<int(hipStream_t)> fn) const
{
    std::string n = std::string(shaderName) + "#" + kernelName;
    cl->ops.push_back({ std::move(n), std::move(fn) });
}

void DispatchCtx::emitSide(const char* kernelName, std::function<int(hipStream_t)> fn, std::initializer_list<Op::Access> touched) const
{
    emit(kernelName, std::move(fn));
    if (!cl->dev->sideStream) return;                  // no side stream: plain in-order op
    cl->ops.back().lane = 1;
    for (const Op::Access& t : touched) if (t.ptr) cl->ops.back().touched.push_back(t);
}

void DispatchCtx::emitSideHeld(const char* kernelName, std::function<int(hipStream_t)> fn, std::initializer_list<Op::Access> touched) const
{
    if (!cl->dev->sideStream) { emit(kernelName, std::move(fn)); return; }
    Op op{ std::string(shaderName) + "#" + kernelName, std::move(fn) };
    op.lane = 1;
    for (const Op::Access& t : touched) if (t.ptr) op.touched.push_back(t);
    cl->heldSide.push_back(std::move(op));
}

} // namespace trhip

using namespace trhip;

namespace
{
// TRHIP_HOST_PROFILE=1: host microseconds spent issuing each kind of command, printed at process exit
// (diagnostics for the submission cost of a frame; off by default).
struct HostProfile
{
    bool on = getenv("TRHIP_HOST_PROFILE") != nullptr;
    std::mutex m;
    std::map<std::string, std::pair<uint64_t, double>> acc;
    void add(const char* kind, double us)
    {
        std::lock_guard<std::mutex> lk(m);
        auto& e = acc[kind];
        e.first++; e.second += us;
    }
    ~HostProfile()
    {
        if (!on) return;
        for (auto& kv : acc)
            fprintf(stderr, "[trhip host profile] %-14s calls %8llu  total %10.1f us  avg %6.2f us\n", kv.first.c_str(),
                    (unsigned long long)kv.second.first, kv.second.second, kv.second.second / (double)kv.second.first);
    }
} g_hostProfile;
}

// ------------------------------------------------------------------------------------------------
thread_local trhip::LaunchTap* trhip::g_launchTap = nullptr;

hipEvent_t trhip_device_t::acquireEvent()
{
    if (!eventPool.empty()) { hipEvent_t e = eventPool.back(); eventPool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

int trhip_device_t::drainProfile()
{
    for (ProfilePending& p : pending) {
        float ms = 0;
        TRHIP_HIP(hipEventSynchronize(p.e1));
        TRHIP_HIP(hipEventElapsedTime(&ms, p.e0, p.e1));
        auto it = accum.find(p.name);
        if (it == accum.end()) { accumOrder.push_back(p.name); it = accum.emplace(p.name, ProfileAccum{}).first; }
        it->second.launches += 1;
        it->second.totalMs += ms;
        eventPool.push_back(p.e0);
        eventPool.push_back(p.e1);
    }
    pending.clear();
    return TRHIP_OK;
}

static void* arenaAlloc(std::vector<trhip_cmdlist_t::ScratchBlock>& arena, int deviceIndex, size_t bytes)
{
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    for (trhip_cmdlist_t::ScratchBlock& b : arena)
        if (b.bytes - b.used >= bytes) { void* p = (char*)b.ptr + b.used; b.used += bytes; return p; }
    size_t blockBytes = std::max(bytes, size_t(1) << 20);
    void* p = nullptr;
    if (hipSetDevice(deviceIndex) != hipSuccess || hipMalloc(&p, blockBytes) != hipSuccess) return nullptr;
    arena.push_back({ p, blockBytes, bytes });
    return p;
}

void* trhip_cmdlist_t::scratchAlloc(size_t bytes) { return arenaAlloc(scratch, dev->index, bytes); }
void* trhip_cmdlist_t::scratchAllocSide(size_t bytes) { return arenaAlloc(sideScratch, dev->index, bytes); }

int trhip_device_t::syncAll()
{
    TRHIP_HIP(hipStreamSynchronize(stream));
    if (sideStream) TRHIP_HIP(hipStreamSynchronize(sideStream));
    return TRHIP_OK;
}

void trhip_cmdlist_t::resetRecording()
{
    heldSide.clear();
    ops.clear();
    for (trhip_buffer_t* b : heldBuffers) trhip_buffer_release(b);
    for (trhip_texture_t* t : heldTextures) trhip_texture_release(t);
    heldBuffers.clear();
    heldTextures.clear();
    markers.clear();
    useMarks.clear();
    openClearBatch.reset();
    openClearOp = SIZE_MAX;
    firstClearBatch.reset();
    firstClearOp = SIZE_MAX;
    clearedTo.clear();
    peephole = Peephole();
    for (ScratchBlock& b : scratch) b.used = 0;
    for (ScratchBlock& b : sideScratch) b.used = 0;
}

namespace
{
struct ClearKernelArgs { void* ptr[trhip_cmdlist_t::ClearBatch::kMax]; uint64_t words[trhip_cmdlist_t::ClearBatch::kMax]; uint32_t value[trhip_cmdlist_t::ClearBatch::kMax]; uint32_t count; };

__global__ __launch_bounds__(256) void multiClearKernel(ClearKernelArgs a)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint32_t k = 0; k < a.count; ++k) {
        uint32_t* p = (uint32_t*)a.ptr[k];
        const uint32_t v = a.value[k];
        for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < a.words[k]; i += stride) p[i] = v;
    }
}
}

int trhip_cmdlist_t::recordClearWords(void* ptr, uint64_t words, uint32_t value, bool fresh)
{
    if (words == 0) return TRHIP_OK;
    if (fresh && firstClearBatch && firstClearBatch->count < ClearBatch::kMax) {
        ClearBatch& f = *firstClearBatch;
        f.ptr[f.count] = ptr; f.words[f.count] = words; f.value[f.count] = value;
        ++f.count;
        return TRHIP_OK;
    }
    if (!(openClearBatch && openClearOp == ops.size() - 1 && openClearBatch->count < ClearBatch::kMax)) {
        openClearBatch = std::make_shared<ClearBatch>();
        std::shared_ptr<ClearBatch> b = openClearBatch;
        const uint32_t cus = dev->computeUnits;
        ops.push_back({ "", [b, cus](hipStream_t s) {
            ClearKernelArgs a;
            uint64_t most = 0;
            a.count = b->count;
            for (uint32_t k = 0; k < b->count; ++k) { a.ptr[k] = b->ptr[k]; a.words[k] = b->words[k]; a.value[k] = b->value[k]; most = std::max(most, b->words[k]); }
            uint64_t grid = (most + 1023u) / 1024u;                      // ~4 words per thread for the largest range
            if (grid > (uint64_t)cus * 8u) grid = (uint64_t)cus * 8u;
            if (grid == 0) grid = 1;
            TRHIP_LAUNCH(multiClearKernel, dim3((uint32_t)grid), dim3(256), 0, s, a);
            return trhip::launchStatus("multiClearKernel"); } });
        ops.back().kind = "clear_buffer";
        openClearOp = ops.size() - 1;
        if (!firstClearBatch) { firstClearBatch = openClearBatch; firstClearOp = openClearOp; }
    }
    ClearBatch& b = *openClearBatch;
    b.ptr[b.count] = ptr; b.words[b.count] = words; b.value[b.count] = value;
    ++b.count;
    return TRHIP_OK;
}

// Every command holds the resources it uses before it is recorded: the one place that notes WHICH command
// uses WHAT, for ordering against side-stream work on the same memory at execute time.
void trhip_cmdlist_t::hold(trhip_buffer_t* b, bool write, size_t op) { if (b) { use(b->ptr, op == SIZE_MAX ? ops.size() : op, write, &b->version); trhip_buffer_retain(b); heldBuffers.push_back(b); } }
void trhip_cmdlist_t::hold(trhip_texture_t* t, bool write) { if (t) { use(t->ptr, ops.size(), write, &t->version); trhip_texture_retain(t); heldTextures.push_back(t); } }

// ------------------------------------------------------------------------------------------------
extern "C" {

const char* trhip_last_error(void) { return tl_error; }
uint32_t trhip_abi_version(void) { return TRHIP_ABI_VERSION; }
uint32_t trhip_shader_count(void) { return (uint32_t)shaders().size(); }
const char* trhip_shader_name(uint32_t i) { return i < shaders().size() ? shaders()[i].name.c_str() : nullptr; }
int trhip_shader_exists(const char* name) { return name && findShader(name) ? 1 : 0; }

static int deviceCreate(int index, void* stream, bool external, trhip_device* out)
{
    if (!out) return fail(TRHIP_ERR_INVALID, "out is null");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(TRHIP_ERR_NO_DEVICE, "no HIP device visible (this back end has no CPU fallback)");
    if (index < 0 || index >= n) return fail(TRHIP_ERR_INVALID, "device index %d out of range (%d devices)", index, n);
    TRHIP_HIP(hipSetDevice(index));
    hipDeviceProp_t props;
    TRHIP_HIP(hipGetDeviceProperties(&props, index));
    auto dev = std::make_unique<trhip_device_t>();
    dev->index = index;
    dev->computeUnits = (uint32_t)props.multiProcessorCount;
    dev->waveSize = (uint32_t)props.warpSize;
    dev->totalMem = (uint64_t)props.totalGlobalMem;
    if (dev->waveSize != 64)
        return fail(TRHIP_ERR_INVALID, "wave size %u: the kernels are written for wave64 (gfx950)", dev->waveSize);
    if (external) {
        dev->stream = (hipStream_t)stream;
        dev->ownsStream = false;
    } else {
        TRHIP_HIP(hipStreamCreateWithFlags(&dev->stream, hipStreamNonBlocking));
        dev->ownsStream = true;
    }
    if (!getenv("TRHIP_NO_SIDE_STREAM")) {             // see trhip_device_t::sideStream
        // High priority: HIP multiplexes the streams of one priority class onto a few hardware queues (4 by default)
        // round robin; in a process with many streams (torch, RCCL) the side stream otherwise lands on the SAME
        // hardware queue as the main stream sooner or later and its kernels serialise with it (seen at 2-8 ranks:
        // +0.06..0.16 ms per frame).  A different priority class is a different queue.
        int lowest = 0, highest = 0;
        TRHIP_HIP(hipDeviceGetStreamPriorityRange(&lowest, &highest));
        int sidePriority = highest;
        if (const char* e = getenv("TRHIP_SIDE_PRIORITY")) sidePriority = atoi(e) < 0 ? highest : atoi(e) > 0 ? lowest : 0;   // experiments
        TRHIP_HIP(hipStreamCreateWithPriority(&dev->sideStream, hipStreamNonBlocking, sidePriority));
        TRHIP_HIP(hipEventCreateWithFlags(&dev->evFork, hipEventDisableTiming));
        for (hipEvent_t& e : dev->runDone) TRHIP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    *out = dev.release();
    return TRHIP_OK;
}

int trhip_device_create(int device_index, trhip_device* out) { return deviceCreate(device_index, nullptr, false, out); }
int trhip_device_create_on_stream(int device_index, void* s, trhip_device* out) { return deviceCreate(device_index, s, true, out); }

void trhip_device_destroy(trhip_device dev)
{
    if (!dev) return;
    (void)hipSetDevice(dev->index);
    (void)dev->syncAll();
    (void)dev->drainProfile();
    for (hipEvent_t e : dev->eventPool) (void)hipEventDestroy(e);
    if (dev->sideStream) { (void)hipStreamSynchronize(dev->sideStream); (void)hipStreamDestroy(dev->sideStream); }
    if (dev->evFork) (void)hipEventDestroy(dev->evFork);
    for (hipEvent_t e : dev->runDone) if (e) (void)hipEventDestroy(e);
    if (dev->ownsStream) (void)hipStreamDestroy(dev->stream);
    delete dev;
}

int trhip_device_wait_idle(trhip_device dev)
{
    if (!dev) return fail(TRHIP_ERR_INVALID, "device is null");
    TRHIP_HIP(hipSetDevice(dev->index));
    return dev->syncAll();
}

int trhip_device_join_side_stream(trhip_device dev)
{
    if (!dev) return fail(TRHIP_ERR_INVALID, "device is null");
    TRHIP_HIP(hipSetDevice(dev->index));
    std::lock_guard<std::mutex> lock(dev->mutex);
    if (dev->sideStream && dev->sideRunCounter > dev->mainWaitedUpTo) {
        TRHIP_HIP(hipStreamWaitEvent(dev->stream, dev->runDone[dev->sideRunCounter % trhip_device_t::kSideRuns], 0));
        dev->mainWaitedUpTo = dev->sideRunCounter;      // the side stream is in order: earlier runs are covered too
    }
    return TRHIP_OK;
}

int trhip_device_info(trhip_device dev, uint32_t* cus, uint32_t* wave, uint64_t* mem)
{
    if (!dev) return fail(TRHIP_ERR_INVALID, "device is null");
    if (cus) *cus = dev->computeUnits;
    if (wave) *wave = dev->waveSize;
    if (mem) *mem = dev->totalMem;
    return TRHIP_OK;
}

void* trhip_device_stream(trhip_device dev) { return dev ? (void*)dev->stream : nullptr; }

// ---- memory --------------------------------------------------------------------------------------
int trhip_heap_create(trhip_device dev, uint64_t bytes, trhip_heap* out)
{
    if (!dev || !out || bytes == 0) return fail(TRHIP_ERR_INVALID, "heap_create: bad arguments");
    TRHIP_HIP(hipSetDevice(dev->index));
    auto h = std::make_unique<trhip_heap_t>();
    h->dev = dev;
    h->bytes = bytes;
    TRHIP_HIP(hipMalloc(&h->base, bytes));
    *out = h.release();
    return TRHIP_OK;
}

static void heapRetain(trhip_heap_t* h) { if (h) h->rc.fetch_add(1); }

void trhip_heap_release(trhip_heap h)
{
    if (!h) return;
    if (h->rc.fetch_sub(1) == 1) {
        (void)hipSetDevice(h->dev->index);
        (void)hipFree(h->base);
        delete h;
    }
}

static void fillBuffer(trhip_buffer_t* b, trhip_device dev, const trhip_buffer_desc* d)
{
    b->dev = dev;
    b->byteSize = d->byteSize;
    b->structStride = d->structStride;
    b->canHaveUAVs = d->canHaveUAVs != 0;
    b->isDrawIndirectArgs = d->isDrawIndirectArgs != 0;
    b->isVirtual = d->isVirtual != 0;
    b->isVolatileConstant = d->isVolatileConstant != 0;
    b->name = d->debugName ? d->debugName : "";
}

int trhip_buffer_create(trhip_device dev, const trhip_buffer_desc* desc, trhip_buffer* out)
{
    if (!dev || !desc || !out) return fail(TRHIP_ERR_INVALID, "buffer_create: null argument");
    if (desc->byteSize == 0) return fail(TRHIP_ERR_INVALID, "buffer_create(%s): byteSize is 0", desc->debugName ? desc->debugName : "");
    auto b = std::make_unique<trhip_buffer_t>();
    fillBuffer(b.get(), dev, desc);
    if (b->isVolatileConstant) {
        b->shadow.assign((size_t)desc->byteSize, 0); // lives in kernel arguments, never in HBM
    } else if (!b->isVirtual) {
        TRHIP_HIP(hipSetDevice(dev->index));
        TRHIP_HIP(hipMalloc(&b->ptr, (size_t)((desc->byteSize + 255) & ~uint64_t(255))));
        b->owns = true;
    }
    *out = b.release();
    return TRHIP_OK;
}

int trhip_buffer_wrap(trhip_device dev, void* ptr, const trhip_buffer_desc* desc, trhip_buffer* out)
{
    if (!dev || !desc || !out || !ptr) return fail(TRHIP_ERR_INVALID, "buffer_wrap: null argument");
    if (desc->byteSize == 0) return fail(TRHIP_ERR_INVALID, "buffer_wrap: byteSize is 0");
    auto b = std::make_unique<trhip_buffer_t>();
    fillBuffer(b.get(), dev, desc);
    b->isVirtual = false;
    b->ptr = ptr;
    b->owns = false;
    *out = b.release();
    return TRHIP_OK;
}

int trhip_buffer_memory_requirements(trhip_buffer buf, uint64_t* size, uint64_t* alignment)
{
    if (!buf) return fail(TRHIP_ERR_INVALID, "buffer is null");
    if (size) *size = (buf->byteSize + 255) & ~uint64_t(255);
    if (alignment) *alignment = 256;
    return TRHIP_OK;
}

int trhip_buffer_bind_memory(trhip_buffer buf, trhip_heap heap, uint64_t offset)
{
    if (!buf || !heap) return fail(TRHIP_ERR_INVALID, "bind_memory: null argument");
    if (!buf->isVirtual) return fail(TRHIP_ERR_STATE, "bind_memory(%s): buffer is not virtual", buf->name.c_str());
    if (offset % 256) return fail(TRHIP_ERR_INVALID, "bind_memory(%s): offset %llu not 256-byte aligned", buf->name.c_str(), (unsigned long long)offset);
    if (offset + buf->byteSize > heap->bytes) return fail(TRHIP_ERR_INVALID, "bind_memory(%s): range exceeds the heap", buf->name.c_str());
    if (buf->heap) trhip_heap_release(buf->heap);
    heapRetain(heap);
    buf->heap = heap;
    buf->ptr = (char*)heap->base + offset;
    buf->version.fetch_add(1);                      // other memory = other contents: derived data (cull cache, meshlet cull stream) is stale
    return TRHIP_OK;
}

void trhip_buffer_retain(trhip_buffer b) { if (b) b->rc.fetch_add(1); }

void trhip_buffer_release(trhip_buffer b)
{
    if (!b) return;
    if (b->rc.fetch_sub(1) == 1) {
        if (b->owns && b->ptr) { (void)hipSetDevice(b->dev->index); (void)hipFree(b->ptr); }
        if (b->sidecar) { (void)hipSetDevice(b->dev->index); (void)hipFree(b->sidecar); }
        if (b->cullCache) { (void)hipSetDevice(b->dev->index); (void)b->dev->syncAll(); (void)hipFree(b->cullCache); }
        if (b->cullStream) { (void)hipSetDevice(b->dev->index); (void)b->dev->syncAll(); (void)hipFree(b->cullStream); }
        if (b->heap) trhip_heap_release(b->heap);
        delete b;
    }
}

void* trhip_buffer_device_ptr(trhip_buffer b) { return b ? b->ptr : nullptr; }
uint64_t trhip_buffer_size(trhip_buffer b) { return b ? b->byteSize : 0; }

int trhip_texture_create(trhip_device dev, const trhip_texture_desc* d, trhip_texture* out)
{
    if (!dev || !d || !out) return fail(TRHIP_ERR_INVALID, "texture_create: null argument");
    if (d->width == 0 || d->height == 0 || d->mipLevels == 0 || d->mipLevels > 16)
        return fail(TRHIP_ERR_INVALID, "texture_create: bad dimensions %ux%u mips %u", d->width, d->height, d->mipLevels);
    if (d->format != TRHIP_FORMAT_R16_FLOAT && d->format != TRHIP_FORMAT_R32_FLOAT)
        return fail(TRHIP_ERR_INVALID, "texture_create: unsupported format %u", d->format);
    auto t = std::make_unique<trhip_texture_t>();
    t->dev = dev;
    t->width = d->width; t->height = d->height; t->mips = d->mipLevels; t->format = d->format;
    t->texelBytes = d->format == TRHIP_FORMAT_R16_FLOAT ? 2 : 4;
    t->isUAV = d->isUAV != 0; t->isVirtual = d->isVirtual != 0;
    t->name = d->debugName ? d->debugName : "";
    uint64_t off = 0;
    for (uint32_t k = 0; k < t->mips; ++k) {
        t->mipOffset[k] = off;
        off += ((uint64_t)t->mipW(k) * t->mipH(k) * t->texelBytes + 255) & ~uint64_t(255);
    }
    t->totalBytes = off;
    if (!t->isVirtual) {
        TRHIP_HIP(hipSetDevice(dev->index));
        TRHIP_HIP(hipMalloc(&t->ptr, (size_t)off));
        t->owns = true;
    }
    *out = t.release();
    return TRHIP_OK;
}

int trhip_texture_memory_requirements(trhip_texture t, uint64_t* size, uint64_t* alignment)
{
    if (!t) return fail(TRHIP_ERR_INVALID, "texture is null");
    if (size) *size = t->totalBytes;
    if (alignment) *alignment = 256;
    return TRHIP_OK;
}

int trhip_texture_bind_memory(trhip_texture t, trhip_heap heap, uint64_t offset)
{
    if (!t || !heap) return fail(TRHIP_ERR_INVALID, "bind_memory: null argument");
    if (!t->isVirtual) return fail(TRHIP_ERR_STATE, "bind_memory(%s): texture is not virtual", t->name.c_str());
    if (offset % 256 || offset + t->totalBytes > heap->bytes) return fail(TRHIP_ERR_INVALID, "bind_memory(%s): bad range", t->name.c_str());
    if (t->heap) trhip_heap_release(t->heap);
    heapRetain(heap);
    t->heap = heap;
    t->ptr = (char*)heap->base + offset;
    t->version.fetch_add(1);                        // other memory = other contents: the footprint-min table is stale
    return TRHIP_OK;
}

void trhip_texture_retain(trhip_texture t) { if (t) t->rc.fetch_add(1); }

void trhip_texture_release(trhip_texture t)
{
    if (!t) return;
    if (t->rc.fetch_sub(1) == 1) {
        if (t->owns && t->ptr) { (void)hipSetDevice(t->dev->index); (void)hipFree(t->ptr); }
        if (t->quad) { (void)hipSetDevice(t->dev->index); (void)t->dev->syncAll(); (void)hipFree(t->quad); }
        if (t->heap) trhip_heap_release(t->heap);
        delete t;
    }
}

void* trhip_texture_device_ptr(trhip_texture t) { return t ? t->ptr : nullptr; }
uint64_t trhip_texture_size(trhip_texture t) { return t ? t->totalBytes : 0; }

int trhip_texture_mip_info(trhip_texture t, uint32_t mip, uint32_t* w, uint32_t* h, uint64_t* off)
{
    if (!t || mip >= t->mips) return fail(TRHIP_ERR_INVALID, "mip_info: bad texture/mip");
    if (w) *w = t->mipW(mip);
    if (h) *h = t->mipH(mip);
    if (off) *off = t->mipOffset[mip];
    return TRHIP_OK;
}

static int syncCopy(trhip_device_t* dev, void* dst, const void* src, uint64_t bytes, hipMemcpyKind kind)
{
    TRHIP_HIP(hipSetDevice(dev->index));
    int rc = dev->syncAll();
    if (rc != TRHIP_OK) return rc;
    TRHIP_HIP(hipMemcpy(dst, src, (size_t)bytes, kind));
    return TRHIP_OK;
}

int trhip_buffer_upload(trhip_buffer b, uint64_t off, const void* src, uint64_t bytes)
{
    if (!b || !src) return fail(TRHIP_ERR_INVALID, "buffer_upload: null argument");
    if (!b->ptr) return fail(TRHIP_ERR_STATE, "buffer_upload(%s): no memory bound", b->name.c_str());
    if (off + bytes > b->byteSize) return fail(TRHIP_ERR_INVALID, "buffer_upload(%s): range exceeds the buffer", b->name.c_str());
    int rc = syncCopy(b->dev, (char*)b->ptr + off, src, bytes, hipMemcpyHostToDevice);
    b->version.fetch_add(1);
    return rc;
}

int trhip_buffer_mark_written(trhip_buffer b)
{
    if (!b) return fail(TRHIP_ERR_INVALID, "buffer_mark_written: null argument");
    b->version.fetch_add(1);                   // derived data (instance cull cache, meshlet cull stream) is rebuilt by the next pass that wants it
    return TRHIP_OK;
}

int trhip_texture_mark_written(trhip_texture t)
{
    if (!t) return fail(TRHIP_ERR_INVALID, "texture_mark_written: null argument");
    t->version.fetch_add(1);                   // the footprint-min table is rebuilt by the next pass that wants it
    return TRHIP_OK;
}

int trhip_buffer_download(trhip_buffer b, uint64_t off, void* dst, uint64_t bytes)
{
    if (!b || !dst) return fail(TRHIP_ERR_INVALID, "buffer_download: null argument");
    if (!b->ptr) return fail(TRHIP_ERR_STATE, "buffer_download(%s): no memory bound", b->name.c_str());
    if (off + bytes > b->byteSize) return fail(TRHIP_ERR_INVALID, "buffer_download(%s): range exceeds the buffer", b->name.c_str());
    return syncCopy(b->dev, dst, (char*)b->ptr + off, bytes, hipMemcpyDeviceToHost);
}

int trhip_texture_upload(trhip_texture t, uint32_t mip, const void* src, uint64_t bytes)
{
    if (!t || !src || mip >= t->mips) return fail(TRHIP_ERR_INVALID, "texture_upload: bad argument");
    if (!t->ptr) return fail(TRHIP_ERR_STATE, "texture_upload(%s): no memory bound", t->name.c_str());
    uint64_t need = (uint64_t)t->mipW(mip) * t->mipH(mip) * t->texelBytes;
    if (bytes != need) return fail(TRHIP_ERR_INVALID, "texture_upload(%s): mip %u is %llu bytes, got %llu", t->name.c_str(), mip, (unsigned long long)need, (unsigned long long)bytes);
    int rc = syncCopy(t->dev, t->mipPtr(mip), src, bytes, hipMemcpyHostToDevice);
    t->version.fetch_add(1);                   // everything submitted has completed: later submissions see the new contents
    return rc;
}

int trhip_texture_download(trhip_texture t, uint32_t mip, void* dst, uint64_t bytes)
{
    if (!t || !dst || mip >= t->mips) return fail(TRHIP_ERR_INVALID, "texture_download: bad argument");
    if (!t->ptr) return fail(TRHIP_ERR_STATE, "texture_download(%s): no memory bound", t->name.c_str());
    uint64_t need = (uint64_t)t->mipW(mip) * t->mipH(mip) * t->texelBytes;
    if (bytes != need) return fail(TRHIP_ERR_INVALID, "texture_download(%s): mip %u is %llu bytes, got %llu", t->name.c_str(), mip, (unsigned long long)need, (unsigned long long)bytes);
    return syncCopy(t->dev, dst, t->mipPtr(mip), bytes, hipMemcpyDeviceToHost);
}

// ---- command lists -------------------------------------------------------------------------------
int trhip_cmd_create(trhip_device dev, trhip_cmdlist* out)
{
    if (!dev || !out) return fail(TRHIP_ERR_INVALID, "cmd_create: null argument");
    auto cl = std::make_unique<trhip_cmdlist_t>();
    cl->dev = dev;
    *out = cl.release();
    return TRHIP_OK;
}

void trhip_cmd_release(trhip_cmdlist cl)
{
    if (!cl) return;
    (void)hipSetDevice(cl->dev->index);
    (void)cl->dev->syncAll();                    // recorded ops may still reference scratch
    cl->resetRecording();
    for (auto& b : cl->scratch) (void)hipFree(b.ptr);
    for (auto& b : cl->sideScratch) (void)hipFree(b.ptr);
    delete cl;
}

int trhip_cmd_open(trhip_cmdlist cl)
{
    if (!cl) return fail(TRHIP_ERR_INVALID, "cmdlist is null");
    if (cl->open) return fail(TRHIP_ERR_STATE, "cmd_open: already open");
    cl->resetRecording();
    cl->open = true;
    return TRHIP_OK;
}

int trhip_cmd_close(trhip_cmdlist cl)
{
    if (!cl) return fail(TRHIP_ERR_INVALID, "cmdlist is null");
    if (!cl->open) return fail(TRHIP_ERR_STATE, "cmd_close: not open");
    if (!cl->markers.empty()) return fail(TRHIP_ERR_STATE, "cmd_close: %zu marker(s) still open", cl->markers.size());
    cl->flushHeldSide();
    cl->open = false;
    // by command index (a clear merged into an earlier launch is noted against that launch)
    std::stable_sort(cl->useMarks.begin(), cl->useMarks.end(), [](const trhip_cmdlist_t::UseMark& a, const trhip_cmdlist_t::UseMark& b) { return a.op < b.op; });
    return TRHIP_OK;
}

#define TRHIP_RECORDING(cl)                                                                  \
    do {                                                                                     \
        if (!(cl)) return fail(TRHIP_ERR_INVALID, "cmdlist is null");                        \
        if (!(cl)->open) return fail(TRHIP_ERR_STATE, "command list is not open");           \
    } while (0)

int trhip_cmd_write_buffer(trhip_cmdlist cl, trhip_buffer buf, uint64_t off, const void* src, uint64_t bytes)
{
    TRHIP_RECORDING(cl);
    if (buf && cl->heldConflicts(buf->ptr, true)) cl->flushHeldSide();   // (a volatile constant buffer has no memory of its own: never a conflict)
    if (!buf || !src) return fail(TRHIP_ERR_INVALID, "write_buffer: null argument");
    if (off + bytes > buf->byteSize) return fail(TRHIP_ERR_INVALID, "write_buffer(%s): %llu+%llu exceeds %llu bytes", buf->name.c_str(), (unsigned long long)off, (unsigned long long)bytes, (unsigned long long)buf->byteSize);
    if (buf->isVolatileConstant) {
        memcpy(buf->shadow.data() + off, src, (size_t)bytes); // version seen by later dispatches of this list
        return TRHIP_OK;
    }
    if (!buf->ptr) return fail(TRHIP_ERR_STATE, "write_buffer(%s): no memory bound", buf->name.c_str());
    auto staged = std::make_shared<std::vector<uint8_t>>((const uint8_t*)src, (const uint8_t*)src + bytes);
    void* dst = (char*)buf->ptr + off;
    cl->hold(buf, true);
    cl->ops.push_back({ "", [staged, dst](hipStream_t s) {
        TRHIP_HIP(hipMemcpyAsync(dst, staged->data(), staged->size(), hipMemcpyHostToDevice, s));
        return (int)TRHIP_OK; } });
    cl->ops.back().kind = "write_buffer";
    return TRHIP_OK;
}

int trhip_cmd_clear_buffer_u32(trhip_cmdlist cl, trhip_buffer buf, uint32_t value)
{
    TRHIP_RECORDING(cl);
    if (buf && cl->heldConflicts(buf->ptr, true)) cl->flushHeldSide();
    if (!buf) return fail(TRHIP_ERR_INVALID, "clear_buffer: null buffer");
    if (!buf->ptr) return fail(TRHIP_ERR_STATE, "clear_buffer(%s): no memory bound", buf->name.c_str());
    if (buf->byteSize % 4) return fail(TRHIP_ERR_INVALID, "clear_buffer(%s): size not a multiple of 4", buf->name.c_str());
    if (cl->stillClearedTo(buf->ptr, value)) return TRHIP_OK;      // cleared to it earlier in this recording, no command has written it since
    const bool hoist = !cl->usedSoFar(buf->ptr) && cl->firstClearBatch && cl->firstClearBatch->count < trhip_cmdlist_t::ClearBatch::kMax;
    const bool merges = cl->openClearBatch && cl->openClearOp == cl->ops.size() - 1 && cl->openClearBatch->count < trhip_cmdlist_t::ClearBatch::kMax;
    cl->hold(buf, true, hoist ? cl->firstClearOp : merges ? cl->openClearOp : cl->ops.size());
    int rc = cl->recordClearWords(buf->ptr, buf->byteSize / 4, value, hoist);
    if (rc == TRHIP_OK) cl->clearedTo[buf->ptr] = { value, cl->useMarks.size() };
    return rc;
}

int trhip_cmd_clear_texture_f32(trhip_cmdlist cl, trhip_texture tex, float value)
{
    TRHIP_RECORDING(cl);
    if (tex && cl->heldConflicts(tex->ptr, true)) cl->flushHeldSide();
    if (!tex) return fail(TRHIP_ERR_INVALID, "clear_texture: null texture");
    if (!tex->ptr) return fail(TRHIP_ERR_STATE, "clear_texture(%s): no memory bound", tex->name.c_str());
    void* p = tex->ptr;
    cl->hold(tex, true);
    if (tex->format == TRHIP_FORMAT_R32_FLOAT) {
        uint32_t bits;
        memcpy(&bits, &value, 4);
        size_t n = (size_t)(tex->totalBytes / 4);
        cl->ops.push_back({ "", [p, n, bits](hipStream_t s) { TRHIP_HIP(hipMemsetD32Async((hipDeviceptr_t)p, (int)bits, n, s)); return (int)TRHIP_OK; } });
        cl->ops.back().kind = "clear_texture";
    } else {
        _Float16 h = (_Float16)value; // round-to-nearest-even
        uint16_t bits;
        memcpy(&bits, &h, 2);
        size_t n = (size_t)(tex->totalBytes / 2);
        cl->ops.push_back({ "", [p, n, bits](hipStream_t s) { TRHIP_HIP(hipMemsetD16Async((hipDeviceptr_t)p, bits, n, s)); return (int)TRHIP_OK; } });
        cl->ops.back().kind = "clear_texture";
    }
    return TRHIP_OK;
}

int trhip_cmd_copy_buffer(trhip_cmdlist cl, trhip_buffer dst, uint64_t dstOff, trhip_buffer src, uint64_t srcOff, uint64_t bytes)
{
    TRHIP_RECORDING(cl);
    cl->flushHeldSide();                               // (held side ops: nothing is assumed about what this command touches)
    if (!dst || !src) return fail(TRHIP_ERR_INVALID, "copy_buffer: null buffer");
    if (!dst->ptr || !src->ptr) return fail(TRHIP_ERR_STATE, "copy_buffer: a buffer has no memory bound");
    if (dstOff + bytes > dst->byteSize || srcOff + bytes > src->byteSize) return fail(TRHIP_ERR_INVALID, "copy_buffer(%s <- %s): range exceeds a buffer", dst->name.c_str(), src->name.c_str());
    void* d = (char*)dst->ptr + dstOff;
    const void* sp = (const char*)src->ptr + srcOff;
    cl->hold(dst, true); cl->hold(src, false);
    cl->ops.push_back({ "", [d, sp, bytes](hipStream_t s) { TRHIP_HIP(hipMemcpyAsync(d, sp, (size_t)bytes, hipMemcpyDeviceToDevice, s)); return (int)TRHIP_OK; } });
    cl->ops.back().kind = "copy_buffer";
    return TRHIP_OK;
}

int trhip_cmd_host_callback(trhip_cmdlist cl, trhip_host_fn fn, void* user)
{
    TRHIP_RECORDING(cl);
    cl->flushHeldSide();                               // (held side ops: nothing is assumed about what this command touches)
    if (!fn) return fail(TRHIP_ERR_INVALID, "host_callback: null function");
    cl->ops.push_back({ "", [fn, user](hipStream_t s) { fn(user, (void*)s); return (int)TRHIP_OK; } });
    cl->ops.back().kind = "host_callback";
    return TRHIP_OK;
}

int trhip_cmd_copy_texture(trhip_cmdlist cl, trhip_texture dst, trhip_texture src)
{
    TRHIP_RECORDING(cl);
    cl->flushHeldSide();                               // (held side ops: nothing is assumed about what this command touches)
    if (!dst || !src) return fail(TRHIP_ERR_INVALID, "copy_texture: null texture");
    if (!dst->ptr || !src->ptr) return fail(TRHIP_ERR_STATE, "copy_texture: a texture has no memory bound");
    if (dst->width != src->width || dst->height != src->height || dst->mips != src->mips || dst->format != src->format)
        return fail(TRHIP_ERR_INVALID, "copy_texture(%s <- %s): descriptions differ", dst->name.c_str(), src->name.c_str());
    void* d = dst->ptr;
    const void* sp = src->ptr;
    const uint64_t bytes = src->totalBytes;
    cl->hold(dst, true); cl->hold(src, false);
    cl->ops.push_back({ "", [d, sp, bytes](hipStream_t s) { TRHIP_HIP(hipMemcpyAsync(d, sp, (size_t)bytes, hipMemcpyDeviceToDevice, s)); return (int)TRHIP_OK; } });
    cl->ops.back().kind = "copy_texture";
    return TRHIP_OK;
}

static int recordDispatch(trhip_cmdlist cl, const char* name, const trhip_binding* b, uint32_t nb, const void* push, uint32_t pushBytes,
                          bool indirect, trhip_buffer args, uint32_t argsOff, uint32_t gx, uint32_t gy, uint32_t gz)
{
    TRHIP_RECORDING(cl);
    if (!name) return fail(TRHIP_ERR_INVALID, "dispatch: shader name is null");
    const ShaderEntry* e = findShader(name);
    if (!e) return fail(TRHIP_ERR_UNKNOWN_SHADER, "dispatch: unknown shader '%s'", name);
    if (nb && !b) return fail(TRHIP_ERR_INVALID, "dispatch(%s): bindings is null", name);
    if (!cl->heldSide.empty()) {                        // a held side op that touches one of this command's resources goes first
        bool conflict = indirect && args && cl->heldConflicts(args->ptr, false);
        for (uint32_t i = 0; i < nb && !conflict; ++i) {
            if (!b[i].resource) continue;
            switch (b[i].type) {
            case TRHIP_BIND_CONSTANT_BUFFER: case TRHIP_BIND_STRUCTURED_SRV: case TRHIP_BIND_STRUCTURED_UAV:
                conflict = cl->heldConflicts(((trhip_buffer_t*)b[i].resource)->ptr, b[i].type == TRHIP_BIND_STRUCTURED_UAV); break;
            case TRHIP_BIND_TEXTURE_SRV: case TRHIP_BIND_TEXTURE_UAV:
                conflict = cl->heldConflicts(((trhip_texture_t*)b[i].resource)->ptr, b[i].type == TRHIP_BIND_TEXTURE_UAV); break;
            default: break;
            }
        }
        if (conflict) cl->flushHeldSide();
    }
    if (indirect) {
        if (!args || !args->ptr) return fail(TRHIP_ERR_INVALID, "dispatch_indirect(%s): no argument buffer", name);
        if (argsOff % 4 || argsOff + 12 > args->byteSize) return fail(TRHIP_ERR_INVALID, "dispatch_indirect(%s): bad argument offset %u", name, argsOff);
        cl->hold(args, false);
    } else if (gx == 0 || gy == 0 || gz == 0) {
        return fail(TRHIP_ERR_INVALID, "dispatch(%s): zero group count (Graphic.cpp:944)", name);
    }
    for (uint32_t i = 0; i < nb; ++i) {
        switch (b[i].type) {
        case TRHIP_BIND_CONSTANT_BUFFER: case TRHIP_BIND_STRUCTURED_SRV: case TRHIP_BIND_STRUCTURED_UAV: {
            trhip_buffer_t* r = (trhip_buffer_t*)b[i].resource;
            if (!r) return fail(TRHIP_ERR_INVALID, "dispatch(%s): binding %u has no buffer", name, i);
            if (b[i].type != TRHIP_BIND_CONSTANT_BUFFER && !r->ptr) return fail(TRHIP_ERR_STATE, "dispatch(%s): buffer '%s' has no memory bound", name, r->name.c_str());
            if (b[i].type == TRHIP_BIND_STRUCTURED_UAV && !r->canHaveUAVs) return fail(TRHIP_ERR_INVALID, "dispatch(%s): buffer '%s' bound as UAV without canHaveUAVs", name, r->name.c_str());
            cl->hold(r, b[i].type == TRHIP_BIND_STRUCTURED_UAV);
            break; }
        case TRHIP_BIND_TEXTURE_SRV: case TRHIP_BIND_TEXTURE_UAV: {
            trhip_texture_t* t = (trhip_texture_t*)b[i].resource;
            if (!t) return fail(TRHIP_ERR_INVALID, "dispatch(%s): binding %u has no texture", name, i);
            if (!t->ptr) return fail(TRHIP_ERR_STATE, "dispatch(%s): texture '%s' has no memory bound", name, t->name.c_str());
            if (b[i].type == TRHIP_BIND_TEXTURE_UAV && b[i].baseMip >= t->mips) return fail(TRHIP_ERR_INVALID, "dispatch(%s): UAV mip %u out of range", name, b[i].baseMip);
            cl->hold(t, b[i].type == TRHIP_BIND_TEXTURE_UAV);
            break; }
        case TRHIP_BIND_PUSH_CONSTANTS: case TRHIP_BIND_SAMPLER: break;
        default: return fail(TRHIP_ERR_INVALID, "dispatch(%s): binding %u has unknown type %u", name, i, b[i].type);
        }
    }
    DispatchCtx ctx{ cl, e->name.c_str(), e->variant, b, nb, push, pushBytes, indirect, args, argsOff, gx, gy, gz };
    return e->fn(ctx);
}

int trhip_cmd_dispatch(trhip_cmdlist cl, const char* name, const trhip_binding* b, uint32_t nb, const void* push, uint32_t pushBytes,
                       uint32_t gx, uint32_t gy, uint32_t gz)
{
    return recordDispatch(cl, name, b, nb, push, pushBytes, false, nullptr, 0, gx, gy, gz);
}

int trhip_cmd_dispatch_indirect(trhip_cmdlist cl, const char* name, const trhip_binding* b, uint32_t nb, const void* push, uint32_t pushBytes,
                                trhip_buffer args, uint32_t argsOff)
{
    return recordDispatch(cl, name, b, nb, push, pushBytes, true, args, argsOff, 0, 0, 0);
}

int trhip_cmd_begin_timer(trhip_cmdlist cl, trhip_timer t)
{
    TRHIP_RECORDING(cl);
    if (!t) return fail(TRHIP_ERR_INVALID, "timer is null");
    cl->ops.push_back({ "", [t](hipStream_t s) { TRHIP_HIP(hipEventRecord(t->e0, s)); t->began = true; t->ended = false; return (int)TRHIP_OK; } });
    cl->ops.back().kind = "timer";
    return TRHIP_OK;
}

int trhip_cmd_end_timer(trhip_cmdlist cl, trhip_timer t)
{
    TRHIP_RECORDING(cl);
    if (!t) return fail(TRHIP_ERR_INVALID, "timer is null");
    cl->ops.push_back({ "", [t](hipStream_t s) { TRHIP_HIP(hipEventRecord(t->e1, s)); t->ended = true; return (int)TRHIP_OK; } });
    cl->ops.back().kind = "timer";
    return TRHIP_OK;
}

int trhip_cmd_begin_marker(trhip_cmdlist cl, const char* name)
{
    TRHIP_RECORDING(cl);
    cl->markers.push_back(name ? name : "");
    return TRHIP_OK;
}

int trhip_cmd_end_marker(trhip_cmdlist cl)
{
    TRHIP_RECORDING(cl);
    if (cl->markers.empty()) return fail(TRHIP_ERR_STATE, "end_marker without begin_marker");
    cl->markers.pop_back();
    return TRHIP_OK;
}

int trhip_queue_execute(trhip_device dev, const trhip_cmdlist* lists, uint32_t n)
{
    if (!dev || (n && !lists)) return fail(TRHIP_ERR_INVALID, "queue_execute: null argument");
    TRHIP_HIP(hipSetDevice(dev->index));
    std::lock_guard<std::mutex> lock(dev->mutex);
    bool inRun = false;
    std::vector<Op::Access> runTouched;
    auto endRun = [&]() -> int {                       // the side stream's current run is complete: publish its event
        if (!inRun) return TRHIP_OK;
        inRun = false;
        const uint64_t id = ++dev->sideRunCounter;
        const uint32_t slot = (uint32_t)(id % trhip_device_t::kSideRuns);
        if (id > trhip_device_t::kSideRuns && id - trhip_device_t::kSideRuns > dev->mainWaitedUpTo) {
            TRHIP_HIP(hipStreamWaitEvent(dev->stream, dev->runDone[slot], 0));   // the run that owned this event slot
            dev->mainWaitedUpTo = id - trhip_device_t::kSideRuns;
        }
        TRHIP_HIP(hipEventRecord(dev->runDone[slot], dev->sideStream));
        for (const Op::Access& t : runTouched) (t.write ? dev->sideWriter : dev->sideReader)[t.ptr] = id;
        runTouched.clear();
        return TRHIP_OK;
    };
    auto ownerOf = [&](const void* ptr, bool write) -> uint64_t {   // a main-stream command reads / writes `ptr`: the side run it must follow
        uint64_t need = 0;                             // read after side write; write after side write or read
        auto w = dev->sideWriter.find(ptr);
        if (w != dev->sideWriter.end()) need = w->second;
        if (write) {
            auto r = dev->sideReader.find(ptr);
            if (r != dev->sideReader.end() && r->second > need) need = r->second;
        }
        return need;
    };
    auto waitForRun = [&](uint64_t need) -> int {
        if (need <= dev->mainWaitedUpTo) return TRHIP_OK;
        TRHIP_HIP(hipStreamWaitEvent(dev->stream, dev->runDone[need % trhip_device_t::kSideRuns], 0));
        dev->mainWaitedUpTo = need;                    // the side stream is in order: earlier runs are covered too
        return TRHIP_OK;
    };
    static const bool g_tapForks = getenv("TRHIP_NO_FORK_TAP") == nullptr;       // experiments: forks by marker packets, as in rounds 1-3
    auto nextIsSide = [&](uint32_t li, size_t oi) -> bool {                       // the command after (li, oi), across list boundaries
        if (!dev->sideStream) return false;
        for (++oi; li < n; ++li, oi = 0)
            if (lists[li] && oi < lists[li]->ops.size()) return lists[li]->ops[oi].lane == 1;
        return false;
    };
    for (uint32_t i = 0; i < n; ++i) {
        trhip_cmdlist_t* cl = lists[i];
        if (!cl) return fail(TRHIP_ERR_INVALID, "queue_execute: list %u is null", i);
        if (cl->open) return fail(TRHIP_ERR_STATE, "queue_execute: list %u is still open", i);
        if (cl->dev != dev) return fail(TRHIP_ERR_INVALID, "queue_execute: list %u belongs to another device", i);
        size_t mark = 0;
        for (size_t oi = 0; oi < cl->ops.size(); ++oi) {
            const Op& op = cl->ops[oi];
            const bool side = op.lane == 1 && dev->sideStream;
            hipStream_t stream = dev->stream;
            if (side) {
                if (!inRun) {                          // fork: the side stream continues from this point of the main stream
                    if (!dev->forkSignalled) TRHIP_HIP(hipEventRecord(dev->evFork, dev->stream));   // (else: the last kernel's completion signal, LaunchTap)
                    dev->forkSignalled = false;
                    TRHIP_HIP(hipStreamWaitEvent(dev->sideStream, dev->evFork, 0));
                    inRun = true;
                }
                runTouched.insert(runTouched.end(), op.touched.begin(), op.touched.end());
                stream = dev->sideStream;
                for (; mark < cl->useMarks.size() && cl->useMarks[mark].op <= oi; ++mark)    // ordered by the fork and by the side stream itself
                    if (cl->useMarks[mark].write && cl->useMarks[mark].version) cl->useMarks[mark].version->fetch_add(1);
            } else {
                int rc = endRun();
                if (rc != TRHIP_OK) return rc;
                // ONE wait for the latest run any of the command's resources needs (a wait is a packet in the main queue: ~4 us
                // of the frame's chain even when it is satisfied on arrival, tools/sync_cost.hip)
                uint64_t need = 0;
                for (; mark < cl->useMarks.size() && cl->useMarks[mark].op <= oi; ++mark) {
                    const trhip_cmdlist_t::UseMark& m = cl->useMarks[mark];
                    need = std::max(need, ownerOf(m.ptr, m.write));
                    if (m.write && m.version) m.version->fetch_add(1);     // the contents change with this command
                }
                rc = waitForRun(need);
                if (rc != TRHIP_OK) return rc;
            }
            const bool prof = dev->profiling && !op.name.empty() && (dev->profileFilter.empty() || dev->profileFilter == op.name);
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (prof) { e0 = dev->acquireEvent(); e1 = dev->acquireEvent(); TRHIP_HIP(hipEventRecord(e0, stream)); }
            const auto h0 = g_hostProfile.on ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point{};
            // the next command forks the side stream from here: let this command's kernel carry the fork event (LaunchTap)
            trhip::LaunchTap tap;
            dev->forkSignalled = false;
            const bool tapped = !side && !prof && g_tapForks && nextIsSide(i, oi);
            if (tapped) { tap.onStream = stream; tap.stopEvent = dev->evFork; trhip::g_launchTap = &tap; }
            int rc = op.fn(stream);
            trhip::g_launchTap = nullptr;
            if (tapped && tap.launches >= 1 && rc == TRHIP_OK) dev->forkSignalled = true;
            if (g_hostProfile.on) g_hostProfile.add(op.kind, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count());
            if (rc != TRHIP_OK) return rc;
            if (prof) { TRHIP_HIP(hipEventRecord(e1, stream)); dev->pending.push_back({ op.name, e0, e1 }); }
        }
    }
    return endRun();
}

// ---- streams / events for callers that order work across streams themselves ------------------------------
int trhip_stream_create_priority(int device_index, int priority_class, void** out)
{
    if (!out) return fail(TRHIP_ERR_INVALID, "stream_create: out is null");
    TRHIP_HIP(hipSetDevice(device_index));
    int lowest = 0, highest = 0;
    TRHIP_HIP(hipDeviceGetStreamPriorityRange(&lowest, &highest));      // numerically: highest <= 0 <= lowest
    hipStream_t s = nullptr;
    TRHIP_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority_class < 0 ? highest : priority_class > 0 ? lowest : 0));
    *out = (void*)s;
    return TRHIP_OK;
}

int trhip_stream_create(int device_index, void** out) { return trhip_stream_create_priority(device_index, 0, out); }

void trhip_stream_destroy(void* s)
{
    if (s) { (void)hipStreamSynchronize((hipStream_t)s); (void)hipStreamDestroy((hipStream_t)s); }
}

int trhip_stream_synchronize(void* s)
{
    TRHIP_HIP(hipStreamSynchronize((hipStream_t)s));
    return TRHIP_OK;
}

int trhip_event_create(int device_index, void** out)
{
    if (!out) return fail(TRHIP_ERR_INVALID, "event_create: out is null");
    TRHIP_HIP(hipSetDevice(device_index));
    hipEvent_t e = nullptr;
    TRHIP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *out = (void*)e;
    return TRHIP_OK;
}

void trhip_event_destroy(void* e)
{
    if (e) (void)hipEventDestroy((hipEvent_t)e);
}

int trhip_event_record(void* e, void* s)
{
    if (!e) return fail(TRHIP_ERR_INVALID, "event_record: event is null");
    TRHIP_HIP(hipEventRecord((hipEvent_t)e, (hipStream_t)s));
    return TRHIP_OK;
}

int trhip_stream_wait_event(void* s, void* e)
{
    if (!e) return fail(TRHIP_ERR_INVALID, "stream_wait_event: event is null");
    TRHIP_HIP(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0));
    return TRHIP_OK;
}

// ---- timers / profile ----------------------------------------------------------------------------
int trhip_timer_create(trhip_device dev, trhip_timer* out)
{
    if (!dev || !out) return fail(TRHIP_ERR_INVALID, "timer_create: null argument");
    TRHIP_HIP(hipSetDevice(dev->index));
    auto t = std::make_unique<trhip_timer_t>();
    t->dev = dev;
    TRHIP_HIP(hipEventCreate(&t->e0));
    TRHIP_HIP(hipEventCreate(&t->e1));
    *out = t.release();
    return TRHIP_OK;
}

void trhip_timer_release(trhip_timer t)
{
    if (!t) return;
    (void)hipEventDestroy(t->e0);
    (void)hipEventDestroy(t->e1);
    delete t;
}

int trhip_timer_get_ms(trhip_timer t, float* ms)
{
    if (!t || !ms) return fail(TRHIP_ERR_INVALID, "timer_get_ms: null argument");
    if (!t->began || !t->ended) return fail(TRHIP_ERR_STATE, "timer_get_ms: timer was not begun and ended in an executed list");
    TRHIP_HIP(hipEventSynchronize(t->e1));
    TRHIP_HIP(hipEventElapsedTime(ms, t->e0, t->e1));
    return TRHIP_OK;
}

int trhip_profile_enable(trhip_device dev, int enabled)
{
    if (!dev) return fail(TRHIP_ERR_INVALID, "device is null");
    std::lock_guard<std::mutex> lock(dev->mutex);
    dev->profiling = enabled != 0;
    return TRHIP_OK;
}

int trhip_profile_filter(trhip_device dev, const char* name)
{
    if (!dev) return fail(TRHIP_ERR_INVALID, "device is null");
    std::lock_guard<std::mutex> lock(dev->mutex);
    dev->profileFilter = name ? name : "";
    return TRHIP_OK;
}

int trhip_profile_reset(trhip_device dev)
{
    if (!dev) return fail(TRHIP_ERR_INVALID, "device is null");
    std::lock_guard<std::mutex> lock(dev->mutex);
    int rc = dev->drainProfile();
    dev->accum.clear();
    dev->accumOrder.clear();
    return rc;
}

int trhip_profile_count(trhip_device dev, uint32_t* n)
{
    if (!dev || !n) return fail(TRHIP_ERR_INVALID, "profile_count: null argument");
    std::lock_guard<std::mutex> lock(dev->mutex);
    int rc = dev->drainProfile();
    *n = (uint32_t)dev->accumOrder.size();
    return rc;
}

int trhip_profile_entry(trhip_device dev, uint32_t i, const char** name, uint64_t* launches, double* totalMs)
{
    if (!dev) return fail(TRHIP_ERR_INVALID, "device is null");
    std::lock_guard<std::mutex> lock(dev->mutex);
    if (i >= dev->accumOrder.size()) return fail(TRHIP_ERR_INVALID, "profile_entry: index out of range");
    const std::string& key = dev->accumOrder[i];
    const ProfileAccum& a = dev->accum[key];
    if (name) *name = key.c_str();
    if (launches) *launches = a.launches;
    if (totalMs) *totalMs = a.totalMs;
    return TRHIP_OK;
}

} // extern "C"
