// trhip_internal.h -- internals of the C-ABI back end (include/trhip.h).  Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <hip/hip_ext.h>

#include "../../include/trhip.h"
#include "ShaderInterop.h"

namespace trhip
{

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int hipfail(hipError_t e, const char* what);

#define TRHIP_HIP(expr)                                                        \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) return ::trhip::hipfail(_e, #expr);              \
    } while (0)

#define TRHIP_REQUIRE(cond, ...)                                               \
    do {                                                                       \
        if (!(cond)) return ::trhip::fail(TRHIP_ERR_INVALID, __VA_ARGS__);     \
    } while (0)

struct ProfilePending { std::string name; hipEvent_t e0, e1; };
struct ProfileAccum { uint64_t launches = 0; double totalMs = 0; };

} // namespace trhip

struct trhip_device_t
{
    int index = 0;
    hipStream_t stream = nullptr;
    bool ownsStream = false;
    uint32_t computeUnits = 0;
    uint32_t waveSize = 0;
    uint64_t totalMem = 0;

    // Second stream for work whose results nothing later in the frame consumes (the ordered-list build of
    // a cull pass): it overlaps the passes that follow.  Consecutive side ops form a RUN that ends with an
    // event; a main-stream command that uses memory a run touched first waits for that run's event
    // (execute-time check, trhip_queue_execute), and every host-side synchronisation covers both streams.
    static constexpr uint32_t kSideRuns = 16;
    hipStream_t sideStream = nullptr;
    hipEvent_t evFork = nullptr;
    bool forkSignalled = false;                // evFork already carries the completion of the main stream's last kernel (trhip::LaunchTap)
    hipEvent_t runDone[kSideRuns] = {};
    uint64_t sideRunCounter = 0;               // id of the last finished run (ids start at 1)
    uint64_t mainWaitedUpTo = 0;               // the main stream is ordered after all runs <= this id
    std::unordered_map<const void*, uint64_t> sideWriter;  // allocation -> last run that wrote it
    std::unordered_map<const void*, uint64_t> sideReader;  // allocation -> last run that read it
    int syncAll();                             // host wait for both streams

    std::mutex mutex;
    bool profiling = false;
    std::string profileFilter;              // "" = every named op; else only ops of exactly this name (trhip_profile_filter)
    std::vector<trhip::ProfilePending> pending;
    std::vector<hipEvent_t> eventPool;
    std::map<std::string, trhip::ProfileAccum> accum;
    std::vector<std::string> accumOrder;

    hipEvent_t acquireEvent();
    int drainProfile();
};

struct trhip_heap_t
{
    trhip_device_t* dev = nullptr;
    void* base = nullptr;
    uint64_t bytes = 0;
    std::atomic<int> rc{1};
};

struct trhip_buffer_t
{
    trhip_device_t* dev = nullptr;
    uint64_t byteSize = 0;
    uint32_t structStride = 0;
    bool canHaveUAVs = false, isDrawIndirectArgs = false, isVirtual = false, isVolatileConstant = false;
    std::string name;
    void* ptr = nullptr;
    bool owns = false;
    trhip_heap_t* heap = nullptr;
    std::vector<uint8_t> shadow; // current version of a volatile constant buffer (record time)
    // Back-end private companion allocation (device memory, freed with the buffer).  Used by the
    // instance-cull pass to hand the meshlet-cull pass a screen-tile-sorted PROCESSING order of the
    // amplification records it wrote into this buffer (k_gpuculling.hip / k_basepass_as.hip).
    void* sidecar = nullptr;
    uint64_t sidecarBytes = 0;
    // Bumped, in SUBMISSION order, by every command that writes the buffer (trhip_queue_execute) and by uploads:
    // derived data (the instance cull cache below, the HZB footprint-min table) records the version it was built
    // from and is rebuilt when that is no longer the current one.
    std::atomic<uint64_t> version{1};
    // Instance buffers only: the INSTANCE CULL CACHE (k_gpuculling.hip), a compact SoA restatement of what the
    // instance cull reads per instance (world-space bounding sphere, max scale, LOD table) -- 88 B instead of the
    // 300 B of AoS instance + mesh records it would otherwise pull through HBM every frame.
    void* cullCache = nullptr;
    uint64_t cullCacheBytes = 0;
    uint64_t cullCacheInstVersion = 0, cullCacheMeshVersion = 0;
    const void* cullCacheMesh = nullptr;
    // Meshlet buffers only: the MESHLET CULL STREAM (k_basepass_as.hip), the 20 bytes of each 32-byte MeshletData the cull
    // reads (bounding sphere, cone word) as two dense arrays; rebuilt when the buffer's version moves.
    void* cullStream = nullptr;
    uint64_t cullStreamBytes = 0;
    uint64_t cullStreamVersion = 0;
    std::atomic<int> rc{1};
};

struct trhip_texture_t
{
    trhip_device_t* dev = nullptr;
    uint32_t width = 0, height = 0, mips = 0, format = 0;
    uint32_t texelBytes = 0;
    bool isUAV = false, isVirtual = false;
    std::string name;
    uint64_t mipOffset[16] = {};
    uint64_t totalBytes = 0;
    void* ptr = nullptr;
    bool owns = false;
    trhip_heap_t* heap = nullptr;
    // Back-end private companion of an R16F min-HZB: the FOOTPRINT-MIN TABLE (k_hzb.hip).  For every mip and
    // every possible bilinear footprint origin (x0,y0) in [-1,w-1]x[-1,h-1] it holds the min of the 2x2
    // edge-clamped footprint, so the meshlet cull kernel resolves SampleLevel(min-reduction) with ONE 2-byte
    // load.  It is current while quadBuiltVersion == version (see trhip_buffer_t::version).
    void* quad = nullptr;
    uint64_t quadBytes = 0;
    uint32_t quadOffset[16] = {};              // first entry of mip k; mip k has ((w_k >> 3) + 1) * ((h_k >> 3) + 1) blocks of 8 x 8 entries
    uint32_t quadTotal = 0;
    uint64_t quadBuiltVersion = 0;
    std::atomic<uint64_t> version{1};
    std::atomic<int> rc{1};

    uint32_t mipW(uint32_t k) const { return (width >> k) ? (width >> k) : 1u; }
    uint32_t mipH(uint32_t k) const { return (height >> k) ? (height >> k) : 1u; }
    void* mipPtr(uint32_t k) const { return (char*)ptr + mipOffset[k]; }
};

struct trhip_timer_t
{
    trhip_device_t* dev = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool began = false, ended = false;
};

namespace trhip
{

struct Op
{
    std::string name;                       // profile key ("" = not profiled)
    std::function<int(hipStream_t)> fn;
    const char* kind = "dispatch";          // what the op is, for the host-side submission profile (TRHIP_HOST_PROFILE=1)
    uint8_t lane = 0;                       // 0 main stream, 1 side stream (see DispatchCtx::emitSide)
    struct Access { const void* ptr; bool write; };
    std::vector<Access> touched;            // side ops: allocations read / written (base pointers)
};

} // namespace trhip

struct trhip_cmdlist_t
{
    trhip_device_t* dev = nullptr;
    bool open = false;
    std::vector<trhip::Op> ops;
    std::vector<trhip_buffer_t*> heldBuffers;
    std::vector<trhip_texture_t*> heldTextures;
    std::vector<std::string> markers;

    struct ScratchBlock { void* ptr; size_t bytes; size_t used; };
    std::vector<ScratchBlock> scratch;

    // Consecutive buffer clears are issued as ONE kernel launch (a frame has ~11 clears of a few bytes
    // each; a launch costs ~5 us on both sides of the queue).
    struct ClearBatch { static constexpr uint32_t kMax = 32; void* ptr[kMax]; uint64_t words[kMax]; uint32_t value[kMax]; uint32_t count = 0; };
    std::shared_ptr<ClearBatch> openClearBatch;    // batch of the LAST op in `ops`, if that op is a clear
    size_t openClearOp = SIZE_MAX;
    // A clear of memory that no earlier command of this recording uses joins the FIRST clear launch of the recording
    // (it is equivalent there, and off the chain of dependent small launches later in the frame); a clear of memory
    // that still holds the value (cleared to it earlier in this recording, not written since) is dropped.
    std::shared_ptr<ClearBatch> firstClearBatch;
    size_t firstClearOp = SIZE_MAX;
    struct ClearedTo { uint32_t value; size_t marks; };       // marks: useMarks.size() right after the clear was noted
    std::unordered_map<const void*, ClearedTo> clearedTo;     // whole allocations, by base pointer
    bool stillClearedTo(const void* ptr, uint32_t value) const
    {
        auto it = clearedTo.find(ptr);
        if (it == clearedTo.end() || it->second.value != value) return false;
        for (size_t i = it->second.marks; i < useMarks.size(); ++i)
            if (useMarks[i].ptr == ptr && useMarks[i].write) return false;
        return true;
    }
    // fresh: the caller guarantees that no earlier command of this recording uses the memory (scratch just allocated)
    int recordClearWords(void* ptr, uint64_t words, uint32_t value, bool fresh = false);
    bool usedSoFar(const void* ptr) const { for (const UseMark& m : useMarks) if (m.ptr == ptr) return true; return false; }

    // Which allocations each command uses (recorded by hold()): checked against the side stream's runs
    // when the list is executed.
    struct UseMark { size_t op; const void* ptr; bool write; std::atomic<uint64_t>* version; };
    std::vector<UseMark> useMarks;
    void use(const void* ptr, size_t op, bool write, std::atomic<uint64_t>* version = nullptr)
    {
        if (!ptr) return;
        useMarks.push_back({ op, ptr, write, version });
    }
    // the command being recorded turned out not to access `ptr` (bound for interface fidelity only): forget its marks
    void forgetUse(const void* ptr, size_t op)
    {
        for (size_t i = useMarks.size(); i-- > 0 && useMarks[i].op >= op;)
            if (useMarks[i].op == op && useMarks[i].ptr == ptr) useMarks.erase(useMarks.begin() + (ptrdiff_t)i);
    }

    // Peephole between two consecutive commands of one recording: a record function may leave a note about the command
    // it just emitted; the next record function may replace that command by a fused one (same command index, so the
    // use marks of both stay attached to it).  Cleared by any other command.
    struct Peephole { size_t op = SIZE_MAX; const char* kind = nullptr; std::shared_ptr<void> data; } peephole;

    // HELD side ops: a side-stream op whose results nothing in this recording needs may be held back and put into the stream
    // LATER than where it was recorded -- behind the next command that forks the side stream anyway -- so that it overlaps a
    // different part of the frame (the early pass's list expansion, 122 MB of stores on C3, ran beside the latency-bound late
    // phase and tripled its kernels' times).  It is flushed, i.e. appended at the current position, before any command that uses
    // a resource it touches (recordDispatch, clears, copies check their operands), by whoever wants it in front of its own side
    // ops, and when the recording is closed.
    std::vector<trhip::Op> heldSide;
    bool heldConflicts(const void* ptr, bool write) const
    {
        for (const trhip::Op& op : heldSide)
            for (const trhip::Op::Access& t : op.touched)
                if (t.ptr == ptr && (t.write || write)) return true;
        return false;
    }
    void flushHeldSide()
    {
        if (heldSide.empty()) return;
        for (trhip::Op& op : heldSide) ops.push_back(std::move(op));
        heldSide.clear();
        openClearBatch.reset(); openClearOp = SIZE_MAX;
        peephole = Peephole{};
    }
    void* scratchAlloc(size_t bytes);       // device memory valid until the list is re-opened/released
    void* scratchAllocSide(size_t bytes);   // same, from an arena only side-stream ops use (they are in order among themselves)
    std::vector<ScratchBlock> sideScratch;
    void resetRecording();
    // write: the command writes the resource; op: index of the command that uses it (default: the next one)
    void hold(trhip_buffer_t* b, bool write, size_t op = SIZE_MAX);
    void hold(trhip_texture_t* t, bool write);
};

namespace trhip
{

// What a shader's record function sees of one AddComputePass (Graphic.cpp:893-947).
struct DispatchCtx
{
    trhip_cmdlist_t* cl;
    const char* shaderName;
    int variant;                            // e.g. LATE_CULL / FILTER value of the permutation
    const trhip_binding* bindings; uint32_t numBindings;
    const void* push; uint32_t pushBytes;
    bool indirect; trhip_buffer_t* argsBuffer; uint32_t argsOffset;
    uint32_t gx, gy, gz;

    trhip_buffer_t* buffer(uint32_t type, uint32_t slot) const;
    trhip_texture_t* texture(uint32_t type, uint32_t slot, uint32_t* baseMip = nullptr) const;
    // Bytes of the constant buffer bound at b<slot> (volatile CB version at record time) or of
    // the push constants; nullptr if absent / wrong size.
    const void* constants(uint32_t slot, size_t bytes) const;
    uint32_t computeUnits() const { return cl->dev->computeUnits; }
    void* scratch(size_t bytes) const { return cl->scratchAlloc(bytes); }
    void* scratchSide(size_t bytes) const { return cl->dev->sideStream ? cl->scratchAllocSide(bytes) : cl->scratchAlloc(bytes); }
    void emit(const char* kernelName, std::function<int(hipStream_t)> fn) const;
    // Same, on the device's side stream, ordered after everything recorded before it.  `touched`: every
    // device allocation the op reads or writes that a later command could also use (base pointers).
    void emitSide(const char* kernelName, std::function<int(hipStream_t)> fn, std::initializer_list<Op::Access> touched) const;
    // Same, HELD (trhip_cmdlist_t::heldSide): enters the stream at the next flush, not here.
    void emitSideHeld(const char* kernelName, std::function<int(hipStream_t)> fn, std::initializer_list<Op::Access> touched) const;
};

// Footprint-min table of an HZB (trhip_texture_t::quad), k_hzb.hip.  ensure: allocate + lay out (record time);
// emit: a command that rebuilds it (side stream when there is one); launch: rebuild right now on `s`.
int hzbQuadEnsure(trhip_texture_t* tex);
int hzbQuadEmitBuild(const DispatchCtx& ctx, trhip_texture_t* tex);
int hzbQuadLaunchBuild(trhip_texture_t* tex, hipStream_t s);

// Instance cull cache (instance_cache.hip.h, k_gpuculling.hip).  ensure: allocate (record time); launch: rebuild on
// `s` unless it is current -- called while commands are submitted, so every earlier write is counted.
int instanceCacheEnsure(trhip_buffer_t* instances);
int instanceCacheLaunchBuild(trhip_buffer_t* instances, trhip_buffer_t* meshData, hipStream_t s);

// Record capacity (groups) from which the early meshlet cull resolves its HZB lookups through the footprint-min table
// (rebuilt per frame: on the side stream beside a large instance pass, by extra workgroups of a small one's launch) instead of
// the texels.  2^17: a rank's share of C3 at 8 ranks (390 k groups: 53 instead of 75 us for its early cull); real assets below
// that keep the texel path.  TRHIP_TABLE_MIN_GROUPS overrides (tuning).
inline uint32_t tableMinGroups()
{
    static const uint32_t v = [] { const char* e = getenv("TRHIP_TABLE_MIN_GROUPS"); return e ? (uint32_t)strtoul(e, nullptr, 0) : (1u << 17); }();
    return v;
}

using RecordFn = int (*)(DispatchCtx&);
void registerShader(const char* name, RecordFn fn, int variant);

struct ShaderRegistrar
{
    ShaderRegistrar(const char* name, RecordFn fn, int variant = 0) { registerShader(name, fn, variant); }
};

// FORKS THROUGH THE KERNEL'S COMPLETION SIGNAL.  The side stream continues from a point of the main stream (a fork).
// hipEventRecord there puts a marker packet into the main queue: 2.9 us of the frame's chain of dependent launches, 5.3 us
// with the side stream's wait (tools/sync_cost.hip, profiles/r4/sync_cost.txt).  hipExtLaunchKernelGGL hands the event the
// completion signal of the kernel itself: 1.6 us.  While trhip_queue_execute runs the main-stream command in front of a fork
// it sets a tap; EVERY launch of that command on that stream (every launch of the back end goes through TRHIP_LAUNCH) is
// given the fork event as its stop event -- like a re-recorded event it ends up standing for the last of them (checked by
// tools/sync_cost.hip: two launches, one event, the waiter sees the second one's result).
struct LaunchTap { hipStream_t onStream = nullptr; hipEvent_t stopEvent = nullptr; int launches = 0; };
extern thread_local LaunchTap* g_launchTap;
#define TRHIP_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                       \
    do {                                                                                                            \
        trhip::LaunchTap* tap_ = trhip::g_launchTap;                                                                \
        if (tap_ && tap_->onStream == (stream) && ++tap_->launches)                                              \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, nullptr, tap_->stopEvent, 0, __VA_ARGS__);    \
        else                                                                                                        \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                    \
    } while (0)

inline int launchStatus(const char* what)
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRHIP_OK : hipfail(e, what);
}

} // namespace trhip
