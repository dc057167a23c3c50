// ShardExchange.cpp -- native per-frame driver of the multi-GPU exchange (not in the reference: single GPU,
// GraphicRHI.cpp:165).  Protocol, slot layout (header, run entries, lane masks) and rationale: toyrenderer_amd/gather.py (the Python statement of the
// same thing, kept for tests) and DESIGN.md section 6.  Everything a frame needs is done here without leaving native
// code: pack (compute stream) -> one all-gather -> unpack + list rebuild (exchange stream), double-buffered and
// ordered by events; and the in-frame late-count exchange (phase 0 after the early instance cull on an auxiliary
// stream, phase 1 = the compute stream waits for it before the late instance cull).
//
// The collectives are function pointers (trhost_allgather_fn): the caller binds ncclAllGather on its own
// communicators (trhost_rccl_allgather below is the ready-made binding), tests bind a host-staged gloo gather.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/trhip.h"
#include "../../../include/trhost.h"
#include "Graphic.h"
#include "VisibilityOutputs.h"
#include "nvrhi_lite.h"

namespace
{

constexpr uint32_t kHeaderWords = 16;
constexpr uint32_t kMaxPassSlots = 4;

void require(int rc, const char* what) { nvrhi::throwIfFailed(rc, what); }

struct Exchange
{
    trhost_exchange_desc desc{};
    uint32_t slotWords = 0;
    int deviceIndex = 0;
    trhip_device compute = nullptr;                 // the renderer's device (not owned)
    trhip_device commDev = nullptr;                 // submits on commStream
    void* commStream = nullptr;
    void* auxStream = nullptr;
    void* packed[2] = {}, *released[2] = {};        // events
    void* latePosted[2] = {}, *lateReady[2] = {};
    trhip_buffer send[2] = {}, recv[2] = {};
    trhip_buffer packState = nullptr;               // the pack kernel's ticket / status words (it leaves them zero)
    uint32_t slotRuns = 0;
    trhip_buffer sendOnCompute[2] = {};              // the same memory, wrapped for the compute device's lists
    trhip_buffer lateCounts[2] = {};                 // gathered late counts per bucket (world words)
    trhip_buffer records[kMaxPassSlots] = {}, masks[kMaxPassSlots] = {}, list[kMaxPassSlots] = {}, args[kMaxPassSlots] = {};
    trhip_cmdlist packList[2] = {}, unpackList[2] = {};
    void* packKey[2][kMaxPassSlots][4] = {};
    bool packRecorded[2] = {};
    uint64_t frame = 0;
    std::string error;

    bool wants(uint32_t s) const { return (desc.pass_slot_mask >> s) & 1u; }
};

std::unique_ptr<Exchange> g_Exchange;

trhip_buffer makeBuffer(trhip_device dev, uint64_t bytes, const char* name)
{
    trhip_buffer_desc d{};
    d.byteSize = bytes < 4 ? 4 : bytes;
    d.structStride = 4;
    d.canHaveUAVs = 1;
    d.debugName = name;
    trhip_buffer b = nullptr;
    require(trhip_buffer_create(dev, &d, &b), name);
    return b;
}

trhip_buffer wrapBuffer(trhip_device dev, trhip_buffer src, const char* name)
{
    trhip_buffer_desc d{};
    d.byteSize = trhip_buffer_size(src);
    d.structStride = 4;
    d.canHaveUAVs = 1;
    d.debugName = name;
    trhip_buffer b = nullptr;
    require(trhip_buffer_wrap(dev, trhip_buffer_device_ptr(src), &d, &b), name);
    return b;
}

trhip_binding bind(uint32_t type, uint32_t slot, void* resource)
{
    trhip_binding b{};
    b.type = type; b.slot = slot; b.resource = resource;
    return b;
}

void recordUnpack(Exchange& x, int b)
{
    std::vector<trhip_binding> binds;
    binds.push_back(bind(TRHIP_BIND_PUSH_CONSTANTS, 0, nullptr));
    binds.push_back(bind(TRHIP_BIND_STRUCTURED_SRV, 0, x.recv[b]));
    for (uint32_t s = 0; s < kMaxPassSlots; ++s)
        if (x.wants(s)) {
            binds.push_back(bind(TRHIP_BIND_STRUCTURED_UAV, 4 * s, x.records[s]));
            binds.push_back(bind(TRHIP_BIND_STRUCTURED_UAV, 4 * s + 1, x.masks[s]));
            binds.push_back(bind(TRHIP_BIND_STRUCTURED_UAV, 4 * s + 2, x.list[s]));
            binds.push_back(bind(TRHIP_BIND_STRUCTURED_UAV, 4 * s + 3, x.args[s]));
        }
    const uint32_t push[4] = { x.desc.world, x.desc.slot_groups, x.slotRuns, x.desc.global_group_capacity };
    require(trhip_cmd_open(x.unpackList[b]), "exchange: open unpack list");
    require(trhip_cmd_dispatch(x.unpackList[b], "visibility_CS_UnpackShards", binds.data(), (uint32_t)binds.size(), push, sizeof push, 1, 1, 1),
            "exchange: record visibility_CS_UnpackShards");
    require(trhip_cmd_close(x.unpackList[b]), "exchange: close unpack list");
}

// The render graph hands out the same buffers frame after frame: the pack list is re-recorded only when they change.
void recordPackIfNeeded(Exchange& x, int b)
{
    void* key[kMaxPassSlots][4] = {};
    for (uint32_t s = 0; s < kMaxPassSlots; ++s) {
        if (!x.wants(s)) continue;
        VisibilityPassBuffers pb;
        if (!GetVisibilityPassBuffers(s, &pb) || !pb.m_bRan) continue;
        key[s][0] = pb.m_MeshletAmplificationDataBuffer->native();
        key[s][1] = pb.m_MeshletVisibilityMaskBuffer->native();
        key[s][2] = pb.m_MeshletDispatchArgumentsBuffer->native();
        key[s][3] = pb.m_VisibleMeshletDrawArgsBuffer->native();
    }
    if (x.packRecorded[b] && memcmp(key, x.packKey[b], sizeof key) == 0) return;
    std::vector<trhip_binding> binds;
    binds.push_back(bind(TRHIP_BIND_PUSH_CONSTANTS, 0, nullptr));
    binds.push_back(bind(TRHIP_BIND_STRUCTURED_UAV, 0, x.sendOnCompute[b]));
    binds.push_back(bind(TRHIP_BIND_STRUCTURED_UAV, 1, x.packState));
    for (uint32_t s = 0; s < kMaxPassSlots; ++s)
        if (key[s][0])
            for (uint32_t k = 0; k < 4; ++k) binds.push_back(bind(TRHIP_BIND_STRUCTURED_SRV, 4 * s + k, key[s][k]));
    const uint32_t push[2] = { x.desc.slot_groups, x.slotRuns };
    require(trhip_cmd_open(x.packList[b]), "exchange: open pack list");
    require(trhip_cmd_dispatch(x.packList[b], "visibility_CS_PackShard", binds.data(), (uint32_t)binds.size(), push, sizeof push, 1, 1, 1),
            "exchange: record visibility_CS_PackShard");
    require(trhip_cmd_close(x.packList[b]), "exchange: close pack list");
    memcpy(x.packKey[b], key, sizeof key);
    x.packRecorded[b] = true;
}

// trhost.h trhost_set_shard_late_exchange contract, natively.
void lateHook(void* user, void* computeStream, void* lateCount, void* shardInfo, int bucket, int phase)
{
    Exchange& x = *(Exchange*)user;
    if (bucket < 0 || bucket > 1) return;
    int rc = TRHIP_OK;
    if (phase == 0) {
        rc = trhip_event_record(x.latePosted[bucket], computeStream);
        if (rc == TRHIP_OK) rc = trhip_stream_wait_event(x.auxStream, x.latePosted[bucket]);
        if (rc == TRHIP_OK && x.desc.late_allgather(x.desc.late_user, lateCount, trhip_buffer_device_ptr(x.lateCounts[bucket]), 1, x.auxStream) != 0) {
            x.error = "late-count all-gather failed";
            return;
        }
        if (rc == TRHIP_OK)
            rc = trhip_launch_shard_late_info(x.auxStream, (const uint32_t*)trhip_buffer_device_ptr(x.lateCounts[bucket]), x.desc.world, x.desc.rank,
                                              (uint32_t*)shardInfo);
        if (rc == TRHIP_OK) rc = trhip_event_record(x.lateReady[bucket], x.auxStream);
    } else {
        rc = trhip_stream_wait_event(computeStream, x.lateReady[bucket]);
    }
    if (rc != TRHIP_OK) x.error = std::string("late-count exchange: ") + trhip_last_error();
}

void destroy()
{
    if (!g_Exchange) return;
    Exchange& x = *g_Exchange;
    SetShardLateExchange(nullptr, nullptr);
    if (x.compute) (void)trhip_device_wait_idle(x.compute);
    if (x.commStream) (void)trhip_stream_synchronize(x.commStream);
    if (x.auxStream) (void)trhip_stream_synchronize(x.auxStream);
    for (int b = 0; b < 2; ++b) {
        if (x.packList[b]) trhip_cmd_release(x.packList[b]);
        if (x.unpackList[b]) trhip_cmd_release(x.unpackList[b]);
        for (trhip_buffer buf : { x.send[b], x.recv[b], x.sendOnCompute[b], x.lateCounts[b] })
            if (buf) trhip_buffer_release(buf);
        for (void* e : { x.packed[b], x.released[b], x.latePosted[b], x.lateReady[b] }) trhip_event_destroy(e);
    }
    for (uint32_t s = 0; s < kMaxPassSlots; ++s)
        for (trhip_buffer buf : { x.records[s], x.masks[s], x.list[s], x.args[s] })
            if (buf) trhip_buffer_release(buf);
    if (x.packState) trhip_buffer_release(x.packState);
    if (x.commDev) trhip_device_destroy(x.commDev);
    trhip_stream_destroy(x.commStream);
    trhip_stream_destroy(x.auxStream);
    g_Exchange.reset();
}

} // namespace

void ShardExchangeCreate(const trhost_exchange_desc& d)
{
    destroy();
    check(d.world >= 1 && d.rank < d.world && d.slots_allgather && d.late_allgather && d.pass_slot_mask != 0 && d.pass_slot_mask < 16);
    const uint32_t slotRuns = d.slot_runs ? d.slot_runs : d.slot_groups;
    check((uint64_t)d.world * (kHeaderWords + 4ull * slotRuns + d.slot_groups) < (1ull << 32));
    g_Exchange = std::make_unique<Exchange>();
    Exchange& x = *g_Exchange;
    x.desc = d;
    x.slotRuns = slotRuns;
    x.slotWords = kHeaderWords + 4u * slotRuns + d.slot_groups;
    x.compute = g_Graphic.m_NVRHIDevice->native();
    x.deviceIndex = g_Graphic.m_DeviceIndex;
    // the late-count exchange is tiny and the frame waits for it: highest priority; the slot exchange is background work
    // behind the next frame: lowest.  Different classes than the renderer's stream = hardware queues of their own.
    require(trhip_stream_create_priority(x.deviceIndex, -1, &x.auxStream), "exchange: aux stream");
    if (d.overlap) {
        require(trhip_stream_create_priority(x.deviceIndex, +1, &x.commStream), "exchange: stream");
        require(trhip_device_create_on_stream(x.deviceIndex, x.commStream, &x.commDev), "exchange: device on the exchange stream");
    } else {
        require(trhip_device_create_on_stream(x.deviceIndex, trhip_device_stream(x.compute), &x.commDev), "exchange: device on the compute stream");
    }
    const uint64_t groupCap = d.group_capacity ? d.group_capacity : (uint64_t)d.world * d.slot_groups;
    const uint64_t listCap = d.list_capacity ? d.list_capacity : 32ull * groupCap;
    check(groupCap <= (1u << 27));
    for (int b = 0; b < 2; ++b) {
        x.send[b] = makeBuffer(x.commDev, 4ull * x.slotWords, "ShardSlotSend");
        x.recv[b] = makeBuffer(x.commDev, 4ull * x.slotWords * d.world, "ShardSlotsRecv");
        x.sendOnCompute[b] = wrapBuffer(x.compute, x.send[b], "ShardSlotSend (compute)");
        x.lateCounts[b] = makeBuffer(x.commDev, 4ull * d.world, "GatheredLateCounts");
        require(trhip_event_create(x.deviceIndex, &x.packed[b]), "exchange: event");
        require(trhip_event_create(x.deviceIndex, &x.released[b]), "exchange: event");
        require(trhip_event_create(x.deviceIndex, &x.latePosted[b]), "exchange: event");
        require(trhip_event_create(x.deviceIndex, &x.lateReady[b]), "exchange: event");
        require(trhip_cmd_create(x.compute, &x.packList[b]), "exchange: pack list");
        require(trhip_cmd_create(x.commDev, &x.unpackList[b]), "exchange: unpack list");
    }
    for (uint32_t s = 0; s < kMaxPassSlots; ++s)
        if (x.wants(s)) {
            x.records[s] = makeBuffer(x.commDev, 12ull * groupCap, "AllRecords");
            x.masks[s] = makeBuffer(x.commDev, 4ull * groupCap, "AllMasks");
            x.list[s] = makeBuffer(x.commDev, 4ull * listCap, "AllVisibleList");
            x.args[s] = makeBuffer(x.commDev, 32, "AllArgs");
            const uint32_t zeros[8] = {};                                       // word 7 = status: read by ShardExchangeWait before any run wrote it
            require(trhip_buffer_upload(x.args[s], 0, zeros, sizeof zeros), "exchange: zero the argument words");
        }
    {   // visibility_CS_PackShard's state words (two halves): zero once, every launch zeroes the half the next one uses
        const uint64_t words = 2ull * (d.slot_groups / 1024u + kMaxPassSlots + 1u);
        x.packState = makeBuffer(x.compute, 8ull * words, "ShardPackState");
        std::vector<uint64_t> zero(words, 0);
        require(trhip_buffer_upload(x.packState, 0, zero.data(), 8ull * words), "exchange: zero the pack state");
    }
    for (int b = 0; b < 2; ++b) recordUnpack(x, b);
    SetShardLateExchange(&lateHook, &x, d.list_presence_mask, d.depth_allreduce_max, d.depth_user);
}

void ShardExchangeRun()
{
    check(g_Exchange);
    Exchange& x = *g_Exchange;
    if (!x.error.empty()) { std::string e; e.swap(x.error); throw nvrhi::Error("shard exchange: " + e); }
    const int b = (int)(x.frame & 1u);
    void* computeStream = trhip_device_stream(x.compute);
    void* commStream = trhip_device_stream(x.commDev);
    if (x.frame >= 2) require(trhip_stream_wait_event(computeStream, x.released[b]), "exchange: wait for the buffers");   // exchange frame-2 done
    recordPackIfNeeded(x, b);
    require(trhip_queue_execute(x.compute, &x.packList[b], 1), "exchange: pack");
    require(trhip_event_record(x.packed[b], computeStream), "exchange: packed");
    require(trhip_stream_wait_event(commStream, x.packed[b]), "exchange: wait for the pack");
    if (x.desc.slots_allgather(x.desc.slots_user, trhip_buffer_device_ptr(x.send[b]), trhip_buffer_device_ptr(x.recv[b]), x.slotWords, commStream) != 0)
        throw nvrhi::Error("shard exchange: slot all-gather failed");
    require(trhip_queue_execute(x.commDev, &x.unpackList[b], 1), "exchange: unpack");
    require(trhip_event_record(x.released[b], commStream), "exchange: released");
    ++x.frame;
}

void ShardExchangeWait()
{
    check(g_Exchange);
    require(trhip_device_wait_idle(g_Exchange->commDev), "exchange: wait");
    require(trhip_stream_synchronize(g_Exchange->auxStream), "exchange: wait (aux)");
    // A slot that overflowed, a corrupt header, a whole-scene buffer too small, a rank that dropped groups at its capacity
    // without a global capacity to cut at: the unpack flags them in word 7 of the pass slot's arguments (gather.py STATUS_*).
    // Such a frame must not be consumed silently.
    if (g_Exchange->frame == 0) return;                                      // nothing has run: nothing to check
    for (uint32_t s = 0; s < kMaxPassSlots; ++s) {
        if (!g_Exchange->wants(s) || !g_Exchange->args[s]) continue;
        uint32_t status = 0;
        require(trhip_buffer_download(g_Exchange->args[s], 7 * sizeof(uint32_t), &status, sizeof status), "exchange: read status");
        if (status != 0)
            throw nvrhi::Error("shard exchange: pass slot " + std::to_string(s) + " failed with status " + std::to_string(status) +
                               " (1 a rank's groups exceed slot_groups, 2 whole-scene capacity exceeded, 4 corrupt slot header, 8 a rank dropped groups at its capacity)");
    }
}

void ShardExchangeOutputs(uint32_t slot, void** records, void** masks, void** list, void** args)
{
    check(g_Exchange && slot < kMaxPassSlots && g_Exchange->wants(slot));
    *records = g_Exchange->records[slot]; *masks = g_Exchange->masks[slot]; *list = g_Exchange->list[slot]; *args = g_Exchange->args[slot];
}

void ShardExchangeDestroy() { destroy(); }

// Ready-made binding of trhost_exchange_desc.depth_allreduce_max to RCCL: user = { address of ncclAllReduce, ncclComm_t }.
// In place, ncclUint32 + ncclMax: reverse-Z depths are non-negative floats, which order like their bit patterns.
extern "C" int trhost_rccl_allreduce_max_u32(void* user, void* words, uint64_t countWords, void* hipStream)
{
    using AllReduce = int (*)(const void*, void*, size_t, int, int, void*, void*);
    void** u = (void**)user;
    return ((AllReduce)u[0])(words, words, (size_t)countWords, 3 /* ncclUint32 */, 2 /* ncclMax */, u[1], hipStream);
}

// Ready-made binding of trhost_allgather_fn to RCCL: user = { address of ncclAllGather, ncclComm_t }.
extern "C" int trhost_rccl_allgather(void* user, const void* send, void* recv, uint64_t countWords, void* hipStream)
{
    using AllGather = int (*)(const void*, void*, size_t, int, void*, void*);
    void** u = (void**)user;
    return ((AllGather)u[0])(send, recv, (size_t)countWords, 2 /* ncclInt32 */, u[1], hipStream);
}
