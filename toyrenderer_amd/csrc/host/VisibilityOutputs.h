// VisibilityOutputs.h -- where a consumer (software rasteriser, multi-GPU gather, tests) finds the
// buffers the last recorded frame's cull passes wrote.  New in this build: in the reference the
// mesh-shader stage consumes the amplification payload on chip and nothing is exposed.
#pragma once

#include "nvrhi_lite.h"

struct VisibilityPassBuffers
{
    bool m_bRan = false;
    nvrhi::BufferHandle m_MeshletAmplificationDataBuffer;   // MeshletAmplificationData[G]
    nvrhi::BufferHandle m_MeshletDispatchArgumentsBuffer;   // {G,1,1, validRecords}
    nvrhi::BufferHandle m_MeshletVisibilityMaskBuffer;      // uint32 per group: lane-visibility ballot
    nvrhi::BufferHandle m_VisibleMeshletListBuffer;         // (g << 5) | lane, canonical order
    nvrhi::BufferHandle m_VisibleMeshletDrawArgsBuffer;     // {numVisible,1,1}
    nvrhi::BufferHandle m_LateCullInstanceCountBuffer, m_LateCullDispatchIndirectArgsBuffer;
};

// slot: 0 early-opaque, 1 late-opaque, 2 early-alpha-mask, 3 late-alpha-mask
bool GetVisibilityPassBuffers(uint32_t slot, VisibilityPassBuffers* out);
void ReleaseVisibilityPassBuffers();

// Multi-GPU (instance list sharded over ranks; trhost.h trhost_set_shard_late_exchange): called while
// the frame is submitted: after each early instance cull (phase 0) and before each late one (phase 1).
using ShardLateFn = void (*)(void* user, void* hipStream, void* lateCount, void* shardInfo, int bucket, int phase);
// listPresenceMask: bit 0 = some rank holds opaque ids, bit 1 = some rank holds alpha-mask ids (0: this rank's own lists);
// every rank posts the in-frame collective of exactly those buckets.  depthFn (may be null): element-wise MAX of the depth
// words over all ranks, enqueued on hipStream -- required when the frame rasterises its own depth (m_bRasterDepth).
using ShardDepthFn = int (*)(void* user, void* depthWords, uint64_t countWords, void* hipStream);
void SetShardLateExchange(ShardLateFn fn, void* user, uint32_t listPresenceMask = 0, ShardDepthFn depthFn = nullptr, void* depthUser = nullptr);

// Outputs of the last GI probe culling dispatch (GIRenderer.cpp; false before the first one).
bool GetGIProbeCullBuffers(nvrhi::BufferHandle* positions, nvrhi::BufferHandle* drawArgs, nvrhi::BufferHandle* instanceToProbe);
void ReleaseGIProbeCullBuffers();

// Depth attachment of the last recorded base pass (read-back for tests; null before the first frame).
nvrhi::TextureHandle GetLastDepthBuffer();

// Native per-frame driver of the multi-GPU exchange (ShardExchange.cpp; C facade in trhost.h).
struct trhost_exchange_desc;
void ShardExchangeCreate(const trhost_exchange_desc& desc);
void ShardExchangeRun();
void ShardExchangeWait();
void ShardExchangeOutputs(uint32_t slot, void** records, void** masks, void** list, void** args);
void ShardExchangeDestroy();
