// k_updateinstance.hip -- "updateinstanceconsts_CS_UpdateInstanceConstsAndBuildTLAS" for gfx950.
//
// Reference: source/shaders/updateinstanceconsts.hlsl:11-52, dispatched by
// UpdateInstanceConstsRenderer::Render (source/BasePassRenderers.cpp:125-162).  Per instance:
// TRS -> matrix, multiply up the parent chain, Prev = World, World = new.  The TLAS instance
// descriptor write (:38-52) is ray tracing and out of scope; u1 is accepted and ignored.
//
// HBM traffic per instance: 4 B node id + 48 B per hierarchy level + 64 B read + 128 B written; + 16 B read and 84 B
// written when the kernel also refreshes the transform-dependent entries of the instance cull cache (instance_cache.hip.h)
// -- otherwise the next cull pass rebuilds the whole cache: 300 B read + 200 B written per instance, every animated frame.
#include <cstddef>

#include "cull_math.hip.h"
#include "instance_cache.hip.h"
#include "trhip_internal.h"

using namespace interop;

namespace
{

struct M44 { float m[4][4]; };

// General 4x4 product as FMA chains (arithmetic convention; toyrenderer_common.hlsli mul()).
__device__ __forceinline__ M44 matmul(const M44& A, const M44& B)
{
    M44 C;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            C.m[i][j] = cm::fma_(A.m[i][3], B.m[3][j], cm::fma_(A.m[i][2], B.m[2][j], cm::fma_(A.m[i][1], B.m[1][j], A.m[i][0] * B.m[0][j])));
    return C;
}

// toyrenderer_common.hlsli:151-203: MakeWorldMatrix = mul(mul(R(q), S), T)
__device__ __forceinline__ M44 makeWorldMatrix(const NodeLocalTransform& t)
{
    const float qx = t.m_Rotation[0], qy = t.m_Rotation[1], qz = t.m_Rotation[2], qw = t.m_Rotation[3];
    const float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz;
    M44 R = {}, S = {}, T = {};
    R.m[0][0] = (1.f - 2.f * qyy) - 2.f * qzz;
    R.m[0][1] = (2.f * qx) * qy + (2.f * qz) * qw;
    R.m[0][2] = (2.f * qx) * qz - (2.f * qy) * qw;
    R.m[1][0] = (2.f * qx) * qy - (2.f * qz) * qw;
    R.m[1][1] = (1.f - 2.f * qxx) - 2.f * qzz;
    R.m[1][2] = (2.f * qy) * qz + (2.f * qx) * qw;
    R.m[2][0] = (2.f * qx) * qz + (2.f * qy) * qw;
    R.m[2][1] = (2.f * qy) * qz - (2.f * qx) * qw;
    R.m[2][2] = (1.f - 2.f * qxx) - 2.f * qyy;
    R.m[3][3] = 1.f;
    S.m[0][0] = t.m_Scale[0]; S.m[1][1] = t.m_Scale[1]; S.m[2][2] = t.m_Scale[2]; S.m[3][3] = 1.f;
    T.m[0][0] = 1.f; T.m[1][1] = 1.f; T.m[2][2] = 1.f; T.m[3][3] = 1.f;
    T.m[3][0] = t.m_Position[0]; T.m[3][1] = t.m_Position[1]; T.m[3][2] = t.m_Position[2];
    return matmul(matmul(R, S), T);
}

// A workgroup owns 256 consecutive instances = 36 864 contiguous bytes of the instance buffer.  The records are 144 bytes
// (two matrices + 16 bytes this pass never touches): a thread per instance reading / writing its own record moves 16 bytes
// per lane at a stride of 144 -- every wave instruction touches 64-72 lines for 1 KB of data, and every line is written in
// eight pieces (100 us on C3, 0.43 of what the bytes need).  Instead the block's records travel whole lines at a time through
// LDS: loaded with consecutive lanes on consecutive 16 bytes, updated in place in LDS (:35-36), stored the same way; the cull
// cache's 64-byte world blocks of the 256 instances (16 KB contiguous) leave through LDS as well.
#ifndef TR_UPD_BLOCK
#define TR_UPD_BLOCK 256
#endif
constexpr uint32_t kUpdBlock = TR_UPD_BLOCK;
constexpr uint32_t kRecVec = sizeof(BasePassInstanceConstants) / 16;            // 9 float4 per record
static_assert(sizeof(BasePassInstanceConstants) % 16 == 0 && offsetof(BasePassInstanceConstants, m_PrevWorldMatrix) == 64, "record layout");

__global__ __launch_bounds__(kUpdBlock) void updateInstanceConstsKernel(const NodeLocalTransform* __restrict__ nodes, uint32_t numNodes,
                                                                        const uint32_t* __restrict__ primToNode,
                                                                        BasePassInstanceConstants* instances, uint32_t first, uint32_t n,
                                                                        bool refreshCache, InstanceCullCache cache)
{
    __shared__ float4 s_rec[kUpdBlock * kRecVec];                                // the block's instance records
    __shared__ float4 s_world[kUpdBlock * 4];                                    // the block's cull-cache world blocks
    const uint32_t tid = threadIdx.x;
    const uint32_t b0 = blockIdx.x * kUpdBlock;
    const uint32_t cnt = n - b0 < kUpdBlock ? n - b0 : kUpdBlock;                // (the grid covers the n instances [first, first + n))
    const uint32_t i0 = first + b0;
    const uint32_t i = i0 + tid;
    const bool mine = tid < cnt;
    // requests first: the node id (and, behind it, the node), the block's records
    const uint32_t nodeID = mine ? primToNode[i] : 0xFFFFFFFFu;                  // :19
    float4* g = reinterpret_cast<float4*>(instances + i0);
    float4 v[kRecVec];
#pragma unroll
    for (uint32_t k = 0; k < kRecVec; ++k) {
        const uint32_t e = k * kUpdBlock + tid;
        v[k] = e < cnt * kRecVec ? g[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool valid = mine && nodeID < numNodes;                                // never read outside the node buffer
    NodeLocalTransform lt = {};
    float4 sph = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) lt = nodes[nodeID];                                               // :20
    if (valid && refreshCache) sph = cache.localSphere[i];
#pragma unroll
    for (uint32_t k = 0; k < kRecVec; ++k) s_rec[k * kUpdBlock + tid] = v[k];
    M44 world = {};
    if (valid) {
        world = makeWorldMatrix(lt);                                             // :22
        uint32_t parent = lt.m_ParentNodeIdx;                                    // :24
        uint32_t guard = 0;
        while (parent != 0xFFFFFFFFu && parent < numNodes && guard++ < 1024) {  // :25-32 (bounded: a cyclic hierarchy must not hang the GPU)
            const NodeLocalTransform pt = nodes[parent];
            world = matmul(world, makeWorldMatrix(pt));
            parent = pt.m_ParentNodeIdx;
        }
    }
    __syncthreads();
    if (valid) {
        float4* rec = s_rec + tid * kRecVec;
        float4 rows[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rows[r] = make_float4(world.m[r][0], world.m[r][1], world.m[r][2], world.m[r][3]);
            rec[4 + r] = rec[r];                                                 // :35 Prev = World
            rec[r] = rows[r];                                                    // :36 World = new
        }
        // the cull cache's view of this instance, from the same four rows the cache builder would read back
        if (refreshCache) instanceCacheWriteTransformPart(cache, i, rows[0], rows[1], rows[2], rows[3], sph, s_world + 4u * tid);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kRecVec; ++k) {
        const uint32_t e = k * kUpdBlock + tid;
        if (e < cnt * kRecVec) g[e] = s_rec[e];
    }
    if (refreshCache) {
        // An instance without a valid node keeps its cache entry: only whole blocks of valid instances leave through LDS.
        float4* cw = const_cast<float4*>(cache.world) + 4ull * i0;
        const bool all = __syncthreads_and(valid || !mine) != 0;
        if (all) {
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t e = k * kUpdBlock + tid;
                if (e < cnt * 4u) cw[e] = s_world[e];
            }
        } else if (valid) {
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) cw[4u * tid + k] = s_world[4u * tid + k];
        }
    }
}

int recordUpdateInstanceConsts(trhip::DispatchCtx& ctx)
{
    // BasePassRenderers.cpp:134-151
    // the reference's constants are { m_NumInstances }; a second word, { m_NumInstances, m_FirstInstance } (this build's host
    // side, sharded scenes: a rank updates the instances it culls, not the whole replicated table), moves the range
    const UpdateInstanceConstsShardConstants* ks = (const UpdateInstanceConstsShardConstants*)ctx.constants(0, sizeof(UpdateInstanceConstsShardConstants));
    UpdateInstanceConstsShardConstants kk = { 0u, 0u };
    if (ks) kk = *ks;
    else if (const UpdateInstanceConstsPassConstants* k1 = (const UpdateInstanceConstsPassConstants*)ctx.constants(0, sizeof(UpdateInstanceConstsPassConstants))) kk.m_NumInstances = k1->m_NumInstances;
    else TRHIP_REQUIRE(false, "%s: push constants (UpdateInstanceConstsPassConstants) missing", ctx.shaderName);
    const UpdateInstanceConstsShardConstants* k = &kk;
    const uint32_t first = kk.m_FirstInstance;
    trhip_buffer_t* nodes = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 0);
    trhip_buffer_t* primToNode = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 1);
    trhip_buffer_t* instances = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 0);
    TRHIP_REQUIRE(nodes && primToNode && instances, "%s: needs SRVs t0 (nodes), t1 (prim->node) and UAV u0 (instances)", ctx.shaderName);
    TRHIP_REQUIRE(nodes->byteSize % sizeof(NodeLocalTransform) == 0, "%s: node buffer size is not a multiple of 48", ctx.shaderName);
    TRHIP_REQUIRE(((uint64_t)first + k->m_NumInstances) * 4 <= primToNode->byteSize, "%s: the instance range exceeds the prim->node buffer", ctx.shaderName);
    TRHIP_REQUIRE(((uint64_t)first + k->m_NumInstances) * sizeof(BasePassInstanceConstants) <= instances->byteSize, "%s: the instance range exceeds the instance buffer", ctx.shaderName);
    TRHIP_REQUIRE(!ctx.indirect, "%s: dispatched directly", ctx.shaderName);
    const uint64_t threads = (uint64_t)ctx.gx * kNumThreadsPerWave;
    const uint32_t n = threads < k->m_NumInstances ? (uint32_t)threads : k->m_NumInstances;
    if (n == 0) return TRHIP_OK;
    const NodeLocalTransform* np = (const NodeLocalTransform*)nodes->ptr;
    const uint32_t numNodes = (uint32_t)(nodes->byteSize / sizeof(NodeLocalTransform));
    const uint32_t* pn = (const uint32_t*)primToNode->ptr;
    BasePassInstanceConstants* ip = (BasePassInstanceConstants*)instances->ptr;
    ctx.emit("main", [=](hipStream_t s) {
        // At submission time: this command's write has already been counted in the buffer's version.  If the cull cache
        // was current for the contents this command replaces, the kernel keeps it current (the entries that depend on
        // the world matrix are re-written from the new one; LOD tables and mesh-space spheres do not change), and the
        // next cull pass finds nothing to rebuild.  A changed mesh buffer is still caught there (its own version).
        const uint64_t now = instances->version;
        const bool refresh = instances->cullCache && instances->cullCacheMesh && instances->cullCacheInstVersion + 1 == now &&
                             instances->cullCacheBytes >= (instances->byteSize / sizeof(BasePassInstanceConstants)) * kInstanceCacheBytesPerInstance &&
                             !getenv("TRHIP_NO_CACHE_REFRESH");
        const InstanceCullCache cache = refresh ? instanceCacheLayout(instances->cullCache, instances->byteSize / sizeof(BasePassInstanceConstants)) : InstanceCullCache{};
        TRHIP_LAUNCH(updateInstanceConstsKernel, dim3((n + kUpdBlock - 1) / kUpdBlock), dim3(kUpdBlock), 0, s, np, numNodes, pn, ip, first, n, refresh, cache);
        if (refresh) instances->cullCacheInstVersion = now;
        return trhip::launchStatus("updateInstanceConstsKernel"); });
    return TRHIP_OK;
}

trhip::ShaderRegistrar r0("updateinstanceconsts_CS_UpdateInstanceConstsAndBuildTLAS", recordUpdateInstanceConsts, 0);

} // namespace
