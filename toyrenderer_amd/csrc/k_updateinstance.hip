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

__global__ __launch_bounds__(256) void updateInstanceConstsKernel(const NodeLocalTransform* __restrict__ nodes, uint32_t numNodes,
                                                                  const uint32_t* __restrict__ primToNode,
                                                                  BasePassInstanceConstants* instances, uint32_t n,
                                                                  bool refreshCache, InstanceCullCache cache)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;                                                           // :13-16
    const uint32_t nodeID = primToNode[i];                                        // :19
    if (nodeID >= numNodes) return;                                               // never read outside the node buffer
    NodeLocalTransform lt = nodes[nodeID];                                        // :20
    M44 world = makeWorldMatrix(lt);                                              // :22
    uint32_t parent = lt.m_ParentNodeIdx;                                         // :24
    uint32_t guard = 0;
    while (parent != 0xFFFFFFFFu && parent < numNodes && guard++ < 1024) {       // :25-32 (bounded: a cyclic hierarchy must not hang the GPU)
        const NodeLocalTransform pt = nodes[parent];
        world = matmul(world, makeWorldMatrix(pt));
        parent = pt.m_ParentNodeIdx;
    }
    float4* w = reinterpret_cast<float4*>(&instances[i].m_WorldMatrix);
    float4* p = reinterpret_cast<float4*>(&instances[i].m_PrevWorldMatrix);
    float4 rows[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rows[r] = make_float4(world.m[r][0], world.m[r][1], world.m[r][2], world.m[r][3]);
        p[r] = w[r];                                                              // :35
        w[r] = rows[r];                                                           // :36
    }
    // the cull cache's view of this instance, from the same four rows the cache builder would read back
    if (refreshCache) instanceCacheWriteTransformPart(cache, i, rows[0], rows[1], rows[2], rows[3], cache.localSphere[i]);
}

int recordUpdateInstanceConsts(trhip::DispatchCtx& ctx)
{
    // BasePassRenderers.cpp:134-151
    const UpdateInstanceConstsPassConstants* k = (const UpdateInstanceConstsPassConstants*)ctx.constants(0, sizeof(UpdateInstanceConstsPassConstants));
    TRHIP_REQUIRE(k, "%s: push constants (UpdateInstanceConstsPassConstants) missing", ctx.shaderName);
    trhip_buffer_t* nodes = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 0);
    trhip_buffer_t* primToNode = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 1);
    trhip_buffer_t* instances = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 0);
    TRHIP_REQUIRE(nodes && primToNode && instances, "%s: needs SRVs t0 (nodes), t1 (prim->node) and UAV u0 (instances)", ctx.shaderName);
    TRHIP_REQUIRE(nodes->byteSize % sizeof(NodeLocalTransform) == 0, "%s: node buffer size is not a multiple of 48", ctx.shaderName);
    TRHIP_REQUIRE((uint64_t)k->m_NumInstances * 4 <= primToNode->byteSize, "%s: m_NumInstances exceeds the prim->node buffer", ctx.shaderName);
    TRHIP_REQUIRE((uint64_t)k->m_NumInstances * sizeof(BasePassInstanceConstants) <= instances->byteSize, "%s: m_NumInstances exceeds the instance buffer", ctx.shaderName);
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
        hipLaunchKernelGGL(updateInstanceConstsKernel, dim3((n + 255) / 256), dim3(256), 0, s, np, numNodes, pn, ip, n, refresh, cache);
        if (refresh) instances->cullCacheInstVersion = now;
        return trhip::launchStatus("updateInstanceConstsKernel"); });
    return TRHIP_OK;
}

trhip::ShaderRegistrar r0("updateinstanceconsts_CS_UpdateInstanceConstsAndBuildTLAS", recordUpdateInstanceConsts, 0);

} // namespace
