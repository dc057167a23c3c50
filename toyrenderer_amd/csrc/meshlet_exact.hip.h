// meshlet_exact.hip.h -- the arguments of the meshlet cull (basepass_AS_Main) and the EXACT evaluation of one meshlet from global
// memory alone: what the deferred mode's passes, the texel kernel's fallbacks and its short passes run (k_basepass_as.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "ShaderInterop.h"
#include "cull_math.hip.h"
#include "instance_cache.hip.h"

using namespace interop;

// The MESHLET CULL STREAM: per meshlet the cull reads the bounding sphere (16 B) and the cone word (4 B) of the 32-byte
// MeshletData record (basepass.hlsl:65, :92) -- the other 12 bytes are the mesh shader's.  The kernel's pace is set by the
// L1 misses a CU can keep in flight (its 256 lines pending most of the time: TCP_PENDING_STALL 65 %), i.e. by the NUMBER OF
// LINES it asks for: a derived copy of just those 20 bytes, spheres and cone words as two dense arrays, is 10 lines per
// 64-meshlet step instead of 16.  Back-end private, like the instance cull cache: built on the device from the bound meshlet
// buffer, rebuilt when that buffer's version moves (meshlets are static per mesh: in practice once per upload), 20 bytes
// per meshlet of extra memory (C4: 20 GB next to the 32 GB of MeshletData).
struct MeshletCullStream
{
    const float4* sphere;                         // [numMeshlets]
    const uint32_t* cone;                         // [numMeshlets]
};

struct MeshletCullArgs
{
    BasePassConstants k;
    const MeshletData* meshlets;
    MeshletCullStream stream;                     // what the cull reads of them: spheres and cone words as dense arrays (below)
    const MeshletAmplificationData* records;
    cm::Hzb hzb;
    cm::HzbQuad quad;                             // footprint-min table of hzb (k_hzb.hip)
    const uint32_t* dispatchArgs;                 // {X,1,1[,validRecords]} read on the device
    uint32_t argsWords;
    uint32_t recordCapacity;
    uint64_t numMeshlets;                         // size of the meshlet buffer (bounds check)
    uint32_t* visMask;
    uint32_t* visibleList; uint32_t listCapacity;
    uint32_t* drawArgs;
    // scratch
    uint32_t* batchSum;                           // per batch of 64 groups: visible meshlets
    uint32_t* superSum;                           // per 256 batches: visible meshlets -> exclusive prefix after the super scan
    uint32_t maxBatches;
    // optional processing order written by the instance pass into the record buffer's sidecar
    // (k_gpuculling.hip): {valid, count} header + a permutation of [0, count) sorted by screen tile
    const uint32_t* permHeader;
    const uint4* perm;                            // {record index, instance, first meshlet, count} in processing order
    InstanceCullCache cache;                      // world matrix, max scale, LOD table per instance (instance_cache.hip.h)
    uint32_t numInstances;
    // The group count the list build works on: written by the cull kernel (G of groupCount()), read by count / scan / expand /
    // compact INSTEAD of the dispatch arguments.  The list build of a large pass runs on the side stream, past the end of the
    // frame's main chain; the next frame starts by clearing the dispatch arguments (BasePassRenderers.cpp:322-332), and with
    // the arguments as the list build's input that clear -- the head of the next frame's chain -- had to wait for it (a
    // cross-stream join, exposed: ~20 us per frame on C3, profiles/r4/experiments.md section 2).
    uint32_t* listGroups;
    cm::ProjBands bands;                          // filtered projection: preconditions and bands (cm::projBands, computed on the host)
    // texel kernel: a pass of at most this many records per half-wave of the grid skips the batch machinery (every half-wave
    // evaluates its records exactly from global memory); 0 = never (tests: TRHIP_NO_SHORT_PASS=1)
    uint32_t shortPassRounds;
};

__device__ __forceinline__ uint32_t groupCount(const MeshletCullArgs& a)
{
    uint32_t G = a.dispatchArgs[0];
    if (a.argsWords > 3 && a.dispatchArgs[3] < G) G = a.dispatchArgs[3];   // Q2: only the defined prefix
    return G < a.recordCapacity ? G : a.recordCapacity;
}

// Exact visibility of meshlet `mi` of instance `cid`, from global memory alone (the path of the texel kernel: every test with the
// compiler's correctly rounded sequences, basepass.hlsl:65-108).  What the deferred mode's re-evaluations run.
template <bool FRUSTUM, bool OCCLUSION, bool CONE>
__device__ __forceinline__ bool exactMeshletAt(const MeshletCullArgs& a, const cm::M43P& VP, const cm::M33P& VR, uint32_t cid, uint64_t mi)
{
    const float4* wr = a.cache.world + 4ull * cid;
    const float4* p = reinterpret_cast<const float4*>(a.meshlets + mi);
    const float4 q0 = wr[0], q1 = wr[1], q2 = wr[2], q3 = wr[3];
    const float4 sphere = p[0];
    const uint32_t cone = __float_as_uint(p[1].x);
    const cm::F3 r0 = { q0.x, q0.y, q0.z }, r1 = { q0.w, q1.x, q1.y }, r2 = { q1.z, q1.w, q2.x };
    const cm::M43P W = cm::packM43(cm::M43{ r0, r1, r2, { q2.y, q2.z, q2.w } });
    const cm::M33P adj = cm::rot(cm::packM43(cm::M43{ cm::cross3(r1, r2), cm::cross3(r2, r0), cm::cross3(r0, r1), { 0.f, 0.f, 0.f } }));
    const cm::F3 cv = cm::toViewP(cm::mulPointP({ sphere.x, sphere.y, sphere.z }, W), VP);
    const float rad = sphere.w * q3.x;
    bool vis = true;
    if (FRUSTUM) vis &= cm::frustumVisible(cv, rad, a.k.m_Frustum.x, a.k.m_Frustum.y, a.k.m_Frustum.z, a.k.m_Frustum.w);
    if (OCCLUSION) vis &= cm::occlusionVisible(cv, rad, a.k.m_NearPlane, a.k.m_P00, a.k.m_P11, a.hzb);
    float unused;
    if (CONE) vis &= !cm::coneBackfacingP(cone, cv, rad, adj, VR, 1.0f, 1.0f, &unused);
    return vis;
}
// ... of lane m of record g (basepass.hlsl:52-63 first)
template <bool FRUSTUM, bool OCCLUSION, bool CONE>
__device__ __forceinline__ bool exactMeshletVisible(const MeshletCullArgs& a, const cm::M43P& VP, const cm::M33P& VR, uint32_t g, uint32_t m)
{
    const MeshletAmplificationData rec = a.records[g];
    const uint32_t cid = rec.m_InstanceConstIdx < a.numInstances ? rec.m_InstanceConstIdx : 0u;
    const uint32_t lodIdx = rec.m_MeshLOD < kMaxNumMeshLODs ? rec.m_MeshLOD : kMaxNumMeshLODs - 1u;
    const uint2 li = a.cache.lod(cid, lodIdx);
    const uint32_t off = rec.m_MeshletGroupOffset;
    uint32_t cnt = li.x > off ? li.x - off : 0u;
    const uint64_t base = (uint64_t)li.y + off;
    cnt = cnt < 32u ? cnt : 32u;
    if (base + cnt > a.numMeshlets) cnt = 0;
    if (m >= cnt) return false;
    return exactMeshletAt<FRUSTUM, OCCLUSION, CONE>(a, VP, VR, cid, base + m);
}

