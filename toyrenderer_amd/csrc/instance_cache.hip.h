// instance_cache.hip.h -- the INSTANCE CULL CACHE (trhip_buffer_t::cullCache of an instance buffer): a compact
// SoA restatement, over instance ids, of what the cull passes read per instance.  Built by instanceCacheKernel
// (k_gpuculling.hip) whenever the instance or the mesh buffer has been written since (version counters,
// trhip_internal.h).  Same arithmetic as the direct path, so every cached value is bit-identical to what the
// kernels would compute from the AoS records (BasePassInstanceConstants 144 B, MeshData 156 B).
#pragma once

#include <hip/hip_runtime.h>

#include "ShaderInterop.h"

struct InstanceCullCache
{
    const float4* sphere;           // world-space bounding sphere: TransformBoundingSphereToWorld (gpuculling.hlsl:116)
    const float4* world;            // [id][4]: ONE 64-byte block per instance -- world matrix rows 0..3 (xyz) as 12 consecutive
                                    // floats, then {maxScale, 0, 0, 0}: everything the meshlet cull reads per record
    const float* maxScale;          // toyrenderer_common.hlsli:134-140
    const uint32_t* numLODs;
    const uint2* lodInfo;           // [lod][id]  {m_NumMeshlets, m_MeshletDataBufferIdx}: one plane per LOD, so that the 8 bytes
                                    // of the selected LOD are a coalesced read over consecutive ids (as [id][lod] a lane pulled
                                    // a 64-byte granule for them: 2/3 of the bytes classify moved on C3)
    const float* error;             // [lod][id]
    const float4* localSphere;      // [id] the mesh's bounding sphere in mesh space: lets the transform update refresh the
                                    // transform-dependent entries (world, sphere, maxScale) without the mesh table
    uint64_t stride;                // ids per plane

    __host__ __device__ const uint2& lod(uint32_t id, uint32_t l) const { return lodInfo[(uint64_t)l * stride + id]; }
    __host__ __device__ const float& err(uint32_t id, uint32_t l) const { return error[(uint64_t)l * stride + id]; }
};

constexpr uint64_t kInstanceCacheBytesPerInstance = 64 + 16 + 4 + 4 + 3 * 4 * interop::kMaxNumMeshLODs + 16;

inline InstanceCullCache instanceCacheLayout(void* base, uint64_t n)
{
    char* p = (char*)base;
    InstanceCullCache c;
    c.world = (const float4*)p;             p += 64 * n;          // first: 64-byte aligned blocks
    c.sphere = (const float4*)p;            p += 16 * n;
    c.maxScale = (const float*)p;           p += 4 * n;
    c.numLODs = (const uint32_t*)p;         p += 4 * n;
    c.lodInfo = (const uint2*)p;            p += 8ull * interop::kMaxNumMeshLODs * n;
    c.error = (const float*)p;              p += 4ull * interop::kMaxNumMeshLODs * n;
    c.localSphere = (const float4*)p;
    c.stride = n;
    return c;
}

// The entries of instance i that depend on its world matrix (rows w0..w3 as stored in BasePassInstanceConstants), from
// the mesh-space bounding sphere: what instanceCacheKernel writes, and what the transform update re-writes in place
// (k_updateinstance.hip) so that an animated frame does not re-read 300 B per instance to rebuild the whole cache.
// Needs cull_math.hip.h before this header.
// worldOut: where the 64-byte world block goes (default: its place in the cache; the transform update stages a workgroup's
// blocks in LDS and stores them whole lines at a time).
__device__ __forceinline__ void instanceCacheWriteTransformPart(const InstanceCullCache& c, uint32_t i, float4 w0, float4 w1, float4 w2, float4 w3, float4 sph,
                                                                float4* worldOut = nullptr)
{
    const cm::M43 W = { { w0.x, w0.y, w0.z }, { w1.x, w1.y, w1.z }, { w2.x, w2.y, w2.z }, { w3.x, w3.y, w3.z } };
    const float ms = cm::maxScale(W.r0, W.r1, W.r2);                                 // toyrenderer_common.hlsli:134-140
    const cm::F3 wc = cm::mulPoint({ sph.x, sph.y, sph.z }, W);                      // gpuculling.hlsl:116 TransformBoundingSphereToWorld
    const_cast<float4*>(c.sphere)[i] = make_float4(wc.x, wc.y, wc.z, sph.w * ms);
    float4* wr = worldOut ? worldOut : const_cast<float4*>(c.world) + 4ull * i;
    wr[0] = make_float4(w0.x, w0.y, w0.z, w1.x);
    wr[1] = make_float4(w1.y, w1.z, w2.x, w2.y);
    wr[2] = make_float4(w2.z, w3.x, w3.y, w3.z);
    wr[3] = make_float4(ms, 0.f, 0.f, 0.f);
    const_cast<float*>(c.maxScale)[i] = ms;
}
