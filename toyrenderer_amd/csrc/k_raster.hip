// k_raster.hip -- "basepass_MS_Main_depth": depth of the visible meshlets of one pass slot, the compute stand-in for
// the mesh shader + fixed-function rasteriser + depth test of the reference's base pass
// (source/shaders/basepass.hlsl:124-188 MS_Main: vertex fetch through g_MeshletVertexIDsBuffer / g_MeshletIndexIDsBuffer,
// position * world * worldToClip; PSO BasePassRenderers.cpp:481-495, reverse-Z depth GREATER).  SURVEY.md 8(f) rank 1:
// it closes the two-phase loop with depth the path produced itself instead of a synthetic depth image.
//
// The rasteriser has no source to restate: its rules are this build's CONVENTION (parity unpinned), stated once in
// oracle/tr_oracle.h (orc_raster_depth) and followed here operation for operation -- no near clipping (a triangle with a
// vertex at w <= near is dropped), pixel-centre samples, inclusive edge functions on both windings, depth interpolated
// with one fma chain and one division, atomic max on the float bits (depth > 0, so unsigned order = float order).  The
// result is a maximum, so it does not depend on the order the triangles are drawn in: bit-exact against the oracle.
//
// One wave per visible meshlet: lanes transform the (<= 64) vertices into the wave's LDS slice, then the wave walks the
// triangles one after the other, 64 pixels of the bounding box per pass.  Sized for correctness first; the bound is
// the bounding-box area, not HBM.
#include "cull_math.hip.h"
#include "trhip_internal.h"

using namespace interop;

namespace
{

constexpr uint32_t kBlock = 256;
constexpr uint32_t kWaves = kBlock / 64;

struct RasterArgs
{
    BasePassConstants k;
    const BasePassInstanceConstants* instances; uint32_t numInstances;
    const MeshData* meshData; uint32_t numMeshes;
    const MeshletData* meshlets; uint64_t numMeshlets;
    const char* vertices; uint64_t numVertices;                 // RawVertexFormat, 20-byte stride
    const uint32_t* vertexIds; uint64_t numVertexIds;
    const uint32_t* triangles; uint64_t numTriangles;
    const MeshletAmplificationData* records; uint32_t recordCapacity;
    const uint32_t* visibleList; uint32_t listCapacity;
    const uint32_t* drawArgs;                                    // {numVisible, 1, 1}
    uint32_t* depth;                                             // R32F as bits
    uint32_t width, height;
};

__device__ __forceinline__ float edgeFn(float ax, float ay, float bx, float by, float px, float py)
{
    return cm::fma_(bx - ax, py - ay, -((by - ay) * (px - ax)));
}

__global__ __launch_bounds__(kBlock) void rasterDepthKernel(RasterArgs a)
{
    __shared__ float s_x[kWaves][64], s_y[kWaves][64], s_d[kWaves][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float* sx = s_x[wave]; float* sy = s_y[wave]; float* sd = s_d[wave];
    uint32_t V = a.drawArgs[0];
    V = V < a.listCapacity ? V : a.listCapacity;
    const uint32_t W = a.width, H = a.height;
    const float halfW = 0.5f * (float)W, halfH = 0.5f * (float)H;
    const cm::M43 clipXYZ = cm::loadM43(a.k.m_WorldToClip);
    for (uint32_t v = blockIdx.x * kWaves + wave; v < V; v += gridDim.x * kWaves) {
        const uint32_t e = a.visibleList[v], g = e >> 5, m = e & 31u;
        if (g >= a.recordCapacity) continue;
        const MeshletAmplificationData rec = a.records[g];                               // basepass.hlsl:138-142
        if (rec.m_InstanceConstIdx >= a.numInstances) continue;
        const BasePassInstanceConstants& inst = a.instances[rec.m_InstanceConstIdx];
        if (inst.m_MeshDataIdx >= a.numMeshes) continue;
        const uint32_t lodIdx = rec.m_MeshLOD < kMaxNumMeshLODs ? rec.m_MeshLOD : kMaxNumMeshLODs - 1u;
        const MeshLODData lod = a.meshData[inst.m_MeshDataIdx].m_MeshLODDatas[lodIdx];
        const uint64_t mi = (uint64_t)lod.m_MeshletDataBufferIdx + rec.m_MeshletGroupOffset + m;
        if (mi >= a.numMeshlets) continue;
        const MeshletData ml = a.meshlets[mi];
        uint32_t nv = ml.m_VertexAndTriangleCount & 0xFFu, nt = (ml.m_VertexAndTriangleCount >> 8) & 0xFFu;   // :144-145
        nv = nv < 64u ? nv : 64u;                                                        // kMaxMeshletVertices
        if ((uint64_t)ml.m_MeshletVertexIDsBufferIdx + nv > a.numVertexIds || (uint64_t)ml.m_MeshletIndexIDsBufferIdx + nt > a.numTriangles) continue;
        const cm::M43 Wm = cm::loadM43(inst.m_WorldMatrix);
        // ---- vertices (:149-158): lane l transforms vertex l ------------------------------------------------
        bool ok = false;
        if (lane < nv) {
            const uint32_t vid = a.vertexIds[ml.m_MeshletVertexIDsBufferIdx + lane];
            if (vid < a.numVertices) {
                const float* p = reinterpret_cast<const float*>(a.vertices + (uint64_t)vid * 20u);
                const cm::F3 wp = cm::mulPoint({ p[0], p[1], p[2] }, Wm);
                const cm::F3 c = cm::mulPoint(wp, clipXYZ);
                const float w = cm::fma_(wp.z, a.k.m_WorldToClip.m[2][3], cm::fma_(wp.y, a.k.m_WorldToClip.m[1][3], wp.x * a.k.m_WorldToClip.m[0][3])) + a.k.m_WorldToClip.m[3][3];
                ok = w > a.k.m_NearPlane;
                sx[lane] = cm::fma_(c.x / w, halfW, halfW);
                sy[lane] = cm::fma_(-(c.y / w), halfH, halfH);
                sd[lane] = c.z / w;
            }
        }
        const unsigned long long okMask = __ballot(ok);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- triangles (:178-187), one after the other; 64 pixels of the bounding box per pass -------------------
        for (uint32_t t = 0; t < nt; ++t) {
            const uint32_t packed = a.triangles[ml.m_MeshletIndexIDsBufferIdx + t];
            const uint32_t ia = packed & 0xFFu, ib = (packed >> 8) & 0xFFu, ic = (packed >> 16) & 0xFFu;
            if (ia >= nv || ib >= nv || ic >= nv) continue;
            if (!((okMask >> ia) & (okMask >> ib) & (okMask >> ic) & 1ull)) continue;
            const float x0 = sx[ia], y0 = sy[ia], x1 = sx[ib], y1 = sy[ib], x2 = sx[ic], y2 = sy[ic];
            const float d0 = sd[ia], d1 = sd[ib], d2 = sd[ic];
            const float area = edgeFn(x0, y0, x1, y1, x2, y2);
            if (!(area != 0.0f)) continue;                                               // degenerate or NaN
            const float sgn = area < 0.0f ? -1.0f : 1.0f;
            const float fminx = cm::min_(cm::min_(x0, x1), x2), fmaxx = cm::max_(cm::max_(x0, x1), x2);
            const float fminy = cm::min_(cm::min_(y0, y1), y2), fmaxy = cm::max_(cm::max_(y0, y1), y2);
            if (!(fmaxx >= 0.0f && fmaxy >= 0.0f && fminx <= (float)W && fminy <= (float)H)) continue;   // off screen or NaN
            const int bx0 = (int)cm::max_(__builtin_floorf(fminx), 0.0f), bx1 = (int)cm::min_(__builtin_ceilf(fmaxx), (float)(W - 1));
            const int by0 = (int)cm::max_(__builtin_floorf(fminy), 0.0f), by1 = (int)cm::min_(__builtin_ceilf(fmaxy), (float)(H - 1));
            if (bx1 < bx0 || by1 < by0) continue;
            const uint32_t bw = (uint32_t)(bx1 - bx0 + 1), bh = (uint32_t)(by1 - by0 + 1);
            const uint64_t total = (uint64_t)bw * bh;
            for (uint64_t i = lane; i < total; i += 64u) {
                const uint32_t row = (uint32_t)(i / bw), col = (uint32_t)(i - (uint64_t)row * bw);
                const uint32_t px = (uint32_t)bx0 + col, py = (uint32_t)by0 + row;
                const float cx = (float)px + 0.5f, cy = (float)py + 0.5f;
                const float e0 = sgn * edgeFn(x1, y1, x2, y2, cx, cy), e1 = sgn * edgeFn(x2, y2, x0, y0, cx, cy), e2 = sgn * edgeFn(x0, y0, x1, y1, cx, cy);
                if (!(e0 >= 0.0f && e1 >= 0.0f && e2 >= 0.0f)) continue;
                const float den = (e0 + e1) + e2;
                if (!(den > 0.0f)) continue;
                const float d = cm::fma_(e2, d2, cm::fma_(e1, d1, e0 * d0)) / den;
                if (d > 0.0f) atomicMax(&a.depth[(uint64_t)py * W + px], __float_as_uint(d));   // GREATER test; NaN never passes
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                           // the LDS slice is reused by the next meshlet
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

int recordRasterDepth(trhip::DispatchCtx& ctx)
{
    // Binding set of BasePassRenderers.cpp:463-479 (t0 instances, t1 vertices, t2 mesh data, t4 meshlets, t5 meshlet
    // vertex ids, t6 meshlet triangles, t7 amplification records) + the outputs of the cull half: t9 visible list,
    // indirect args = its draw args; u0 = the depth buffer (R32_FLOAT).
    const BasePassConstants* k = (const BasePassConstants*)ctx.constants(0, sizeof(BasePassConstants));
    TRHIP_REQUIRE(k, "%s: constant buffer b0 (BasePassConstants, 256 bytes) missing", ctx.shaderName);
    trhip_buffer_t* instances = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 0);
    trhip_buffer_t* vertices = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 1);
    trhip_buffer_t* meshData = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 2);
    trhip_buffer_t* meshlets = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 4);
    trhip_buffer_t* vids = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 5);
    trhip_buffer_t* tris = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 6);
    trhip_buffer_t* records = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 7);
    trhip_buffer_t* list = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 9);
    uint32_t mip = 0;
    trhip_texture_t* depth = ctx.texture(TRHIP_BIND_TEXTURE_UAV, 0, &mip);
    TRHIP_REQUIRE(instances && vertices && meshData && meshlets && vids && tris && records && list,
                  "%s: needs SRVs t0 (instances), t1 (vertices), t2 (mesh data), t4 (meshlets), t5 (meshlet vertex ids), t6 (meshlet triangles), t7 (records), t9 (visible list)", ctx.shaderName);
    TRHIP_REQUIRE(depth && mip == 0 && depth->format == TRHIP_FORMAT_R32_FLOAT, "%s: needs Texture_UAV u0 = the R32_FLOAT depth buffer, mip 0", ctx.shaderName);
    TRHIP_REQUIRE(ctx.indirect && ctx.argsBuffer->byteSize - ctx.argsOffset >= 12, "%s: dispatched indirectly on the visible list's draw args", ctx.shaderName);
    TRHIP_REQUIRE(k->m_OutputResolution.x == depth->width && k->m_OutputResolution.y == depth->height,
                  "%s: m_OutputResolution %ux%u does not match the depth buffer %ux%u", ctx.shaderName, k->m_OutputResolution.x, k->m_OutputResolution.y, depth->width, depth->height);
    RasterArgs a;
    memset(&a, 0, sizeof a);
    a.k = *k;
    a.instances = (const BasePassInstanceConstants*)instances->ptr; a.numInstances = (uint32_t)std::min<uint64_t>(instances->byteSize / sizeof(BasePassInstanceConstants), 0xFFFFFFFFull);
    a.meshData = (const MeshData*)meshData->ptr; a.numMeshes = (uint32_t)std::min<uint64_t>(meshData->byteSize / sizeof(MeshData), 0xFFFFFFFFull);
    a.meshlets = (const MeshletData*)meshlets->ptr; a.numMeshlets = meshlets->byteSize / sizeof(MeshletData);
    a.vertices = (const char*)vertices->ptr; a.numVertices = vertices->byteSize / 20u;
    a.vertexIds = (const uint32_t*)vids->ptr; a.numVertexIds = vids->byteSize / 4;
    a.triangles = (const uint32_t*)tris->ptr; a.numTriangles = tris->byteSize / 4;
    a.records = (const MeshletAmplificationData*)records->ptr; a.recordCapacity = (uint32_t)std::min<uint64_t>(records->byteSize / sizeof(MeshletAmplificationData), 0xFFFFFFFFull);
    a.visibleList = (const uint32_t*)list->ptr; a.listCapacity = (uint32_t)std::min<uint64_t>(list->byteSize / 4, 0xFFFFFFFFull);
    a.drawArgs = (const uint32_t*)((const char*)ctx.argsBuffer->ptr + ctx.argsOffset);
    a.depth = (uint32_t*)depth->ptr;
    a.width = depth->width; a.height = depth->height;
    const uint32_t grid = ctx.computeUnits() * 4u;
    ctx.emit("main", [a, grid](hipStream_t s) {
        hipLaunchKernelGGL(rasterDepthKernel, dim3(grid), dim3(kBlock), 0, s, a);
        return trhip::launchStatus("rasterDepthKernel"); });
    return TRHIP_OK;
}

trhip::ShaderRegistrar r0("basepass_MS_Main_depth", recordRasterDepth, 0);

} // namespace
