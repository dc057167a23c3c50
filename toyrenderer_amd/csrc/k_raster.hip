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
// Two launches.  "main": one wave per visible meshlet -- lanes transform the (<= 64) vertices into the wave's LDS
// slice, then set up the triangles (lane t: triangle t); a triangle whose box exceeds kSmallBox pixels is appended to a
// queue (screen positions, depths, box; one counter update per wave), the others are drawn by the wave one after the
// other, 64 pixels of the bounding box per pass.  "tiles": one
// workgroup per 64x64-pixel screen tile collects the queued triangles that touch it and rasterises them into an LDS copy
// of the tile, 4096 pixels at a time, then merges the tile into the depth buffer; it looks for them in the list of its
// 256x256-pixel coarse bin, which "main" fills while it queues.  Without the second launch a dozen
// waves walked the 10^4-pixel boxes of a near wall while the chip idled (3.2 ms per pass at 3840x2160,
// tools/raster_time.py).  Every pixel value is computed by the same operations whichever launch produces it.
#include "cull_math.hip.h"
#include "trhip_internal.h"

using namespace interop;

namespace
{

constexpr uint32_t kBlock = 256;
constexpr uint32_t kWaves = kBlock / 64;
constexpr uint32_t kSmallBox = 1024;            // pixels: larger bounding boxes go to the tile pass
constexpr uint32_t kTile = 64;                  // pixels per side
constexpr uint32_t kQueueCapacity = 1u << 20;   // 48 MB; beyond it triangles are drawn in place (slow, still exact)
constexpr uint32_t kTileList = 1024;            // queued triangles a tile handles per round
constexpr uint32_t kBinShift = 8;               // coarse bins of 256x256 pixels (4x4 tiles): a tile scans its bin's list, not the whole queue
constexpr uint32_t kBinCapacity = 1u << 16;     // queue indices per bin; a fuller bin makes its tiles scan the whole queue

struct BigTriangle                              // 48 bytes
{
    float x0, y0, d0, x1, y1, d1, x2, y2, d2, sgn;
    uint32_t boxX, boxY;                        // x0 | x1 << 16, y0 | y1 << 16 (inclusive pixel bounds)
};

__device__ __forceinline__ float edgeFn(float ax, float ay, float bx, float by, float px, float py)
{
    return cm::fma_(bx - ax, py - ay, -((by - ay) * (px - ax)));
}

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
    BigTriangle* queue;                                          // scratch: [kQueueCapacity]
    uint32_t* queueCount;                                        // scratch, zeroed before "main"
    uint32_t* binCount;                                          // scratch, zeroed: [binsX * binsY] entries appended (may exceed the capacity)
    uint32_t* binList;                                           // scratch: [binsX * binsY][kBinCapacity] queue indices
    uint32_t binsX, binsY;
};

// One triangle over the pixels [bx0, bx1] x [by0, by1], `threads` lanes striding over them from `first`; every covered
// pixel goes to `sink(px, py, depthBits)`.  The arithmetic of orc_raster_depth, operation for operation.
template <typename Sink>
__device__ __forceinline__ void coverBox(float x0, float y0, float d0, float x1, float y1, float d1, float x2, float y2, float d2, float sgn,
                                         uint32_t bx0, uint32_t by0, uint32_t bw, uint32_t bh, uint32_t first, uint32_t threads, Sink sink)
{
    const uint64_t total = (uint64_t)bw * bh;
    for (uint64_t i = first; i < total; i += threads) {
        const uint32_t row = (uint32_t)(i / bw), col = (uint32_t)(i - (uint64_t)row * bw);
        const uint32_t px = bx0 + col, py = by0 + row;
        const float cx = (float)px + 0.5f, cy = (float)py + 0.5f;
        const float e0 = sgn * edgeFn(x1, y1, x2, y2, cx, cy), e1 = sgn * edgeFn(x2, y2, x0, y0, cx, cy), e2 = sgn * edgeFn(x0, y0, x1, y1, cx, cy);
        if (!(e0 >= 0.0f && e1 >= 0.0f && e2 >= 0.0f)) continue;
        const float den = (e0 + e1) + e2;
        if (!(den > 0.0f)) continue;
        const float d = cm::fma_(e2, d2, cm::fma_(e1, d1, e0 * d0)) / den;
        if (d > 0.0f) sink(px, py, __float_as_uint(d));                                   // GREATER test; NaN never passes
    }
}

__global__ __launch_bounds__(kBlock) void rasterDepthKernel(RasterArgs a)
{
    __shared__ float s_x[kWaves][64], s_y[kWaves][64], s_d[kWaves][64];
    __shared__ BigTriangle s_tri[kWaves][64];
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
        // ---- triangles (:178-187).  Set-up in parallel: lane t prepares triangle t (64 at a time); the large ones are
        //      queued with ONE counter update per wave, the others are then drawn one after the other, 64 pixels of the
        //      bounding box per pass -----------------------------------------------------------------------------------
        BigTriangle* st = s_tri[wave];
        for (uint32_t tb = 0; tb < nt; tb += 64u) {
            const uint32_t t = tb + lane;
            bool live = false;
            BigTriangle q = {};
            uint32_t bw = 0, bh = 0;
            if (t < nt) {
                const uint32_t packed = a.triangles[ml.m_MeshletIndexIDsBufferIdx + t];
                const uint32_t ia = packed & 0xFFu, ib = (packed >> 8) & 0xFFu, ic = (packed >> 16) & 0xFFu;
                if (ia < nv && ib < nv && ic < nv && ((okMask >> ia) & (okMask >> ib) & (okMask >> ic) & 1ull)) {
                    q.x0 = sx[ia]; q.y0 = sy[ia]; q.x1 = sx[ib]; q.y1 = sy[ib]; q.x2 = sx[ic]; q.y2 = sy[ic];
                    q.d0 = sd[ia]; q.d1 = sd[ib]; q.d2 = sd[ic];
                    const float area = edgeFn(q.x0, q.y0, q.x1, q.y1, q.x2, q.y2);
                    const float fminx = cm::min_(cm::min_(q.x0, q.x1), q.x2), fmaxx = cm::max_(cm::max_(q.x0, q.x1), q.x2);
                    const float fminy = cm::min_(cm::min_(q.y0, q.y1), q.y2), fmaxy = cm::max_(cm::max_(q.y0, q.y1), q.y2);
                    if (area != 0.0f                                                          // not degenerate, not NaN
                        && fmaxx >= 0.0f && fmaxy >= 0.0f && fminx <= (float)W && fminy <= (float)H) {   // on screen, not NaN
                        q.sgn = area < 0.0f ? -1.0f : 1.0f;
                        const int bx0 = (int)cm::max_(__builtin_floorf(fminx), 0.0f), bx1 = (int)cm::min_(__builtin_ceilf(fmaxx), (float)(W - 1));
                        const int by0 = (int)cm::max_(__builtin_floorf(fminy), 0.0f), by1 = (int)cm::min_(__builtin_ceilf(fmaxy), (float)(H - 1));
                        if (bx1 >= bx0 && by1 >= by0) {
                            live = true;
                            bw = (uint32_t)(bx1 - bx0 + 1); bh = (uint32_t)(by1 - by0 + 1);
                            q.boxX = (uint32_t)bx0 | ((uint32_t)bx1 << 16); q.boxY = (uint32_t)by0 | ((uint32_t)by1 << 16);
                        }
                    }
                }
            }
            // large on screen: the tile pass draws them
            bool big = live && (uint64_t)bw * bh > kSmallBox && a.queue != nullptr;
            const unsigned long long bigMask = __ballot(big);
            if (bigMask) {
                uint32_t first = 0;
                if (lane == 0) first = atomicAdd(a.queueCount, (uint32_t)__popcll(bigMask));
                first = __shfl(first, 0);
                const uint32_t slot = first + (uint32_t)__popcll(bigMask & ((1ull << lane) - 1ull));
                if (big && slot < kQueueCapacity) {
                    a.queue[slot] = q;
                    // its index goes to every coarse bin the box touches
                    const uint32_t cx0 = (q.boxX & 0xFFFFu) >> kBinShift, cx1 = (q.boxX >> 16) >> kBinShift;
                    const uint32_t cy0 = (q.boxY & 0xFFFFu) >> kBinShift, cy1 = (q.boxY >> 16) >> kBinShift;
                    for (uint32_t cy = cy0; cy <= cy1; ++cy)
                        for (uint32_t cx = cx0; cx <= cx1; ++cx) {
                            const uint32_t bin = cy * a.binsX + cx;
                            const uint32_t k = atomicAdd(&a.binCount[bin], 1u);
                            if (k < kBinCapacity) a.binList[(uint64_t)bin * kBinCapacity + k] = slot;
                        }
                } else {
                    big = false;                                                             // queue full: drawn in place
                }
            }
            // the others, in place
            const unsigned long long smallMask = __ballot(live && !big);
            if (smallMask) {
                st[lane] = q;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                uint32_t* depth = a.depth;
                for (unsigned long long mrem = smallMask; mrem; mrem &= mrem - 1ull) {
                    const BigTriangle c = st[__builtin_ctzll(mrem)];
                    const uint32_t bx0 = c.boxX & 0xFFFFu, by0 = c.boxY & 0xFFFFu;
                    coverBox(c.x0, c.y0, c.d0, c.x1, c.y1, c.d1, c.x2, c.y2, c.d2, c.sgn, bx0, by0, (c.boxX >> 16) - bx0 + 1u, (c.boxY >> 16) - by0 + 1u, lane, 64u,
                             [depth, W](uint32_t px, uint32_t py, uint32_t bits) { atomicMax(&depth[(uint64_t)py * W + px], bits); });
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                       // st is rewritten by the next 64 triangles
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                           // the LDS slice is reused by the next meshlet
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}


// The queued triangles, by screen tile.  Rounds of at most kTileList triangles: collect (all threads scan the queue,
// LDS append), rasterise into the LDS tile, next round; then merge.  A tile is owned by one workgroup and this launch
// follows "main" on the stream, so the merge needs no atomics.
__global__ __launch_bounds__(kBlock) void rasterTilesKernel(RasterArgs a)
{
    __shared__ uint32_t s_depth[kTile * kTile];
    __shared__ uint32_t s_list[kTileList];
    __shared__ uint32_t s_count;
    __shared__ uint32_t s_waveMin[kWaves];
    const uint32_t tid = threadIdx.x;
    uint32_t n = *a.queueCount;
    n = n < kQueueCapacity ? n : kQueueCapacity;
    if (n == 0) return;
    const uint32_t tilesX = (a.width + kTile - 1) / kTile, tilesY = (a.height + kTile - 1) / kTile;
    for (uint32_t tile = blockIdx.x; tile < tilesX * tilesY; tile += gridDim.x) {
        const uint32_t tx0 = (tile % tilesX) * kTile, ty0 = (tile / tilesX) * kTile;
        const uint32_t tx1 = min(tx0 + kTile, a.width) - 1u, ty1 = min(ty0 + kTile, a.height) - 1u;
        for (uint32_t i = tid; i < kTile * kTile; i += kBlock) s_depth[i] = 0u;
        bool any = false;
        float tileFar = 0.0f;                                                            // farthest depth in the tile (0 = something still uncovered)
        // candidates: the list of the tile's coarse bin, or the whole queue when that list overflowed
        const uint32_t bin = (ty0 >> kBinShift) * a.binsX + (tx0 >> kBinShift);
        const uint32_t binned = a.binCount[bin];
        const bool wholeQueue = binned > kBinCapacity;
        const uint32_t* candidates = a.binList + (uint64_t)bin * kBinCapacity;
        const uint32_t nCand = wholeQueue ? n : binned;
        for (uint32_t base = 0; base < nCand;) {
            if (tid == 0) s_count = 0;
            __syncthreads();
            // collect: the scan stops early when the list is full; `base` advances to the first triangle not yet looked at
            uint32_t next = base;
            for (; next < nCand; next += kBlock) {
                const uint32_t c = next + tid;
                if (c < nCand) {
                    const uint32_t i = wholeQueue ? c : candidates[c];
                    const uint32_t bx = a.queue[i].boxX, by = a.queue[i].boxY;
                    if ((bx & 0xFFFFu) <= tx1 && (bx >> 16) >= tx0 && (by & 0xFFFFu) <= ty1 && (by >> 16) >= ty0) {
                        const uint32_t k = atomicAdd(&s_count, 1u);
                        s_list[k] = i;                                                     // room for it: checked below before the next chunk
                    }
                }
                __syncthreads();
                const bool full = s_count + kBlock > kTileList;                            // the next chunk might not fit
                __syncthreads();
                if (full) { next += kBlock; break; }
            }
            base = next;
            const uint32_t m = s_count;
            any |= m != 0;
            for (uint32_t k = 0; k < m; ++k) {
                // Every 32 triangles: the farthest depth the tile holds so far.  A triangle none of whose samples can be
                // nearer than that cannot change a maximum and is skipped (its samples are at most max(d0,d1,d2) times
                // (1 + 6 * 2^-24): three roundings in the fma chain, two in the sum of the weights, one in the division).
                if ((k & 31u) == 0u && (k != 0u || any)) {
                    uint32_t mn = 0xFFFFFFFFu;
                    const uint32_t w = tx1 - tx0 + 1u, h = ty1 - ty0 + 1u;
                    for (uint32_t i = tid; i < w * h; i += kBlock) { const uint32_t y = i / w, x = i - y * w; mn = min(mn, s_depth[y * kTile + x]); }
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, d));
                    __syncthreads();                                                     // the previous value has been read by everyone
                    if ((tid & 63u) == 0u) s_waveMin[tid >> 6] = mn;
                    __syncthreads();
                    tileFar = __uint_as_float(min(min(s_waveMin[0], s_waveMin[1]), min(s_waveMin[2], s_waveMin[3])));   // depths are > 0: bit order = value order
                }
                const BigTriangle q = a.queue[s_list[k]];
                if (cm::max_(cm::max_(q.d0, q.d1), q.d2) * 0x1.00001p+0f < tileFar) continue;     // NaN or inf: never skipped
                const uint32_t bx0 = max(q.boxX & 0xFFFFu, tx0), bx1 = min(q.boxX >> 16, tx1);
                const uint32_t by0 = max(q.boxY & 0xFFFFu, ty0), by1 = min(q.boxY >> 16, ty1);
                coverBox(q.x0, q.y0, q.d0, q.x1, q.y1, q.d1, q.x2, q.y2, q.d2, q.sgn, bx0, by0, bx1 - bx0 + 1u, by1 - by0 + 1u, tid, kBlock,
                         [tx0, ty0](uint32_t px, uint32_t py, uint32_t bits) { atomicMax(&s_depth[(py - ty0) * kTile + (px - tx0)], bits); });
            }
            __syncthreads();
        }
        if (any) {
            const uint32_t w = tx1 - tx0 + 1u, h = ty1 - ty0 + 1u;
            for (uint32_t i = tid; i < w * h; i += kBlock) {
                const uint32_t y = i / w, x = i - y * w;
                const uint32_t v = s_depth[y * kTile + x];
                if (v) { uint32_t* g = &a.depth[(uint64_t)(ty0 + y) * a.width + tx0 + x]; if (v > *g) *g = v; }
            }
        }
        __syncthreads();
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
    TRHIP_REQUIRE(a.width <= 0xFFFFu && a.height <= 0xFFFFu, "%s: depth buffer %ux%u: at most 65535 pixels per side", ctx.shaderName, a.width, a.height);
    // queue of the triangles that are large on screen: scratch of this command; its counter is zeroed by the recording's
    // first clear launch
    a.queue = (BigTriangle*)ctx.scratch((size_t)kQueueCapacity * sizeof(BigTriangle));
    a.queueCount = (uint32_t*)ctx.scratch(16);
    TRHIP_REQUIRE(a.queue && a.queueCount, "%s: scratch allocation failed", ctx.shaderName);
    a.binsX = (a.width + (1u << kBinShift) - 1u) >> kBinShift;
    a.binsY = (a.height + (1u << kBinShift) - 1u) >> kBinShift;
    const uint32_t bins = a.binsX * a.binsY;
    a.binCount = (uint32_t*)ctx.scratch((size_t)bins * 4);
    a.binList = (uint32_t*)ctx.scratch((size_t)bins * kBinCapacity * 4);
    TRHIP_REQUIRE(a.binCount && a.binList, "%s: scratch allocation failed", ctx.shaderName);
    int rc = ctx.cl->recordClearWords(a.queueCount, 4, 0, true);
    if (rc == TRHIP_OK) rc = ctx.cl->recordClearWords(a.binCount, bins, 0, true);
    if (rc != TRHIP_OK) return rc;
    const uint32_t grid = ctx.computeUnits() * 4u;
    ctx.emit("main", [a, grid](hipStream_t s) {
        TRHIP_LAUNCH(rasterDepthKernel, dim3(grid), dim3(kBlock), 0, s, a);
        return trhip::launchStatus("rasterDepthKernel"); });
    const uint32_t tiles = ((a.width + kTile - 1) / kTile) * ((a.height + kTile - 1) / kTile);
    const uint32_t tileGrid = tiles < ctx.computeUnits() * 8u ? tiles : ctx.computeUnits() * 8u;
    ctx.emit("tiles", [a, tileGrid](hipStream_t s) {
        TRHIP_LAUNCH(rasterTilesKernel, dim3(tileGrid), dim3(kBlock), 0, s, a);
        return trhip::launchStatus("rasterTilesKernel"); });
    return TRHIP_OK;
}

trhip::ShaderRegistrar r0("basepass_MS_Main_depth", recordRasterDepth, 0);

} // namespace
