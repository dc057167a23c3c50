// k_gpuculling.hip -- "gpuculling_CS_GPUCulling LATE_CULL={0,1}" and
// "gpuculling_CS_BuildLateCullIndirectArgs" for gfx950.
//
// Reference: source/shaders/gpuculling.hlsl:35-195, dispatched by
// BasePassRenderer::GPUCulling (source/BasePassRenderers.cpp:298-404).
//
// The reference appends amplification records and late-list entries with global atomics, so its
// output ORDER is whatever order the atomics resolve in.  This build defines the canonical order
// = ascending dispatch-thread id (SURVEY.md section 7 "Determinism vs. atomics") and produces it
// with prefix sums instead of atomics, in three launches on one stream:
//   A  classify   one thread per list entry: frustum + HZB test + LOD select; block-level
//                 exclusive scan of (groups to emit, late flags) -> per-block sums
//   B  scan       one block: exclusive scan of the per-block sums, final counters
//   C  emit       one thread per list entry: writes its records / late-list entry at its offset
// HBM traffic: 4 B id + 68 B of the 144-B instance record + MeshData per entry (A), 8 B of
// scratch per entry (A write, C read), 12 B per record (C).  Bound: HBM; the pass is <5 % of a
// frame on the 100 M-meshlet config (DESIGN.md "Kernels").
#include "cull_math.hip.h"
#include "hzb_quad.hip.h"
#include "instance_cache.hip.h"
#include "trhip_internal.h"

using namespace interop;

namespace
{

constexpr uint32_t kBlock = 256;
constexpr uint32_t kWordSubmit = 1u << 30;
constexpr uint32_t kWordLate = 1u << 31;
constexpr uint32_t kGroupMask = (1u << 27) - 1u;

struct InstanceCullArgs
{
    GPUCullingPassConstants k;
    const BasePassInstanceConstants* instances;
    const uint32_t* ids;            // primitive ids (early) / late-list ids (late)
    const MeshData* meshData;
    cm::Hzb hzb;
    MeshletAmplificationData* records;
    uint32_t* dispatchArgs;         // {X,Y,Z[,validRecords]}
    uint32_t* lateCount;
    uint32_t* lateIds;
    const uint32_t* indirectArgs;   // late: {ceil(count/64),1,1} (Q1)
    InstanceCullCache cache;
    uint32_t numInstances;          // entries of the instance buffer (= of the cache)
    const uint32_t* shardLate;      // late, multi-GPU only: {late entries of the lower ranks, of all ranks} (trhip.h)
    uint32_t directThreads;         // early: gx * 32
    uint32_t maxGroups;             // capacity of `records` (65535 in the reference, Q2)
    uint32_t argsWords;             // 3 or 4
    // scratch
    uint32_t* word;                 // per entry: groups | lod<<27 | submit<<30 | late<<31
    uint32_t* localOff;             // per entry: exclusive offset inside its block
    uint32_t* blockGroups;          // per block: sum of groups      -> exclusive prefix after B
    uint64_t* blockLateSubmit;      // per block: late | submits<<32 -> exclusive prefix after B
    uint32_t* bases;                // [0] X before the pass, [1] late count before the pass
    uint32_t numBlocks;
    // screen-tile binning of the submitted instances (PROCESSING order of the meshlet pass only; the
    // record order above is untouched): a counting sort by tile, see "Large passes" below; perm lives in the records
    // buffer's sidecar: {valid, count, ...} header (64 words) followed by one 16-byte entry per group.
    uint32_t* tileHist;             // large passes: [workgroup][kNumTiles] groups per tile -> (scan) groups of the tile in earlier workgroups
    uint32_t* tileTotal;            // [kNumTiles]
    uint16_t* tileOf;               // per entry
    uint2* lodSel;                  // per entry: {meshlets, first meshlet} of the LOD the instance was submitted at
    uint32_t* permHeader;
    uint4* perm;                    // {record index, instance, first meshlet, meshlets in the group}: the record RESOLVED, in processing order
    uint32_t permCapacity;
    uint32_t* lateArgsOut;          // early, optional: gpuculling_CS_BuildLateCullIndirectArgs folded into the scan (recordBuildLateArgs)
    // small passes (instanceFusedKernel): one status line per tile of 256 entries + the ticket counter, zeroed per launch
    unsigned long long* fusedStatus;
    uint32_t* fusedTicket;
};

// What an early recordGPUCulling leaves for a recordBuildLateArgs that follows it immediately (trhip_cmdlist_t::peephole).
struct QuadShare { bool stale = false; };   // large passes: the scan launch decides whether the table is rebuilt, the emit launch builds its share
struct EarlyCullNote { InstanceCullArgs a; size_t scanOp; bool fused; trhip_texture_t* quadOwner = nullptr; std::shared_ptr<QuadShare> share; };

__global__ __launch_bounds__(256) void instanceCacheKernel(const BasePassInstanceConstants* instances, uint32_t n,
                                                           const MeshData* meshData, uint32_t numMeshes, InstanceCullCache c)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const BasePassInstanceConstants& inst = instances[i];
    const float4 w0 = *reinterpret_cast<const float4*>(inst.m_WorldMatrix.m[0]);
    const float4 w1 = *reinterpret_cast<const float4*>(inst.m_WorldMatrix.m[1]);
    const float4 w2 = *reinterpret_cast<const float4*>(inst.m_WorldMatrix.m[2]);
    const float4 w3 = *reinterpret_cast<const float4*>(inst.m_WorldMatrix.m[3]);
    const uint32_t meshIdx = inst.m_MeshDataIdx;
    float4 sph = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t numLODs = 0;
    uint32_t nm[kMaxNumMeshLODs], mb[kMaxNumMeshLODs];
    float err[kMaxNumMeshLODs];
#pragma unroll
    for (uint32_t l = 0; l < kMaxNumMeshLODs; ++l) { nm[l] = 0; mb[l] = 0; err[l] = 0.f; }
    if (meshIdx < numMeshes) {                                                       // never read outside the mesh table
        const MeshData& mesh = meshData[meshIdx];
        sph = *reinterpret_cast<const float4*>(&mesh.m_BoundingSphere);
        numLODs = mesh.m_NumLODs;
#pragma unroll
        for (uint32_t l = 0; l < kMaxNumMeshLODs; ++l) {
            nm[l] = mesh.m_MeshLODDatas[l].m_NumMeshlets; mb[l] = mesh.m_MeshLODDatas[l].m_MeshletDataBufferIdx; err[l] = mesh.m_MeshLODDatas[l].m_Error;
        }
    }
    instanceCacheWriteTransformPart(c, i, w0, w1, w2, w3, sph);
    const_cast<float4*>(c.localSphere)[i] = sph;
    const_cast<uint32_t*>(c.numLODs)[i] = numLODs;
#pragma unroll
    for (uint32_t l = 0; l < kMaxNumMeshLODs; ++l) {
        const_cast<uint2&>(c.lod(i, l)) = make_uint2(nm[l], mb[l]);
        const_cast<float&>(c.err(i, l)) = err[l];
    }
}

// 16 x 16 screen tiles: on C3 the meshlet cull runs equally fast with 8 x 8, 16 x 16 and 32 x 32 (measured, profiles/r2/
// experiments.md), and everything that walks the [workgroup][tile] histogram gets cheaper with fewer tiles.
#ifndef TR_TILES_PER_AXIS
#define TR_TILES_PER_AXIS 16
#endif
constexpr uint32_t kTilesPerAxis = TR_TILES_PER_AXIS;
constexpr uint32_t kNumTiles = kTilesPerAxis * kTilesPerAxis;
constexpr uint32_t kPermHeaderWords = 64;
#ifndef TR_MIN_BINNED
#define TR_MIN_BINNED 4096
#endif
constexpr uint32_t kMinBinnedEntries = TR_MIN_BINNED;      // below this many list entries the processing order is left alone (device-side test)

// Which screen tile (kTilesPerAxis x kTilesPerAxis of them) the instance's centre projects to.  Scheduling heuristic only
// (approximate reciprocal, no exactness requirement): it never influences an output value.
__device__ __forceinline__ uint32_t screenTile(cm::F3 cv, float P00, float P11)
{
    const float iz = __builtin_amdgcn_rcpf(cm::max_(cv.z, 1e-6f));
    const float u = cm::fma_(cv.x * iz * P00, 0.5f, 0.5f), v = cm::fma_(cv.y * iz * P11, -0.5f, 0.5f);
    const int tx = min(max((int)(u * (float)kTilesPerAxis), 0), (int)kTilesPerAxis - 1);
    const int ty = min(max((int)(v * (float)kTilesPerAxis), 0), (int)kTilesPerAxis - 1);
    return (uint32_t)ty * kTilesPerAxis + (uint32_t)tx;
}

template <int LATE>
__device__ __forceinline__ uint32_t threadCount(const InstanceCullArgs& a)
{
    if (LATE) {
        // gpuculling.hlsl:94-103 with the dispatch size of :182-195 (Q1: ceil(count/64) groups of 32)
        uint32_t count = *a.lateCount;
        uint64_t launched = (uint64_t)a.indirectArgs[0] * kNumThreadsPerWave;
        if (a.shardLate) {
            // sharded instance list: the threads the single-GPU dispatch launches cover the first
            // ceil(total/64)*32 entries of the rank-major concatenation of the late lists; this
            // rank's entries start at shardLate[0] in it
            const uint64_t all = (((uint64_t)a.shardLate[1] + 63u) / 64u) * kNumThreadsPerWave;
            launched = all > a.shardLate[0] ? all - a.shardLate[0] : 0u;
        }
        uint32_t n = launched < count ? (uint32_t)launched : count;
        return n < a.k.m_NbInstances ? n : a.k.m_NbInstances;
    }
    return a.directThreads < a.k.m_NbInstances ? a.directThreads : a.k.m_NbInstances;
}

__device__ __forceinline__ uint32_t waveInclusiveScan(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(v, d);
        if (lane >= (uint32_t)d) v += n;
    }
    return v;
}

// gpuculling.hlsl:105-178 up to the ordered side effects.
template <int LATE>
__device__ __forceinline__ uint32_t classify(const InstanceCullArgs& a, uint32_t id, uint32_t* tileOut, uint2* lodOut)
{
    const GPUCullingPassConstants& k = a.k;
    const bool doFrustum = (k.m_CullingFlags & kCullingFlagFrustumCullingEnable) != 0;
    const bool doOcclusion = (k.m_CullingFlags & kCullingFlagOcclusionCullingEnable) != 0;

    // :114-116 through the instance cull cache (same values as from a.instances[id] / a.meshData[...])
    const uint32_t cid = id < a.numInstances ? id : 0u;                             // never read outside the cache
    const float4 ws = a.cache.sphere[cid];
    const float ms = a.cache.maxScale[cid];
    const cm::F3 wc = { ws.x, ws.y, ws.z };
    const float r = ws.w;
    cm::F3 cv = cm::toView(wc, cm::loadM43(k.m_WorldToView));                      // :118-119

    if (!LATE && doFrustum &&                                                      // :124-134
        !cm::frustumVisible(cv, r, k.m_Frustum.x, k.m_Frustum.y, k.m_Frustum.z, k.m_Frustum.w))
        return 0;

    if (doOcclusion) {
        if (!LATE) cv = cm::toView(wc, cm::loadM43(k.m_PrevWorldToView));          // :143-146 (Q3)
        if (!cm::occlusionVisible(cv, r, k.m_NearPlane, k.m_P00, k.m_P11, a.hzb))  // :148-158
            return LATE ? 0 : kWordLate;                                           // :162-178
    }

    // SubmitInstance :35-62
    const uint32_t numLODs = a.cache.numLODs[cid];
    uint32_t lod = 0;
    if (k.m_ForcedMeshLOD != kInvalidMeshLOD) {
        const uint32_t last = numLODs - 1u;
        lod = k.m_ForcedMeshLOD < last ? k.m_ForcedMeshLOD : last;
    } else {
        const float distance = cm::max_(cm::sqrt_(cm::dot3(cv, cv)) - r, 0.0f);
        const float threshold = distance * k.m_MeshLODTarget / ms;
        const uint32_t n = numLODs < kMaxNumMeshLODs ? numLODs : kMaxNumMeshLODs;
        for (uint32_t i = 1; i < n; ++i)
            if (a.cache.err(cid, i) < threshold) lod = i;
    }
    lod = lod < kMaxNumMeshLODs ? lod : kMaxNumMeshLODs - 1u;                       // never index past the table
    const uint2 li = a.cache.lod(cid, lod);                                         // {m_NumMeshlets, m_MeshletDataBufferIdx}
    *lodOut = li;
    const uint32_t numMeshlets = li.x;
    const uint32_t groups = (numMeshlets + kNumThreadsPerWave - 1u) / kNumThreadsPerWave; // DivideAndRoundUp
    *tileOut = screenTile(cv, k.m_P00, k.m_P11);
    return kWordSubmit | (lod << 27) | (groups & kGroupMask);
}

// ---------------------------------------------------------------------------------------------------------------
// Large passes: classify -> scan -> emit, three launches, NO device-scope atomics (round 1 and the first half of round 2
// counted the tile histogram and handed out the tile-run places with 2 x 440 k global atomics on C3, which paced both
// kernels).  A workgroup of kBigThreads threads owns kBigChunk consecutive list entries:
//   classify   per entry: the tests + LOD (classify<>), the word / block-local offset / tile / LOD selection to scratch;
//              per workgroup: its sums and its tile histogram (LDS atomics) as row `block` of H[blocks][tiles];
//   scan       workgroup 0: exclusive scan of the workgroups' sums, final counters (what the reference's atomics
//              leave), late-cull arguments; workgroups 1..: sixteen tile columns of H each -> H[b][t] = groups
//              of tile t in workgroups < b (64 stripes of workgroups, two walks), tile totals T[t];
//   emit       per workgroup: run starts = exclusive scan of T + its row of H, kept as LDS cursors; records in canonical
//              order, tile-ordered resolved entries at cursor positions (LDS atomics: the order inside a
//              (workgroup, tile) run is arbitrary, which a PROCESSING order may be).
// C3 (early 781 k entries, late ~100 k): 1024 x 2 entries per workgroup: early 49 + 6 + 24 us, late 14 + 7 + 26 us (too few
// workgroups for the late list); 512 x 1: 37 + 13 + 23 / 7 + 5 + 6 us (shipped); 256 x 1: 36 + 21 + 25 / 7 + 5 + 8 us (the
// histogram matrix the scan walks grows with the number of workgroups).  Round-1 scheme with global atomics: 47 + 11 + 30 / 6 + 10 + 12.
// End of round 2 (16 x 16 tiles, LOD tables as planes, cooperative entry stores, rows requested before the list length):
// early 23 + 8 + 16 us; the late pass runs the single-launch kernel below whatever its capacity (8 us).
#ifndef TR_BIG_THREADS
#define TR_BIG_THREADS 512
#endif
#ifndef TR_BIG_PER
#define TR_BIG_PER 1
#endif
constexpr uint32_t kBigThreads = TR_BIG_THREADS;
constexpr uint32_t kBigPerThread = TR_BIG_PER;
constexpr uint32_t kScanThreads = 1024;
constexpr uint32_t kBigChunk = kBigThreads * kBigPerThread;
constexpr uint32_t kBigWaves = kBigThreads / 64;
constexpr uint32_t kScanTilesPerGroup = kNumTiles < 16 ? kNumTiles : 16;  // tile columns per scan workgroup (64 contiguous bytes of a row)
constexpr uint32_t kScanTileGroups = kNumTiles / kScanTilesPerGroup;      // scan workgroups 1..kScanTileGroups
constexpr uint32_t kScanStripes = kScanThreads / kScanTilesPerGroup;      // 64 stripes of workgroups

template <int LATE>
__device__ __forceinline__ uint32_t activeBigBlocks(const InstanceCullArgs& a, uint32_t n)
{
    const uint32_t need = (n + kBigChunk - 1) / kBigChunk;
    return need < a.numBlocks ? need : a.numBlocks;
}

template <int LATE>
__global__ __launch_bounds__(kBigThreads) void instanceClassifyKernel(InstanceCullArgs a)
{
    __shared__ uint32_t s_hist[kNumTiles];
    __shared__ uint32_t s_waveG[kBigWaves], s_waveL[kBigWaves], s_waveS[kBigWaves];
    const uint32_t n = threadCount<LATE>(a);
    if (blockIdx.x >= activeBigBlocks<LATE>(a, n)) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool binned = n >= kMinBinnedEntries;
    for (uint32_t i = tid; i < kNumTiles; i += kBigThreads) s_hist[i] = 0;
    uint32_t carryG = 0, carryL = 0, carryS = 0;
    for (uint32_t e = 0; e < kBigPerThread; ++e) {
        __syncthreads();                                   // histogram zeroed / the wave sums of the previous round consumed
        const uint32_t t = blockIdx.x * kBigChunk + e * kBigThreads + tid;
        uint32_t word = 0, tile = 0;
        uint2 lodSel = make_uint2(0u, 0u);
        if (t < n) word = classify<LATE>(a, a.ids[t], &tile, &lodSel);
        const uint32_t g = (word & kWordSubmit) ? (word & kGroupMask) : 0u;
        if (g != 0 && binned) {
            atomicAdd(&s_hist[tile], g);                   // histogram of groups per screen tile (LDS)
            a.tileOf[t] = (uint16_t)tile;
            a.lodSel[t] = lodSel;
        }
        const uint32_t late = word >> 31;
        const uint32_t submit = (word >> 30) & 1u;
        const uint32_t incG = waveInclusiveScan(g, lane);
        const uint32_t incL = waveInclusiveScan(late, lane);
        const uint32_t cntS = (uint32_t)__popcll(__ballot(submit != 0));
        if (lane == 63) { s_waveG[wave] = incG; s_waveL[wave] = incL; s_waveS[wave] = cntS; }
        __syncthreads();
        uint32_t baseG = 0, baseL = 0, totG = 0, totL = 0, totS = 0;
#pragma unroll
        for (uint32_t w = 0; w < kBigWaves; ++w) {
            if (w < wave) { baseG += s_waveG[w]; baseL += s_waveL[w]; }
            totG += s_waveG[w]; totL += s_waveL[w]; totS += s_waveS[w];
        }
        if (t < n) {
            a.word[t] = word;
            a.localOff[t] = late ? (carryL + baseL + incL - late) : (carryG + baseG + incG - g);
        }
        carryG += totG; carryL += totL; carryS += totS;
    }
    __syncthreads();
    if (binned)
        for (uint32_t i = tid; i < kNumTiles; i += kBigThreads) a.tileHist[(uint64_t)blockIdx.x * kNumTiles + i] = s_hist[i];
    if (tid == 0) {
        a.blockGroups[blockIdx.x] = carryG;
        a.blockLateSubmit[blockIdx.x] = (uint64_t)carryL | ((uint64_t)carryS << 32);
    }
}

constexpr uint32_t kScanStripsPerGroup = kScanThreads / 256u;             // table strips a workgroup past the scan's own builds
template <int LATE>
__global__ __launch_bounds__(kScanThreads) void instanceScanKernel(InstanceCullArgs a, trhip::QuadArgs q, uint32_t stripEnd)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // Workgroups past the scan's own (early pass only, when the HZB's footprint-min table is stale): four strips of the table each
    // (hzb_quad.hip.h).  The meshlet cull behind this pass reads the table.  The scan is seventeen workgroups of dependent round
    // trips: the first strips [0, stripEnd) of the table (11 MB read, 11 MB written in all) are built on the CUs idle beside it,
    // the rest by extra workgroups of the emit launch -- no launch of its own, no fork and join in front of the meshlet cull
    // (side stream beside classify: 20 + 9 + 16 us and a 7-us join; all strips here: 17 + 17 + 16; all in emit: 18 + 9 + 22;
    // half and half: profiles/r4/experiments.md section 7).
    if (!LATE && blockIdx.x > kScanTileGroups) {
        __shared__ float s_t[kScanStripsPerGroup][9][trhip::kQuadStripCols + 1];
        const uint32_t strip = (blockIdx.x - kScanTileGroups - 1u) * kScanStripsPerGroup + (tid >> 8);
        trhip::hzbQuadStrip(q, strip, s_t[tid >> 8], tid & 255u, strip < stripEnd);
        return;
    }
    if (blockIdx.x != 0) {
        // ---- tile columns: H[b][t] <- groups of tile t in workgroups < b; T[t] <- all of them -------------------------
        // The kernel is a chain of dependent round trips (list length -> rows -> sums -> rows again) around very little
        // work: 11 us for the early AND for the 8x shorter late list.  So the rows are striped by the CAPACITY of the pass
        // (known at launch) and loaded before the list length is: one round trip for everything, rows beyond the active
        // workgroups are masked afterwards (they hold an earlier frame's histogram), values stay in registers for the
        // second walk.
        __shared__ uint32_t s_stripe[kScanStripes][kScanTilesPerGroup];
        const uint32_t tl = tid % kScanTilesPerGroup, stripe = tid / kScanTilesPerGroup;
        const uint32_t tile = (blockIdx.x - 1u) * kScanTilesPerGroup + tl;
        constexpr uint32_t kMaxPer = 32;
        const uint32_t perCap = (a.numBlocks + kScanStripes - 1) / kScanStripes;
        // (early pass only: the late list is ~8x shorter than its capacity, and reading the capacity's rows next to the
        // list build on the side stream made the late scan slower, 20 -> 26 us between events)
        if (!LATE && perCap <= kMaxPer) {
            uint32_t v[kMaxPer];
            const uint32_t r0 = stripe * perCap;
#pragma unroll
            for (uint32_t r = 0; r < kMaxPer; ++r)
                v[r] = (r < perCap && r0 + r < a.numBlocks) ? a.tileHist[(uint64_t)(r0 + r) * kNumTiles + tile] : 0u;
            const uint32_t n = threadCount<LATE>(a);
            if (n < kMinBinnedEntries) return;
            const uint32_t activeBlocks = activeBigBlocks<LATE>(a, n);
            uint32_t sum = 0;
#pragma unroll
            for (uint32_t r = 0; r < kMaxPer; ++r) { v[r] = r0 + r < activeBlocks ? v[r] : 0u; sum += v[r]; }
            s_stripe[stripe][tl] = sum;
            __syncthreads();
            uint32_t run = 0;
            for (uint32_t sIdx = 0; sIdx < stripe; ++sIdx) run += s_stripe[sIdx][tl];
#pragma unroll
            for (uint32_t r = 0; r < kMaxPer; ++r) {
                if (r < perCap && r0 + r < activeBlocks) a.tileHist[(uint64_t)(r0 + r) * kNumTiles + tile] = run;
                run += v[r];
            }
            if (stripe == kScanStripes - 1u) a.tileTotal[tile] = run;
            return;
        }
        const uint32_t n = threadCount<LATE>(a);
        if (n < kMinBinnedEntries) return;
        const uint32_t activeBlocks = activeBigBlocks<LATE>(a, n);
        const uint32_t per = (activeBlocks + kScanStripes - 1) / kScanStripes;
        const uint32_t b0 = stripe * per < activeBlocks ? stripe * per : activeBlocks;
        const uint32_t b1 = b0 + per < activeBlocks ? b0 + per : activeBlocks;
        uint32_t sum = 0;
        for (uint32_t b = b0; b < b1; ++b) sum += a.tileHist[(uint64_t)b * kNumTiles + tile];
        s_stripe[stripe][tl] = sum;
        __syncthreads();
        uint32_t run = 0;
        for (uint32_t sIdx = 0; sIdx < stripe; ++sIdx) run += s_stripe[sIdx][tl];
        for (uint32_t b = b0; b < b1; ++b) {
            const uint32_t v = a.tileHist[(uint64_t)b * kNumTiles + tile];
            a.tileHist[(uint64_t)b * kNumTiles + tile] = run;
            run += v;
        }
        if (stripe == kScanStripes - 1u) a.tileTotal[tile] = run;
        return;
    }
    // workgroup 0: the loads its last step needs and the first slice of the workgroup sums are requested together with the
    // list length (all within the capacity of the pass), not after it
    const uint32_t argsX0 = a.dispatchArgs[0];
    const uint32_t late0 = LATE ? 0u : *a.lateCount;
    const uint32_t firstG = !LATE && tid < a.numBlocks ? a.blockGroups[tid] : 0u;
    const uint64_t firstLS = !LATE && tid < a.numBlocks ? a.blockLateSubmit[tid] : 0ull;
    const uint32_t n = threadCount<LATE>(a);
    const uint32_t activeBlocks = activeBigBlocks<LATE>(a, n);
    // ---- workgroup 0: exclusive scan over the per-workgroup sums; final counters ------------------------------------
    __shared__ uint32_t s_wg[2][(kScanThreads / 64)];
    __shared__ uint64_t s_wls[2][(kScanThreads / 64)];
    __shared__ uint32_t s_carryG;
    __shared__ uint64_t s_carryLS;
    uint32_t carryG = 0, flip = 0;
    uint64_t carryLS = 0;
    for (uint32_t base = 0; base < activeBlocks; base += kScanThreads) {
        const uint32_t i = base + tid;
        const uint32_t g = i < activeBlocks ? (!LATE && base == 0 ? firstG : a.blockGroups[i]) : 0u;
        const uint64_t ls = i < activeBlocks ? (!LATE && base == 0 ? firstLS : a.blockLateSubmit[i]) : 0ull;
        const uint32_t incG = waveInclusiveScan(g, lane);
        // 64-bit inclusive scan of (late | submits << 32): two 32-bit halves, no carry between them (each < 2^32)
        const uint32_t incLo = waveInclusiveScan((uint32_t)ls, lane), incHi = waveInclusiveScan((uint32_t)(ls >> 32), lane);
        const uint64_t incLS = (uint64_t)incLo | ((uint64_t)incHi << 32);
        if (lane == 63) { s_wg[flip][wave] = incG; s_wls[flip][wave] = incLS; }
        __syncthreads();
        uint32_t preG = 0, totG = 0;
        uint64_t preLS = 0, totLS = 0;
#pragma unroll
        for (uint32_t w = 0; w < (kScanThreads / 64); ++w) {
            const uint32_t xg = s_wg[flip][w];
            const uint64_t xl = s_wls[flip][w];
            if (w < wave) { preG += xg; preLS += xl; }
            totG += xg; totLS += xl;
        }
        if (i < activeBlocks) {
            a.blockGroups[i] = carryG + preG + incG - g;
            a.blockLateSubmit[i] = carryLS + preLS + incLS - ls;
        }
        carryG += totG; carryLS += totLS;
        flip ^= 1u;
    }
    if (tid == 0) { s_carryG = carryG; s_carryLS = carryLS; }
    __syncthreads();
    if (tid == 0) {
        const uint32_t baseX = argsX0;
        const uint32_t baseLate = late0;
        a.bases[0] = baseX;
        a.bases[1] = baseLate;
        const uint32_t X = baseX + s_carryG;                  // gpuculling.hlsl:65 (counter still counts drops, Q2)
        a.dispatchArgs[0] = X;
        if ((uint32_t)(s_carryLS >> 32) != 0) {               // :66-67, only when something was submitted
            a.dispatchArgs[1] = 1;
            a.dispatchArgs[2] = 1;
        }
        if (a.argsWords > 3) a.dispatchArgs[3] = X;           // valid records; the first dropped instance lowers it in emit
        if (!LATE) {
            const uint32_t late = baseLate + (uint32_t)s_carryLS;
            *a.lateCount = late;                                // :165
            if (a.lateArgsOut) {                                // gpuculling.hlsl:182-195 (Q1), folded in
                a.lateArgsOut[0] = (late + 63u) / 64u;
                a.lateArgsOut[1] = 1;
                a.lateArgsOut[2] = 1;
            }
        }
        // The tile-sorted processing order is published only when every group is emitted (no Q2 drop,
        // pass started from a cleared counter) and fits: then it is a permutation of [0, X).
        a.permHeader[0] = (n >= kMinBinnedEntries && baseX == 0 && X < a.maxGroups && X <= a.permCapacity) ? 1u : 0u;
        a.permHeader[1] = X;
    }
}

constexpr uint32_t kEmitStripsPerGroup = kBigThreads / 256u;
template <int LATE>
__global__ __launch_bounds__(kBigThreads) void instanceEmitKernel(InstanceCullArgs a, trhip::QuadArgs q, uint32_t stripBegin)
{
    // Workgroups past the pass's own: the strips [stripBegin, all) of the HZB's footprint-min table the scan launch left (see there)
    if (!LATE && blockIdx.x >= a.numBlocks) {
        __shared__ float s_t[kEmitStripsPerGroup][9][trhip::kQuadStripCols + 1];
        const uint32_t strip = stripBegin + (blockIdx.x - a.numBlocks) * kEmitStripsPerGroup + (threadIdx.x >> 8);
        trhip::hzbQuadStrip(q, strip, s_t[threadIdx.x >> 8], threadIdx.x & 255u, strip < q.firstStrip[q.mips]);
        return;
    }
    __shared__ uint32_t s_cursor[kNumTiles];
    __shared__ uint32_t s_wave[kBigWaves];
    const uint32_t n = threadCount<LATE>(a);
    if (blockIdx.x >= activeBigBlocks<LATE>(a, n)) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool binned = n >= kMinBinnedEntries;
    if (binned) {
        // run start of (this workgroup, tile) = groups of the tiles before it + groups of this tile in earlier workgroups
        uint32_t carry = 0;
        for (uint32_t base = 0; base < kNumTiles; base += kBigThreads) {
            const bool inRange = base + tid < kNumTiles;
            const uint32_t tot = inRange ? a.tileTotal[base + tid] : 0u;
            const uint32_t inc = waveInclusiveScan(tot, lane);
            if (lane == 63) s_wave[wave] = inc;
            __syncthreads();
            uint32_t pre = 0, all = 0;
            for (uint32_t w = 0; w < kBigWaves; ++w) { if (w < wave) pre += s_wave[w]; all += s_wave[w]; }
            if (inRange) s_cursor[base + tid] = carry + pre + inc - tot + a.tileHist[(uint64_t)blockIdx.x * kNumTiles + base + tid];
            carry += all;
            __syncthreads();
        }
    }
    const uint32_t baseG = a.bases[0] + a.blockGroups[blockIdx.x];
    const uint32_t baseL = a.bases[1] + (uint32_t)a.blockLateSubmit[blockIdx.x];
    for (uint32_t e = 0; e < kBigPerThread; ++e) {
        const uint32_t t = blockIdx.x * kBigChunk + e * kBigThreads + tid;
        const uint32_t word = t < n ? a.word[t] : 0u;
        if (word & kWordLate) a.lateIds[baseL + a.localOff[t]] = a.ids[t];              // :162-167
        const bool submit = (word & kWordSubmit) != 0 && !(word & kWordLate);
        const uint32_t groups = word & kGroupMask;
        const uint32_t lod = (word >> 27) & 7u;
        const uint32_t off = submit ? baseG + a.localOff[t] : 0u;                       // :65
        const bool dropped = submit && off + groups >= a.maxGroups;                     // :69-74 (Q2)
        if (dropped && groups != 0 && off < a.maxGroups && a.argsWords > 3) a.dispatchArgs[3] = off;
        const bool emitRec = submit && !dropped;
        const uint32_t id = emitRec ? a.ids[t] : 0u;
        if (emitRec) {                                                                  // :76-84
            // four 12-byte records = three 16-byte stores (dword-aligned: the back end runs in unaligned-access mode): a
            // quarter of the store instructions, each of which touches a line per lane or two
            struct __attribute__((packed, aligned(4))) Quad { uint32_t x, y, z, w; };
            uint32_t* dst = reinterpret_cast<uint32_t*>(a.records + off);
            uint32_t i = 0;
            for (; i + 4u <= groups; i += 4u) {
                const uint32_t o = i * kNumThreadsPerWave;
                *reinterpret_cast<Quad*>(dst + 3u * i) = Quad{ id, lod, o, id };
                *reinterpret_cast<Quad*>(dst + 3u * i + 4u) = Quad{ lod, o + kNumThreadsPerWave, id, lod };
                *reinterpret_cast<Quad*>(dst + 3u * i + 8u) = Quad{ o + 2u * kNumThreadsPerWave, id, lod, o + 3u * kNumThreadsPerWave };
            }
            for (; i < groups; ++i) {
                MeshletAmplificationData rec = { id, lod, i * kNumThreadsPerWave };
                a.records[off + i] = rec;
            }
        }
        // Place in the tile-ordered list: the record resolved through the LOD table by classify, once per instance
        // (basepass.hlsl:54-63): the meshlet cull then reads 16 bytes of this list and one 64-byte block of the instance
        // cache per record, nothing else.  A run's entries are contiguous, the runs of a wave's lanes are not: written lane
        // by lane, each of the (typically four) 16-byte stores touches 64 lines.  So FOUR LANES write one run's first four
        // entries (64 contiguous bytes): 16 lines per store instruction.  Longer runs finish lane by lane.
        const bool place = emitRec && groups != 0 && binned;
        uint32_t p = 0, numMeshlets = 0, mbase = 0;
        if (place) {
            p = atomicAdd(&s_cursor[a.tileOf[t]], groups);
            const uint2 li = a.lodSel[t];
            numMeshlets = li.x; mbase = li.y;
        }
        const uint32_t head = place ? (groups < 4u ? groups : 4u) : 0u;
        const uint32_t j = lane & 3u;
#pragma unroll
        for (uint32_t it = 0; it < 4; ++it) {
            const int src = (int)(it * 16u + (lane >> 2));
            const uint32_t sp = __shfl(p, src), soff = __shfl(off, src), sid = __shfl(id, src);
            const uint32_t sbase = __shfl(mbase, src), snm = __shfl(numMeshlets, src), shead = __shfl(head, src);
            if (j < shead) {
                const uint32_t first = j * kNumThreadsPerWave;
                const uint32_t cnt = snm > first ? (snm - first < kNumThreadsPerWave ? snm - first : kNumThreadsPerWave) : 0u;
                if (sp + j < a.permCapacity) a.perm[sp + j] = make_uint4(soff + j, sid, sbase + first, cnt);
            }
        }
        if (place)
            for (uint32_t i = 4; i < groups; ++i) {
                const uint32_t first = i * kNumThreadsPerWave;
                const uint32_t cnt = numMeshlets > first ? (numMeshlets - first < kNumThreadsPerWave ? numMeshlets - first : kNumThreadsPerWave) : 0u;
                if (p + i < a.permCapacity) a.perm[p + i] = make_uint4(off + i, id, mbase + first, cnt);
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Small passes (at most kFusedMaxEntries list entries: one rank's share of a sharded scene, real assets): classify, scan
// and emit in ONE launch -- two links less in the frame's chain of dependent launches (DESIGN.md "Launch chain").  A
// workgroup takes tiles of 256 entries in ticket order; a tile publishes its three sums (groups | late, submits) in
// two 64-bit relaxed agent-scope atomics of its own 128-byte status line and then adds up the sums of ALL its
// predecessors (at most 511: two per thread, one round trip) -- a tile only waits for tiles with smaller tickets, which
// are held by workgroups already running.  Same canonical order as the three-kernel path; no tile binning (the
// processing order of the meshlet pass stays canonical: measured a wash at this size).
constexpr uint32_t kFusedMaxEntries = 1u << 17;
constexpr uint32_t kFusedMaxTiles = kFusedMaxEntries / kBlock;       // 512
// The LATE pass takes this kernel whatever its capacity (up to 4096 tiles = 1 M list entries): its list is the late list,
// typically an eighth of the capacity or less, its length is only known on the device, and its three-kernel form was a
// chain of 30 us (6 + 18 + 6 on C3, the scan's dependent round trips stretched by the early list build running next to
// them) for 100 k entries.  Workgroups past the list's end leave before they take a ticket, so the grid can cover the
// capacity; the tile order of the meshlet cull is given up for the late pass (measured: worth nothing there).
constexpr uint32_t kFusedLateMaxTiles = 4096;
constexpr uint32_t kFusedStatusStride = 16;                          // 64-bit words per tile: one 128-byte line
constexpr unsigned long long kFusedFlag = 1ull << 63, kFusedPoison = 1ull << 60;

template <int LATE>
__global__ __launch_bounds__(kBlock) void instanceFusedKernel(InstanceCullArgs a, trhip::QuadArgs q)
{
    // Workgroups past the pass's own (early pass only, when the HZB's footprint-min table is stale): one strip of the table
    // each (hzb_quad.hip.h).  The meshlet cull behind this pass reads the table; built here it costs no launch of its own and
    // no cross-stream dependency -- on a rank's share of a sharded scene the side-stream rebuild (fork + join) cost the frame
    // as much as the table saved it (profiles/r4/experiments.md).
    if (!LATE && blockIdx.x >= a.numBlocks) {
        __shared__ float s_t[9][trhip::kQuadStripCols + 1];
        trhip::hzbQuadStrip(q, blockIdx.x - a.numBlocks, s_t);
        return;
    }
    __shared__ uint32_t s_waveG[kBlock / 64], s_waveL[kBlock / 64], s_waveS[kBlock / 64];
    __shared__ unsigned long long s_preG[kBlock / 64], s_preLS[kBlock / 64];
    __shared__ uint32_t s_tile;
    const uint32_t n = threadCount<LATE>(a);
    const uint32_t numTiles = (n + kBlock - 1) / kBlock < a.numBlocks ? (n + kBlock - 1) / kBlock : a.numBlocks;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // The counters the pass continues from.  The LAST tile of this launch rewrites these words; every workgroup that stays
    // takes its ticket after reading them, and the last tile closes only after all tickets are taken -- but that order
    // rests on program order alone unless the loads are atomics the compiler may not sink below the ticket / the spin loop
    // (relaxed, agent scope: an L2 read, no fence needed -- the values were written by an earlier launch).
    const uint32_t baseX = __hip_atomic_load(&a.dispatchArgs[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t baseLate = LATE ? 0u : __hip_atomic_load(a.lateCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The grid covers the capacity (>= numTiles workgroups): exactly max(numTiles, 1) of them stay, each takes ONE ticket
    // = one tile.  (An empty pass is closed by workgroup 0 as "tile 0".)
    if (blockIdx.x >= (numTiles ? numTiles : 1u)) return;
    {
        if (tid == 0) s_tile = atomicAdd(a.fusedTicket, 1u);
        __syncthreads();
        const uint32_t tile = s_tile;
        const uint32_t t = tile * kBlock + tid;
        uint32_t word = 0, tileUnused = 0;
        uint2 lodUnused;
        if (t < n) word = classify<LATE>(a, a.ids[t], &tileUnused, &lodUnused);
        const uint32_t g = (word & kWordSubmit) ? (word & kGroupMask) : 0u;
        const uint32_t late = word >> 31;
        const uint32_t submit = (word >> 30) & 1u;
        const uint32_t incG = waveInclusiveScan(g, lane);
        const uint32_t incL = waveInclusiveScan(late, lane);
        const uint32_t cntS = (uint32_t)__popcll(__ballot(submit != 0));
        if (lane == 63) { s_waveG[wave] = incG; s_waveL[wave] = incL; s_waveS[wave] = cntS; }
        __syncthreads();
        uint32_t baseG = 0, baseL = 0, totG = 0, totL = 0, totS = 0;
#pragma unroll
        for (uint32_t w = 0; w < kBlock / 64; ++w) {
            if (w < wave) { baseG += s_waveG[w]; baseL += s_waveL[w]; }
            totG += s_waveG[w]; totL += s_waveL[w]; totS += s_waveS[w];
        }
        const uint32_t localOff = late ? (baseL + incL - late) : (baseG + incG - g);
        // ---- publish this tile's sums, add up the predecessors' -----------------------------------------------------
        unsigned long long* line = a.fusedStatus + (uint64_t)tile * kFusedStatusStride;
        if (tid == 0 && numTiles != 0) {
            __hip_atomic_store(&line[0], kFusedFlag | (unsigned long long)totG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&line[1], kFusedFlag | ((unsigned long long)totS << 32) | (unsigned long long)totL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned long long sumG = 0, sumLS = 0;
        for (uint32_t j = tid; j < tile; j += kBlock) {                              // at most two predecessors per thread
            const unsigned long long* pl = a.fusedStatus + (uint64_t)j * kFusedStatusStride;
            unsigned long long v0 = 0, v1 = 0;
            uint32_t spins = 0;
            for (;;) {
                v0 = __hip_atomic_load(&pl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v1 = __hip_atomic_load(&pl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v0 & v1 & kFusedFlag) != 0ull) { v0 &= ~kFusedFlag; v1 &= ~kFusedFlag; break; }
                if (++spins > (1u << 22)) { v0 = kFusedPoison; v1 = 0; break; }      // never seen: a wrong count fails parity, a hang would take the GPU down
                __builtin_amdgcn_s_sleep(1);
            }
            sumG += v0; sumLS += v1;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { sumG += __shfl_xor(sumG, d); sumLS += __shfl_xor(sumLS, d); }
        if (lane == 0) { s_preG[wave] = sumG; s_preLS[wave] = sumLS; }
        __syncthreads();
        unsigned long long preG = 0, preLS = 0;
#pragma unroll
        for (uint32_t w = 0; w < kBlock / 64; ++w) { preG += s_preG[w]; preLS += s_preLS[w]; }
        const uint32_t prefixG = (uint32_t)preG, prefixL = (uint32_t)preLS;           // (a poisoned sum shows up in X below)
        // ---- emit (gpuculling.hlsl:60-84, :162-167), as instanceEmitKernel ---------------------------------------------
        if (t < n) {
            if (word & kWordLate) {
                a.lateIds[baseLate + prefixL + localOff] = a.ids[t];
            } else if (word & kWordSubmit) {
                const uint32_t groups = word & kGroupMask;
                const uint32_t lod = (word >> 27) & 7u;
                const uint32_t off = baseX + prefixG + localOff;                         // :65
                if (off + groups >= a.maxGroups) {                                       // :69-74 (Q2)
                    if (groups != 0 && off < a.maxGroups && a.argsWords > 3) a.dispatchArgs[3] = off;
                } else {
                    const uint32_t id = a.ids[t];
                    for (uint32_t i = 0; i < groups; ++i) {                              // :76-84
                        MeshletAmplificationData rec = { id, lod, i * kNumThreadsPerWave };
                        a.records[off + i] = rec;
                    }
                }
            }
        }
        // ---- the last tile closes the pass (what instanceScanKernel's thread 0 does) -----------------------------------
        if (tid == 0 && (tile == numTiles - 1u || numTiles == 0)) {
            const unsigned long long allG = preG + totG;
            const uint32_t allL = prefixL + totL, allS = (uint32_t)(preLS >> 32) + totS;
            const uint32_t X = (allG >> 40) ? 0xFFFFFFFFu : baseX + (uint32_t)allG;      // poisoned -> an impossible count
            a.bases[0] = baseX;
            a.bases[1] = baseLate;
            a.dispatchArgs[0] = X;                                                       // :65 (counter still counts drops, Q2)
            if (allS != 0) { a.dispatchArgs[1] = 1; a.dispatchArgs[2] = 1; }            // :66-67
            // valid records: X unless an instance is dropped at the capacity -- then that instance writes its offset
            // above (it exists iff X reaches the capacity from below it)
            if (a.argsWords > 3 && (X < a.maxGroups || baseX >= a.maxGroups)) a.dispatchArgs[3] = X;
            if (!LATE) {
                const uint32_t lateTotal = baseLate + allL;
                *a.lateCount = lateTotal;                                                // :165
                if (a.lateArgsOut) { a.lateArgsOut[0] = (lateTotal + 63u) / 64u; a.lateArgsOut[1] = 1; a.lateArgsOut[2] = 1; }
            }
            a.permHeader[0] = 0u;                                                        // canonical processing order
            a.permHeader[1] = X;
        }
    }
}

__global__ void buildLateCullIndirectArgsKernel(const uint32_t* count, uint32_t* args)
{
    args[0] = (count[0] + 63u) / 64u;                                               // gpuculling.hlsl:192 (Q1)
    args[1] = 1;
    args[2] = 1;
}

__global__ void shardLateInfoKernel(const uint32_t* counts, uint32_t world, uint32_t rank, uint32_t* info)
{
    uint32_t below = 0, all = 0;
    for (uint32_t p = 0; p < world; ++p) {
        const uint32_t c = counts[p];
        if (p < rank) below += c;
        all += c;
    }
    info[0] = below;
    info[1] = all;
}

int fillHzb(const trhip::DispatchCtx& ctx, trhip_texture_t* tex, const Vector2U& dims, bool occlusion, cm::Hzb* out)
{
    memset(out, 0, sizeof *out);
    if (!occlusion) return TRHIP_OK;
    TRHIP_REQUIRE(tex, "%s: occlusion culling enabled but no HZB texture bound", ctx.shaderName);
    TRHIP_REQUIRE(tex->format == TRHIP_FORMAT_R16_FLOAT, "%s: HZB '%s' is not R16_FLOAT", ctx.shaderName, tex->name.c_str());
    TRHIP_REQUIRE(tex->width == dims.x && tex->height == dims.y,
                  "%s: m_HZBDimensions %ux%u does not match HZB texture %ux%u", ctx.shaderName, dims.x, dims.y, tex->width, tex->height);
    out->base = (const _Float16*)tex->ptr;
    out->width = tex->width; out->height = tex->height; out->mips = tex->mips;
    for (uint32_t k = 0; k < tex->mips; ++k) out->mipOffset[k] = (uint32_t)(tex->mipOffset[k] / 2);
    return TRHIP_OK;
}

// The early fused pass's launch: its own workgroups + (quadOwner, stale table) one workgroup per strip of the HZB's footprint-min table
std::function<int(hipStream_t)> fusedEarlyLaunch(const InstanceCullArgs& a, trhip_texture_t* quadOwner)
{
    return [a, quadOwner](hipStream_t s) {
        trhip::QuadArgs q;
        memset(&q, 0, sizeof q);
        uint32_t strips = 0;
        if (quadOwner && quadOwner->quadBuiltVersion != quadOwner->version) {            // (submission order: every earlier write of the HZB is counted)
            q = trhip::quadArgs(quadOwner);
            strips = q.firstStrip[q.mips];
            quadOwner->quadBuiltVersion = quadOwner->version;
        }
        TRHIP_LAUNCH(instanceFusedKernel<0>, dim3(a.numBlocks + strips), dim3(kBlock), 0, s, a, q);
        return trhip::launchStatus("instanceFusedKernel"); };
}

// Share of the table's strips built in the scan launch (the rest in the emit launch): TRHIP_QUAD_SCAN_PERCENT, experiments
inline uint32_t quadScanStrips(uint32_t all)
{
    static const uint32_t pct = [] { const char* e = getenv("TRHIP_QUAD_SCAN_PERCENT"); const int v = e ? atoi(e) : 50; return (uint32_t)(v < 0 ? 0 : v > 100 ? 100 : v); }();
    return (uint32_t)((uint64_t)all * pct / 100u);
}

// The early three-kernel pass's scan launch: its own workgroups + (quadOwner, stale table) one workgroup per four table strips of
// its share; `share` tells the emit launch of the same pass whether the table is being rebuilt
std::function<int(hipStream_t)> scanEarlyLaunch(const InstanceCullArgs& a, trhip_texture_t* quadOwner, std::shared_ptr<QuadShare> share)
{
    return [a, quadOwner, share](hipStream_t s) {
        trhip::QuadArgs q;
        memset(&q, 0, sizeof q);
        uint32_t extra = 0, stripEnd = 0;
        if (share) share->stale = false;
        if (quadOwner && quadOwner->quadBuiltVersion != quadOwner->version) {            // (submission order: every earlier write of the HZB is counted)
            q = trhip::quadArgs(quadOwner);
            stripEnd = share ? quadScanStrips(q.firstStrip[q.mips]) : q.firstStrip[q.mips];
            extra = (stripEnd + kScanStripsPerGroup - 1u) / kScanStripsPerGroup;
            quadOwner->quadBuiltVersion = quadOwner->version;
            if (share) share->stale = true;
        }
        TRHIP_LAUNCH(instanceScanKernel<0>, dim3(1 + kScanTileGroups + extra), dim3(kScanThreads), 0, s, a, q, stripEnd);
        return trhip::launchStatus("instanceScanKernel"); };
}

template <int LATE>
int recordGPUCulling(trhip::DispatchCtx& ctx)
{
    // Binding set of BasePassRenderers.cpp:351-362.
    const GPUCullingPassConstants* k = (const GPUCullingPassConstants*)ctx.constants(0, sizeof(GPUCullingPassConstants));
    TRHIP_REQUIRE(k, "%s: constant buffer b0 (GPUCullingPassConstants, 180 bytes) missing", ctx.shaderName);
    trhip_buffer_t* instances = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 0);
    trhip_buffer_t* ids = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 1);
    trhip_buffer_t* meshData = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 2);
    trhip_texture_t* hzb = ctx.texture(TRHIP_BIND_TEXTURE_SRV, 3);
    trhip_buffer_t* records = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 0);
    trhip_buffer_t* args = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 1);
    trhip_buffer_t* lateCount = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 2);
    trhip_buffer_t* lateIds = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 3);
    trhip_buffer_t* shardLate = LATE ? ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 4) : nullptr;   // optional, multi-GPU only
    TRHIP_REQUIRE(!shardLate || shardLate->byteSize >= 8, "%s: shard late info (t4) smaller than 8 bytes", ctx.shaderName);
    TRHIP_REQUIRE(instances && ids && meshData && records && args && lateCount && lateIds,
                  "%s: needs SRVs t0..t2 and UAVs u0..u3 (BasePassRenderers.cpp:351-362)", ctx.shaderName);
    const bool occlusion = (k->m_CullingFlags & kCullingFlagOcclusionCullingEnable) != 0;
    TRHIP_REQUIRE(instances->byteSize % sizeof(BasePassInstanceConstants) == 0, "%s: instance buffer size is not a multiple of 144", ctx.shaderName);
    TRHIP_REQUIRE(meshData->byteSize % sizeof(MeshData) == 0, "%s: mesh data buffer size is not a multiple of 156", ctx.shaderName);
    TRHIP_REQUIRE(records->byteSize >= sizeof(MeshletAmplificationData), "%s: amplification buffer too small", ctx.shaderName);
    TRHIP_REQUIRE(args->byteSize >= 12, "%s: dispatch-arguments buffer smaller than 12 bytes", ctx.shaderName);
    if (LATE) {
        TRHIP_REQUIRE(ctx.indirect, "%s: LATE_CULL=1 is dispatched indirectly (BasePassRenderers.cpp:399)", ctx.shaderName);
        TRHIP_REQUIRE(occlusion, "%s: LATE_CULL=1 without occlusion culling", ctx.shaderName);
        TRHIP_REQUIRE((uint64_t)k->m_NbInstances * 4 <= lateIds->byteSize, "%s: late id buffer smaller than m_NbInstances", ctx.shaderName);
    } else {
        TRHIP_REQUIRE(!ctx.indirect, "%s: LATE_CULL=0 is dispatched directly", ctx.shaderName);
        TRHIP_REQUIRE((uint64_t)k->m_NbInstances * 4 <= ids->byteSize, "%s: m_NbInstances %u exceeds the id buffer", ctx.shaderName, k->m_NbInstances);
        if (occlusion)
            TRHIP_REQUIRE((uint64_t)k->m_NbInstances * 4 <= lateIds->byteSize, "%s: late id buffer smaller than m_NbInstances", ctx.shaderName);
    }
    TRHIP_REQUIRE(lateCount->byteSize >= 4, "%s: late counter buffer too small", ctx.shaderName);

    InstanceCullArgs a;
    memset(&a, 0, sizeof a);
    a.k = *k;
    int rc = fillHzb(ctx, hzb, k->m_HZBDimensions, occlusion, &a.hzb);
    if (rc != TRHIP_OK) return rc;
    a.instances = (const BasePassInstanceConstants*)instances->ptr;
    a.ids = LATE ? (const uint32_t*)lateIds->ptr : (const uint32_t*)ids->ptr;
    a.meshData = (const MeshData*)meshData->ptr;
    a.records = (MeshletAmplificationData*)records->ptr;
    a.dispatchArgs = (uint32_t*)args->ptr;
    a.lateCount = (uint32_t*)lateCount->ptr;
    a.lateIds = (uint32_t*)lateIds->ptr;
    a.indirectArgs = LATE ? (const uint32_t*)((const char*)ctx.argsBuffer->ptr + ctx.argsOffset) : nullptr;
    a.shardLate = shardLate ? (const uint32_t*)shardLate->ptr : nullptr;
    const uint64_t direct = (uint64_t)ctx.gx * kNumThreadsPerWave;
    a.directThreads = direct > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)direct;
    const uint64_t cap = records->byteSize / sizeof(MeshletAmplificationData);
    a.maxGroups = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
    a.argsWords = args->byteSize >= 16 ? 4u : 3u;

    const uint32_t nMax = k->m_NbInstances;
    if (nMax == 0) return TRHIP_OK;

    // instance cull cache: (re)built, in submission order, when the instance or the mesh buffer has been written
    rc = trhip::instanceCacheEnsure(instances);
    if (rc != TRHIP_OK) return rc;
    a.cache = instanceCacheLayout(instances->cullCache, instances->byteSize / sizeof(BasePassInstanceConstants));
    a.numInstances = (uint32_t)(instances->byteSize / sizeof(BasePassInstanceConstants));
    ctx.emit("instance_cache", [instances, meshData](hipStream_t s) { return trhip::instanceCacheLaunchBuild(instances, meshData, s); });
    static const bool noFused = getenv("TRHIP_NO_FUSED_INSTANCE") != nullptr;          // tests: the three-kernel path on small passes
    const bool fusedPath = (nMax <= kFusedMaxEntries || (LATE && (nMax + kBlock - 1) / kBlock <= kFusedLateMaxTiles)) && !noFused;
    a.numBlocks = fusedPath ? (nMax + kBlock - 1) / kBlock : (nMax + kBigChunk - 1) / kBigChunk;
    a.word = (uint32_t*)ctx.scratch((size_t)nMax * 4);
    a.localOff = (uint32_t*)ctx.scratch((size_t)nMax * 4);
    a.blockGroups = (uint32_t*)ctx.scratch((size_t)a.numBlocks * 4);
    a.blockLateSubmit = (uint64_t*)ctx.scratch((size_t)a.numBlocks * 8);
    a.bases = (uint32_t*)ctx.scratch(16);
    a.tileHist = (uint32_t*)ctx.scratch(fusedPath ? 16 : (size_t)a.numBlocks * kNumTiles * 4);   // fully written by classify before scan reads it
    a.tileTotal = (uint32_t*)ctx.scratch(kNumTiles * 4);
    a.tileOf = (uint16_t*)ctx.scratch((size_t)nMax * 2);
    a.lodSel = (uint2*)ctx.scratch((size_t)nMax * 8);
    TRHIP_REQUIRE(a.word && a.localOff && a.blockGroups && a.blockLateSubmit && a.bases && a.tileHist && a.tileTotal && a.tileOf && a.lodSel, "%s: scratch allocation failed", ctx.shaderName);
    // sidecar of the amplification buffer: header + one u32 per record slot
    {
        const uint64_t need = (uint64_t)kPermHeaderWords * 4 + (uint64_t)a.maxGroups * 16;
        if (records->sidecarBytes < need) {
            if (records->sidecar) { (void)hipStreamSynchronize(ctx.cl->dev->stream); (void)hipFree(records->sidecar); records->sidecar = nullptr; records->sidecarBytes = 0; }
            void* p = nullptr;
            TRHIP_HIP(hipMalloc(&p, (size_t)need));
            records->sidecar = p;
            records->sidecarBytes = need;
        }
        a.permHeader = (uint32_t*)records->sidecar;
        a.perm = (uint4*)(a.permHeader + kPermHeaderWords);
        a.permCapacity = a.maxGroups;
    }
    // The early meshlet cull that follows this pass resolves its HZB lookups through the footprint-min table of
    // this same HZB (hzb_quad.hip.h): bring the table up to date with extra workgroups of this pass's own scan / fused launch.
    static const bool noInlineQuad = getenv("TRHIP_NO_INLINE_QUAD") != nullptr;         // experiments: its own launch on the side stream, beside the instance pass
    const bool wantTable = !LATE && occlusion && a.maxGroups >= trhip::tableMinGroups(); // the rule of recordASMain (k_basepass_as.hip)
    const bool inlineQuad = wantTable && !noInlineQuad;                                 // extra workgroups of the fused launch (small pass) / of the scan launch (large pass) build it
    if (wantTable) {
        rc = inlineQuad ? trhip::hzbQuadEnsure(hzb) : trhip::hzbQuadEmitBuild(ctx, hzb);
        if (rc != TRHIP_OK) return rc;
    }

    if (fusedPath) {
        const size_t words = (size_t)a.numBlocks * kFusedStatusStride * 2 + 4;
        uint32_t* mem = (uint32_t*)ctx.scratch(words * 4);
        TRHIP_REQUIRE(mem, "%s: scratch allocation failed", ctx.shaderName);
        rc = ctx.cl->recordClearWords(mem, words, 0, true);
        if (rc != TRHIP_OK) return rc;
        a.fusedStatus = (unsigned long long*)mem;
        a.fusedTicket = mem + (size_t)a.numBlocks * kFusedStatusStride * 2;
        trhip_texture_t* quadOwner = inlineQuad ? hzb : nullptr;
        if (quadOwner) ctx.cl->use(quadOwner->quad, ctx.cl->ops.size(), true);           // this command (re)writes the table
        if (LATE) ctx.emit("fused", [a](hipStream_t s) {
            trhip::QuadArgs q;
            memset(&q, 0, sizeof q);
            TRHIP_LAUNCH(instanceFusedKernel<LATE>, dim3(a.numBlocks), dim3(kBlock), 0, s, a, q);
            return trhip::launchStatus("instanceFusedKernel"); });
        else ctx.emit("fused", fusedEarlyLaunch(a, quadOwner));
        if (!LATE && occlusion) ctx.cl->peephole = { ctx.cl->ops.size() - 1, "gpuculling_early", std::make_shared<EarlyCullNote>(EarlyCullNote{ a, ctx.cl->ops.size() - 1, true, quadOwner }) };
        return TRHIP_OK;
    }

    ctx.emit("classify", [a](hipStream_t s) {
        TRHIP_LAUNCH(instanceClassifyKernel<LATE>, dim3(a.numBlocks), dim3(kBigThreads), 0, s, a);
        return trhip::launchStatus("instanceClassifyKernel"); });
    trhip_texture_t* quadOwner = inlineQuad ? hzb : nullptr;
    std::shared_ptr<QuadShare> share = quadOwner ? std::make_shared<QuadShare>() : nullptr;
    if (quadOwner) ctx.cl->use(quadOwner->quad, ctx.cl->ops.size(), true);               // the scan command (re)writes the table (and the emit command behind it)
    if (LATE) ctx.emit("scan", [a](hipStream_t s) {
        trhip::QuadArgs q;
        memset(&q, 0, sizeof q);
        TRHIP_LAUNCH(instanceScanKernel<LATE>, dim3(1 + kScanTileGroups), dim3(kScanThreads), 0, s, a, q, 0u);
        return trhip::launchStatus("instanceScanKernel"); });
    else ctx.emit("scan", scanEarlyLaunch(a, quadOwner, share));
    const size_t scanOp = ctx.cl->ops.size() - 1;
    if (quadOwner) ctx.cl->use(quadOwner->quad, ctx.cl->ops.size(), true);               // the emit command writes the rest of the table
    ctx.emit("emit", [a, quadOwner, share](hipStream_t s) {
        trhip::QuadArgs q;
        memset(&q, 0, sizeof q);
        uint32_t extra = 0, stripBegin = 0;
        if (!LATE && quadOwner && share && share->stale) {                                  // the scan launch of this pass rebuilt the first strips
            q = trhip::quadArgs(quadOwner);
            stripBegin = quadScanStrips(q.firstStrip[q.mips]);
            extra = (q.firstStrip[q.mips] - stripBegin + kEmitStripsPerGroup - 1u) / kEmitStripsPerGroup;
        }
        TRHIP_LAUNCH(instanceEmitKernel<LATE>, dim3(a.numBlocks + extra), dim3(kBigThreads), 0, s, a, q, stripBegin);
        return trhip::launchStatus("instanceEmitKernel"); });
    if (!LATE && occlusion) ctx.cl->peephole = { ctx.cl->ops.size() - 1, "gpuculling_early", std::make_shared<EarlyCullNote>(EarlyCullNote{ a, scanOp, false, quadOwner, share }) };
    return TRHIP_OK;
}

int recordBuildLateArgs(trhip::DispatchCtx& ctx)
{
    // BasePassRenderers.cpp:379-388
    trhip_buffer_t* count = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 0);
    trhip_buffer_t* args = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 0);
    TRHIP_REQUIRE(count && args, "%s: needs SRV t0 (late count) and UAV u0 (indirect args)", ctx.shaderName);
    TRHIP_REQUIRE(count->byteSize >= 4 && args->byteSize >= 12, "%s: buffers too small", ctx.shaderName);
    TRHIP_REQUIRE(!ctx.indirect && ctx.gx == 1 && ctx.gy == 1 && ctx.gz == 1, "%s: dispatched as 1x1x1", ctx.shaderName);
    const uint32_t* c = (const uint32_t*)count->ptr;
    uint32_t* a = (uint32_t*)args->ptr;
    // The reference records this dispatch right after the early instance cull that produced the count
    // (BasePassRenderers.cpp:367-389): then the cull's scan kernel, which writes the count, writes the arguments too.
    const trhip_cmdlist_t::Peephole ph = ctx.cl->peephole;
    if (ph.kind && !strcmp(ph.kind, "gpuculling_early") && ph.op != SIZE_MAX && ph.op == ctx.cl->ops.size() - 1) {
        const EarlyCullNote* note = (const EarlyCullNote*)ph.data.get();
        if (note->a.lateCount == c && note->scanOp < ctx.cl->ops.size() && ctx.cl->ops[note->scanOp].lane == 0) {
            InstanceCullArgs fused = note->a;
            fused.lateArgsOut = a;
            const size_t scanOp = note->scanOp;
            if (note->fused) ctx.cl->ops[scanOp].fn = fusedEarlyLaunch(fused, note->quadOwner);
            else ctx.cl->ops[scanOp].fn = scanEarlyLaunch(fused, note->quadOwner, note->share);
            // this dispatch's accesses (count read, arguments written) now happen in the scan command
            for (size_t i = ctx.cl->useMarks.size(); i-- > 0 && ctx.cl->useMarks[i].op >= ctx.cl->ops.size();) ctx.cl->useMarks[i].op = scanOp;
            ctx.cl->peephole = trhip_cmdlist_t::Peephole();
            return TRHIP_OK;
        }
    }
    ctx.emit("main", [c, a](hipStream_t s) {
        TRHIP_LAUNCH(buildLateCullIndirectArgsKernel, dim3(1), dim3(1), 0, s, c, a);
        return trhip::launchStatus("buildLateCullIndirectArgsKernel"); });
    return TRHIP_OK;
}

trhip::ShaderRegistrar r0("gpuculling_CS_GPUCulling LATE_CULL=0", recordGPUCulling<0>, 0);
trhip::ShaderRegistrar r1("gpuculling_CS_GPUCulling LATE_CULL=1", recordGPUCulling<1>, 1);
trhip::ShaderRegistrar r2("gpuculling_CS_BuildLateCullIndirectArgs", recordBuildLateArgs, 0);

} // namespace

namespace trhip
{

int instanceCacheEnsure(trhip_buffer_t* instances)
{
    const uint64_t n = instances->byteSize / sizeof(BasePassInstanceConstants);
    TRHIP_REQUIRE(n >= 1 && n <= 0xFFFFFFFFull, "instance cull cache: instance buffer '%s' size out of range", instances->name.c_str());
    if (instances->cullCacheBytes < n * kInstanceCacheBytesPerInstance) {
        TRHIP_HIP(hipSetDevice(instances->dev->index));
        if (instances->cullCache) {
            int rc = instances->dev->syncAll();
            if (rc != TRHIP_OK) return rc;
            (void)hipFree(instances->cullCache);
            instances->cullCache = nullptr; instances->cullCacheBytes = 0;
        }
        TRHIP_HIP(hipMalloc(&instances->cullCache, (size_t)(n * kInstanceCacheBytesPerInstance)));
        instances->cullCacheBytes = n * kInstanceCacheBytesPerInstance;
        instances->cullCacheInstVersion = 0;
    }
    return TRHIP_OK;
}

int instanceCacheLaunchBuild(trhip_buffer_t* instances, trhip_buffer_t* meshData, hipStream_t s)
{
    const uint64_t vi = instances->version, vm = meshData->version;
    if (instances->cullCacheInstVersion == vi && instances->cullCacheMeshVersion == vm && instances->cullCacheMesh == meshData->ptr) return TRHIP_OK;
    const uint32_t n = (uint32_t)(instances->byteSize / sizeof(BasePassInstanceConstants));
    const uint64_t numMeshes = meshData->byteSize / sizeof(MeshData);
    TRHIP_LAUNCH(instanceCacheKernel, dim3((n + 255u) / 256u), dim3(256), 0, s, (const BasePassInstanceConstants*)instances->ptr, n,
                       (const MeshData*)meshData->ptr, (uint32_t)(numMeshes > 0xFFFFFFFFull ? 0xFFFFFFFFull : numMeshes),
                       instanceCacheLayout(instances->cullCache, n));
    instances->cullCacheInstVersion = vi; instances->cullCacheMeshVersion = vm; instances->cullCacheMesh = meshData->ptr;
    return launchStatus("instanceCacheKernel");
}

} // namespace trhip

extern "C" int trhip_launch_shard_late_info(void* hip_stream, const uint32_t* gathered_counts, uint32_t world, uint32_t rank, uint32_t* info)
{
    if (!gathered_counts || !info || world == 0 || rank >= world)
        return trhip::fail(TRHIP_ERR_INVALID, "launch_shard_late_info: bad arguments (world %u, rank %u)", world, rank);
    TRHIP_LAUNCH(shardLateInfoKernel, dim3(1), dim3(1), 0, (hipStream_t)hip_stream, gathered_counts, world, rank, info);
    return trhip::launchStatus("shardLateInfoKernel");
}
