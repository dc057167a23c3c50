// k_basepass_as.hip -- compute replacement of the amplification shader AS_Main
// ("basepass_AS_Main LATE_CULL={0,1}", alias "basepass_AS_Main_cull") for gfx950.
//
// Reference: source/shaders/basepass.hlsl:40-122, launched by dispatchMeshIndirect from
// BasePassRenderer::RenderInstances (source/BasePassRenderers.cpp:406-503).  One reference group
// = 32 lanes = 32 consecutive meshlets of one (instance, LOD) record; it ends in
// DispatchMesh(numVisible, payload{indices in lane order}).  Here the per-group payload becomes
//   visMask[g]       32-bit lane-visibility ballot of group g            (4 B / 32 meshlets)
//   visibleList[k]   (g << 5) | lane, groups ascending, lanes ascending   (4 B / visible meshlet)
//   drawArgs         {numVisibleTotal, 1, 1}
// i.e. WavePrefixCountBits order inside a group (basepass.hlsl:116) and ascending group order
// across groups (the canonical order of SURVEY.md section 7).  LATE_CULL is unused by the
// reference's AS (Q5), so both permutations are the same kernel.
//
// This is the hot kernel of the path (32 B of MeshletData per meshlet tested).  The kernel as shipped
// (DESIGN.md section 5 has the measurements behind each point; what was tried on the way is in profiles/r*/experiments.md):
//   * persistent grid, 5 workgroups of 4 waves per CU.  A workgroup walks WINDOWS of 128 consecutive records of the tile-
//     ordered list the instance pass wrote; a wave owns a BATCH of 32 of them (no workgroup barrier in the loop): lane l
//     resolves record l from its 16-byte list entry {record, instance, first meshlet, count} and the instance's 64-byte
//     block of the cull cache (instance_cache.hip.h), computes the adjugate and parks 96 B of per-record invariants in the
//     wave's LDS slice.  Entry and block are requested a batch ahead (PIPELINED BATCHES below);
//   * step loop, 16 steps: two records per step (lanes 0-31 / 32-63), per-record data are LDS broadcast reads.  The kernel
//     streams the MESHLET CULL STREAM (MeshletCullStream: the 20 bytes per meshlet the cull reads, as two dense arrays):
//     one global_load_lds_dwordx4 + one global_load_lds_dword per step, whole lines, each requested once, into a 2-slot
//     ring in LDS, refilled at the top of the step; every lane reads ITS OWN meshlet from there.  The ring's loads, the HZB
//     lookup and their waits are hand-counted inline assembly (issueMeshletLoads; tools/check_inflight.py lints them);
//   * every test is evaluated branch-free and the result is the AND (the tests are pure, so this equals the reference's
//     short-circuit order :73-108); predicates are lane masks straight from the vector compares;
//   * footprint-table kernel (TABLE: large early passes) = DEFERRED MODE: projection and cone from v_rsq_f32 / v_rcp_f32
//     with proven bands (cm::projectFiltered, cm::coneBackSure); the HZB lookup is ONE 2-byte load from the footprint-min
//     table (hzb_quad.hip.h), consumed a step later; a lane whose fast values are not certain to decide what the reference
//     decides goes on the wave's list in LDS and is re-evaluated exactly, 32-64 lanes at a time, at batch boundaries;
//   * texel kernel (small passes, late passes, the deferred mode's last resort): cm::stepQuotients -- the exact IEEE square
//     roots and divisions without the v_div_scale / v_div_fmas / v_div_fixup glue whenever the whole wave's operands allow
//     it -- and two texel-pair loads per lookup.  A SHORT pass (at most two records per half-wave of the grid: every late
//     pass of C3) skips the batches: each half-wave evaluates its records exactly from global memory (exactMeshletVisible);
//   * the loop issues no stores: the 32 masks of a batch (WaveActiveCountBits / WavePrefixCountBits :116-120 = the two
//     halves of the ballot) are staged in LDS and leave in one store per wave and batch.
#include <algorithm>
#include <cstdlib>
#include <string>
#include <type_traits>

#if defined(TR_STAMPS) || defined(TR_COUNT_SLOW) || defined(TR_COUNT_PATHS)   // diagnostic builds only (never shipped, never timed)
__device__ unsigned long long g_stampSums[8];
#define g_pathCount (g_stampSums + 4)          // [4] steps on the fast arithmetic path, [5] on the exact path (TR_COUNT_PATHS)
#endif
#include <atomic>
#include <memory>
#include <mutex>
#include <unordered_map>

#include "cull_math.hip.h"
#include "instance_cache.hip.h"
#include "meshlet_exact.hip.h"
#include "trhip_internal.h"

using namespace interop;

namespace
{

constexpr uint32_t kBlock = 256;
constexpr uint32_t kWaves = kBlock / 64;
constexpr uint32_t kBatch = 64;                  // records per wave batch
constexpr uint32_t kSteps = kBatch / 2;          // two records per wave step
#ifndef TR_RING_SLOTS
#define TR_RING_SLOTS 2      /* steps of the meshlet cull stream in flight per wave (1.25 KB each, staged in LDS) */
#endif
#ifndef TR_CULL_BATCH
/* records per wave and prologue: a multiple of 2 * TR_RING_SLOTS.  Measured on C3 with 2 ring slots: 64 records (47 KB of
 * LDS per workgroup, 3 workgroups per CU) 0.511 ms, 32 (34 KB, 4 per CU) 0.482 ms, 16 (5 per CU) 0.498 ms. */
#define TR_CULL_BATCH (TR_RING_SLOTS == 3 ? 30 : 32)
#endif
#ifndef TR_WINDOW_MAP
#define TR_WINDOW_MAP 0      /* see windowOf */
#endif
#ifndef TR_EARLY_PREFETCH
/* 1: a step refills its ring slot as soon as it has read its own meshlets out of it (the top of the step) instead of
 * behind its arithmetic: 0.85 of a step more lead for every prefetch at no cost in LDS.  Needs TR_DEFER (a lookup
 * consumed in its own step would sit behind the prefetch in the in-order return queue). */
#define TR_EARLY_PREFETCH 1
#endif
#ifndef TR_DEFER
/* 1: the occlusion lookup of step s is consumed at the END OF STEP s + 1 (its value rides through a whole step of
 * arithmetic in a register).  Loads return in order, so a wait for the lookup of step s also waits for every ring slot
 * requested before it: consumed in its own step (0, rounds 1-2) the wait pinned the ring to ONE step of lead whatever its
 * depth -- 26 % of the wave-cycles sat in that wait even with the lookups compiled out (profiles/r3/experiments.md). */
#define TR_DEFER 1
#endif
constexpr uint32_t kCullBatch = TR_CULL_BATCH;   // meshlet cull: records a wave resolves per prologue (<= 64: one per lane)
constexpr uint32_t kCullSteps = kCullBatch / 2;
constexpr uint32_t kRingSlots = TR_RING_SLOTS;
static_assert(kCullSteps % kRingSlots == 0 && kCullBatch <= 64, "a batch is a whole number of trips round the ring");
#ifndef TR_CULL_WAVES
#define TR_CULL_WAVES 4
#endif
constexpr uint32_t kCullWaves = TR_CULL_WAVES;   // meshlet cull: waves per workgroup (= per window of 64 * kCullWaves... records)
constexpr uint32_t kCullBlock = 64 * kCullWaves;
#ifndef TR_SHORT_PASS_ROUNDS
#define TR_SHORT_PASS_ROUNDS 2
#endif
constexpr uint32_t kShortPassRounds = TR_SHORT_PASS_ROUNDS;   // texel kernel: passes of at most this many records per half-wave of the grid skip the batch machinery
constexpr uint32_t kSlotBytes = 1280;            // a ring slot: 64 spheres + 64 cone words (issueMeshletLoads)
// Deferred mode (meshletCullKernel): the meshlets whose fast evaluation is not certain wait in the wave's list in LDS (kDefStage
// entries) for their exact re-evaluation: a pass over the whole list at the first batch boundary that finds kDefPassAt entries
// or more, and when the wave ends.  An entry: { record index << 5 | lane, view-space centre x, y, z, radius } -- everything the
// exact OCCLUSION test needs (the frustum test was exact, the cone test certain); centre x = 0xFFFFFFFF marks a meshlet whose
// CONE test was not certain: that one is re-evaluated from its record (exactMeshletVisible).
constexpr uint32_t kDefWords = 5;
constexpr uint32_t kDefStage = 64;
#ifndef TR_DEF_PASS_AT
#define TR_DEF_PASS_AT 32
#endif
constexpr uint32_t kDefPassAt = TR_DEF_PASS_AT;  // ... of EARLIER batches (C3: ~10 entries per batch of 32 records -> a pass of ~30 entries every third batch)
constexpr uint32_t kDefFromRecord = 0xFFFFFFFFu;
static_assert(kDefPassAt <= kDefStage, "deferred list sizes");

struct RecordInfo                                 // per-record invariants parked in LDS (96 B, read as 128-bit words)
{
    float wxy[8];                                 // world matrix rows 0..3: (x, y) pairs (8-byte aligned: read as packed operands)
    float wz[4];                                  //                         z column
    float adjxy[6];                               // MakeAdjugateMatrix rows 0..2: (x, y) pairs
    float adjz[3];                                //                               z column
    float maxScale;
    uint32_t first;                               // m_MeshletDataBufferIdx + m_MeshletGroupOffset (0 when the record tests nothing)
    uint32_t lastOff;                             // 16 * count - 8 (count = lanes with meshletIdx < m_NumMeshlets, 1..32), or 0 for count 0:
                                                  // lane `sub` is active iff 16 * sub + 8 <= lastOff; min(16 * sub, lastOff) & ~15 is the byte
                                                  // offset of the sphere it stages (its own, or the record's last one)
};
__device__ __forceinline__ uint32_t lastOffOf(uint32_t count) { return count ? 16u * count - 8u : 0u; }
__device__ __forceinline__ uint32_t countOf(uint32_t lastOff) { return (lastOff + 8u) >> 4; }
static_assert(sizeof(RecordInfo) == 96, "RecordInfo layout");

__device__ __forceinline__ cm::M43P worldOf(const RecordInfo& ri)
{
    return { { ri.wxy[0], ri.wxy[1] }, { ri.wxy[2], ri.wxy[3] }, { ri.wxy[4], ri.wxy[5] }, { ri.wxy[6], ri.wxy[7] },
             ri.wz[0], ri.wz[1], ri.wz[2], ri.wz[3] };
}
__device__ __forceinline__ cm::M33P adjugateOf(const RecordInfo& ri)
{
    return { { ri.adjxy[0], ri.adjxy[1] }, { ri.adjxy[2], ri.adjxy[3] }, { ri.adjxy[4], ri.adjxy[5] }, ri.adjz[0], ri.adjz[1], ri.adjz[2] };
}

__device__ __forceinline__ uint32_t listGroupCount(const MeshletCullArgs& a)      // the list build's view of it (MeshletCullArgs::listGroups)
{
    return a.listGroups ? a.listGroups[0] : groupCount(a);
}

typedef float v4f __attribute__((ext_vector_type(4)));
// The two records of a step (2 x 32 spheres = 1 KB, 2 x 32 cone words = 256 B of the meshlet cull stream) go from memory
// STRAIGHT INTO LDS (global_load_lds_dwordx4 / _dword: no VGPRs in between) as whole cache lines: lane (half, sub) moves
// sphere `sub` and cone word `sub` of record `half`, so every line is requested from the L2 exactly once, by one
// instruction.  That matters more than anything else in this kernel: its pace is set by the L1 misses a CU can keep in
// flight (TCP_PENDING_STALL 60-70 % of the cycles), not by HBM or by VALU issue.  Each lane then reads ITS OWN meshlet's
// sphere (ds_read_b128, consecutive lanes on consecutive 16 bytes) and cone word (ds_read_b32) out of the staged block.
// (Rounds 2-3 staged the 32-byte MeshletData records themselves: 16 lines per step.  Round 1 kept the chunks in VGPRs and
// completed the (sphere, cone) pairs with DPP swaps, selects and a shuffled ballot: 14 vector + 40 scalar instructions per
// step.  Loading 16 + 4 bytes per lane directly from the records asks the L2 for every line twice: 5 % slower than round 1.)
//
// The compiler's wait-count pass cannot tell LDS-DMA targets apart (any LDS read after a global_load_lds builtin waits
// for vmcnt(0), which would serialise the prefetch), so the ring is driven by hand: the loads, the table lookup that
// shares their counter and every wait on vmcnt in the loop are inline assembly, and the loop contains no other vector
// memory instruction.  Loads return in order, so a wait "until at most N are outstanding" is exact.
//
// LDS layout of a ring slot (per wave): [0,1024) the spheres of lanes 0..63 (record A: lanes 0-31, record B: 32-63),
// [1024,1280) their cone words.
template <bool AFTER_READS = false>
__device__ __forceinline__ void issueMeshletLoads(char* slotLds /* wave-uniform */, const MeshletCullStream& stream, uint32_t firstIdx, uint32_t lastOff, uint32_t sub16 /* 16 * sub */,
                                                  v4f readA = v4f{ 0.f, 0.f, 0.f, 0.f }, uint32_t readB = 0u /* AFTER_READS: what this wave has just read out of the slot */)
{
    // Every lane always loads (lanes past the record's end re-read its last meshlet): the number of loads in flight never
    // depends on the data, and nothing outside the record is read.
#ifdef TR_EXP_NOMEM      /* experiment, results WRONG: every wave streams the same block (cache hits): the kernel without its HBM traffic */
    firstIdx &= 63u;
#endif
    const uint32_t idx = firstIdx + (min(sub16, lastOff) >> 4);                      // first + min(sub, count - 1): basepass.hlsl:65 (< numMeshlets <= 2^32)
    const char* pa = reinterpret_cast<const char*>(stream.sphere) + ((uint64_t)idx << 4);
    const char* pb = reinterpret_cast<const char*>(stream.cone) + ((uint64_t)idx << 2);
    const uint32_t ldsOff = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)slotLds;
    // nt: the 1.1 GB stream is read once and must not evict the HZB table and the per-record data from the L2s
#ifndef TR_DMA_POLICY_ID
#define TR_DMA_POLICY_ID 1       /* experiments (profiles/r2/experiments.md): cache policy of the stream */
#endif
#if TR_DMA_POLICY_ID == 0
#define TR_DMA_POLICY ""
#elif TR_DMA_POLICY_ID == 1
#define TR_DMA_POLICY "nt"
#elif TR_DMA_POLICY_ID == 2
#define TR_DMA_POLICY "sc1"
#elif TR_DMA_POLICY_ID == 3
#define TR_DMA_POLICY "sc0 sc1"
#elif TR_DMA_POLICY_ID == 4
#define TR_DMA_POLICY "sc0 sc1 nt"
#elif TR_DMA_POLICY_ID == 5
#define TR_DMA_POLICY "sc0"
#else
#define TR_DMA_POLICY "sc1 nt"
#endif
    if (AFTER_READS)
        // the slot is overwritten as soon as the loads land: the values read out of it are operands of this statement, so the
        // compiler has waited for them (lgkmcnt) before it
        asm volatile("s_mov_b32 m0, %2\n\t"
                     "global_load_lds_dwordx4 %0, off " TR_DMA_POLICY "\n\t"
                     "s_add_u32 m0, %2, 0x400\n\t"
                     "global_load_lds_dword %1, off " TR_DMA_POLICY
                     :: "v"(pa), "v"(pb), "s"(ldsOff), "v"(readA), "v"(readB) : "memory", "m0", "scc");
    else
        asm volatile("s_mov_b32 m0, %2\n\t"
                     "global_load_lds_dwordx4 %0, off " TR_DMA_POLICY "\n\t"
                     "s_add_u32 m0, %2, 0x400\n\t"
                     "global_load_lds_dword %1, off " TR_DMA_POLICY
                     :: "v"(pa), "v"(pb), "s"(ldsOff) : "memory", "m0", "scc");
}
#define TR_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

#ifdef TR_STAMPS
#define TR_STAMP(i) do { unsigned long long _t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); stampSum[i] += _t - stampLast; stampLast = _t; } while (0)
#else
#define TR_STAMP(i) do {} while (0)
#endif

// TABLE: resolve the HZB lookup through the footprint-min table (one 2-byte load; the early pass, where the
// table is rebuilt once per frame behind the instance pass) or through the texels themselves (two texel-pair
// loads; the late pass, which is small and follows an HZB rebuild).  Same results either way.
#ifndef TR_LK_POLICY_ID         /* cache-policy bits of the table lookup (experiments; measured alike, profiles/r3/experiments.md section 8) */
#define TR_LK_POLICY_ID 0
#endif
#if TR_LK_POLICY_ID == 0
#define TR_LK_POLICY ""
#elif TR_LK_POLICY_ID == 1
#define TR_LK_POLICY " sc0"
#elif TR_LK_POLICY_ID == 2
#define TR_LK_POLICY " sc1"
#elif TR_LK_POLICY_ID == 3
#define TR_LK_POLICY " sc0 sc1"
#elif TR_LK_POLICY_ID == 4
#define TR_LK_POLICY " nt"
#else
#define TR_LK_POLICY " sc1 nt"
#endif
#ifndef TR_CULL_WAVES_PER_EU
#define TR_CULL_WAVES_PER_EU 5   /* waves per SIMD the register allocation aims at (96 VGPRs; a few dwords spilled outside the loop) = workgroups of <= 32 000 B LDS per CU.  With the 32-byte records staged (rounds 2-3) 4, 5 and 6 per CU ran alike; with the 20-byte stream 5 is 2.3 % faster than 4, and the deferred mode at 4 per CU (LDS 32 720 B) ran 11 % slower than at 5 */
#endif
template <bool FRUSTUM, bool OCCLUSION, bool CONE, bool TABLE>
__global__ __launch_bounds__(kCullBlock, TR_CULL_WAVES_PER_EU) void meshletCullKernel(MeshletCullArgs a)
{
#ifdef TR_STAMPS
    unsigned long long stampSum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stampLast = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    __shared__ RecordInfo s_recAll[kCullWaves][kCullBatch + 2 * kRingSlots];   // + 2 per ring slot: the prefetch of the last steps reads past the batch (count 0)
    __shared__ uint32_t s_gIdxAll[kCullWaves][kCullBatch];
    __shared__ uint32_t s_quadOff[16];                 // texel path: mip offsets
    __shared__ uint4 s_mipTab[17];                     // table path: per-mip constants indexed by exponent + 1 (cm::occTailQuad)
    __shared__ float s_coneTab[cm::kConeTabEntries];   // cone byte -> byte / 255 (cm::coneTableEntry)
    constexpr bool kDeferred = OCCLUSION && TABLE;     // the deferred mode (below); the other instantiations decide every lane exactly in place
    __shared__ uint2 s_mipBand[kDeferred ? 17 : 1];    // deferred mode: per level, the band of the level decision (cm::projMipDelta)
    __shared__ uint32_t s_defAll[kCullWaves][kDeferred ? kDefStage * kDefWords : 1u];   // per wave: the deferred list (kDefWords words per entry)
    __shared__ __attribute__((aligned(8))) uint32_t s_maskAll[kCullWaves][kCullBatch + 2];   // [0], [1]: where the deferred resolve of "the step before the first" lands
    __shared__ __attribute__((aligned(16))) char s_ring[kCullWaves][kRingSlots][kSlotBytes];   // per wave: the ring slots of the staged meshlet cull stream

    const uint32_t G = groupCount(a);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t half = lane >> 5, sub = lane & 31u;
    if (a.listGroups && blockIdx.x == 0 && tid == 0) a.listGroups[0] = G;           // the list build's input (see MeshletCullArgs::listGroups)
    const cm::M43 V = cm::loadM43(a.k.m_WorldToView);
    const cm::M43P VP = cm::packM43(V);
    const cm::M33P VR = cm::rot(VP);
    const float coneSlack = cm::coneSlackFactor(V);
    // SHORT PASSES (texel kernel: late passes, small early passes): the batch machinery below gives a wave kCullBatch records to
    // walk in kCullSteps serial steps -- right for a million records, but the late pass of C3 has 2 544 (0.15 % of the frame's
    // meshlets): twenty workgroups walked them for 8-10 us while the rest of the grid looked on.  When the grid has a half-wave
    // for every record (kShortPassRounds times over), every half-wave evaluates its record's 32 meshlets exactly from global
    // memory (exactMeshletVisible: the sequences the step loop's fallbacks run, same results) -- one chain of dependent loads for
    // the whole pass instead of sixteen in a row.
    if (!TABLE && G <= gridDim.x * (2u * kCullWaves) * a.shortPassRounds) {
        for (uint32_t g = (blockIdx.x * kCullWaves + wave) * 2u + half; g < G; g += gridDim.x * kCullWaves * 2u) {
            const bool vis = exactMeshletVisible<FRUSTUM, OCCLUSION, CONE>(a, VP, VR, g, sub);
            const unsigned long long b = __ballot(vis);
            if (sub == 0) a.visMask[g] = (uint32_t)(half ? b >> 32 : b);
        }
        return;
    }
    RecordInfo* s_rec = s_recAll[wave];
    uint32_t* s_gIdx = s_gIdxAll[wave];

    if (tid < 16) s_quadOff[tid] = a.hzb.mipOffset[tid];
    if (TABLE && tid >= 1 && tid <= 16) {
        const uint32_t mip = tid - 1u < a.hzb.mips ? tid - 1u : 0u;
        const uint32_t mw = (a.hzb.width >> mip) ? (a.hzb.width >> mip) : 1u, mh = (a.hzb.height >> mip) ? (a.hzb.height >> mip) : 1u;
        s_mipTab[tid] = make_uint4(a.quad.offset[mip], ((mw >> 3) + 1u) * 64u, __float_as_uint(0.5f * (float)mw), __float_as_uint(0.5f * (float)mh));
    }
    if (kDeferred && tid >= 1 && tid <= 16) { const uint32_t d = cm::projMipDelta(a.bands, tid); s_mipBand[tid] = make_uint2(d, 2u * d); }
    // the fast arithmetic path wants nearPlane in [2^-20, 2^20] (cm::stepQuotients)
    const bool nearInRange = a.k.m_NearPlane >= 0x1p-20f && a.k.m_NearPlane <= 0x1p20f;
    if (CONE) for (uint32_t i = tid; i < cm::kConeTabEntries; i += kCullBlock) s_coneTab[i] = cm::coneTableEntry(i);   // (any workgroup size: TR_CULL_WAVES)
    if (lane < 2 * kRingSlots) { s_recAll[wave][kCullBatch + lane].first = 0; s_recAll[wave][kCullBatch + lane].lastOff = 0; }
    uint32_t* s_def = s_defAll[wave];
    uint32_t* s_mask = s_maskAll[wave];
    __syncthreads();                                                                 // the only workgroup barrier

    // Work decomposition.  Records are processed in screen-tile order when the instance pass published one for exactly
    // this record count (any permutation of [0,G) is a valid processing order: masks are stored by record index);
    // otherwise in record order.  The order is cut into WINDOWS of 64 * kCullWaves = 256 consecutive records; workgroup b takes
    // windows b, b + gridDim, ...; inside a window, wave w runs records {2 * kCullWaves * s + 2 * w + half} at step s.  The
    // waves of a workgroup thus stay inside 256 consecutive records (in tile order: a few dozen instances of one screen
    // region) for a whole batch, and the HZB lookups of a CU keep hitting the same few table rows in its L1.
    // (Round 1 walked ONE window of 2 * numWaves records with all waves of the chip -- good for the L2s, but every CU then
    // touched a different screen region at every step and 3 of 4 lookups missed its L1: -DTR_TEAM_ALL, 3 % slower.)
    const bool usePerm = a.permHeader != nullptr && a.permHeader[0] == 1u && a.permHeader[1] == G;
#ifndef TR_TEAM_ALL
    const uint32_t teamWaves = kCullWaves, waveInTeam = wave, team = blockIdx.x, teams = gridDim.x;
#else
    const uint32_t teamWaves = gridDim.x * kCullWaves, waveInTeam = blockIdx.x * kCullWaves + wave, team = 0u, teams = 1u;
#endif
    const uint32_t superSize = teamWaves * kCullBatch;
    const uint32_t numSuper = (G + superSize - 1) / superSize;
    // The batch entry of lane l = the record of step l/2, half l&1 (0xFFFFFFFF = none): {record index, instance, first
    // meshlet, count} from the tile-ordered list the instance pass wrote, or {record index, instance, lod, group offset}
    // from the record buffer itself.
    auto loadEntry = [&](uint32_t sb_) -> uint4 {
#ifdef TR_BLOCKED_MAP    /* experiment: wave w takes the window's records [32 w, 32 w + 32) -- the four groups of an instance in two consecutive steps of ONE wave */
        const uint64_t e64 = (uint64_t)sb_ * superSize + waveInTeam * kCullBatch + lane;
#else
        const uint64_t e64 = (uint64_t)sb_ * superSize + (lane >> 1) * 2 * teamWaves + 2 * waveInTeam + (lane & 1);
#endif
        if (lane >= kCullBatch || sb_ >= numSuper || e64 >= G) return make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
        const uint32_t e = (uint32_t)e64;
        if (usePerm) return a.perm[e];
        const MeshletAmplificationData rec = a.records[e];
        return make_uint4(e, rec.m_InstanceConstIdx, rec.m_MeshLOD, rec.m_MeshletGroupOffset);
    };
    // Windows are dealt round-robin: team t takes windows t, t + teams, ... for `fullRounds` rounds.  The `left` windows of
    // the last, incomplete round would keep a fraction of the chip busy for a whole window's time while the rest idles
    // (C3: 13.4 windows per workgroup = 14 rounds for 13.4 rounds of work); they are CUT into pieces of `stepsPer` steps
    // instead, one piece per team, so that the last round takes left / teams of a window's time.
#ifndef TR_NO_SPLIT_TAIL
    const uint32_t fullRounds = numSuper / teams, left = numSuper - fullRounds * teams;
    uint32_t piecesPer = 1u, stepsPer = kCullSteps;
    if (left) {
        const uint32_t k = std::min(teams / left, kCullSteps / kRingSlots);
        if (k > 1u) {
            stepsPer = ((kCullSteps + k - 1u) / k + kRingSlots - 1u) / kRingSlots * kRingSlots;
            piecesPer = (kCullSteps + stepsPer - 1u) / stepsPer;
        }
    }
#else
    const uint32_t fullRounds = (numSuper + teams - 1u) / teams, left = 0u, piecesPer = 1u, stepsPer = kCullSteps;
#endif
    // iteration `it` of this team: window (0xFFFFFFFF = none) and the steps [s0, s1) of it
    auto windowOf = [&](uint32_t it, uint32_t& s0, uint32_t& s1) -> uint32_t {
        s0 = 0u; s1 = kCullSteps;
        if (it < fullRounds) {
            // Which window: the workgroups that share a CU (the dispatcher deals workgroups b, b + CUs, b + 2 CUs ... onto the same
            // CU while the grid is CUs x TR_CULL_WAVES_PER_EU) take ADJACENT windows of the tile-ordered list -- the same screen
            // region, the same few hundred lines of the footprint table in the CU's L1 (TR_WINDOW_MAP 1: adjacent in every round;
            // 2: a CU keeps one contiguous range of the list for its whole life; 0: round-robin over all workgroups, rounds 1-3)
            uint64_t w = (uint64_t)it * teams + team;
#if TR_WINDOW_MAP
            constexpr uint32_t kCo = TR_CULL_WAVES_PER_EU;
            if (teams % kCo == 0u) {
                const uint32_t perRow = teams / kCo, c = team % perRow, j = team / perRow;
                w = TR_WINDOW_MAP == 2 ? ((uint64_t)c * fullRounds + it) * kCo + j : (uint64_t)it * teams + (uint64_t)c * kCo + j;
            }
#endif
            return w < numSuper ? (uint32_t)w : 0xFFFFFFFFu;
        }
        if (it == fullRounds && left && team < left * piecesPer) {
            s0 = (team % piecesPer) * stepsPer;
            s1 = std::min(s0 + stepsPer, kCullSteps);
            return fullRounds * teams + team / piecesPer;
        }
        return 0xFFFFFFFFu;
    };
    // steps [s0, nSteps) of window sb that hold records for this wave (wave-uniform; 0 = nothing left for it)
    auto stepsOf = [&](uint32_t sb_, uint32_t s0_, uint32_t s1_) -> uint32_t {
        if (sb_ == 0xFFFFFFFFu) return 0u;
        const uint32_t sbBase = sb_ * superSize;
#ifdef TR_BLOCKED_MAP
        if (sbBase + waveInTeam * kCullBatch >= G) return 0u;
        const uint32_t remaining = G - sbBase - waveInTeam * kCullBatch;
        uint32_t n = (remaining + 1u) / 2u;
#else
        if (sbBase + 2 * waveInTeam >= G) return 0u;                                 // nothing left for this wave
        const uint32_t remaining = G - sbBase - 2 * waveInTeam;
        uint32_t n = (remaining + 2 * teamWaves - 1) / (2 * teamWaves);
#endif
        n = n < kCullSteps ? (n + kRingSlots - 1u) / kRingSlots * kRingSlots : kCullSteps;   // rounded up to whole trips round the ring
        n = n < s1_ ? n : s1_;                                                       // this team's piece of the window
        return s0_ < n ? n : 0u;                                                     // (only a piece of the partial last window can be empty)
    };
    // PIPELINED BATCHES.  A batch used to begin with three latencies in a row: the drain of the previous batch's last
    // prefetches, the 64-byte instance blocks of its own records (their addresses come from the list entry), the first two ring
    // slots (their addresses come from the resolved records) -- 20 % of a wave's cycles outside its step loop for 7 % of its
    // instructions (profiles/r3/stamps_final_kernel.txt).  Now everything the NEXT batch needs from memory is requested at the
    // end of the current one, behind its last step: the instance blocks (into registers that live only across the batch
    // boundary) and, with the tile-ordered list, whose entries carry {first meshlet, count}, the first two ring slots too --
    // they land in LDS behind the padding prefetches of the last steps, in order.  One wait, in the next prologue, covers the
    // three.  The list entry of batch n + 1 is made available in prologue n (loaded in prologue n - 1), the one of batch n + 2
    // requested there.
#ifndef TR_PIPE
#define TR_PIPE 1
#endif
    constexpr bool kPipe = TR_PIPE != 0 && TR_EARLY_PREFETCH != 0 && kRingSlots == 2;
    uint32_t s0 = 0, s1 = kCullSteps, sbN, s0N, s1N, sbNN, s0NN, s1NN;
    uint32_t sb = windowOf(0u, s0, s1);
    sbN = windowOf(1u, s0N, s1N);
    uint4 entry = loadEntry(sb), entryN = loadEntry(sbN);
    auto blockAddr = [&](const uint4& e) -> const float4* {
#ifdef TR_EXP_INST0     /* experiment, results WRONG: every record reads instance block 0 (what the scattered 64-byte reads cost) */
        const uint32_t cid = 0u;
#else
        const uint32_t cid = e.y < a.numInstances ? e.y : 0u;                        // never read outside the cache (no entry: block 0, unused)
#endif
        return a.cache.world + 4ull * cid;                                           // one 64-byte block: world rows + max scale
    };
    float4 q0, q1, q2, q3;                                                           // the instance blocks of the batch about to begin (lane l: record l)
    { const float4* wr = blockAddr(entry); q0 = wr[0]; q1 = wr[1]; q2 = wr[2]; q3 = wr[3]; }
    bool primed = false;                                                             // wave-uniform: the ring's first two slots were requested at the end of the previous batch
    // ---- DEFERRED MODE (kDeferred: the footprint-table kernel of large passes) ---------------------------------------------
    // The step loop evaluates every meshlet with FAST arithmetic only (cm::stepDeferred: closed-form projection from one
    // v_rsq_f32 per axis and one v_rcp_f32, cone from v_rsq_f32) and, with it, whether the fast values are certain to decide
    // what the reference's exact square roots and divisions decide (cm::projectFiltered has the proof).  A meshlet that is
    // not certain and still matters goes on the wave's deferred list (kDefWords above) and the bit the loop stored for it is
    // OVERWRITTEN later by the exact evaluation: 64 listed meshlets at a time, at batch boundaries, behind the batch's mask
    // store (a wave owns its records: only it touches their mask words), the rest when the wave ends.  On C3 about 0.9 % of the
    // tested meshlets take that way: 1.6 passes of ~200 instructions per wave and launch instead of the exact sequences in every
    // one of the wave's 170 steps (rounds 2-3).  The exact pass resolves its lookups through the table, like the loop.
    uint32_t stCount = 0u;                             // wave-uniform: entries in the list (s_def)
    uint32_t stAtBatch = 0u;                           // ... when the current batch began
    bool batchExact = false;                           // wave-uniform: the list overflowed inside this batch -> the batch is redone exactly, in place, and its entries dropped
    auto exactMeshlet = [&](uint32_t g, uint32_t m) -> bool { return exactMeshletVisible<FRUSTUM, OCCLUSION, CONE>(a, VP, VR, g, m); };
    // the exact occlusion test of a listed meshlet from its view-space sphere (culling.hlsli:36-82 with the compiler's correctly
    // rounded sequences), the lookup through the table like the loop's (cm::occTailQuad), through the texels where the table
    // cannot serve it (a zero bilinear weight)
    auto exactOcclusion = [&](cm::F3 c, float r) -> bool {
        cm::StepQuot q;
        const float crx = c.x * r, cry = c.y * r, crz = c.z * r;                          // :53
        const float czr2 = cm::fma_(c.z, c.z, -(r * r));                                  // :54
        const float vx = cm::sqrt_(cm::fma_(c.x, c.x, czr2)), vy = cm::sqrt_(cm::fma_(c.y, c.y, czr2));   // :56, :60
        q.mn = cm::v2f{ cm::div_(cm::fma_(vx, c.x, -crz), cm::fma_(vx, c.z, crx)), cm::div_(cm::fma_(vy, c.y, -crz), cm::fma_(vy, c.z, cry)) };   // :57, :61
        q.mx = cm::v2f{ cm::div_(cm::fma_(vx, c.x, crz), cm::fma_(vx, c.z, -crx)), cm::div_(cm::fma_(vy, c.y, crz), cm::fma_(vy, c.z, -cry)) };   // :58, :62
        q.depthSphere = cm::div_(a.k.m_NearPlane, c.z - r);                               // :79
        const cm::OccQuad oq = cm::occTailQuad(q, c, r, a.k.m_NearPlane, a.k.m_P00, a.k.m_P11, a.hzb, s_mipTab, a.quad.total);
        const float footprintMin = (float)a.quad.base[oq.iq];
        bool vis = (((oq.accept >> lane) & 1ull) != 0ull) | (oq.depthSphere >= footprintMin);                                 // :48-49, :81
        if ((oq.slow >> lane) & 1ull) vis = cm::occlusionVisible(c, r, a.k.m_NearPlane, a.k.m_P00, a.k.m_P11, a.hzb);
        return vis;
    };
    // re-evaluate the first `n` listed meshlets (n <= 64) and overwrite their bits in the mask array, then close the gap in the
    // list.  The caller guarantees that the masks of those meshlets' batches are in memory (a wave owns its records: only it
    // touches their mask words): inside the kernel a pass runs at the end of a batch, IN FRONT of its mask store, over the
    // entries of EARLIER batches -- their stores were issued a batch ago and the batch's final s_waitcnt vmcnt(0) has covered
    // them -- so a pass waits for nothing but its own table lookups.
    auto runPass = [&](uint32_t n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef TR_COUNT_PATHS
        if (lane == 0) { atomicAdd(&g_stampSums[0], (unsigned long long)n); atomicAdd(&g_stampSums[1], 1ull); }   // deferred meshlets, passes
#endif
#ifndef TR_EXP_NOFIX      /* experiment, results WRONG: the listed meshlets are dropped -- what their exact re-evaluation costs */
        {
            // every lane runs along (cm::occTailQuad works on wave-wide lane masks): the lanes past n re-read entry 0
            const uint32_t* src = s_def + (lane < n ? lane : 0u) * kDefWords;
            const uint32_t e = src[0], w1 = src[1], g = e >> 5, m = e & 31u;
            const cm::F3 c = { __uint_as_float(w1), __uint_as_float(src[2]), __uint_as_float(src[3]) };
            const float r = __uint_as_float(src[4]);
            bool vis = exactOcclusion(c, r);
            if (w1 == kDefFromRecord) vis = exactMeshlet(g, m);                          // (cone not certain; or, harmlessly, a centre whose x has this bit pattern)
            if (lane < n) {
                if (vis) atomicOr(&a.visMask[g], 1u << m);
                else atomicAnd(&a.visMask[g], ~(1u << m));
            }
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // close the gap: entries [n, stCount) move to the front (the list holds at most 64 * 5 words: five words per lane)
        const uint32_t left = (stCount - n) * kDefWords;
        uint32_t v[kDefWords];
#pragma unroll
        for (uint32_t k = 0; k < kDefWords; ++k) v[k] = lane + 64u * k < left ? s_def[n * kDefWords + lane + 64u * k] : 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t k = 0; k < kDefWords; ++k) if (lane + 64u * k < left) s_def[lane + 64u * k] = v[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        stCount -= n;
    };
    for (uint32_t it = 0; sb != 0xFFFFFFFFu; ++it, sb = sbN, s0 = s0N, s1 = s1N, sbN = sbNN, s0N = s0NN, s1N = s1NN) {
        const uint32_t nSteps = stepsOf(sb, s0, s1);                                 // this team's piece of the window: steps [s0, nSteps)
        if (nSteps == 0u) break;
        sbNN = windowOf(it + 2u, s0NN, s1NN);
        TR_STAMP(0);   // between batches
        stAtBatch = stCount;
        // ---- prologue: lane l resolves its record (basepass.hlsl:52-58) from its list entry and its instance block, both
        //      requested a batch ago ---------------------------------------------------------------------------------
        {
            RecordInfo ri;
            ri.lastOff = 0; ri.first = 0;
            const uint4 cur = entry;
            entry = entryN;
            asm volatile("" : "+v"(entry.x), "+v"(entry.y), "+v"(entry.z), "+v"(entry.w));   // the next batch's entry is AVAILABLE from here on (the compiler's wait for it lands here, where the blocks are waited for anyway)
            entryN = loadEntry(sbNN);                                                // the entry of the batch after next: in flight during this batch and the next
            const uint32_t g = cur.x < G && (lane >> 1) >= s0 && (lane >> 1) < nSteps ? cur.x : 0xFFFFFFFFu;
            if (lane < kCullBatch) s_gIdx[lane] = g;
            if (g < G) {
                const uint32_t cid = cur.y < a.numInstances ? cur.y : 0u;
                const cm::F3 r0 = { q0.x, q0.y, q0.z }, r1 = { q0.w, q1.x, q1.y }, r2 = { q1.z, q1.w, q2.x };
                ri.wxy[0] = r0.x; ri.wxy[1] = r0.y; ri.wz[0] = r0.z;
                ri.wxy[2] = r1.x; ri.wxy[3] = r1.y; ri.wz[1] = r1.z;
                ri.wxy[4] = r2.x; ri.wxy[5] = r2.y; ri.wz[2] = r2.z;
                ri.wxy[6] = q2.y; ri.wxy[7] = q2.z; ri.wz[3] = q2.w;
                ri.maxScale = q3.x;                                                  // toyrenderer_common.hlsli:134-140 (cached)
                const cm::F3 a0 = cm::cross3(r1, r2), a1 = cm::cross3(r2, r0), a2 = cm::cross3(r0, r1); // :124-132
                ri.adjxy[0] = a0.x; ri.adjxy[1] = a0.y; ri.adjz[0] = a0.z;
                ri.adjxy[2] = a1.x; ri.adjxy[3] = a1.y; ri.adjz[1] = a1.z;
                ri.adjxy[4] = a2.x; ri.adjxy[5] = a2.y; ri.adjz[2] = a2.z;
                // lanes with meshletIdx = groupOffset + lane < numMeshlets (basepass.hlsl:62-63): {first meshlet, count}
                // already resolved by the instance pass (tile-ordered list), or through the LOD table here
                uint64_t base = cur.z;
                uint32_t cnt = cur.w;
                if (!usePerm) {
                    const uint32_t lodIdx = cur.z < kMaxNumMeshLODs ? cur.z : kMaxNumMeshLODs - 1u;
                    const uint2 li = a.cache.lod(cid, lodIdx);
                    const uint32_t off = cur.w;
                    cnt = li.x > off ? li.x - off : 0u;
                    base = (uint64_t)li.y + off;
                }
                cnt = cnt < 32u ? cnt : 32u;
                if (base + cnt > a.numMeshlets) cnt = 0;                             // never read outside the meshlet buffer
                ri.lastOff = lastOffOf(cnt);
                if (cnt) ri.first = (uint32_t)base;                                  // < numMeshlets <= 2^32 (recordASMain)
            } else if (lane < kCullBatch) {                                          // no record (past the list, or outside this team's piece): tests nothing
                ri.maxScale = 0.f;                                                   // (zeroed here, not in front of the branch: full batches never come this way)
#pragma unroll
                for (int i = 0; i < 8; ++i) ri.wxy[i] = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) ri.wz[i] = 0.f;
#pragma unroll
                for (int i = 0; i < 6; ++i) ri.adjxy[i] = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) ri.adjz[i] = 0.f;
            }
            if (lane < kCullBatch) s_rec[lane] = ri;
        }
        // LDS traffic of one wave is executed in order: the wave-level fence only stops the
        // compiler from moving the reads below above the writes above.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        TR_STAMP(1);   // prologue
        // ---- main loop: two records per step, their MeshletData staged in LDS kRingSlots steps ahead --------------
        // Loads outstanding at the top of a step: 2 per ring slot, this step's slot first; after the step's table lookup and
        // its prefetch: {the other slots' 2 each, lookup, this slot's 2}.  Hence the two waits: vmcnt(2 * (slots - 1))
        // at the top, vmcnt(2) for the lookup.
        char* const ring = s_ring[__builtin_amdgcn_readfirstlane((int)wave)][0];
        const uint32_t ringOff = lane * 16u, ringOffCone = 1024u + lane * 4u;        // this lane's sphere / cone word inside a ring slot
        const uint32_t sub16 = sub * 16u, subEnd = sub * 16u + 8u;                   // lane constants of the stream addressing / the "active" test
        // The occlusion lookup of a step (TABLE: one 2-byte table entry; texel path: two texel pairs), issued for the lanes
        // still in the race -- and ALWAYS for lane 0 (any in-range address), so that the instruction issues whatever the
        // data: the hand-counted waits below rely on a fixed number of loads per step.
        constexpr uint32_t kLk = !OCCLUSION ? 0u : TABLE ? 1u : 2u;                  // lookup loads per step
        constexpr bool kDefer = TR_DEFER && OCCLUSION;
        constexpr bool kEarly = TR_EARLY_PREFETCH && (kDefer || !OCCLUSION);
        // In flight across a step boundary (kDefer): the raw lookup words of step s live in lk0 / lk1[s % slots] until the
        // end of step s + 1; the rest of that step's decision rides along in ordinary registers.
        uint32_t lk0[kRingSlots], lk1[kRingSlots];
#pragma unroll
        for (uint32_t k = 0; k < kRingSlots; ++k) { lk0[k] = 0u; lk1[k] = 0u; }
        cm::lmask pVis = 0ull, pAccept = 0ull;
        bool pPair = false;
        float pDepth = 0.f;
        auto issueLookup = [&](uint32_t& w0, uint32_t& w1, const void* p0, const void* p1, cm::lmask want) {
            const unsigned long long m = want | 1ull;
            unsigned long long sv;
            if (TABLE)
                asm volatile("s_mov_b64 %[sv], exec\n\ts_and_b64 exec, exec, %[m]\n\tglobal_load_ushort %[d], %[a], off" TR_LK_POLICY "\n\ts_mov_b64 exec, %[sv]"
                             : [d] "+v"(w0), [sv] "=&s"(sv) : [a] "v"(p0), [m] "s"(m) : "memory", "scc");
            else
                asm volatile("s_mov_b64 %[sv], exec\n\ts_and_b64 exec, exec, %[m]\n\tglobal_load_dword %[d0], %[a0], off\n\tglobal_load_dword %[d1], %[a1], off\n\ts_mov_b64 exec, %[sv]"
                             : [d0] "+v"(w0), [d1] "+v"(w1), [sv] "=&s"(sv) : [a0] "v"(p0), [a1] "v"(p1), [m] "s"(m) : "memory", "scc");
        };
        // The masks of the two records of a step -- :116,120 WavePrefix/ActiveCountBits: lanes 0-31 ran record 2s, lanes 32-63
        // record 2s + 1, in meshlet order -- are the two halves of the step's lane mask; lane 0 stores both (record r of the
        // batch lives at s_mask[r + 2]: pair = 2s + 2, even).  LDS: the loop issues no stores to memory.
        auto resolve = [&](uint32_t pair, cm::lmask vis, cm::lmask accept, bool pairCol, float depthSphere, uint32_t w0, uint32_t w1) {
            if (OCCLUSION) {
                if (TABLE) {
                    const float footprintMin = (float)__builtin_bit_cast(_Float16, (uint16_t)w0);
                    vis &= accept | cm::mGe(depthSphere, footprintMin);                            // :81
                } else {
                    const float d00 = cm::texelLo(w0), d10 = cm::texelLo(w1);                      // cm::occlusionResolve
                    const float d01 = pairCol ? cm::texelHi(w0) : d00, d11 = pairCol ? cm::texelHi(w1) : d10;
                    vis &= accept | cm::mGe(depthSphere, cm::min_(cm::min_(cm::min_(d00, d01), d10), d11));
                }
            }
            if (lane == 0) *reinterpret_cast<uint2*>(&s_mask[pair]) = make_uint2((uint32_t)vis, (uint32_t)(vis >> 32));
        };
        auto step = [&](auto kc, uint32_t s) {
            constexpr uint32_t kSlot = decltype(kc)::value, kPrev = (kSlot + kRingSlots - 1u) % kRingSlots;
            char* const slot = ring + kSlotBytes * kSlot;
            TR_STAMP(7);   // loop overhead / previous tail
            const uint32_t r = 2 * s + half;                                         // record within the batch
            const RecordInfo& ri = s_rec[r];
            // Loads outstanding at the top of a step, oldest first: this step's slot (2), then per other slot its 2 -- with the
            // previous step's lookup (kLk loads, kDefer) in front of the youngest slot's.
            if (kDefer) { if (kRingSlots == 3) { if (kLk == 1) TR_WAIT_VMCNT(5); else TR_WAIT_VMCNT(6); } else { if (kLk == 1) TR_WAIT_VMCNT(3); else TR_WAIT_VMCNT(4); } }
            else { if (kRingSlots == 3) TR_WAIT_VMCNT(4); else TR_WAIT_VMCNT(2); }   // this slot has landed
            const v4f sph = *reinterpret_cast<const v4f*>(slot + ringOff);
            const uint32_t cone = *reinterpret_cast<const uint32_t*>(slot + ringOffCone);
            const float4 sphere = make_float4(sph.x, sph.y, sph.z, sph.w);
            // prefetch step s + kRingSlots into this slot (past the batch: the padding entries, a harmless re-read of meshlet 0 that
            // keeps the loads unconditional)
            if (kEarly) issueMeshletLoads<true>(slot, a.stream, s_rec[r + 2 * kRingSlots].first, s_rec[r + 2 * kRingSlots].lastOff, sub16, sph, cone);
            const cm::lmask active = cm::mLeU(subEnd, ri.lastOff);                                 // :62-63 meshletIdx < numMeshlets
            cm::lmask vis = active;
            const cm::M43P W = worldOf(ri);
            const cm::F3 cw = cm::mulPointP({ sphere.x, sphere.y, sphere.z }, W);                  // :67
            const cm::F3 cv = cm::toViewP(cw, VP);                                                 // :68-69
            const float rad = sphere.w * ri.maxScale;                                              // :71
            if (FRUSTUM)
                vis &= cm::frustumVisibleM(cv, rad, a.k.m_Frustum.x, a.k.m_Frustum.y, a.k.m_Frustum.z, a.k.m_Frustum.w); // :73
            TR_STAMP(2);   // wait for data + transform + frustum
            cm::lmask accept = 0ull;
            bool pair = false;
            float depthSphere = 0.f;
            cm::StepQuot q;
            if (kDeferred) {
                // ---- deferred mode: fast values + certainty; the uncertain lanes that still matter go on the list ------------
                cm::lmask sureOcc, sureCone;
                cm::stepDeferred<CONE>(cv, rad, cone, adjugateOf(ri), a.k.m_NearPlane, a.bands, s_coneTab, q, sureOcc, sureCone);
                cm::lmask coneUnsure = 0ull;
                if (CONE) {                                                                        // :104-108
                    const cm::lmask back = cm::coneBackSure(q, cv, rad, VR, coneSlack, sureCone);
                    coneUnsure = vis & ~sureCone;                                                  // passed the frustum; the cone test is within its band
                    vis &= ~back;
                }
                cm::lmask unsure = coneUnsure;
                TR_STAMP(3);   // quotients + cone
                const cm::OccQuad oq = cm::occTailQuadFiltered(q.mn, q.mx, q.depthSphere, cv, rad, a.k.m_NearPlane, a.k.m_P00, a.k.m_P11, a.hzb,
                                                                s_mipTab, s_mipBand, a.quad.total, a.bands, sureOcc);   // :75-88 (Q4)
                if (!nearInRange) sureOcc = 0ull;
                accept = oq.accept; depthSphere = oq.depthSphere;
                unsure |= vis & ~oq.accept & ~sureOcc;                                             // its lookup matters and level / footprint are within their bands
                // The one 2-byte load of the lookup -- only for the lanes whose meshlet is still in the race and not
                // accepted at the near plane (:48-49): the tests are pure, so skipping a lookup whose result cannot
                // matter changes nothing, and a third fewer scattered requests reach the L1.
                const uint16_t* entry = reinterpret_cast<const uint16_t*>(a.quad.base) + oq.iq;
#ifdef TR_NO_LOOKUP      /* experiment, results WRONG: what the kernel would cost if the lookups were free */
                issueLookup(lk0[kSlot], lk1[kSlot], a.quad.base, a.quad.base, 0ull);
                asm volatile("" :: "v"(entry));
#else
                issueLookup(lk0[kSlot], lk1[kSlot], entry, entry, vis & ~oq.accept);
#endif
                if (__builtin_expect(unsure != 0ull, 0)) {
                    const uint32_t n = (uint32_t)__builtin_popcountll(unsure);
                    if (stCount + n > kDefStage) {
                        batchExact = true;                                                         // (dozens of uncertain lanes in one batch: hostile data, or meshlets at the camera)
                    } else {
                        const uint32_t at = stCount + __builtin_amdgcn_mbcnt_hi((uint32_t)(unsure >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)unsure, 0u));
                        if ((unsure >> lane) & 1ull) {
                            uint32_t* d = s_def + at * kDefWords;
                            d[0] = (s_gIdx[r] << 5) | sub;
                            d[1] = CONE && ((coneUnsure >> lane) & 1ull) ? kDefFromRecord : __float_as_uint(cv.x);
                            d[2] = __float_as_uint(cv.y); d[3] = __float_as_uint(cv.z); d[4] = __float_as_uint(rad);
                        }
                        stCount += n;
                    }
                }
            } else {
            // every square root and division of the step (:56-62, :79, normalize :103): fast when the whole wave can
            if (OCCLUSION || CONE)
                cm::stepQuotients<OCCLUSION, CONE, true>(active, cv, rad, cone, adjugateOf(ri), a.k.m_NearPlane, nearInRange, q, s_coneTab);
            if (CONE)                                                                              // :104-108
                vis &= ~cm::coneBack(q, cv, rad, VR, coneSlack, vis, cone, adjugateOf(ri));
            TR_STAMP(3);   // quotients + cone
            }
            if (OCCLUSION && !TABLE) {
                const cm::OccSample os = cm::occTailTexel(q, cv, rad, a.k.m_NearPlane, a.k.m_P00, a.k.m_P11, a.hzb, s_quadOff);
                accept = cm::mLt(cv.z - a.k.m_NearPlane, rad);                                     // == os.accept (:48-49)
                depthSphere = os.depthSphere; pair = os.pair;
                issueLookup(lk0[kSlot], lk1[kSlot], a.hzb.base + os.i0, a.hzb.base + os.i1, vis & ~accept);   // two texel pairs (cm::loadTexelPair)
            }
            // (late prefetch: the slot's LDS reads above have returned, their values were used)
            if (!kEarly) issueMeshletLoads(slot, a.stream, s_rec[r + 2 * kRingSlots].first, s_rec[r + 2 * kRingSlots].lastOff, sub16);
            TR_STAMP(4);   // lookup + prefetch issue
            if (!OCCLUSION) {
                resolve(2u * s + 2u, vis, 0ull, false, 0.f, 0u, 0u);
            } else if (!kDefer) {
                // Loads are counted in order: "at most 2 outstanding" = everything before this step's prefetch has landed.
                asm volatile("s_waitcnt vmcnt(2)" : "+v"(lk0[kSlot]), "+v"(lk1[kSlot]) :: "memory");
                TR_STAMP(5);   // lookup wait
                resolve(2u * s + 2u, vis, accept, pair, depthSphere, lk0[kSlot], lk1[kSlot]);
            } else {
                // the PREVIOUS step's lookup: younger than it are this step's prefetch (2) and lookup (kLk) -- and, when the
                // prefetch is issued behind the arithmetic (!kEarly), its own step's prefetch (2).  (Step 0 with kEarly: the
                // primed ring's youngest slot sits behind the dummy lookup too; waiting for it here is harmless.)
                if (kEarly) {
                    if (kLk == 1) asm volatile("s_waitcnt vmcnt(3)" : "+v"(lk0[kPrev]), "+v"(lk1[kPrev]) :: "memory");
                    else asm volatile("s_waitcnt vmcnt(4)" : "+v"(lk0[kPrev]), "+v"(lk1[kPrev]) :: "memory");
                } else if (kLk == 1) asm volatile("s_waitcnt vmcnt(5)" : "+v"(lk0[kPrev]), "+v"(lk1[kPrev]) :: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" : "+v"(lk0[kPrev]), "+v"(lk1[kPrev]) :: "memory");
                TR_STAMP(5);   // lookup wait
                resolve(2u * s, pVis, pAccept, pPair, pDepth, lk0[kPrev], lk1[kPrev]);            // the records of step s - 1 (s = 0: the dummy pair)
                pVis = vis; pAccept = accept; pPair = pair; pDepth = depthSphere;
            }
            TR_STAMP(6);   // resolve + ballot + mask store
        };
        // Prime the ring.  kDefer: a lookup "of the step before the first" (lane 0, entry 0) goes where a step's lookup sits in
        // the load order -- in front of the youngest slot's loads -- so that every step sees the same sequence; its resolve
        // lands in s_mask[0], [1].
        // (primed: the two slots were requested at the end of the previous batch and have landed -- the prologue's wait covered them;
        // the first step's deferred resolve then reads the zeros lk0 / lk1 were set to: it lands in s_mask[0], [1] all the same)
        if (!primed) {
#pragma unroll
            for (uint32_t k = 0; k + 1 < kRingSlots; ++k)
                issueMeshletLoads(ring + kSlotBytes * k, a.stream, s_rec[2 * (s0 + k) + half].first, s_rec[2 * (s0 + k) + half].lastOff, sub16);
            if (kDefer) {
                const void* p = TABLE ? (const void*)a.quad.base : (const void*)a.hzb.base;
                issueLookup(lk0[kRingSlots - 1u], lk1[kRingSlots - 1u], p, p, 0ull);
            }
            issueMeshletLoads(ring + kSlotBytes * (kRingSlots - 1u), a.stream, s_rec[2 * (s0 + kRingSlots - 1u) + half].first, s_rec[2 * (s0 + kRingSlots - 1u) + half].lastOff, sub16);
        }
#pragma unroll 1
        for (uint32_t s = s0; s < nSteps; s += kRingSlots) {
            step(std::integral_constant<uint32_t, 0>{}, s);
            step(std::integral_constant<uint32_t, 1>{}, s + 1);
            if (kRingSlots == 3) step(std::integral_constant<uint32_t, 2>{}, s + 2);
        }
        if (kDefer) {                                    // the last step's lookup: the youngest load when the prefetch goes first (kEarly), else in front of the last prefetch
            if (kEarly) asm volatile("s_waitcnt vmcnt(0)" : "+v"(lk0[kRingSlots - 1u]), "+v"(lk1[kRingSlots - 1u]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" : "+v"(lk0[kRingSlots - 1u]), "+v"(lk1[kRingSlots - 1u]) :: "memory");
            resolve(2u * nSteps, pVis, pAccept, pPair, pDepth, lk0[kRingSlots - 1u], lk1[kRingSlots - 1u]);
        }
        // (not pipelined: the last, padding prefetches must have landed before the next batch primes the ring.  Pipelined: the
        // next batch's requests go out BEHIND them, below, and land behind them)
        if (!kPipe) TR_WAIT_VMCNT(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (kDeferred && __builtin_expect(batchExact, 0)) {
            // ---- the list overflowed inside this batch (dozens of uncertain lanes in 32 records: hostile data): the whole
            //      batch is redone exactly while its masks are still in LDS, and its list entries are dropped ---------------
            for (uint32_t s = s0; s < nSteps; ++s) {
                const uint32_t r = 2 * s + half;
                const uint32_t g = s_gIdx[r];
                const unsigned long long ballot = __ballot(g < G && exactMeshlet(g, sub));
                if (sub == 0) s_mask[r + 2u] = half ? (uint32_t)(ballot >> 32) : (uint32_t)ballot;
            }
            stCount = stAtBatch;
            batchExact = false;
#ifdef TR_COUNT_PATHS
            if (lane == 0) atomicAdd(&g_stampSums[2], 1ull);                            // batches redone exactly
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (kDeferred && __builtin_expect(stAtBatch >= kDefPassAt, 0)) {                 // a pass over the entries of earlier batches
            const uint32_t n = stAtBatch < 64u ? stAtBatch : 64u;
            runPass(n);
            stAtBatch -= n;
        }
        // ---- the batch's 64 masks leave in one store (lane l: record l of the batch) ----------------------------
        {
            const uint32_t g = lane < kCullBatch ? s_gIdx[lane] : 0xFFFFFFFFu;
            if (g < G) a.visMask[g] = s_mask[(lane < kCullBatch ? lane : 0u) + 2u];     // (g is none outside this team's piece of the window)
        }
        // ---- the next batch's requests (see PIPELINED BATCHES): its instance blocks, and the ring's first two slots ----------
        primed = false;
        const uint32_t nStepsN = stepsOf(sbN, s0N, s1N);
        if (nStepsN) {
            const float4* wr = blockAddr(entry);                                     // (entry: the next batch's, available since this batch's prologue)
            q0 = wr[0]; q1 = wr[1]; q2 = wr[2]; q3 = wr[3];
            if (kPipe && usePerm) {
                // {first meshlet, lastOff} of the next batch's record `lane`, exactly as its prologue will resolve them
                const bool has = entry.x < G && (lane >> 1) >= s0N && (lane >> 1) < nStepsN;
                uint32_t cnt = entry.w < 32u ? entry.w : 32u;
                if (!has || (uint64_t)entry.z + cnt > a.numMeshlets) cnt = 0u;
                const uint32_t pf = cnt ? entry.z : 0u, pl = lastOffOf(cnt);
                // lane (half, sub) of step s0N + k stages record 2 (s0N + k) + half
#pragma unroll
                for (uint32_t k = 0; k < kRingSlots; ++k) {
                    const int from = (int)((2u * (s0N + k) + half) * 4u);
                    const uint32_t f = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)pf), l = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)pl);
                    issueMeshletLoads(ring + kSlotBytes * k, a.stream, f, l, sub16);
                }
                primed = true;
            }
        } else if (kPipe) {
            TR_WAIT_VMCNT(0);                                                        // the wave's last batch: nothing of the ring is in flight when it ends
        }
    }
    if (kDeferred && stCount) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // the last batches' mask stores have completed
        while (stCount) runPass(stCount < 64u ? stCount : 64u);
    }
#ifdef TR_STAMPS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_stampSums[i], stampSum[i]);
#endif
}

// Sum over the 64 lanes of a wave, returned in every lane: four DPP row shifts (a row = 16 lanes), two row broadcasts (gfx9
// DPP), one readlane -- the __shfl_xor butterfly costs six trips through the LDS crossbar (ds_bpermute) per value.
__device__ __forceinline__ uint32_t waveSum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);     // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);     // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);     // row_shr:8: lane 15 of a row = the row's sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);    // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);    // row_bcast:31 into rows 2 and 3: lane 63 = the sum
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Visible meshlets per batch of 64 consecutive records (canonical order): one wave per batch.  Two levels of sums -- per
// batch and per "super" of 256 batches, a workgroup per super so that both are plain stores -- and a scan over the supers
// only (visSuperScanKernel: 107 values on C3): the expansion finds a batch's list offset from the super's prefix + the
// batches before it in its super (at most four loads per lane).  The single-workgroup scan over ALL batches was 20 us on
// C3 (27 k batches), during which the expansion it delayed ran into the late meshlet cull.  (Per-super sums by
// device-scope atomics instead: 27 k atomics on 107 addresses took 100 us and stalled the HZB build next to them.)
#ifndef TR_SUPER_SHIFT
#define TR_SUPER_SHIFT 8
#endif
constexpr uint32_t kSuperShift = TR_SUPER_SHIFT;
constexpr uint32_t kSuperBatches = 1u << kSuperShift;
constexpr uint32_t kCountThreads = 1024;
constexpr uint32_t kCountWaves = kCountThreads / 64;
__global__ __launch_bounds__(kCountThreads) void visCountKernel(MeshletCullArgs a)
{
    __shared__ uint32_t s_part[kCountWaves];
    const uint32_t G = listGroupCount(a);
    const uint32_t numBatches = (G + kBatch - 1) / kBatch;
    const uint32_t numSupers = (numBatches + kSuperBatches - 1) >> kSuperShift;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t kPerWave = kSuperBatches / kCountWaves;                       // 16 batches
    for (uint32_t sb = blockIdx.x; sb < numSupers; sb += gridDim.x) {
        const uint32_t b0 = (sb << kSuperShift) + wave * kPerWave;
        uint32_t pc[kPerWave];
#pragma unroll
        for (uint32_t i = 0; i < kPerWave; ++i) {
            const uint32_t g = (b0 + i) * kBatch + lane;
            pc[i] = g < G ? (uint32_t)__popc(a.visMask[g]) : 0u;
        }
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t i = 0; i < kPerWave; ++i) {
            pc[i] = waveSum(pc[i]);
            if (lane == 0 && b0 + i < numBatches) a.batchSum[b0 + i] = pc[i];
            mine += pc[i];
        }
        __syncthreads();                                                             // s_part consumed by the previous round
        if (lane == 0) s_part[wave] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
#pragma unroll
            for (uint32_t w = 0; w < kCountWaves; ++w) total += s_part[w];
            a.superSum[sb] = total;
        }
    }
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

// One workgroup: exclusive scan of the super sums in place; the total replaces DispatchMesh(numVisible,1,1) summed over
// groups (basepass.hlsl:120-121).
__global__ __launch_bounds__(1024) void visSuperScanKernel(MeshletCullArgs a)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t G = listGroupCount(a);
    const uint32_t numBatches = (G + kBatch - 1) / kBatch;
    const uint32_t numSupers = (numBatches + kSuperBatches - 1) >> kSuperShift;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < numSupers; base += 1024u) {
        const uint32_t i = base + tid;
        const uint32_t v = i < numSupers ? a.superSum[i] : 0u;
        const uint32_t inc = waveInclusiveScan(v, lane);
        __syncthreads();
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t pre = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) { if (w < wave) pre += s_wave[w]; all += s_wave[w]; }
        if (i < numSupers) a.superSum[i] = carry + pre + inc - v;
        carry += all;
    }
    if (tid == 0) { a.drawArgs[0] = carry; a.drawArgs[1] = 1; a.drawArgs[2] = 1; }
}

// Ordered compaction, one wave per batch: lane l holds the mask of record l, a wave scan gives
// the record offsets, then one thread per (group, lane) slot: a visible slot's position is the
// batch offset + popcounts of the earlier groups of its batch + WavePrefixCountBits in its group.
__global__ __launch_bounds__(kBlock) void visExpandKernel(MeshletCullArgs a)
{
    __shared__ uint2 s_mo[kWaves][kBatch];                    // per record of the wave's batch: {mask, list offset}
    const uint32_t G = listGroupCount(a);
    const uint32_t numBatches = (G + kBatch - 1) / kBatch;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t half = lane >> 5, sub = lane & 31u;
    const uint32_t bit = 1u << sub, below = bit - 1u;
    uint2* mo = s_mo[wave];
    for (uint32_t batch = blockIdx.x * kWaves + wave; batch < numBatches; batch += gridDim.x * kWaves) {
        const uint32_t g0 = batch * kBatch;
        const uint32_t mask = g0 + lane < G ? a.visMask[g0 + lane] : 0u;
        // list offset of the batch: its super's prefix + the batches before it in its super
        const uint32_t sb = batch >> kSuperShift;
        uint32_t before = 0;
        for (uint32_t j = (sb << kSuperShift) + lane; j < batch; j += 64u) before += a.batchSum[j];
        before = waveSum(before) + a.superSum[sb];
        const uint32_t pc = (uint32_t)__popc(mask);
        const uint32_t exc = waveInclusiveScan(pc, lane) - pc + before;
        mo[lane] = make_uint2(mask, exc);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // lane (half, sub) looks at bit `sub` of records half, 2 + half, ...: the record's {mask, offset} is one broadcast
        // LDS read per half-wave, the position one v_bcnt (popcount + add)
        uint32_t value = ((g0 + half) << 5) | sub;
#pragma unroll 8
        for (uint32_t s = 0; s < kSteps; ++s) {
            const uint2 e = mo[2 * s + half];
            if (e.x & bit) {
                const uint32_t pos = e.y + (uint32_t)__popc(e.x & below);
                if (pos < a.listCapacity) a.visibleList[pos] = value;
            }
            value += 2u << 5;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                       // the slice is rewritten by the next batch
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---------------------------------------------------------------------------------------------
// Single-launch ordered compaction for SMALL passes (record capacity < 2^19: one rank's share of a sharded scene, real
// assets): count, scan and expand of the three kernels above in one, so that the list costs one link of the frame's
// chain of dependent launches instead of three (a link is ~6 us whatever it does; DESIGN.md "Launch chain").
// A workgroup takes tiles of kCompactTile consecutive groups in TICKET order; a tile publishes its count as soon as
// it has it and then sums the counts of ALL its predecessors (at most 255: one thread each, one round trip).  A tile
// only ever waits for tiles with smaller tickets, which are held by workgroups already running: no deadlock whatever
// the grid size.  A status word carries flag and count in ONE 64-bit relaxed agent-scope atomic, so no fence is
// needed (the data a tile writes is consumed by later launches only).  The spin is bounded: on expiry the tile poisons its prefix (bit 48),
// the poison travels through the sums to the last tile, which then writes 0xFFFFFFFF draw arguments (never seen in
// practice; a wrong list fails parity, a hang would take the GPU down).
constexpr uint32_t kCompactTile = 2048;                 // groups per tile
constexpr uint32_t kCompactThreads = 1024;              // 2 groups per thread; 16 waves keep the expansion's LDS/store latency covered
constexpr uint32_t kCompactWaves = kCompactThreads / 64;
constexpr uint32_t kCompactMaxTiles = (1u << 19) / kCompactTile;
constexpr uint32_t kStatusStride = 16;                  // one status word per 128-byte line: 255 waiters per word, all tiles at once
constexpr uint64_t kFlagA = 1ull << 62, kFlagMask = 3ull << 62, kPoison = 1ull << 48;

__global__ __launch_bounds__(kCompactThreads) void visCompactKernel(MeshletCullArgs a, unsigned long long* status, uint32_t* ticket)
{
    __shared__ uint32_t s_mask[kCompactTile];
    __shared__ uint32_t s_off[kCompactTile];
    __shared__ uint32_t s_waveTot[kCompactWaves];
    __shared__ uint32_t s_tile;
    __shared__ unsigned long long s_pre[kCompactMaxTiles / 64u];
    const uint32_t G = listGroupCount(a);
    const uint32_t numTiles = (G + kCompactTile - 1) / kCompactTile;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (;;) {
        if (tid == 0) s_tile = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= numTiles) {
            if (tile == 0 && tid == 0) { a.drawArgs[0] = 0; a.drawArgs[1] = 1; a.drawArgs[2] = 1; }    // empty pass
            return;
        }
        // ---- masks of the tile -> LDS; thread t owns groups 2t, 2t+1 of the tile -----------------------------------
        const uint32_t g0 = tile * kCompactTile;
        const uint32_t b = g0 + tid * 2u;
        uint32_t m0 = 0, m1 = 0;
        if (b + 2u <= G) { const uint2 v = *reinterpret_cast<const uint2*>(a.visMask + b); m0 = v.x; m1 = v.y; }
        else if (b < G) m0 = a.visMask[b];
        const uint32_t c0 = (uint32_t)__popc(m0), mine = c0 + (uint32_t)__popc(m1);
        const uint32_t inc = waveInclusiveScan(mine, lane);
        if (lane == 63) s_waveTot[wave] = inc;
        __syncthreads();
        uint32_t pre = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kCompactWaves; ++w) { const uint32_t x = s_waveTot[w]; if (w < wave) pre += x; total += x; }
        const uint32_t exc = pre + inc - mine;
        *reinterpret_cast<uint2*>(&s_mask[tid * 2u]) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(&s_off[tid * 2u]) = make_uint2(exc, exc + c0);
        // ---- prefix of the tile: the counts of ALL its predecessors, read in one parallel round trip (at most
        // kCompactMaxTiles - 1 = 255 of them: thread j waits for tile j's count) -------------------------------------
        if (tid == 0) __hip_atomic_store(&status[tile * kStatusStride], kFlagA | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave < kCompactMaxTiles / 64u) {
            unsigned long long v = 0;
            if (tid < tile) {
                uint32_t spins = 0;
                for (;;) {
                    v = __hip_atomic_load(&status[tid * kStatusStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v & kFlagMask) { v &= ~kFlagMask; break; }
                    if (++spins > (1u << 22)) { v = kPoison; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
            if (lane == 0) s_pre[wave] = v;
        }
        __syncthreads();
        unsigned long long base = 0;
#pragma unroll
        for (uint32_t w = 0; w < kCompactMaxTiles / 64u; ++w) base += s_pre[w];
        if (tid == 0 && tile == numTiles - 1) {                                      // the last tile knows the total
            const unsigned long long all = base + total;
            const bool bad = (all >> 40) != 0ull;                                     // poisoned by a timed-out wait
            a.drawArgs[0] = bad ? 0xFFFFFFFFu : (uint32_t)all;                        // basepass.hlsl:120-121 summed over groups
            a.drawArgs[1] = bad ? 0xFFFFFFFFu : 1u;
            a.drawArgs[2] = bad ? 0xFFFFFFFFu : 1u;
        }
        // ---- expansion: one thread per (group, lane) slot, canonical order ------------------------------------------
        const uint32_t groupsHere = G - g0 < kCompactTile ? G - g0 : kCompactTile;
#pragma unroll 4
        for (uint32_t slot = tid; slot < groupsHere * 32u; slot += kCompactThreads) {
            const uint32_t g = slot >> 5, sub = slot & 31u;
            const uint32_t mask = s_mask[g];
            if (mask & (1u << sub)) {
                const unsigned long long pos = base + s_off[g] + (uint32_t)__popc(mask & ((1u << sub) - 1u));
                if (pos < a.listCapacity) a.visibleList[pos] = ((g0 + g) << 5) | sub;
            }
        }
        __syncthreads();                                                             // LDS is reused by the next tile
    }
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU exchange (not in the reference: single GPU, GraphicRHI.cpp:165).
// A rank ships its pass slots in COMPACT form inside one fixed-capacity "shard slot", so that the frame needs ONE
// equal-size all-gather and no host read-back:
//   * the amplification records as RUNS.  The instance pass emits, per submitted instance, consecutive records
//     {instance, lod, 0}, {instance, lod, 32}, {instance, lod, 64}, ... (gpuculling.hlsl:139-157), so a maximal run of
//     records that continue each other (same instance, same lod, group offset + 32) is described by 16 bytes --
//     {instance, lod, first group offset, index of its first record in the pass slot} -- whatever its length: one entry
//     per submitted INSTANCE instead of 12 bytes per GROUP.  The encoding is lossless for any record array (a record
//     that continues nothing starts a run of its own);
//   * the lane mask of every group (4 bytes: 1 bit per tested meshlet instead of 4 bytes per visible meshlet).
// Every rank then rebuilds the whole-scene records rank-major straight from the received run entries (device-side
// offsets from the slot headers), concatenates the masks and rebuilds the ordered visible list with the same
// count/scan/expand kernels the single-GPU path uses: the result is bit-identical to a single-GPU frame.
//
// Shard slot (u32 words), S = slotGroups, R = slotRuns:
//   [0..15]         header: {G_s, X_s} for pass slot s = 0..3 at words 2s, 2s+1 (groups sent; groups the dispatch counter
//                   counted, dropped ones included: Q2 made global, gather.py); word 8 = overflow flag (groups > S or
//                   runs > R), word 9 = the rank dropped groups at its record capacity (Q2), words 10..13 = runs of
//                   pass slots 0..s (cumulative end of pass slot s in the run array)
//   [16, 16+4R)     run entries of pass slot 0, then 1, ... back to back (4 words each)
//   [16+4R, +S)     lane masks of pass slot 0, then 1, ... back to back
constexpr uint32_t kMaxPassSlots = 4;
constexpr uint32_t kSlotHeaderWords = 16;
constexpr uint32_t kMaxRanks = 64;
constexpr uint32_t kPackThreads = 256;
constexpr uint32_t kPackRounds = 4;                                   // records per thread and tile: all loads of a tile in flight at once
constexpr uint32_t kPackTile = kPackThreads * kPackRounds;            // 1024 records
constexpr unsigned long long kPackFlag = 1ull << 63, kPackPoison = 1ull << 40;

struct ShardPackArgs
{
    const uint32_t* records[kMaxPassSlots];
    const uint32_t* masks[kMaxPassSlots];
    const uint32_t* dispatchArgs[kMaxPassSlots];
    const uint32_t* drawArgs[kMaxPassSlots];
    uint32_t argsWords[kMaxPassSlots];
    uint32_t recordCapacity[kMaxPassSlots];
    uint32_t* slot;
    uint32_t slotGroups, slotRuns;
    unsigned long long* status;            // status word of tile t; all zero when the launch starts
    unsigned long long* statusNext;        // the other half of the state buffer: zeroed by this launch for the next one
    uint32_t maxTiles;
    uint32_t tileBlocks;                   // workgroups [0, tileBlocks) compact the runs, the rest copy the masks
};

// The run starts of all pass slots are compacted in order by tiles of 1024 records: a tile publishes its number of run
// starts and sums those of ALL its predecessors.  Tile t belongs to workgroup t mod tileBlocks, and the host keeps the
// grid within what the chip holds at once (<= 4 workgroups of 256 threads per CU), so every tile a workgroup waits for
// is held by a workgroup that is running or has finished -- no tickets: a same-address atomic per workgroup is what
// the first version of this kernel spent most of its 20 us on.  The other workgroups copy the masks meanwhile.
// The status words live in one half of the state buffer; every launch zeroes the OTHER half, which the next launch on
// that buffer uses (the host alternates per launch, recordPackShard), so the recorded command is re-executed frame
// after frame without a clear launch in front of it and without a last-one-out counter.
__global__ __launch_bounds__(kPackThreads) void shardPackKernel(ShardPackArgs a)
{
    __shared__ uint32_t s_cnt[kPackRounds * (kPackThreads / 64)];
    __shared__ uint32_t s_total;
    __shared__ unsigned long long s_pre[kPackThreads / 64][kMaxPassSlots + 1];
    uint32_t G[kMaxPassSlots], start[kMaxPassSlots + 1], tileStart[kMaxPassSlots + 1];
    uint32_t overflow = 0, dropped = 0;
    start[0] = 0; tileStart[0] = 0;
#pragma unroll
    for (uint32_t s = 0; s < kMaxPassSlots; ++s) {
        uint32_t g = 0;
        if (a.dispatchArgs[s]) {                                                    // same rule as groupCount()
            g = a.dispatchArgs[s][0];
            if (a.argsWords[s] > 3 && a.dispatchArgs[s][3] < g) g = a.dispatchArgs[s][3];
            g = g < a.recordCapacity[s] ? g : a.recordCapacity[s];
            if (g != a.dispatchArgs[s][0]) dropped = 1;                             // the counter counts dropped groups too (Q2)
        }
        if (start[s] + g > a.slotGroups) { g = a.slotGroups - start[s]; overflow = 1; }
        G[s] = g;
        start[s + 1] = start[s] + g;
        tileStart[s + 1] = tileStart[s] + (g + kPackTile - 1u) / kPackTile;
    }
    const uint32_t total = start[kMaxPassSlots], numTiles = tileStart[kMaxPassSlots];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t* runOut = a.slot + kSlotHeaderWords;
    uint32_t* maskOut = runOut + 4ull * a.slotRuns;
    unsigned long long* status = a.status;
    if (blockIdx.x < a.tileBlocks)
        for (uint32_t j = blockIdx.x * kPackThreads + tid; j < a.maxTiles; j += a.tileBlocks * kPackThreads) a.statusNext[j] = 0ull;
    const uint32_t tilesToDo = numTiles ? numTiles : 1u;                              // an empty shard still writes its header
    for (uint32_t tile = blockIdx.x; blockIdx.x < a.tileBlocks && tile < tilesToDo; tile += a.tileBlocks) {
        __syncthreads();
        uint32_t s = 0;
#pragma unroll
        for (uint32_t q = 1; q < kMaxPassSlots; ++q) s += tile >= tileStart[q] && numTiles ? 1u : 0u;
        const uint32_t Gs = numTiles ? G[s] : 0u;
        const uint32_t* rec = a.records[s];
        const uint32_t g0 = (tile - tileStart[s]) * kPackTile;
        // ---- which records of the tile start a run (round i looks at records g0 + 256 i + tid: coalesced) -------------
        uint32_t bits = 0;
        uint32_t cur[kPackRounds][3], prev[kPackRounds][3];
#pragma unroll
        for (uint32_t i = 0; i < kPackRounds; ++i) {
            const uint32_t g = g0 + i * kPackThreads + tid;
#pragma unroll
            for (uint32_t w = 0; w < 3; ++w) {
                cur[i][w] = g < Gs ? rec[3ull * g + w] : 0u;
                prev[i][w] = g < Gs && g > 0u ? rec[3ull * g + w - 3u] : 0u;
            }
        }
#pragma unroll
        for (uint32_t i = 0; i < kPackRounds; ++i) {
            const uint32_t g = g0 + i * kPackThreads + tid;
            const bool st = g < Gs && (g == 0u || !(cur[i][0] == prev[i][0] && cur[i][1] == prev[i][1] && cur[i][2] == prev[i][2] + 32u));
            const unsigned long long b = __ballot(st);
            if (lane == 0) s_cnt[i * (kPackThreads / 64) + wave] = (uint32_t)__popcll(b);
            bits |= st ? 1u << i : 0u;
        }
        __syncthreads();
        if (wave == 0) {
            constexpr uint32_t kCounters = kPackRounds * (kPackThreads / 64);
            const uint32_t c = lane < kCounters ? s_cnt[lane] : 0u;
            const uint32_t inc = waveInclusiveScan(c, lane);
            if (lane < kCounters) s_cnt[lane] = inc - c;
            if (lane == 63) s_total = inc;
        }
        __syncthreads();
        const uint32_t tileRuns = s_total;
        if (tid == 0 && numTiles) __hip_atomic_store(&status[tile], kPackFlag | (unsigned long long)tileRuns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- run starts before this tile; the LAST tile also needs them per pass slot (header words 10..13) ------------
        const bool last = numTiles == 0u || tile == numTiles - 1u;
        unsigned long long sum = 0, part[kMaxPassSlots] = {};
        for (uint32_t j = tid; j < tile; j += kPackThreads) {
            unsigned long long v = 0;
            uint32_t spins = 0;
            for (;;) {
                v = __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v & kPackFlag) { v &= ~kPackFlag; break; }
                if (++spins > (1u << 22)) { v = kPackPoison; break; }               // never seen; a hang would take the GPU down
                __builtin_amdgcn_s_sleep(1);                                        // leave the issue slots to the tiles being waited for
            }
            sum += v;
            if (last) {
#pragma unroll
                for (uint32_t q = 0; q < kMaxPassSlots; ++q) part[q] += j < tileStart[q + 1] ? v : 0ull;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        if (last) {
#pragma unroll
            for (uint32_t q = 0; q < kMaxPassSlots; ++q) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) part[q] += __shfl_xor(part[q], d);
            }
        }
        if (lane == 0) {
            s_pre[wave][0] = sum;
            if (last) {
#pragma unroll
                for (uint32_t q = 0; q < kMaxPassSlots; ++q) s_pre[wave][q + 1] = part[q];
            }
        }
        __syncthreads();
        unsigned long long prefix = 0;
#pragma unroll
        for (uint32_t w = 0; w < kPackThreads / 64; ++w) prefix += s_pre[w][0];
        const bool poisoned = (prefix >> 40) != 0ull;
        // ---- the tile's run entries ----------------------------------------------------------------------------------
#pragma unroll
        for (uint32_t i = 0; i < kPackRounds; ++i) {
            const bool st = (bits >> i) & 1u;
            const unsigned long long b = __ballot(st);
            if (st) {
                const unsigned long long idx = prefix + s_cnt[i * (kPackThreads / 64) + wave] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
                if (idx < a.slotRuns && !poisoned)
                    *reinterpret_cast<uint4*>(runOut + 4ull * idx) = make_uint4(cur[i][0], cur[i][1], cur[i][2], g0 + i * kPackThreads + tid);
            }
        }
        // ---- header (the last tile knows every count) -------------------------------------------------------------------
        if (last && tid < kSlotHeaderWords) {
            unsigned long long end[kMaxPassSlots];
#pragma unroll
            for (uint32_t q = 0; q < kMaxPassSlots; ++q) {
                end[q] = 0;
#pragma unroll
                for (uint32_t w = 0; w < kPackThreads / 64; ++w) end[q] += s_pre[w][q + 1];
                if (numTiles && tile < tileStart[q + 1]) end[q] += tileRuns;          // this tile belongs to pass slot <= q
            }
            const unsigned long long allRuns = prefix + tileRuns;
            uint32_t v = 0;
#pragma unroll
            for (uint32_t q = 0; q < kMaxPassSlots; ++q) {
                if (tid == 2 * q) v = G[q];
                if (tid == 2 * q + 1) v = a.dispatchArgs[q] ? a.dispatchArgs[q][0] : 0u;   // the counter: dropped groups included (Q2)
                if (tid == 10 + q) v = poisoned ? 0xFFFFFFFFu : (uint32_t)end[q];
            }
            if (tid == 8) v = (overflow || allRuns > a.slotRuns || poisoned) ? 1u : 0u;
            if (tid == 9) v = dropped;
            a.slot[tid] = v;
        }
    }
    for (uint32_t g = (blockIdx.x - a.tileBlocks) * kPackThreads + tid; blockIdx.x >= a.tileBlocks && g < total; g += (gridDim.x - a.tileBlocks) * kPackThreads) {   // masks
        uint32_t s = 0;
#pragma unroll
        for (uint32_t q = 1; q < kMaxPassSlots; ++q) s += g >= start[q] ? 1u : 0u;
        maskOut[g] = a.masks[s][g - start[s]];
    }
}

struct ShardUnpackArgs
{
    const uint32_t* recv;                  // world x (16 + 4 * slotRuns + slotGroups) words
    uint32_t world, slotGroups, slotRuns;
    uint32_t* records[kMaxPassSlots];      // whole-scene outputs per pass slot (nullptr = not gathered)
    uint32_t* masks[kMaxPassSlots];
    uint32_t* args[kMaxPassSlots];         // 8 words: {G,1,1,G} dispatch args, {V,1,1} draw args, status
    uint32_t capacity[kMaxPassSlots];      // groups
    uint32_t globalCap;                    // > 0: Q2 made global -- the single-GPU run's group capacity (gather.py)
};

// Rank-major rebuild.  Segment k = (rank p, pass slot s); every block derives the segment table from the slot headers
// (world <= 64): where the segment's masks and run entries lie in the received slot, where its groups go in the
// whole-scene arrays.  The grid then copies the mask words of all segments and expands the run entries of all segments
// (one thread per run; a run longer than 8 records is written by the whole wave).
__global__ __launch_bounds__(256) void shardUnpackKernel(ShardUnpackArgs a)
{
    constexpr uint32_t kSeg = kMaxPassSlots * kMaxRanks;
    __shared__ uint32_t s_G[kMaxRanks][kMaxPassSlots];
    __shared__ uint32_t s_E[kMaxRanks][kMaxPassSlots];
    __shared__ uint32_t s_maskStart[kSeg + 1], s_runStart[kSeg + 1];
    __shared__ uint32_t s_maskSrc[kSeg], s_runSrc[kSeg], s_dst[kSeg], s_len[kSeg];
    const uint32_t slotWords = kSlotHeaderWords + 4u * a.slotRuns + a.slotGroups;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid < a.world * kMaxPassSlots) {
        const uint32_t p = tid / kMaxPassSlots, s = tid % kMaxPassSlots;
        s_G[p][s] = a.records[s] ? a.recv[(uint64_t)p * slotWords + 2u * s] : 0u;
        s_E[p][s] = a.recv[(uint64_t)p * slotWords + 10u + s];
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t status = 0, maskRun = 0, runRun = 0, k = 0;
        uint32_t off[kMaxPassSlots] = {};
        unsigned long long counted[kMaxPassSlots] = {};                            // Q2 made global: groups counted by the lower ranks
        bool cutDone[kMaxPassSlots] = {};
        for (uint32_t p = 0; p < a.world; ++p) {
            if (a.recv[(uint64_t)p * slotWords + 8u]) status |= 1u;                // that rank's slot overflowed
            if (a.recv[(uint64_t)p * slotWords + 9u] && !a.globalCap) status |= 8u; // that rank dropped groups (Q2) and nobody resolves it
            uint32_t inSlot = 0, runBegin = 0;
            bool bad = false;
            for (uint32_t s = 0; s < kMaxPassSlots; ++s) {
                // a pass slot this rank did not gather still occupies its place in the sender's slot
                const uint32_t sentG = a.recv[(uint64_t)p * slotWords + 2u * s];
                uint32_t g = s_G[p][s];
                const uint32_t srcG = inSlot;
                inSlot += sentG;
                uint32_t runEnd = s_E[p][s];
                if (inSlot > a.slotGroups || inSlot < sentG || runEnd < runBegin || runEnd > a.slotRuns) bad = true;   // corrupt header: copy nothing
                if (bad) { status |= 4u; g = 0; runEnd = runBegin; }
                if (a.globalCap && a.records[s] && !bad) {
                    // the first instance the single-GPU pass drops lies on the first rank whose counted groups reach the
                    // capacity: the run containing local record cap - before - 1 (gather.py); behind it nothing counts
                    const unsigned long long x = a.recv[(uint64_t)p * slotWords + 2u * s + 1u];
                    if (cutDone[s]) g = 0;
                    else if (counted[s] + x >= a.globalCap) {
                        cutDone[s] = true;
                        const unsigned long long j = a.globalCap - counted[s] - 1ull;
                        if (j < g) {
                            const uint32_t* run = a.recv + (uint64_t)p * slotWords + kSlotHeaderWords + 4ull * runBegin;
                            uint32_t lo = 0, hi = runEnd - runBegin;                    // last run with first record <= j (run 0 starts at 0)
                            while (hi - lo > 1u) {
                                const uint32_t mid = (lo + hi) >> 1;
                                if (run[4ull * mid + 3u] <= j) lo = mid; else hi = mid;
                            }
                            g = hi > lo ? run[4ull * lo + 3u] : 0u;
                        }
                    }
                    counted[s] += x;
                }
                uint32_t runs = a.records[s] ? runEnd - runBegin : 0u;
                if (off[s] + g > a.capacity[s]) { g = a.capacity[s] - off[s]; status |= 2u; }
                const uint32_t base = p * slotWords + kSlotHeaderWords;            // < 2^32 words (checked on the host)
                s_maskStart[k] = maskRun; s_runStart[k] = runRun;
                s_maskSrc[k] = base + 4u * a.slotRuns + srcG; s_runSrc[k] = base + 4u * runBegin;
                s_dst[k] = off[s]; s_len[k] = g;
                maskRun += g; runRun += runs; ++k;
                off[s] += g;
                runBegin = runEnd;
            }
        }
        s_maskStart[k] = maskRun; s_runStart[k] = runRun;
        if (blockIdx.x == 0)
            for (uint32_t s = 0; s < kMaxPassSlots; ++s)
                if (a.args[s]) {
                    // {X, 1, 1, validRecords}: with the global capacity X is the sum of the ranks' counters, as on one GPU
                    const unsigned long long X = a.globalCap ? counted[s] : off[s];
                    a.args[s][0] = X > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)X; a.args[s][1] = 1; a.args[s][2] = 1; a.args[s][3] = off[s];
                    a.args[s][7] = status;
                }
    }
    __syncthreads();
    const uint32_t numSeg = kMaxPassSlots * a.world;
    const uint32_t totalMasks = s_maskStart[numSeg];
    for (uint32_t i = blockIdx.x * 256u + tid; i < totalMasks; i += gridDim.x * 256u) {
        uint32_t lo = 0, hi = numSeg;                                              // last k with s_maskStart[k] <= i
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_maskStart[mid] <= i) lo = mid; else hi = mid;
        }
        a.masks[lo % kMaxPassSlots][s_dst[lo] + (i - s_maskStart[lo])] = a.recv[(uint64_t)s_maskSrc[lo] + (i - s_maskStart[lo])];
    }
    const uint32_t totalRuns = s_runStart[numSeg];
    for (uint32_t it = blockIdx.x * 256u + (tid & ~63u); it < totalRuns; it += gridDim.x * 256u) {    // wave-uniform bound
        const uint32_t i = it + lane;
        uint32_t id = 0, lod = 0, off0 = 0, first = 0, cnt = 0;
        uint32_t* dst = nullptr;
        if (i < totalRuns) {
            uint32_t lo = 0, hi = numSeg;
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_runStart[mid] <= i) lo = mid; else hi = mid;
            }
            const uint32_t e = i - s_runStart[lo], runs = s_runStart[lo + 1] - s_runStart[lo], len = s_len[lo];
            const uint4 entry = *reinterpret_cast<const uint4*>(a.recv + (uint64_t)s_runSrc[lo] + 4ull * e);
            // the run ends where the next one of the same segment starts; the last one at the segment's group count.
            // Clamped to the groups this segment may write (capacity), so nothing lands outside its range.
            uint32_t next = e + 1u < runs ? a.recv[(uint64_t)s_runSrc[lo] + 4ull * (e + 1u) + 3u] : 0xFFFFFFFFu;
            next = next < len ? next : len;
            id = entry.x; lod = entry.y; off0 = entry.z; first = entry.w;
            cnt = first < next ? next - first : 0u;
            dst = a.records[lo % kMaxPassSlots] + 3ull * (s_dst[lo] + first);
        }
        if (cnt <= 8u)
            for (uint32_t j = 0; j < cnt; ++j) { dst[3u * j] = id; dst[3u * j + 1u] = lod; dst[3u * j + 2u] = off0 + 32u * j; }
        unsigned long long big = __ballot(cnt > 8u);
        while (big) {
            const int l = __builtin_ctzll(big);
            big &= big - 1ull;
            const uint32_t bid = __shfl(id, l), blod = __shfl(lod, l), boff = __shfl(off0, l), bcnt = __shfl(cnt, l);
            uint32_t* bdst = (uint32_t*)(((unsigned long long)__shfl((uint32_t)((unsigned long long)dst >> 32), l) << 32) | __shfl((uint32_t)(unsigned long long)dst, l));
            for (uint32_t j = lane; j < bcnt; j += 64u) { bdst[3u * j] = bid; bdst[3u * j + 1u] = blod; bdst[3u * j + 2u] = boff + 32u * j; }
        }
    }
}

// Launches the ordered-list build (count -> scan of the 256-batch sums -> expand; one compact launch for small passes) over a mask array.  side: on the device's side
// stream -- nothing later in a frame consumes the list, so it overlaps the passes that follow (HZB build,
// late phase); the back end joins it before any command that touches the same buffers and at the end of
// the submission (trhip_internal.h).
void emitListBuild(const trhip::DispatchCtx& ctx, const MeshletCullArgs& a, const char* prefix, bool side, const void* argsBase)
{
    auto emit = [&](const std::string& name, std::function<int(hipStream_t)> fn) {
        if (side) ctx.emitSide(name.c_str(), std::move(fn), { { a.listGroups ? (const void*)a.listGroups : argsBase, false }, { a.visMask, false }, { a.visibleList, true }, { a.drawArgs, true } });
        else ctx.emit(name.c_str(), std::move(fn));
    };
    if (!side && a.recordCapacity <= kCompactMaxTiles * kCompactTile) {
        // small pass: one launch (visCompactKernel); status words + ticket are scratch of this command, zeroed by the
        // recording's first clear launch
        const uint32_t tiles = (a.recordCapacity + kCompactTile - 1) / kCompactTile;
        const size_t words = (size_t)kCompactMaxTiles * kStatusStride * 2 + 4;
        uint32_t* mem = (uint32_t*)ctx.scratch(words * 4);
        if (mem && ctx.cl->recordClearWords(mem, words, 0, true) == TRHIP_OK) {
            unsigned long long* status = (unsigned long long*)mem;
            uint32_t* ticket = mem + kCompactMaxTiles * kStatusStride * 2;
            uint32_t grid = ctx.computeUnits();               // one 1024-thread workgroup per CU
            if (grid > tiles) grid = tiles;
            if (grid == 0) grid = 1;
            emit(std::string(prefix) + "compact", [a, status, ticket, grid](hipStream_t s) {
                TRHIP_LAUNCH(visCompactKernel, dim3(grid), dim3(kCompactThreads), 0, s, a, status, ticket);
                return trhip::launchStatus("visCompactKernel"); });
            return;
        }
    }
    const uint32_t needBlocks = (a.maxBatches + kWaves - 1) / kWaves;
    uint32_t gridSmall = ctx.computeUnits() * 8u;      // count / expand: light kernels, one wave per 64 groups
    if (side) {                                         // experiment: TRHIP_EXPAND_BLOCKS_PER_CU (side-stream list builds only)
        static const int perCU = [] { const char* e = getenv("TRHIP_EXPAND_BLOCKS_PER_CU"); return e ? atoi(e) : 0; }();
        if (perCU > 0) gridSmall = ctx.computeUnits() * (uint32_t)perCU;
    }
    if (gridSmall > needBlocks) gridSmall = needBlocks;
    if (gridSmall == 0) gridSmall = 1;
    const std::string p = prefix;
    const uint32_t supers = (a.maxBatches >> kSuperShift) + 1u;
    emit(p + "count", [a, supers](hipStream_t s) {
        TRHIP_LAUNCH(visCountKernel, dim3(supers), dim3(kCountThreads), 0, s, a);
        return trhip::launchStatus("visCountKernel"); });
    emit(p + "scan", [a](hipStream_t s) {
        TRHIP_LAUNCH(visSuperScanKernel, dim3(1), dim3(1024), 0, s, a);
        return trhip::launchStatus("visSuperScanKernel"); });
    // Experiment (TRHIP_DEFER_EXPAND=1; off): the expansion of a LARGE EARLY pass (122 MB of stores on C3, 26 us) HELD BACK -- it
    // enters the side stream behind the late meshlet cull (recordASMain flushes it in front of the late list build) instead of
    // beside the late instance pass and the late meshlet cull.  Measured (profiles/r4/experiments.md): those two drop from 11 +
    // 30 to 9 + 10 us, but the expansion then runs beside the second HZB build (13 + 5 -> 14 + 17 us), pushes the late list build
    // and the next frame's table rebuild into the next frame's instance pass (emit 16 -> 25 us): frame 0.470 -> 0.480 ms.  The
    // stores have to happen somewhere in the 0.13 ms of latency-bound launches around the cull kernel.
    static const bool deferExpand = [] { const char* e = getenv("TRHIP_DEFER_EXPAND"); return e ? atoi(e) != 0 : false; }();
    auto expandFn = [a, gridSmall](hipStream_t s) {
        TRHIP_LAUNCH(visExpandKernel, dim3(gridSmall), dim3(kBlock), 0, s, a);
        return trhip::launchStatus("visExpandKernel"); };
    if (side && deferExpand && ctx.variant == 0 && prefix[0] == 0)
        ctx.emitSideHeld("expand", std::move(expandFn), { { a.listGroups ? (const void*)a.listGroups : argsBase, false }, { a.visMask, false }, { a.visibleList, true }, { a.drawArgs, true } });
    else emit(p + "expand", std::move(expandFn));
}

template <bool F, bool O, bool C>
void launchCull(const MeshletCullArgs& a, uint32_t grid, bool table, hipStream_t s)
{
    if (O && table) TRHIP_LAUNCH((meshletCullKernel<F, O, C, true>), dim3(grid), dim3(kCullBlock), 0, s, a);      // the deferred mode
    else TRHIP_LAUNCH((meshletCullKernel<F, O, C, false>), dim3(grid), dim3(kCullBlock), 0, s, a);
}

// ---- the meshlet cull stream (MeshletCullStream): layout, allocation, build ------------------------------------
constexpr uint64_t kStreamBytesPerMeshlet = 20;
__host__ __device__ inline MeshletCullStream meshletStreamLayout(void* mem, uint64_t numMeshlets)
{
    const uint64_t coneAt = (numMeshlets * 16u + 255u) & ~255ull;                   // the cone words start on a 256-byte boundary
    return { reinterpret_cast<const float4*>(mem), reinterpret_cast<const uint32_t*>(reinterpret_cast<char*>(mem) + coneAt) };
}
inline uint64_t meshletStreamBytes(uint64_t numMeshlets) { return ((numMeshlets * 16u + 255u) & ~255ull) + numMeshlets * 4u; }

__global__ __launch_bounds__(256) void meshletStreamKernel(const MeshletData* __restrict__ meshlets, uint64_t n, MeshletCullStream out)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += stride) {
        const float4* p = reinterpret_cast<const float4*>(meshlets + i);
        const float4 sphere = p[0];
        const uint32_t cone = __float_as_uint(p[1].x);                               // m_ConeAxisAndCutoff: the word behind the sphere
        const_cast<float4*>(out.sphere)[i] = sphere;
        const_cast<uint32_t*>(out.cone)[i] = cone;
    }
}

int meshletStreamEnsure(trhip_buffer_t* meshlets)
{
    const uint64_t n = meshlets->byteSize / sizeof(MeshletData);
    const uint64_t need = meshletStreamBytes(n);
    if (meshlets->cullStreamBytes < need) {
        TRHIP_HIP(hipSetDevice(meshlets->dev->index));
        if (meshlets->cullStream) {
            int rc = meshlets->dev->syncAll();
            if (rc != TRHIP_OK) return rc;
            (void)hipFree(meshlets->cullStream);
            meshlets->cullStream = nullptr; meshlets->cullStreamBytes = 0;
        }
        TRHIP_HIP(hipMalloc(&meshlets->cullStream, (size_t)need));
        meshlets->cullStreamBytes = need;
        meshlets->cullStreamVersion = 0;
    }
    return TRHIP_OK;
}

// at submission time, on the stream of the cull that follows: no-op unless the meshlet buffer was written since the stream was built
int meshletStreamLaunchBuild(trhip_buffer_t* meshlets, hipStream_t s)
{
    const uint64_t v = meshlets->version;
    if (meshlets->cullStreamVersion == v) return TRHIP_OK;
    const uint64_t n = meshlets->byteSize / sizeof(MeshletData);
    const uint64_t blocks = (n + 255u) / 256u;
    TRHIP_LAUNCH(meshletStreamKernel, dim3((uint32_t)(blocks < 65536u ? (blocks ? blocks : 1u) : 65536u)), dim3(256), 0, s,
                       (const MeshletData*)meshlets->ptr, n, meshletStreamLayout(meshlets->cullStream, n));
    meshlets->cullStreamVersion = v;
    return trhip::launchStatus("meshletStreamKernel");
}

int recordASMain(trhip::DispatchCtx& ctx)
{
    // Binding set of BasePassRenderers.cpp:463-479 (t0 instances, t2 mesh data, t4 meshlets,
    // t7 amplification records, t8 HZB) + this build's outputs u0 visMask, u1 visibleList, u2 drawArgs.
    const BasePassConstants* k = (const BasePassConstants*)ctx.constants(0, sizeof(BasePassConstants));
    TRHIP_REQUIRE(k, "%s: constant buffer b0 (BasePassConstants, 256 bytes) missing", ctx.shaderName);
    trhip_buffer_t* instances = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 0);
    trhip_buffer_t* meshData = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 2);
    trhip_buffer_t* meshlets = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 4);
    trhip_buffer_t* records = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 7);
    trhip_texture_t* hzb = ctx.texture(TRHIP_BIND_TEXTURE_SRV, 8);
    trhip_buffer_t* visMask = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 0);
    trhip_buffer_t* visList = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 1);
    trhip_buffer_t* drawArgs = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 2);
    TRHIP_REQUIRE(instances && meshData && meshlets && records, "%s: needs SRVs t0, t2, t4, t7 (BasePassRenderers.cpp:463-479)", ctx.shaderName);
    TRHIP_REQUIRE(visMask && visList && drawArgs, "%s: needs UAVs u0 (visMask), u1 (visibleList), u2 (drawArgs)", ctx.shaderName);
    TRHIP_REQUIRE(ctx.indirect, "%s: launched by dispatchMeshIndirect (BasePassRenderers.cpp:497-502): use dispatch_indirect", ctx.shaderName);
    TRHIP_REQUIRE(instances->byteSize % sizeof(BasePassInstanceConstants) == 0, "%s: instance buffer size is not a multiple of 144", ctx.shaderName);
    TRHIP_REQUIRE(meshlets->byteSize % sizeof(MeshletData) == 0, "%s: meshlet buffer size is not a multiple of 32", ctx.shaderName);
    TRHIP_REQUIRE(drawArgs->byteSize >= 12, "%s: drawArgs buffer smaller than 12 bytes", ctx.shaderName);

    MeshletCullArgs a;
    memset(&a, 0, sizeof a);
    a.k = *k;
    const bool occlusion = (k->m_CullingFlags & kCullingFlagOcclusionCullingEnable) != 0;
    // LATE_CULL=1 follows an HZB rebuild and covers only what the early phase rejected: texel path, no table.
    // Small passes (capacity below 2^19 groups) also take the texel path: the table rebuild is a fixed ~20 us per
    // frame on the side stream and pays off only when the pass is long (same rule in k_gpuculling.hip).
    const bool useTable = occlusion && ctx.variant == 0 && records->byteSize / sizeof(MeshletAmplificationData) >= trhip::tableMinGroups();
    int rc = TRHIP_OK;
    {
        memset(&a.hzb, 0, sizeof a.hzb);
        if (occlusion) {
            TRHIP_REQUIRE(hzb, "%s: occlusion culling enabled but no HZB texture bound at t8", ctx.shaderName);
            TRHIP_REQUIRE(hzb->format == TRHIP_FORMAT_R16_FLOAT, "%s: HZB is not R16_FLOAT", ctx.shaderName);
            TRHIP_REQUIRE(hzb->width == k->m_HZBDimensions.x && hzb->height == k->m_HZBDimensions.y,
                          "%s: m_HZBDimensions %ux%u does not match the HZB texture %ux%u", ctx.shaderName,
                          k->m_HZBDimensions.x, k->m_HZBDimensions.y, hzb->width, hzb->height);
            a.hzb.base = (const _Float16*)hzb->ptr;
            a.hzb.width = hzb->width; a.hzb.height = hzb->height; a.hzb.mips = hzb->mips;
            for (uint32_t m = 0; m < hzb->mips; ++m) a.hzb.mipOffset[m] = (uint32_t)(hzb->mipOffset[m] / 2);
            if (useTable) {
                rc = trhip::hzbQuadEnsure(hzb);
                if (rc != TRHIP_OK) return rc;
                a.quad.base = (const _Float16*)hzb->quad;
                a.quad.total = hzb->quadTotal;
                for (uint32_t m = 0; m < hzb->mips; ++m) a.quad.offset[m] = hzb->quadOffset[m];
            }
        }
    }
    if (rc != TRHIP_OK) return rc;
    a.meshlets = (const MeshletData*)meshlets->ptr;
    TRHIP_REQUIRE(meshlets->byteSize >= sizeof(MeshletData), "%s: empty meshlet buffer", ctx.shaderName);
    rc = meshletStreamEnsure(meshlets);
    if (rc != TRHIP_OK) return rc;
    a.stream = meshletStreamLayout(meshlets->cullStream, meshlets->byteSize / sizeof(MeshletData));
    a.records = (const MeshletAmplificationData*)records->ptr;
    a.dispatchArgs = (const uint32_t*)((const char*)ctx.argsBuffer->ptr + ctx.argsOffset);
    a.argsWords = (ctx.argsBuffer->byteSize - ctx.argsOffset) >= 16 ? 4u : 3u;
    const uint64_t cap = records->byteSize / sizeof(MeshletAmplificationData);
    a.recordCapacity = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
    TRHIP_REQUIRE(a.recordCapacity <= (1u << 27), "%s: more than 2^27 records cannot be encoded as (g<<5)|lane", ctx.shaderName);
    TRHIP_REQUIRE(visMask->byteSize / 4 >= a.recordCapacity, "%s: visMask holds %llu groups, records buffer %u", ctx.shaderName,
                  (unsigned long long)(visMask->byteSize / 4), a.recordCapacity);
    a.numMeshlets = meshlets->byteSize / sizeof(MeshletData);
    TRHIP_REQUIRE(a.numMeshlets <= (1ull << 32), "%s: more than 2^32 meshlets (m_MeshletDataBufferIdx is 32 bits wide)", ctx.shaderName);
    a.visMask = (uint32_t*)visMask->ptr;
    a.visibleList = (uint32_t*)visList->ptr;
    const uint64_t lcap = visList->byteSize / 4;
    a.listCapacity = lcap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)lcap;
    a.drawArgs = (uint32_t*)drawArgs->ptr;
    static const bool noPerm = getenv("TRHIP_AS_NO_PERM") != nullptr;                // experiments: process the records in list order
    if (!noPerm && records->sidecar && records->sidecarBytes >= 256 + (uint64_t)a.recordCapacity * 16) {
        a.permHeader = (const uint32_t*)records->sidecar;
        a.perm = (const uint4*)(a.permHeader + 64);
    }
    rc = trhip::instanceCacheEnsure(instances);
    if (rc != TRHIP_OK) return rc;
    a.numInstances = (uint32_t)(instances->byteSize / sizeof(BasePassInstanceConstants));
    a.cache = instanceCacheLayout(instances->cullCache, a.numInstances);
    a.maxBatches = (a.recordCapacity + kBatch - 1) / kBatch;
    a.batchSum = (uint32_t*)(a.recordCapacity >= (1u << 19) ? ctx.scratchSide((size_t)a.maxBatches * 4)   // only the list build uses it
                                                            : ctx.scratch((size_t)a.maxBatches * 4));
    TRHIP_REQUIRE(a.batchSum, "%s: scratch allocation failed", ctx.shaderName);
    {
        const size_t supers = (a.maxBatches >> kSuperShift) + 1u;
        a.superSum = (uint32_t*)(a.recordCapacity >= (1u << 19) ? ctx.scratchSide(supers * 4) : ctx.scratch(supers * 4));
        TRHIP_REQUIRE(a.superSum, "%s: scratch allocation failed", ctx.shaderName);
    }

    // Persistent grid: the group count lives on the device (indirect), so launch enough
    // workgroups to fill the chip and let them stride over the chunks.
    // Five workgroups per CU: 31.7 KB of LDS each (13.8 KB of per-record data, 10 KB of ring, 5 KB of deferred list) and 96
    // VGPRs.  (LDS: the allocation granule makes 32 000 bytes the limit for five -- 32 720 ran four per CU, 11 % slower.)
    uint32_t blocksPerCU = TR_CULL_WAVES_PER_EU;
    if (const char* e = getenv("TRHIP_AS_BLOCKS_PER_CU")) blocksPerCU = (uint32_t)atoi(e) ? (uint32_t)atoi(e) : blocksPerCU;   // tuning experiments
    uint32_t grid = ctx.computeUnits() * blocksPerCU;
    const uint32_t needBlocks = (a.recordCapacity + kCullBatch * kCullWaves - 1) / (kCullBatch * kCullWaves);
    if (grid > needBlocks) grid = needBlocks;
    if (grid == 0) grid = 1;
    // the list build's group count: a word of back-end private memory that lives with the mask buffer (its sidecar): written by
    // the cull on the main stream, read by the list build, possibly on the side stream -- the same address in every recording,
    // so the hazard tracking orders the next frame's cull after this frame's list build through it
    if (!visMask->sidecar) {
        TRHIP_HIP(hipSetDevice(visMask->dev->index));
        TRHIP_HIP(hipMalloc(&visMask->sidecar, 256));
        visMask->sidecarBytes = 256;
    }
    a.listGroups = (uint32_t*)visMask->sidecar;
    ctx.cl->use(a.listGroups, ctx.cl->ops.size(), true);
    if (useTable) a.bands = cm::projBands(k->m_P00, k->m_P11, hzb->width, hzb->height);
    static const bool noShortPass = getenv("TRHIP_NO_SHORT_PASS") != nullptr;          // tests: the texel kernel's batch path on small passes too
    a.shortPassRounds = noShortPass ? 0u : kShortPassRounds;
    const uint32_t flags = k->m_CullingFlags & 7u;
    trhip_texture_t* quadOwner = useTable ? hzb : nullptr;
    const bool table = useTable;
    if (quadOwner) ctx.cl->use(quadOwner->quad, ctx.cl->ops.size(), false);     // the kernel reads the table: ordered after a side-stream rebuild
    ctx.emit("cull", [a, grid, flags, quadOwner, table, instances, meshData, meshlets](hipStream_t s) {
        // A virtual buffer bound to other memory after this list was recorded (trhip_buffer_bind_memory): the recorded pointers
        // are the old memory's, the derived stream would be built from the new one.  nvrhi rebuilds binding sets on such a
        // change; here the list must be recorded again.
        if ((const void*)a.meshlets != meshlets->ptr)
            return trhip::fail(TRHIP_ERR_STATE, "basepass_AS_Main: the meshlet buffer was bound to other memory after this command list was recorded: record it again");
        {                                                  // no-op unless the instance or mesh buffer was written since the cache was built
            int crc = trhip::instanceCacheLaunchBuild(instances, meshData, s);
            if (crc != TRHIP_OK) return crc;
            crc = meshletStreamLaunchBuild(meshlets, s);  // no-op unless the meshlet buffer was written since its stream was built
            if (crc != TRHIP_OK) return crc;
        }
        if (quadOwner) {                                   // no-op unless the HZB was written since its table was built
            int brc = trhip::hzbQuadLaunchBuild(quadOwner, s);
            if (brc != TRHIP_OK) return brc;
        }
        switch (flags) {
        case 0: launchCull<false, false, false>(a, grid, table, s); break;
        case 1: launchCull<true, false, false>(a, grid, table, s); break;
        case 2: launchCull<false, true, false>(a, grid, table, s); break;
        case 3: launchCull<true, true, false>(a, grid, table, s); break;
        case 4: launchCull<false, false, true>(a, grid, table, s); break;
        case 5: launchCull<true, false, true>(a, grid, table, s); break;
        case 6: launchCull<false, true, true>(a, grid, table, s); break;
        default: launchCull<true, true, true>(a, grid, table, s); break;
        }
        return trhip::launchStatus("meshletCullKernel"); });
    // The side stream costs two events and two cross-stream waits per run (~20 us of host time): worth it
    // when the list build is long (>= 2^19 groups of capacity), not for small passes.
    ctx.cl->flushHeldSide();                            // (the early pass's held list expansion goes in front of this pass's list build)
    emitListBuild(ctx, a, "", a.recordCapacity >= (1u << 19), ctx.argsBuffer->ptr);
    return TRHIP_OK;
}

// "visibility_CS_PackShard": pass slot s binds t(4s) records, t(4s+1) visMask, t(4s+2) dispatch args,
// t(4s+3) draw args (all four or none); u0 = the shard slot; u1 = the kernel's state words (16 bytes x
// (slotGroups / 1024 + 5): two halves, ZERO before the first dispatch; launches on one state buffer must be ordered (one
// stream): each uses one half and zeroes the other for the next); push constants {slotGroups, slotRuns}.
int recordPackShard(trhip::DispatchCtx& ctx)
{
    const uint32_t* push = (const uint32_t*)ctx.constants(0, 8);
    trhip_buffer_t* slot = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 0);
    trhip_buffer_t* state = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 1);
    TRHIP_REQUIRE(push && slot && state, "%s: needs push constants {slotGroups, slotRuns}, UAV u0 (shard slot) and UAV u1 (state words, zeroed once)", ctx.shaderName);
    ShardPackArgs a;
    memset(&a, 0, sizeof a);
    a.slotGroups = push[0];
    a.slotRuns = push[1];
    const uint64_t slotWords = (uint64_t)kSlotHeaderWords + 4ull * a.slotRuns + a.slotGroups;
    TRHIP_REQUIRE(slot->byteSize >= slotWords * 4, "%s: shard slot buffer smaller than 16 + 4 * %u + %u words", ctx.shaderName, a.slotRuns, a.slotGroups);
    const uint64_t maxTiles = (uint64_t)a.slotGroups / kPackTile + kMaxPassSlots + 1u;       // every pass slot may end in a partial tile
    TRHIP_REQUIRE(maxTiles < (1u << 30) && state->byteSize >= maxTiles * 16u, "%s: state buffer u1 smaller than %llu bytes", ctx.shaderName, (unsigned long long)(maxTiles * 16u));
    TRHIP_REQUIRE(((uintptr_t)slot->ptr & 15u) == 0 && ((uintptr_t)state->ptr & 7u) == 0, "%s: u0 / u1 not 16 / 8-byte aligned", ctx.shaderName);
    a.slot = (uint32_t*)slot->ptr;
    a.maxTiles = (uint32_t)maxTiles;
    unsigned long long* halves = (unsigned long long*)state->ptr;
    for (uint32_t s = 0; s < kMaxPassSlots; ++s) {
        trhip_buffer_t* rec = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 4 * s);
        trhip_buffer_t* mask = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 4 * s + 1);
        trhip_buffer_t* args = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 4 * s + 2);
        trhip_buffer_t* draw = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 4 * s + 3);
        if (!rec && !mask && !args && !draw) continue;
        TRHIP_REQUIRE(rec && mask && args && draw, "%s: pass slot %u needs t%u..t%u (records, visMask, dispatch args, draw args)", ctx.shaderName, s, 4 * s, 4 * s + 3);
        TRHIP_REQUIRE(args->byteSize >= 12 && draw->byteSize >= 12, "%s: argument buffers smaller than 12 bytes", ctx.shaderName);
        const uint64_t cap = std::min<uint64_t>(rec->byteSize / sizeof(MeshletAmplificationData), mask->byteSize / 4);
        a.records[s] = (const uint32_t*)rec->ptr;
        a.masks[s] = (const uint32_t*)mask->ptr;
        a.dispatchArgs[s] = (const uint32_t*)args->ptr;
        a.drawArgs[s] = (const uint32_t*)draw->ptr;
        a.argsWords[s] = args->byteSize >= 16 ? 4u : 3u;
        a.recordCapacity[s] = cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
    }
    // all workgroups resident at once (see the kernel): <= 4 per CU, 3 of them for the tiles
    a.tileBlocks = (uint32_t)std::min<uint64_t>(maxTiles, (uint64_t)ctx.computeUnits() * 3u);
    const uint32_t maskBlocks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(((uint64_t)a.slotGroups + 4u * kPackThreads - 1u) / (4u * kPackThreads), 1u), ctx.computeUnits());
    const uint32_t grid = a.tileBlocks + maskBlocks;
    // which half this launch uses: one counter per state buffer, advanced at LAUNCH time (a recorded command is executed
    // many times, and several recorded commands may share one state buffer)
    static std::mutex mu;
    static std::unordered_map<void*, std::shared_ptr<std::atomic<uint32_t>>> launches;
    std::shared_ptr<std::atomic<uint32_t>> count;
    {
        std::lock_guard<std::mutex> lock(mu);
        std::shared_ptr<std::atomic<uint32_t>>& c = launches[(void*)halves];
        if (!c) c = std::make_shared<std::atomic<uint32_t>>(0u);
        count = c;
    }
    ctx.emit("main", [a, grid, halves, count](hipStream_t s) {
        ShardPackArgs l = a;
        const uint32_t half = count->fetch_add(1u) & 1u;
        l.status = halves + (size_t)half * a.maxTiles;
        l.statusNext = halves + (size_t)(half ^ 1u) * a.maxTiles;
        TRHIP_LAUNCH(shardPackKernel, dim3(grid), dim3(kPackThreads), 0, s, l);
        return trhip::launchStatus("shardPackKernel"); });
    return TRHIP_OK;
}

// "visibility_CS_UnpackShards": t0 = the gathered slots (world x slot words); pass slot s binds
// u(4s) records, u(4s+1) masks, u(4s+2) visible list, u(4s+3) args (8 words: {G,1,1,G}, {V,1,1}, status);
// push constants {world, slotGroups, slotRuns}.
int recordUnpackShards(trhip::DispatchCtx& ctx)
{
    const uint32_t* push = (const uint32_t*)ctx.constants(0, 12);
    trhip_buffer_t* recv = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 0);
    TRHIP_REQUIRE(push && recv, "%s: needs push constants {world, slotGroups, slotRuns[, globalGroupCapacity]} and SRV t0 (gathered slots)", ctx.shaderName);
    ShardUnpackArgs a;
    memset(&a, 0, sizeof a);
    a.world = push[0];
    a.slotGroups = push[1];
    a.slotRuns = push[2];
    if (const uint32_t* push4 = (const uint32_t*)ctx.constants(0, 16)) a.globalCap = push4[3];   // optional 4th word
    TRHIP_REQUIRE(a.world >= 1 && a.world <= kMaxRanks, "%s: world size %u outside [1, %u]", ctx.shaderName, a.world, kMaxRanks);
    const uint64_t words = (uint64_t)a.world * (kSlotHeaderWords + 4ull * a.slotRuns + a.slotGroups);
    TRHIP_REQUIRE(words < (1ull << 32), "%s: %u slots of %u groups / %u runs exceed 2^32 words", ctx.shaderName, a.world, a.slotGroups, a.slotRuns);
    TRHIP_REQUIRE(recv->byteSize >= words * 4, "%s: gathered buffer smaller than world x slot", ctx.shaderName);
    TRHIP_REQUIRE(((uintptr_t)recv->ptr & 15u) == 0, "%s: t0 not 16-byte aligned", ctx.shaderName);
    a.recv = (const uint32_t*)recv->ptr;
    MeshletCullArgs lists[kMaxPassSlots];
    uint64_t copyWords = 0;
    for (uint32_t s = 0; s < kMaxPassSlots; ++s) {
        trhip_buffer_t* rec = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 4 * s);
        trhip_buffer_t* mask = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 4 * s + 1);
        trhip_buffer_t* list = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 4 * s + 2);
        trhip_buffer_t* args = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 4 * s + 3);
        if (!rec && !mask && !list && !args) continue;
        TRHIP_REQUIRE(rec && mask && list && args, "%s: pass slot %u needs u%u..u%u (records, masks, visible list, args)", ctx.shaderName, s, 4 * s, 4 * s + 3);
        TRHIP_REQUIRE(args->byteSize >= 32, "%s: args buffer of pass slot %u smaller than 32 bytes", ctx.shaderName, s);
        const uint64_t cap = std::min<uint64_t>(rec->byteSize / sizeof(MeshletAmplificationData), mask->byteSize / 4);
        TRHIP_REQUIRE(cap <= (1u << 27), "%s: more than 2^27 records cannot be encoded as (g<<5)|lane", ctx.shaderName);
        a.records[s] = (uint32_t*)rec->ptr;
        a.masks[s] = (uint32_t*)mask->ptr;
        a.args[s] = (uint32_t*)args->ptr;
        a.capacity[s] = (uint32_t)cap;
        copyWords += std::min<uint64_t>(cap, (uint64_t)a.world * a.slotGroups);
        MeshletCullArgs& l = lists[s];
        memset(&l, 0, sizeof l);
        l.dispatchArgs = a.args[s];
        l.argsWords = 4;
        l.recordCapacity = a.capacity[s];
        l.visMask = a.masks[s];
        l.visibleList = (uint32_t*)list->ptr;
        const uint64_t lcap = list->byteSize / 4;
        l.listCapacity = lcap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)lcap;
        l.drawArgs = a.args[s] + 4;
        l.maxBatches = (l.recordCapacity + kBatch - 1) / kBatch;
        l.batchSum = (uint32_t*)ctx.scratch((size_t)l.maxBatches * 4);
        const size_t supers = (l.maxBatches >> kSuperShift) + 1u;
        l.superSum = (uint32_t*)ctx.scratch(supers * 4);
        TRHIP_REQUIRE(l.batchSum && l.superSum, "%s: scratch allocation failed", ctx.shaderName);

    }
    uint32_t grid = ctx.computeUnits() * 4u;
    const uint64_t need = (copyWords + 255u) / 256u;
    if (grid > need) grid = (uint32_t)need;
    if (grid == 0) grid = 1;
    ctx.emit("unpack", [a, grid](hipStream_t s) {
        TRHIP_LAUNCH(shardUnpackKernel, dim3(grid), dim3(256), 0, s, a);
        return trhip::launchStatus("shardUnpackKernel"); });
    for (uint32_t s = 0; s < kMaxPassSlots; ++s)
        if (a.records[s]) {
            const char prefix[] = { 's', 'l', 'o', 't', (char)('0' + s), '_', 0 };
            emitListBuild(ctx, lists[s], prefix, false, nullptr);
        }
    return TRHIP_OK;
}

trhip::ShaderRegistrar rp("visibility_CS_PackShard", recordPackShard, 0);
trhip::ShaderRegistrar ru("visibility_CS_UnpackShards", recordUnpackShards, 0);
trhip::ShaderRegistrar r0("basepass_AS_Main LATE_CULL=0", recordASMain, 0);
trhip::ShaderRegistrar r1("basepass_AS_Main LATE_CULL=1", recordASMain, 1);
trhip::ShaderRegistrar r2("basepass_AS_Main_cull", recordASMain, 0);

} // namespace

#if defined(TR_STAMPS) || defined(TR_COUNT_SLOW) || defined(TR_COUNT_PATHS)
extern "C" int trhip_debug_read_stamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stampSums), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stampSums), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif
