// k_giprobe.hip -- "giprobevisualization_CS_VisualizeGIProbesCulling" for gfx950: the SECOND consumer of the cull
// primitives (FrustumCull / OcclusionCull + compaction into indirect draw arguments), SURVEY.md 8(f) rank 3.
//
// Reference: source/shaders/giprobevisualization.hlsl:16-69, dispatched by GIDebugRenderer::RenderDDGIDebug
// (source/GIRenderer.cpp:690-735) with ceil(numProbes / 32) groups.  Per probe: [state / world position from the
// RTXGI-DDGI volume :29-37] -> view space (:39-40) -> FrustumCull (:42) -> OcclusionCull (:47-60) ->
// InterlockedAdd(g_OutProbeIndirectArgs[0].m_InstanceCount) (:62-63) -> position + probe index appended (:66-67).
//
// What differs, and why:
//   * DDGILoadProbeState / DDGIGetProbeCoords / DDGIGetProbeWorldPosition are RTXGI SDK code (extern/nvidia/RTXGI-DDGI,
//     an empty submodule): the probe states and world positions are INPUT buffers here -- t10 = float3 per probe,
//     t11 = one float per probe (RTXGI_DDGI_PROBE_STATE_INACTIVE = 1) -- in place of t10 (volume descriptors) and u10
//     (probe-data texture array).  Everything after `probeWorldPosition` is the reference's arithmetic.
//   * the reference appends in whatever order its atomics resolve; here the visible probes are appended in ascending
//     probe order (the canonical order the oracle states), by an ordered single-launch compaction: workgroups take tiles
//     of 256 probes in ticket order, a tile publishes its count and sums the counts of all its predecessors (they hold
//     smaller tickets, i.e. belong to workgroups already running: no deadlock whatever the grid size).
#include "cull_math.hip.h"
#include "trhip_internal.h"

using namespace interop;

namespace
{

constexpr uint32_t kProbeTile = 256;
constexpr uint32_t kProbeMaxTiles = 4096;                 // 2^20 probes
constexpr uint32_t kProbeStatusStride = 2;                // 64-bit words per tile status (16 bytes)
constexpr unsigned long long kProbeFlag = 1ull << 63, kProbePoison = 1ull << 40;

struct ProbeCullArgs
{
    GIProbeVisualizationUpdateConsts k;
    const float* positions;            // 3 floats per probe
    const float* states;               // 1 float per probe
    cm::Hzb hzb;
    float* outPositions;  uint32_t outPositionCapacity;    // entries (float3)
    uint32_t* drawArgs;                // DrawIndexedIndirectArguments
    uint32_t* outInstanceToProbe; uint32_t outIndexCapacity;
    unsigned long long* status;        // zeroed per launch
    uint32_t* ticket;                  // zeroed per launch
    uint32_t numProbes, numTiles;
};

__global__ __launch_bounds__(kProbeTile) void giProbeCullKernel(ProbeCullArgs a)
{
    __shared__ uint32_t s_wave[kProbeTile / 64];
    __shared__ unsigned long long s_pre[kProbeTile / 64];
    __shared__ uint32_t s_tile;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // the counter the pass continues from (GIRenderer.cpp:687-689 writes 0 before the dispatch): read by every workgroup
    // before any tile can finish the pass and rewrite it
    const uint32_t base = a.drawArgs[1];
    const cm::M43 V = cm::loadM43(a.k.m_WorldToView);
    for (;;) {
        __syncthreads();
        if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= a.numTiles) return;
        const uint32_t probeIndex = tile * kProbeTile + tid;                                   // :18-23
        bool visible = probeIndex < a.numProbes;
        cm::F3 wp = { 0.f, 0.f, 0.f };
        if (visible) {
            if (a.k.m_bHideInactiveProbes && a.states[probeIndex] == 1.0f) visible = false;    // :29-34
            wp = { a.positions[3u * probeIndex], a.positions[3u * probeIndex + 1u], a.positions[3u * probeIndex + 2u] };   // :36-37
        }
        if (visible) {
            const cm::F3 v = cm::toView(wp, V);                                                // :39-40
            visible = cm::frustumVisible(v, a.k.m_ProbeRadius, a.k.m_Frustum.x, a.k.m_Frustum.y, a.k.m_Frustum.z, a.k.m_Frustum.w);   // :42-45
            if (visible) visible = cm::occlusionVisible(v, a.k.m_ProbeRadius, a.k.m_NearPlane, a.k.m_P00, a.k.m_P11, a.hzb);          // :47-60
        }
        // ---- ordered compaction (replaces InterlockedAdd :62-63) --------------------------------------------------
        const unsigned long long ballot = __ballot(visible);
        const uint32_t rankInWave = (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(ballot);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kProbeTile / 64; ++w) { if (w < wave) before += s_wave[w]; total += s_wave[w]; }
        if (tid == 0) __hip_atomic_store(&a.status[(uint64_t)tile * kProbeStatusStride], kProbeFlag | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long sum = 0;
        for (uint32_t j = tid; j < tile; j += kProbeTile) {
            unsigned long long v = 0;
            uint32_t spins = 0;
            for (;;) {
                v = __hip_atomic_load(&a.status[(uint64_t)j * kProbeStatusStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v & kProbeFlag) { v &= ~kProbeFlag; break; }
                if (++spins > (1u << 22)) { v = kProbePoison; break; }               // never seen; a hang would take the GPU down
                __builtin_amdgcn_s_sleep(1);
            }
            sum += v;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        if (lane == 0) s_pre[wave] = sum;
        __syncthreads();
        unsigned long long prefix = 0;
#pragma unroll
        for (uint32_t w = 0; w < kProbeTile / 64; ++w) prefix += s_pre[w];
        const bool poisoned = (prefix >> 40) != 0ull;
        if (visible && !poisoned) {
            const uint64_t outInstanceIndex = (uint64_t)base + prefix + before + rankInWave;
            if (outInstanceIndex < a.outPositionCapacity) {                                       // :66
                a.outPositions[3u * outInstanceIndex] = wp.x;
                a.outPositions[3u * outInstanceIndex + 1u] = wp.y;
                a.outPositions[3u * outInstanceIndex + 2u] = wp.z;
            }
            if (outInstanceIndex < a.outIndexCapacity) a.outInstanceToProbe[outInstanceIndex] = probeIndex;   // :67
        }
        if (tid == 0 && tile == a.numTiles - 1u)
            a.drawArgs[1] = poisoned ? 0xFFFFFFFFu : base + (uint32_t)prefix + total;             // m_InstanceCount
    }
}

int recordGIProbeCull(trhip::DispatchCtx& ctx)
{
    // GIRenderer.cpp:707-733
    const GIProbeVisualizationUpdateConsts* k = (const GIProbeVisualizationUpdateConsts*)ctx.constants(0, sizeof(GIProbeVisualizationUpdateConsts));
    TRHIP_REQUIRE(k, "%s: constant buffer b0 (GIProbeVisualizationUpdateConsts, 124 bytes) missing", ctx.shaderName);
    trhip_texture_t* hzb = ctx.texture(TRHIP_BIND_TEXTURE_SRV, 0);
    trhip_buffer_t* positions = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 10);
    trhip_buffer_t* states = ctx.buffer(TRHIP_BIND_STRUCTURED_SRV, 11);
    trhip_buffer_t* outPositions = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 0);
    trhip_buffer_t* drawArgs = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 1);
    trhip_buffer_t* outIndex = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 2);
    TRHIP_REQUIRE(hzb && positions && states, "%s: needs t0 (HZB), t10 (probe positions, float3) and t11 (probe states, float): the inputs that replace the DDGI volume", ctx.shaderName);
    TRHIP_REQUIRE(outPositions && drawArgs && outIndex, "%s: needs UAVs u0 (probe positions), u1 (DrawIndexedIndirectArguments), u2 (instance -> probe index)", ctx.shaderName);
    TRHIP_REQUIRE(hzb->format == TRHIP_FORMAT_R16_FLOAT && hzb->ptr, "%s: HZB is not a bound R16_FLOAT texture", ctx.shaderName);
    TRHIP_REQUIRE(hzb->width == k->m_HZBDimensions.x && hzb->height == k->m_HZBDimensions.y, "%s: m_HZBDimensions %ux%u does not match the HZB texture %ux%u",
                  ctx.shaderName, k->m_HZBDimensions.x, k->m_HZBDimensions.y, hzb->width, hzb->height);
    TRHIP_REQUIRE(drawArgs->byteSize >= sizeof(DrawIndexedIndirectArguments), "%s: u1 smaller than DrawIndexedIndirectArguments (20 bytes)", ctx.shaderName);
    TRHIP_REQUIRE(!ctx.indirect, "%s: dispatched directly (GIRenderer.cpp:729)", ctx.shaderName);
    const uint64_t threads = (uint64_t)ctx.gx * kNumThreadsPerWave;
    const uint32_t n = threads < k->m_NumProbes ? (uint32_t)threads : k->m_NumProbes;
    TRHIP_REQUIRE((uint64_t)n * 12 <= positions->byteSize && (uint64_t)n * 4 <= states->byteSize, "%s: m_NumProbes exceeds the probe input buffers", ctx.shaderName);
    TRHIP_REQUIRE(n <= kProbeTile * kProbeMaxTiles, "%s: more than %u probes", ctx.shaderName, kProbeTile * kProbeMaxTiles);
    if (n == 0) return TRHIP_OK;
    ProbeCullArgs a;
    memset(&a, 0, sizeof a);
    a.k = *k;
    a.positions = (const float*)positions->ptr;
    a.states = (const float*)states->ptr;
    a.hzb.base = (const _Float16*)hzb->ptr;
    a.hzb.width = hzb->width; a.hzb.height = hzb->height; a.hzb.mips = hzb->mips;
    for (uint32_t m = 0; m < hzb->mips; ++m) a.hzb.mipOffset[m] = (uint32_t)(hzb->mipOffset[m] / 2);
    a.outPositions = (float*)outPositions->ptr;
    a.outPositionCapacity = (uint32_t)std::min<uint64_t>(outPositions->byteSize / 12, 0xFFFFFFFFull);
    a.drawArgs = (uint32_t*)drawArgs->ptr;
    a.outInstanceToProbe = (uint32_t*)outIndex->ptr;
    a.outIndexCapacity = (uint32_t)std::min<uint64_t>(outIndex->byteSize / 4, 0xFFFFFFFFull);
    a.numProbes = n;
    a.numTiles = (n + kProbeTile - 1) / kProbeTile;
    const size_t words = (size_t)kProbeMaxTiles * kProbeStatusStride * 2 + 4;
    uint32_t* mem = (uint32_t*)ctx.scratch(words * 4);
    TRHIP_REQUIRE(mem, "%s: scratch allocation failed", ctx.shaderName);
    int rc = ctx.cl->recordClearWords(mem, words, 0, true);
    if (rc != TRHIP_OK) return rc;
    a.status = (unsigned long long*)mem;
    a.ticket = mem + (size_t)kProbeMaxTiles * kProbeStatusStride * 2;
    uint32_t grid = ctx.computeUnits() * 4u;
    if (grid > a.numTiles) grid = a.numTiles;
    ctx.emit("main", [a, grid](hipStream_t s) {
        TRHIP_LAUNCH(giProbeCullKernel, dim3(grid), dim3(kProbeTile), 0, s, a);
        return trhip::launchStatus("giProbeCullKernel"); });
    return TRHIP_OK;
}

trhip::ShaderRegistrar r0("giprobevisualization_CS_VisualizeGIProbesCulling", recordGIProbeCull, 0);

} // namespace
