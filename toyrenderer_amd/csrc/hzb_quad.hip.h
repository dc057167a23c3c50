// hzb_quad.hip.h -- the footprint-min table of an HZB (trhip_texture_t::quad): layout, arguments, and the strip routine that
// builds it.  Its own header because several launches run the routine: hzbQuadBuildKernel (k_hzb.hip: the stand-alone build),
// and, as extra workgroups of the early instance pass's own launches -- no launch of its own, no cross-stream dependency in
// front of the meshlet cull -- instanceFusedKernel<0> (small passes) and instanceScanKernel<0> + instanceEmitKernel<0> (large
// passes: half of the strips beside the scan's seventeen workgroups of dependent round trips, half behind the emit's own
// workgroups) in k_gpuculling.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <cstring>

#include "cull_math.hip.h"
#include "trhip_internal.h"

namespace trhip
{

// Footprint-min table (trhip_texture_t::quad): entry (X, Y) of mip k, X in [0, w], Y in [0, h], is the min of
// the texels {clamp(X-1), clamp(X)} x {clamp(Y-1), clamp(Y)} -- the footprint of a bilinear lookup whose
// origin floor(uv*dim - 0.5) is (X-1, Y-1) and whose two weights per axis are non-zero (culling.hlsli:78
// with the min-reduction sampler, CommonResources.cpp:276-287).  One thread per entry, all mips in one launch.
// Layout: 8 x 8 BLOCKS of entries (64 x 2 B = one 128-byte cache line per block), blocks row-major, ((w >> 3) + 1) per
// block row: the lookups of the meshlet cull scatter over a 2-D screen region, and a region covers 2-4x fewer lines
// this way than with row-major entries (a line = 64 x 1 texels).  Entry (X, Y) of mip k sits at
// quadOffset[k] + ((Y >> 3) * blocksPerRow + (X >> 3)) * 64 + (Y & 7) * 8 + (X & 7); blocks past the edge are padding.
struct QuadArgs
{
    const _Float16* base;
    _Float16* out;
    uint32_t width, height, mips, total;
    uint32_t mipOffset[16];       // texels
    uint32_t quadOffset[16];      // entries
    uint32_t blocksPerRow[16];    // (mip width >> 3) + 1
    uint32_t stripsPerRow[16];    // ceil(blocksPerRow / kQuadStripBlocks)
    uint32_t firstStrip[17];      // strips (= workgroups) of the mips before k
};

// A workgroup builds one STRIP of the table: kQuadStripBlocks consecutive 8 x 8 blocks of one block row = 256 x 8
// entries, contiguous in the table.  Their footprints are 257 x 9 texels: loaded once (rows of consecutive texels, clamped
// to the edge, which is exactly what the padding entries and the border entries need) into LDS, then every thread takes
// eight entries = four LDS reads each.  (One thread per entry with four 2-byte global loads took 33 us next to the
// instance pass for 11 MB of traffic, four entries per thread 30 us: a chain of dependent little loads.  Row stride 257
// words: the 8 x 8 lanes of a wave spread over 15 banks.)
constexpr uint32_t kQuadStripBlocks = 32;
constexpr uint32_t kQuadStripCols = kQuadStripBlocks * 8;          // 256 entries per entry row

// Builds strip `strip_` (0 .. firstStrip[mips] - 1) of the table with 256 threads (tid = 0 .. 255), s_t: 9 x 257 floats of LDS.
// `valid` = false: no strip, the threads only take part in the barrier (every thread of the workgroup calls this once:
// workgroups of 256 x n threads build n strips side by side).
__device__ __forceinline__ void hzbQuadStrip(const QuadArgs& a, uint32_t strip_, float (*s_t)[kQuadStripCols + 1], uint32_t tid = threadIdx.x, bool valid = true)
{
    if (!valid) { __syncthreads(); return; }
    uint32_t k = 0;
    for (uint32_t m = 1; m < a.mips; ++m) k += strip_ >= a.firstStrip[m] ? 1u : 0u;
    const uint32_t mw = (a.width >> k) ? (a.width >> k) : 1u, mh = (a.height >> k) ? (a.height >> k) : 1u;
    const uint32_t strip = strip_ - a.firstStrip[k];
    const uint32_t brow = strip / a.stripsPerRow[k], seg = strip - brow * a.stripsPerRow[k];
    const uint32_t X0 = seg * kQuadStripCols, Y0 = brow * 8u;
    const _Float16* __restrict__ t = a.base + a.mipOffset[k];
    // texel (Y0 - 1 + r, X0 - 1 + c), clamped
    {
        const int tx = min(max((int)(X0 + tid) - 1, 0), (int)mw - 1);
        float v[9];
#pragma unroll
        for (uint32_t r = 0; r < 9; ++r) {
            const int ty = min(max((int)(Y0 + r) - 1, 0), (int)mh - 1);
            v[r] = (float)t[(uint32_t)ty * mw + (uint32_t)tx];
        }
        float edge = 0.f;
        if (tid < 9) {
            const int ty = min(max((int)(Y0 + tid) - 1, 0), (int)mh - 1);
            const int ex = min((int)(X0 + kQuadStripCols) - 1, (int)mw - 1);
            edge = (float)t[(uint32_t)ty * mw + (uint32_t)ex];
        }
#pragma unroll
        for (uint32_t r = 0; r < 9; ++r) s_t[r][tid] = v[r];
        if (tid < 9) s_t[tid][kQuadStripCols] = edge;
    }
    __syncthreads();
    const uint32_t blocksHere = min(kQuadStripBlocks, a.blocksPerRow[k] - seg * kQuadStripBlocks);
    _Float16* __restrict__ dst = a.out + a.quadOffset[k] + ((uint64_t)brow * a.blocksPerRow[k] + (uint64_t)seg * kQuadStripBlocks) * 64u;
#pragma unroll
    for (uint32_t e = 0; e < 8; ++e) {
        const uint32_t i = e * 256u + tid;                       // entry of the strip: block i >> 6, row (i >> 3) & 7, column i & 7
        const uint32_t blk = i >> 6;
        if (blk >= blocksHere) continue;
        const uint32_t xl = blk * 8u + (i & 7u), yl = (i >> 3) & 7u;
        dst[i] = (_Float16)cm::min_(cm::min_(cm::min_(s_t[yl][xl], s_t[yl][xl + 1]), s_t[yl + 1][xl]), s_t[yl + 1][xl + 1]);     // min of fp16 values: exact
    }
}

inline QuadArgs quadArgs(const trhip_texture_t* tex)
{
    QuadArgs a;
    memset(&a, 0, sizeof a);
    a.base = (const _Float16*)tex->ptr;
    a.out = (_Float16*)tex->quad;
    a.width = tex->width; a.height = tex->height; a.mips = tex->mips; a.total = tex->quadTotal;
    uint32_t strips = 0;
    for (uint32_t i = 0; i < tex->mips; ++i) {
        a.mipOffset[i] = (uint32_t)(tex->mipOffset[i] / 2);
        a.quadOffset[i] = tex->quadOffset[i];
        a.blocksPerRow[i] = (tex->mipW(i) >> 3) + 1u;
        a.stripsPerRow[i] = (a.blocksPerRow[i] + kQuadStripBlocks - 1u) / kQuadStripBlocks;
        a.firstStrip[i] = strips;
        strips += a.stripsPerRow[i] * ((tex->mipH(i) >> 3) + 1u);
    }
    a.firstStrip[tex->mips] = strips;
    return a;
}

} // namespace trhip
