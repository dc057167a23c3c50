// k_hzb.hip -- HZB construction for gfx950:
//   "minmaxdownsample_CS_Main"                                  depth -> HZB mip 0
//   "ffx_spd_downsample_pass_CS FFX_SPD_OPTION_DOWNSAMPLE_FILTER={1 min, 2 max}"  mips 1..N-1
//
// Reference: source/shaders/minmaxdownsample.hlsl:10-35 and FFXHelpers::SPD::Execute
// (source/FFXHelpers.cpp:36-115) driven by BasePassRenderer::GenerateHZB
// (source/BasePassRenderers.cpp:505-542).  The FidelityFX SPD shader itself is not in the
// reference tree (empty submodule) -> its published behaviour is restated: every texel of mip k+1
// is the min (max) of the 2x2 block of mip k, all mips from one dispatch.  Parity unpinned for
// SPD; the convention is fixed by the oracle (tr_oracle.c orc_hzb_build).
//
// HBM traffic: depth W*H*4 B read once, HZB mip chain (w*h*2 B * 4/3) written once and mip 0 read
// once.  Bound: HBM; ~45 MB per build at 3840x2160.
#include "cull_math.hip.h"
#include "hzb_quad.hip.h"
#include "trhip_internal.h"

using namespace interop;

namespace
{

// minmaxdownsample.hlsl:15-34.  Gather at uv=(tid+0.5)/outDim on the WxH depth image with a
// point-clamp sampler = the 2x2 quad floor(uv*dim-0.5)+{0,1}, clamped to the edge (Q11).
template <bool MAX>
__global__ __launch_bounds__(256) void minMaxDownsampleKernel(const float* __restrict__ depth, uint32_t W, uint32_t H,
                                                              _Float16* __restrict__ out, uint32_t ow, uint32_t oh)
{
    const uint32_t x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= ow || y >= oh) return;                                                // :15-18
    const float u = ((float)x + 0.5f) / (float)ow, v = ((float)y + 0.5f) / (float)oh; // :20
    const float fx = cm::fma_(u, (float)W, -0.5f), fy = cm::fma_(v, (float)H, -0.5f);
    int x0 = (int)__builtin_floorf(fx), y0 = (int)__builtin_floorf(fy);
    int x1 = min(max(x0 + 1, 0), (int)W - 1), y1 = min(max(y0 + 1, 0), (int)H - 1);
    x0 = min(max(x0, 0), (int)W - 1);
    y0 = min(max(y0, 0), (int)H - 1);
    const float a = depth[(uint64_t)y0 * W + x0], b = depth[(uint64_t)y0 * W + x1];
    const float c = depth[(uint64_t)y1 * W + x0], d = depth[(uint64_t)y1 * W + x1];
    const float r = MAX ? cm::max_(cm::max_(a, b), cm::max_(c, d))                  // Max4 / Min4, :25-32
                        : cm::min_(cm::min_(a, b), cm::min_(c, d));
    out[(uint64_t)y * ow + x] = (_Float16)r;                                        // R16_FLOAT store, RNE (Q10)
}

struct SpdArgs
{
    _Float16* base;
    uint32_t width, height, mips;
    uint32_t mipOffset[16];       // texels
};

template <bool MAX>
__device__ __forceinline__ float red4(float a, float b, float c, float d)
{
    return MAX ? cm::max_(cm::max_(a, b), cm::max_(c, d)) : cm::min_(cm::min_(a, b), cm::min_(c, d));
}

// Tile pass: one workgroup reduces a 64x64 tile of mip 0 to mips 1..6 through LDS.
// Needs width and height to be multiples of 64.
template <bool MAX>
__global__ __launch_bounds__(256) void spdTileKernel(SpdArgs a, uint32_t lastMip)
{
    __shared__ float s[2][32 * 32];
    const uint32_t tid = threadIdx.x;
    const uint32_t tx = blockIdx.x, ty = blockIdx.y;
    const _Float16* m0 = a.base + a.mipOffset[0];
    // mip 1: each thread produces a 2x2 patch of the 32x32 tile from a 4x4 patch of mip 0
    {
        const uint32_t px = (tid & 15) * 2, py = (tid >> 4) * 2;
        for (uint32_t dy = 0; dy < 2; ++dy)
            for (uint32_t dx = 0; dx < 2; ++dx) {
                const uint32_t ox = px + dx, oy = py + dy;
                const uint32_t sx = tx * 64 + ox * 2, sy = ty * 64 + oy * 2;
                const float v = red4<MAX>((float)m0[(uint64_t)sy * a.width + sx], (float)m0[(uint64_t)sy * a.width + sx + 1],
                                          (float)m0[(uint64_t)(sy + 1) * a.width + sx], (float)m0[(uint64_t)(sy + 1) * a.width + sx + 1]);
                s[0][oy * 32 + ox] = v;
                if (lastMip >= 1) {
                    const uint32_t mw = a.width >> 1;
                    a.base[a.mipOffset[1] + (uint64_t)(ty * 32 + oy) * mw + tx * 32 + ox] = (_Float16)v;
                }
            }
    }
    __syncthreads();
    uint32_t cur = 0, dim = 32;
    for (uint32_t mip = 2; mip <= 6 && mip <= lastMip; ++mip) {
        const uint32_t od = dim >> 1;
        if (tid < od * od) {
            const uint32_t ox = tid % od, oy = tid / od;
            const float* p = s[cur];
            const float v = red4<MAX>(p[(2 * oy) * dim + 2 * ox], p[(2 * oy) * dim + 2 * ox + 1],
                                      p[(2 * oy + 1) * dim + 2 * ox], p[(2 * oy + 1) * dim + 2 * ox + 1]);
            s[cur ^ 1][oy * od + ox] = v;
            const uint32_t mw = a.width >> mip;
            a.base[a.mipOffset[mip] + (uint64_t)(ty * od + oy) * mw + tx * od + ox] = (_Float16)v;
        }
        __syncthreads();
        cur ^= 1;
        dim = od;
    }
}

// Depth -> mip 0 -> mips 1..6 of one 64x64 tile in ONE launch: minMaxDownsampleKernel + spdTileKernel when the two
// dispatches follow each other on the same HZB (recordSPD).  Mip 0 is produced row by row (lane = consecutive x, the
// access pattern of minMaxDownsampleKernel), stored, and kept in LDS AS STORED (rounded to fp16): the reduction above
// it reads what a separate SPD pass would read back.
template <bool MAX>
__global__ __launch_bounds__(256) void hzbDepthTileKernel(const float* __restrict__ depth, uint32_t W, uint32_t H, SpdArgs a, uint32_t lastMip)
{
    __shared__ float s0[64 * 64];
    __shared__ float s[2][32 * 32];
    const uint32_t tid = threadIdx.x;
    const uint32_t tx = blockIdx.x, ty = blockIdx.y;
    _Float16* m0 = a.base + a.mipOffset[0];
    const uint32_t ow = a.width, oh = a.height;
    {
        const uint32_t lx = tid & 63u, x = tx * 64 + lx;
        const float u = ((float)x + 0.5f) / (float)ow;                                // minmaxdownsample.hlsl:20
        const float fx = cm::fma_(u, (float)W, -0.5f);
        int x0 = (int)__builtin_floorf(fx);
        const int x1 = min(max(x0 + 1, 0), (int)W - 1);
        x0 = min(max(x0, 0), (int)W - 1);
        const int xl = min(x0, (int)W - 2);                                           // base of the 8-byte load (W >= 2)
        // All 32 loads of the thread's 16 texels are requested before the first result is stored (the stores to the HZB
        // may alias the depth image as far as the compiler knows, so it would not move a load above them by itself).
        struct __attribute__((packed, aligned(4))) Pair { float a, b; };
        const bool wide = W >= 2u;               // the two texels of a row are neighbours (or the same one at a clamped edge):
                                                 // ONE 8-byte load per row (dword-aligned) instead of two scalar loads
        Pair v0[16], v1[16];
        // The source rows of a wave's 16 output rows are the same for all its lanes: lane j works them out for row j
        // (one correctly rounded division each, minmaxdownsample.hlsl:20) and the loop reads them with v_readlane -- a
        // sixteenth of the row arithmetic, and the load addresses get a scalar row base (13.5 -> 13.0 us per build;
        // the kernel without its depth reads takes 10: profiles/r3/experiments.md section 11).
        int rowY0, rowY1;
        {
            const uint32_t ly = (tid & 15u) * 4 + (tid >> 6), y = ty * 64 + ly;
            const float v = ((float)y + 0.5f) / (float)oh;
            const float fy = cm::fma_(v, (float)H, -0.5f);
            const int f0 = (int)__builtin_floorf(fy);
            rowY1 = min(max(f0 + 1, 0), (int)H - 1);
            rowY0 = min(max(f0, 0), (int)H - 1);
        }
#pragma unroll
        for (uint32_t it = 0; it < 16; ++it) {
            const int y0 = __builtin_amdgcn_readlane(rowY0, (int)it), y1 = __builtin_amdgcn_readlane(rowY1, (int)it);
            if (wide) {
                v0[it] = *reinterpret_cast<const Pair*>(depth + (uint64_t)y0 * W + xl);
                v1[it] = *reinterpret_cast<const Pair*>(depth + (uint64_t)y1 * W + xl);
            } else {
                v0[it] = { depth[(uint64_t)y0 * W + x0], depth[(uint64_t)y0 * W + x1] };
                v1[it] = { depth[(uint64_t)y1 * W + x0], depth[(uint64_t)y1 * W + x1] };
            }
        }
#pragma unroll
        for (uint32_t it = 0; it < 16; ++it) {
            const uint32_t ly = it * 4 + (tid >> 6), y = ty * 64 + ly;
            const float p = !wide || x0 == xl ? v0[it].a : v0[it].b, q = wide && x1 == xl ? v0[it].a : v0[it].b;
            const float r = !wide || x0 == xl ? v1[it].a : v1[it].b, t = wide && x1 == xl ? v1[it].a : v1[it].b;
            const _Float16 h = (_Float16)red4<MAX>(p, q, r, t);                       // R16_FLOAT store, RNE (Q10)
            m0[(uint64_t)y * ow + x] = h;
            s0[ly * 64 + lx] = (float)h;
        }
    }
    __syncthreads();
    {
        const uint32_t px = (tid & 15) * 2, py = (tid >> 4) * 2;
        for (uint32_t dy = 0; dy < 2; ++dy)
            for (uint32_t dx = 0; dx < 2; ++dx) {
                const uint32_t ox = px + dx, oy = py + dy;
                const float v = red4<MAX>(s0[(2 * oy) * 64 + 2 * ox], s0[(2 * oy) * 64 + 2 * ox + 1],
                                          s0[(2 * oy + 1) * 64 + 2 * ox], s0[(2 * oy + 1) * 64 + 2 * ox + 1]);
                s[0][oy * 32 + ox] = v;
                if (lastMip >= 1) {
                    const uint32_t mw = a.width >> 1;
                    a.base[a.mipOffset[1] + (uint64_t)(ty * 32 + oy) * mw + tx * 32 + ox] = (_Float16)v;
                }
            }
    }
    __syncthreads();
    uint32_t cur = 0, dim = 32;
    for (uint32_t mip = 2; mip <= 6 && mip <= lastMip; ++mip) {
        const uint32_t od = dim >> 1;
        if (tid < od * od) {
            const uint32_t ox = tid % od, oy = tid / od;
            const float* p = s[cur];
            const float v = red4<MAX>(p[(2 * oy) * dim + 2 * ox], p[(2 * oy) * dim + 2 * ox + 1],
                                      p[(2 * oy + 1) * dim + 2 * ox], p[(2 * oy + 1) * dim + 2 * ox + 1]);
            s[cur ^ 1][oy * od + ox] = v;
            const uint32_t mw = a.width >> mip;
            a.base[a.mipOffset[mip] + (uint64_t)(ty * od + oy) * mw + tx * od + ox] = (_Float16)v;
        }
        __syncthreads();
        cur ^= 1;
        dim = od;
    }
}

// One mip from the one below it, a thread per texel, the tail kernel's clamped 2x2 rule: the general path for chains the
// 64x64 tiling does not fit (a dimension below 64 while the mip is still larger than 64x64 texels, e.g. the 1024x32
// HZB of a 2048x64 render).  One launch per mip until the tail kernel can take over.
template <bool MAX>
__global__ __launch_bounds__(256) void spdMipKernel(SpdArgs a, uint32_t mip)
{
    const uint32_t pw = (a.width >> (mip - 1)) ? (a.width >> (mip - 1)) : 1u, ph = (a.height >> (mip - 1)) ? (a.height >> (mip - 1)) : 1u;
    const uint32_t mw = (a.width >> mip) ? (a.width >> mip) : 1u, mh = (a.height >> mip) ? (a.height >> mip) : 1u;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= mw * mh) return;
    const uint32_t x = i % mw, y = i / mw;
    const uint32_t x0 = min(2 * x, pw - 1), x1 = min(2 * x + 1, pw - 1);
    const uint32_t y0 = min(2 * y, ph - 1), y1 = min(2 * y + 1, ph - 1);
    const _Float16* p = a.base + a.mipOffset[mip - 1];
    a.base[a.mipOffset[mip] + i] = (_Float16)red4<MAX>((float)p[y0 * pw + x0], (float)p[y0 * pw + x1], (float)p[y1 * pw + x0], (float)p[y1 * pw + x1]);
}

// What recordMinMaxDownsample leaves for a recordSPD that follows it immediately (trhip_cmdlist_t::peephole).
struct MinMaxNote { const float* depth; uint32_t W, H; _Float16* out; uint32_t ow, oh; bool mx; };

// Tail pass: one workgroup, starting from mip `first` (at most 64x64 texels), produces every
// remaining mip with the clamped 2x2 rule (non-square chains degenerate to 2x1 / 1x2 blocks).
template <bool MAX>
__global__ __launch_bounds__(1024) void spdTailKernel(SpdArgs a, uint32_t first)
{
    __shared__ float s[2][64 * 64];
    const uint32_t tid = threadIdx.x;
    uint32_t pw = (a.width >> first) ? (a.width >> first) : 1u, ph = (a.height >> first) ? (a.height >> first) : 1u;
    for (uint32_t i = tid; i < pw * ph; i += 1024) s[0][i] = (float)a.base[a.mipOffset[first] + i];
    __syncthreads();
    uint32_t cur = 0;
    for (uint32_t mip = first + 1; mip < a.mips; ++mip) {
        const uint32_t mw = (a.width >> mip) ? (a.width >> mip) : 1u, mh = (a.height >> mip) ? (a.height >> mip) : 1u;
        const float* p = s[cur];
        for (uint32_t i = tid; i < mw * mh; i += 1024) {
            const uint32_t x = i % mw, y = i / mw;
            const uint32_t x0 = min(2 * x, pw - 1), x1 = min(2 * x + 1, pw - 1);
            const uint32_t y0 = min(2 * y, ph - 1), y1 = min(2 * y + 1, ph - 1);
            const float v = red4<MAX>(p[y0 * pw + x0], p[y0 * pw + x1], p[y1 * pw + x0], p[y1 * pw + x1]);
            s[cur ^ 1][i] = v;
            a.base[a.mipOffset[mip] + i] = (_Float16)v;
        }
        __syncthreads();
        cur ^= 1;
        pw = mw; ph = mh;
    }
}

__global__ __launch_bounds__(256) void hzbQuadBuildKernel(trhip::QuadArgs a)
{
    __shared__ float s_t[9][trhip::kQuadStripCols + 1];
    trhip::hzbQuadStrip(a, blockIdx.x, s_t);
}


int recordMinMaxDownsample(trhip::DispatchCtx& ctx)
{
    // BasePassRenderers.cpp:515-536
    const MinMaxDownsampleConsts* k = (const MinMaxDownsampleConsts*)ctx.constants(0, sizeof(MinMaxDownsampleConsts));
    TRHIP_REQUIRE(k, "%s: push constants (MinMaxDownsampleConsts, 12 bytes) missing", ctx.shaderName);
    trhip_texture_t* src = ctx.texture(TRHIP_BIND_TEXTURE_SRV, 0);
    uint32_t mip = 0;
    trhip_texture_t* dst = ctx.texture(TRHIP_BIND_TEXTURE_UAV, 0, &mip);
    TRHIP_REQUIRE(src && dst, "%s: needs Texture_SRV t0 (depth) and Texture_UAV u0 (HZB)", ctx.shaderName);
    TRHIP_REQUIRE(src->format == TRHIP_FORMAT_R32_FLOAT, "%s: depth source must be R32_FLOAT", ctx.shaderName);
    TRHIP_REQUIRE(dst->format == TRHIP_FORMAT_R16_FLOAT, "%s: HZB must be R16_FLOAT", ctx.shaderName);
    TRHIP_REQUIRE(k->m_OutputDimensions.x == dst->mipW(mip) && k->m_OutputDimensions.y == dst->mipH(mip),
                  "%s: m_OutputDimensions %ux%u does not match HZB mip %u (%ux%u)", ctx.shaderName,
                  k->m_OutputDimensions.x, k->m_OutputDimensions.y, mip, dst->mipW(mip), dst->mipH(mip));
    TRHIP_REQUIRE(!ctx.indirect, "%s: dispatched directly", ctx.shaderName);
    TRHIP_REQUIRE(ctx.gx >= (dst->mipW(mip) + 7) / 8 && ctx.gy >= (dst->mipH(mip) + 7) / 8,
                  "%s: dispatch %ux%u groups of 8x8 does not cover %ux%u (BasePassRenderers.cpp:533)", ctx.shaderName,
                  ctx.gx, ctx.gy, dst->mipW(mip), dst->mipH(mip));
    const float* depth = (const float*)src->mipPtr(0);
    _Float16* out = (_Float16*)dst->mipPtr(mip);
    const uint32_t W = src->width, H = src->height, ow = dst->mipW(mip), oh = dst->mipH(mip);
    const bool mx = k->m_bDownsampleMax != 0;
    ctx.emit("main", [=](hipStream_t s) {
        dim3 grid((ow + 31) / 32, (oh + 7) / 8);
        if (mx) TRHIP_LAUNCH(minMaxDownsampleKernel<true>, grid, dim3(256), 0, s, depth, W, H, out, ow, oh);
        else TRHIP_LAUNCH(minMaxDownsampleKernel<false>, grid, dim3(256), 0, s, depth, W, H, out, ow, oh);
        return trhip::launchStatus("minMaxDownsampleKernel"); });
    ctx.cl->peephole = { ctx.cl->ops.size() - 1, "minmaxdownsample", std::make_shared<MinMaxNote>(MinMaxNote{ depth, W, H, out, ow, oh, mx }) };
    return TRHIP_OK;
}

int recordSPD(trhip::DispatchCtx& ctx)
{
    // FFXHelpers.cpp:66-114: u2 = HZB mip 0 (rw_input_downsample_src_mips[0]), u3.. = mips 1..N-1.
    const SPDConstants* k = (const SPDConstants*)ctx.constants(0, sizeof(SPDConstants));
    TRHIP_REQUIRE(k, "%s: push constants (SPDConstants, 32 bytes) missing", ctx.shaderName);
    uint32_t mip0 = 0;
    trhip_texture_t* tex = ctx.texture(TRHIP_BIND_TEXTURE_UAV, 2, &mip0);
    TRHIP_REQUIRE(tex, "%s: needs Texture_UAV u2 = HZB mip 0 (FFXHelpers.cpp:71)", ctx.shaderName);
    TRHIP_REQUIRE(mip0 == 0, "%s: u2 must be mip 0", ctx.shaderName);
    TRHIP_REQUIRE(tex->format == TRHIP_FORMAT_R16_FLOAT, "%s: HZB must be R16_FLOAT", ctx.shaderName);
    TRHIP_REQUIRE(k->mips == tex->mips - 1, "%s: SPDConstants.mips %u != HZB mips-1 %u (FFXHelpers.cpp:63-64)", ctx.shaderName, k->mips, tex->mips - 1);
    for (uint32_t i = 1; i < tex->mips; ++i) {
        uint32_t m = 0;
        trhip_texture_t* t = ctx.texture(TRHIP_BIND_TEXTURE_UAV, 2 + i, &m);
        TRHIP_REQUIRE(t == tex && m == i, "%s: UAV u%u must be HZB mip %u (FFXHelpers.cpp:76-81)", ctx.shaderName, 2 + i, i);
    }
    TRHIP_REQUIRE(!ctx.indirect, "%s: dispatched directly", ctx.shaderName);
    // u0 = the SPD global atomic counter (FFXHelpers.cpp:69): bound for interface fidelity; these kernels order the
    // tail after the tiles by a second launch and never touch it, so it stays as cleared.
    if (trhip_buffer_t* counter = ctx.buffer(TRHIP_BIND_STRUCTURED_UAV, 0)) ctx.cl->forgetUse(counter->ptr, ctx.cl->ops.size());
    const bool mx = ctx.variant == 2;
    SpdArgs a;
    memset(&a, 0, sizeof a);
    a.base = (_Float16*)tex->ptr;
    a.width = tex->width; a.height = tex->height; a.mips = tex->mips;
    for (uint32_t i = 0; i < tex->mips; ++i) a.mipOffset[i] = (uint32_t)(tex->mipOffset[i] / 2);
    if (tex->mips <= 1) return TRHIP_OK;

    uint32_t first = 0;
    const bool tiled = (tex->width % 64 == 0) && (tex->height % 64 == 0);
    // The reference's GenerateHZB (BasePassRenderers.cpp:505-542) is minmaxdownsample immediately followed by SPD on
    // the same texture: when exactly that was recorded, both become one launch in the first command's place.
    const trhip_cmdlist_t::Peephole ph = ctx.cl->peephole;
    const MinMaxNote* note = (ph.kind && !strcmp(ph.kind, "minmaxdownsample") && ph.op == ctx.cl->ops.size() - 1 && ph.op != SIZE_MAX) ? (const MinMaxNote*)ph.data.get() : nullptr;
    const bool fuse = tiled && note && note->out == a.base + a.mipOffset[0] && note->ow == tex->width && note->oh == tex->height && note->mx == mx
                      && ctx.cl->ops.back().lane == 0;
    if (fuse) {
        const MinMaxNote n = *note;
        const uint32_t lastMip = tex->mips - 1 < 6 ? tex->mips - 1 : 6;
        ctx.cl->ops.pop_back();
        ctx.emit("depth_tile", [a, n, lastMip, mx](hipStream_t s) {
            dim3 grid(a.width / 64, a.height / 64);
            if (mx) TRHIP_LAUNCH(hzbDepthTileKernel<true>, grid, dim3(256), 0, s, n.depth, n.W, n.H, a, lastMip);
            else TRHIP_LAUNCH(hzbDepthTileKernel<false>, grid, dim3(256), 0, s, n.depth, n.W, n.H, a, lastMip);
            return trhip::launchStatus("hzbDepthTileKernel"); });
        first = lastMip;
    } else if (tiled) {
        const uint32_t lastMip = tex->mips - 1 < 6 ? tex->mips - 1 : 6;
        ctx.emit("tile", [a, lastMip, mx](hipStream_t s) {
            dim3 grid(a.width / 64, a.height / 64);
            if (mx) TRHIP_LAUNCH(spdTileKernel<true>, grid, dim3(256), 0, s, a, lastMip);
            else TRHIP_LAUNCH(spdTileKernel<false>, grid, dim3(256), 0, s, a, lastMip);
            return trhip::launchStatus("spdTileKernel"); });
        first = lastMip;
    }
    while ((uint64_t)tex->mipW(first) * tex->mipH(first) > 64 * 64 && first + 1 < tex->mips) {     // untiled chain: mip by mip
        const uint32_t mip = first + 1;
        const uint32_t texels = tex->mipW(mip) * tex->mipH(mip);
        ctx.emit("mip", [a, mip, texels, mx](hipStream_t s) {
            if (mx) TRHIP_LAUNCH(spdMipKernel<true>, dim3((texels + 255u) / 256u), dim3(256), 0, s, a, mip);
            else TRHIP_LAUNCH(spdMipKernel<false>, dim3((texels + 255u) / 256u), dim3(256), 0, s, a, mip);
            return trhip::launchStatus("spdMipKernel"); });
        first = mip;
    }
    if (first + 1 < tex->mips) {
        ctx.emit("tail", [a, first, mx](hipStream_t s) {
            if (mx) TRHIP_LAUNCH(spdTailKernel<true>, dim3(1), dim3(1024), 0, s, a, first);
            else TRHIP_LAUNCH(spdTailKernel<false>, dim3(1), dim3(1024), 0, s, a, first);
            return trhip::launchStatus("spdTailKernel"); });
    }
    return TRHIP_OK;
}

} // namespace

namespace trhip
{

int hzbQuadEnsure(trhip_texture_t* tex)
{
    TRHIP_REQUIRE(tex && tex->ptr && tex->format == TRHIP_FORMAT_R16_FLOAT && tex->mips <= 16, "footprint-min table: needs a bound R16_FLOAT texture");
    uint64_t total = 0;
    for (uint32_t k = 0; k < tex->mips; ++k) {
        tex->quadOffset[k] = (uint32_t)total;
        total += (uint64_t)((tex->mipW(k) >> 3) + 1) * ((tex->mipH(k) >> 3) + 1) * 64u;      // 8 x 8 blocks, see hzbQuadBuildKernel
    }
    TRHIP_REQUIRE(total < (1ull << 31), "footprint-min table: HZB %ux%u too large", tex->width, tex->height);
    tex->quadTotal = (uint32_t)total;
    if (tex->quadBytes < total * 2) {
        TRHIP_HIP(hipSetDevice(tex->dev->index));
        if (tex->quad) { int rc = tex->dev->syncAll(); if (rc != TRHIP_OK) return rc; (void)hipFree(tex->quad); tex->quad = nullptr; tex->quadBytes = 0; }
        TRHIP_HIP(hipMalloc(&tex->quad, (size_t)total * 2));
        tex->quadBytes = total * 2;
        tex->quadBuiltVersion = 0;
    }
    return TRHIP_OK;
}

int hzbQuadLaunchBuild(trhip_texture_t* tex, hipStream_t s)
{
    const uint64_t v = tex->version;                   // called while commands are submitted: every earlier write is counted
    if (tex->quadBuiltVersion == v) return TRHIP_OK;   // nothing wrote the HZB since the last build
    const QuadArgs a = quadArgs(tex);
    TRHIP_LAUNCH(hzbQuadBuildKernel, dim3(a.firstStrip[a.mips]), dim3(256), 0, s, a);
    tex->quadBuiltVersion = v;
    return launchStatus("hzbQuadBuildKernel");
}

int hzbQuadEmitBuild(const DispatchCtx& ctx, trhip_texture_t* tex)
{
    int rc = hzbQuadEnsure(tex);
    if (rc != TRHIP_OK) return rc;
    ctx.emitSide("footprint_min", [tex](hipStream_t s) { return hzbQuadLaunchBuild(tex, s); }, { { tex->ptr, false }, { tex->quad, true } });
    return TRHIP_OK;
}

} // namespace trhip

namespace
{

trhip::ShaderRegistrar r0("minmaxdownsample_CS_Main", recordMinMaxDownsample, 0);
trhip::ShaderRegistrar r1("ffx_spd_downsample_pass_CS FFX_SPD_OPTION_DOWNSAMPLE_FILTER=1", recordSPD, 1);
trhip::ShaderRegistrar r2("ffx_spd_downsample_pass_CS FFX_SPD_OPTION_DOWNSAMPLE_FILTER=2", recordSPD, 2);

} // namespace
