// cull_math.hip.h -- device-side arithmetic of the visibility tests (gfx950).
//
// Follows the reference HLSL (source/shaders/culling.hlsli, toyrenderer_common.hlsli,
// basepass.hlsl:90-108, gpuculling.hlsl:39-57) under the arithmetic convention of DESIGN.md
// "Arithmetic": IEEE binary32, no implicit contraction (-ffp-contract=off on this translation
// unit), matrix and dot products as explicit v_fma_f32 chains, correctly rounded '/' and sqrt
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), floor(log2) by exponent extraction,
// fp16 HZB texels.  Results are compared bit for bit with the CPU oracle by the -m gpu tests.
#pragma once

#include <hip/hip_runtime.h>

#include "ShaderInterop.h"

namespace cm
{

struct F3 { float x, y, z; };

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float min_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float max_(float a, float b) { return __builtin_fmaxf(a, b); }
#ifdef TR_EXPERIMENT_FAST_MATH   // timing experiment only (results are NOT bit-exact): approximate sqrt / division
__device__ __forceinline__ float sqrt_(float a) { return __builtin_amdgcn_sqrtf(a); }
__device__ __forceinline__ float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#else
__device__ __forceinline__ float sqrt_(float a) { return __builtin_sqrtf(a); }
__device__ __forceinline__ float div_(float a, float b) { return a / b; }
#endif
__device__ __forceinline__ float clamp_(float x, float lo, float hi) { return min_(max_(x, lo), hi); }

// Lane masks.  A `bool` of the compiler lives in an SGPR pair too, but the moment its VALUE is wanted (a ballot that is
// stored or combined, not just branched on) hipcc materialises it per lane (v_cndmask 0/1) and compares again -- two
// vector instructions per ballot, one of them a compare (4.7 SIMD cycles: profiles/r3/valu_rate.txt).  The hot loop
// therefore takes its predicates straight from the vector compares: one bit per lane, 0 for lanes outside EXEC, combined
// with scalar instructions.  Every wave of the kernels that use these is full (EXEC = all ones), so `~m` is "not".
typedef unsigned long long lmask;
__device__ __forceinline__ lmask mLt(float a, float b) { lmask m; asm("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }   // a < b (false for NaN)
__device__ __forceinline__ lmask mGe(float a, float b) { lmask m; asm("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }   // a >= b (false for NaN)
__device__ __forceinline__ lmask mAbsGt(float a, float b) { lmask m; asm("v_cmp_gt_f32_e64 %0, |%1|, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }   // |a| > b (false for NaN)
__device__ __forceinline__ lmask mLe(float a, float b) { lmask m; asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }   // a <= b (false for NaN)
__device__ __forceinline__ lmask mLtU(uint32_t a, uint32_t b) { lmask m; asm("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ lmask mLeU(uint32_t a, uint32_t b) { lmask m; asm("v_cmp_le_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ lmask mGeU(uint32_t a, uint32_t b) { lmask m; asm("v_cmp_ge_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ lmask mAbsLe(float a, float b) { lmask m; asm("v_cmp_le_f32_e64 %0, |%1|, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }   // |a| <= b (false for NaN)
__device__ __forceinline__ lmask mNotGt(float a, float b) { lmask m; asm("v_cmp_ngt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }   // !(a > b) (true for NaN)

// Two independent fp32 values in one 64-bit register pair: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 process both
// in one issue slot (the hot kernel is VALU-issue bound).  Every packed operation below is the same IEEE operation
// on each component as its scalar counterpart, so results do not change.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat2(float x) { return v2f{ x, x }; }

// Correctly rounded square roots of two values.  For x in [2^-96, FLT_MAX] the 8-operation sequence below
// (v_rsq_f32, one coupled Newton step on (sqrt, 1/(2 sqrt)), one residual correction; 6 of the 8 packed) returns
// exactly the IEEE result -- verified EXHAUSTIVELY on gfx950 against the compiler's expansion for all 1 879 048 192
// floats of that range (tools/sqrt_exhaustive.hip; below 2^-96 it does not hold, which is why the compiler's 16-
// operation expansion rescales).  Anything outside the range anywhere in the wave (zero, tiny, negative, inf, NaN)
// sends the whole wave through the compiler's sqrt.
__device__ __forceinline__ v2f sqrt2(v2f x)
{
#ifdef TR_EXPERIMENT_FAST_MATH
    return v2f{ __builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y) };
#else
    const bool inRange = (int)((__float_as_uint(x.x) - 0x0F800000u) < 0x70000000u) & (int)((__float_as_uint(x.y) - 0x0F800000u) < 0x70000000u);
    if (__builtin_expect(__ballot(!inRange) != 0ull, 0)) return v2f{ __builtin_sqrtf(x.x), __builtin_sqrtf(x.y) };
    const v2f y = { __builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y) };
    const v2f g0 = x * y, h0 = y * splat2(0.5f);
    const v2f r0 = fma2(-h0, g0, splat2(0.5f));
    const v2f g1 = fma2(g0, r0, g0), h1 = fma2(h0, r0, h0);
    const v2f d1 = fma2(-g1, g1, x);
    return fma2(d1, h1, g1);
#endif
}

// Correctly rounded a / b per component (== the compiler's IEEE fdiv expansion: v_div_scale, v_rcp, Newton steps,
// v_div_fmas, v_div_fixup) with the six fma / mul steps of the two divisions issued as packed instructions.
__device__ __forceinline__ v2f div2(v2f n, v2f d)
{
#ifdef TR_EXPERIMENT_FAST_MATH
    return v2f{ n.x * __builtin_amdgcn_rcpf(d.x), n.y * __builtin_amdgcn_rcpf(d.y) };
#else
    bool vccX, vccY, unused;
    const v2f ds = { __builtin_amdgcn_div_scalef(n.x, d.x, false, &unused), __builtin_amdgcn_div_scalef(n.y, d.y, false, &unused) };
    const v2f ns = { __builtin_amdgcn_div_scalef(n.x, d.x, true, &vccX), __builtin_amdgcn_div_scalef(n.y, d.y, true, &vccY) };
    const v2f r = { __builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y) };
    const v2f e0 = fma2(-ds, r, splat2(1.0f));
    const v2f r1 = fma2(e0, r, r);
    const v2f q = ns * r1;
    const v2f e1 = fma2(-ds, q, ns);
    const v2f q1 = fma2(e1, r1, q);
    const v2f e2 = fma2(-ds, q1, ns);
    return v2f{ __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e2.x, r1.x, q1.x, vccX), d.x, n.x),
                __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e2.y, r1.y, q1.y, vccY), d.y, n.y) };
#endif
}

__device__ __forceinline__ float dot3(F3 a, F3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }

__device__ __forceinline__ F3 cross3(F3 a, F3 b)
{
    return { fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)) };
}

// Rows 0..3 (xyz) of a row-major 4x4: everything mul(float4(p,1), M).xyz needs.
struct M43 { F3 r0, r1, r2, r3; };

// mul(float4(p,1), M).xyz
__device__ __forceinline__ F3 mulPoint(F3 p, const M43& m)
{
    return { fma_(p.z, m.r2.x, fma_(p.y, m.r1.x, p.x * m.r0.x)) + m.r3.x,
             fma_(p.z, m.r2.y, fma_(p.y, m.r1.y, p.x * m.r0.y)) + m.r3.y,
             fma_(p.z, m.r2.z, fma_(p.y, m.r1.z, p.x * m.r0.z)) + m.r3.z };
}

// mul(float3, float3x3(r0,r1,r2))
__device__ __forceinline__ F3 mulVec(F3 v, F3 r0, F3 r1, F3 r2)
{
    return { fma_(v.z, r2.x, fma_(v.y, r1.x, v.x * r0.x)),
             fma_(v.z, r2.y, fma_(v.y, r1.y, v.x * r0.y)),
             fma_(v.z, r2.z, fma_(v.y, r1.z, v.x * r0.z)) };
}

// gpuculling.hlsl:118-119 / basepass.hlsl:68-69: view transform, then z *= -1
__device__ __forceinline__ F3 toView(F3 p, const M43& v)
{
    F3 o = mulPoint(p, v);
    o.z = -o.z;
    return o;
}

// The same products with the x and y columns issued as packed pairs (per component the identical fma chain).
struct M43P { v2f r0, r1, r2, r3; float z0, z1, z2, z3; };   // rows 0..3: (x, y) pairs + the z column
struct M33P { v2f r0, r1, r2; float z0, z1, z2; };

__device__ __forceinline__ M43P packM43(const M43& m)
{
    return { { m.r0.x, m.r0.y }, { m.r1.x, m.r1.y }, { m.r2.x, m.r2.y }, { m.r3.x, m.r3.y }, m.r0.z, m.r1.z, m.r2.z, m.r3.z };
}
__device__ __forceinline__ M33P rot(const M43P& m) { return { m.r0, m.r1, m.r2, m.z0, m.z1, m.z2 }; }

__device__ __forceinline__ F3 mulPointP(F3 p, const M43P& m)          // == mulPoint
{
    const v2f xy = fma2(splat2(p.z), m.r2, fma2(splat2(p.y), m.r1, splat2(p.x) * m.r0)) + m.r3;
    return { xy.x, xy.y, fma_(p.z, m.z2, fma_(p.y, m.z1, p.x * m.z0)) + m.z3 };
}
__device__ __forceinline__ F3 mulVecP(F3 v, const M33P& m)            // == mulVec
{
    const v2f xy = fma2(splat2(v.z), m.r2, fma2(splat2(v.y), m.r1, splat2(v.x) * m.r0));
    return { xy.x, xy.y, fma_(v.z, m.z2, fma_(v.y, m.z1, v.x * m.z0)) };
}
__device__ __forceinline__ F3 toViewP(F3 p, const M43P& v)            // == toView
{
    F3 o = mulPointP(p, v);
    o.z = -o.z;
    return o;
}

// toyrenderer_common.hlsli:134-140
__device__ __forceinline__ float maxScale(F3 r0, F3 r1, F3 r2)
{
    return sqrt_(max_(max_(dot3(r0, r0), dot3(r1, r1)), dot3(r2, r2)));
}

// culling.hlsli:6-21 (Q7); returns true = visible
__device__ __forceinline__ bool frustumVisible(F3 c, float r, float fx, float fy, float fz, float fw)
{
    bool a = fma_(c.z, fy, __builtin_fabsf(c.x) * fx) < r;
    bool b = fma_(c.z, fw, __builtin_fabsf(c.y) * fz) < r;
    return a & b;
}

__device__ __forceinline__ lmask frustumVisibleM(F3 c, float r, float fx, float fy, float fz, float fw)   // == frustumVisible, as a lane mask
{
    return mLt(fma_(c.z, fy, __builtin_fabsf(c.x) * fx), r) & mLt(fma_(c.z, fw, __builtin_fabsf(c.y) * fz), r);
}

// HZB (R16_FLOAT mip chain) as the kernels see it.
struct Hzb
{
    const _Float16* base;   // mip 0
    uint32_t width, height, mips;
    uint32_t mipOffset[16]; // in texels, relative to base
};

// floor(log2(max(w,h))) clamped by SampleLevel to [0,mips-1]; culling.hlsli:75 (Q6)
__device__ __forceinline__ int hzbLevel(float width, float height, uint32_t mips)
{
    float m = max_(width, height);
    if (!(m >= 1.0f)) return 0;
    int e = (int)((__float_as_uint(m) >> 23) & 0xFFu) - 127;
    int last = (int)mips - 1;
    return e > last ? last : e;
}

// SampleLevel with the linear-clamp MIN-reduction sampler (culling.hlsli:78,
// CommonResources.cpp:276-287,298): min over the bilinear footprint texels of non-zero weight.
__device__ __forceinline__ float sampleHzbMin(const Hzb& h, float u, float v, int mip)
{
    uint32_t mw = (h.width >> mip) ? (h.width >> mip) : 1u;
    uint32_t mh = (h.height >> mip) ? (h.height >> mip) : 1u;
    const _Float16* t = h.base + h.mipOffset[mip];
    float fx = fma_(u, (float)mw, -0.5f);
    float fy = fma_(v, (float)mh, -0.5f);
    float flx = __builtin_floorf(fx), fly = __builtin_floorf(fy);
    int x0 = (int)flx, y0 = (int)fly;
    bool wx1 = (fx - flx) > 0.0f, wy1 = (fy - fly) > 0.0f;
    int xm = (int)mw - 1, ym = (int)mh - 1;
    int x1 = min(max(x0 + 1, 0), xm), y1 = min(max(y0 + 1, 0), ym);
    x0 = min(max(x0, 0), xm);
    y0 = min(max(y0, 0), ym);
    // a texel of zero weight is replaced by the (always taken) texel 00: same minimum
    x1 = wx1 ? x1 : x0;
    y1 = wy1 ? y1 : y0;
    float d00 = (float)t[(uint32_t)y0 * mw + (uint32_t)x0];
    float d01 = (float)t[(uint32_t)y0 * mw + (uint32_t)x1];
    float d10 = (float)t[(uint32_t)y1 * mw + (uint32_t)x0];
    float d11 = (float)t[(uint32_t)y1 * mw + (uint32_t)x1];
    return min_(min_(min_(d00, d01), d10), d11);
}

// culling.hlsli:36-82 split in two so that a kernel can issue the four texel loads, do unrelated
// ALU work (the cone test) while they are in flight, and only then compare.  Evaluated for every
// lane (no branch): the near-plane accept (:48-49) is carried as a flag, and NaN/inf from a sphere
// that straddles the camera plane are absorbed by the clamps, so the addresses are always valid.
struct OccSample
{
    bool accept;              // sphere intersects the near plane -> visible (:48-49)
    float depthSphere;        // :79
    uint32_t i0, i1;          // texel index (relative to Hzb::base) of the footprint's left texel in row y0 / y1
    bool pair;                // the footprint has a second column (x0 + 1)
};

// One 32-bit load fetches the two horizontally adjacent texels of a footprint row (the address is
// only 2-byte aligned: gfx950 runs in unaligned-access mode).  The high half is texel x0 + 1; it is
// ignored when the footprint has a single column (at the right edge it may belong to the next row or
// to the mip's alignment padding, which is inside the allocation).
__device__ __forceinline__ uint32_t loadTexelPair(const _Float16* base, uint32_t idx)
{
    uint32_t v;
    __builtin_memcpy(&v, base + idx, 4);
    return v;
}
__device__ __forceinline__ float texelLo(uint32_t v) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(v & 0xFFFFu)); }
__device__ __forceinline__ float texelHi(uint32_t v) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(v >> 16)); }

// mipOff: the mip offset table (texels); the hot kernel passes a copy parked in LDS so that the
// per-lane lookup is one ds_read instead of a select chain over kernel arguments.
__device__ __forceinline__ OccSample occlusionPrepare(F3 c, float r, float nearPlane, float P00, float P11, const Hzb& h, const uint32_t* mipOff)
{
    OccSample o;
    o.accept = (c.z - nearPlane) < r;                                // :48-49
    float crx = c.x * r, cry = c.y * r, crz = c.z * r;               // :53
    float czr2 = fma_(c.z, c.z, -(r * r));                           // :54
    float vx = sqrt_(fma_(c.x, c.x, czr2));                          // :56
    float minx = div_(fma_(vx, c.x, -crz), fma_(vx, c.z, crx));           // :57
    float maxx = div_(fma_(vx, c.x, crz), fma_(vx, c.z, -crx));           // :58
    float vy = sqrt_(fma_(c.y, c.y, czr2));                          // :60
    float miny = div_(fma_(vy, c.y, -crz), fma_(vy, c.z, cry));           // :61
    float maxy = div_(fma_(vy, c.y, crz), fma_(vy, c.z, -cry));           // :62
    float ax = clamp_(minx * P00, -1.0f, 1.0f);                      // :64-67
    float ay = clamp_(miny * P11, -1.0f, 1.0f);
    float az = clamp_(maxx * P00, -1.0f, 1.0f);
    float aw = clamp_(maxy * P11, -1.0f, 1.0f);
    ax = fma_(ax, 0.5f, 0.5f);                                       // :70-71 ClipXYToUV
    ay = fma_(ay, -0.5f, 0.5f);
    az = fma_(az, 0.5f, 0.5f);
    aw = fma_(aw, -0.5f, 0.5f);
    float width = (az - ax) * (float)h.width;                        // :73
    float height = (aw - ay) * (float)h.height;                      // :74
    int mip = hzbLevel(width, height, h.mips);                       // :75
    float u = (ax + az) * 0.5f, v = (ay + aw) * 0.5f;                // :78
    // SampleLevel footprint (see sampleHzbMin)
    uint32_t mw = (h.width >> mip) ? (h.width >> mip) : 1u;
    uint32_t mh = (h.height >> mip) ? (h.height >> mip) : 1u;
    float fx = fma_(u, (float)mw, -0.5f);
    float fy = fma_(v, (float)mh, -0.5f);
    float flx = __builtin_floorf(fx), fly = __builtin_floorf(fy);
    int x0 = (int)flx, y0 = (int)fly;
    bool wx1 = (fx - flx) > 0.0f, wy1 = (fy - fly) > 0.0f;
    int xm = (int)mw - 1, ym = (int)mh - 1;
    int x1 = min(max(x0 + 1, 0), xm), y1 = min(max(y0 + 1, 0), ym);
    x0 = min(max(x0, 0), xm);
    y0 = min(max(y0, 0), ym);
    x1 = wx1 ? x1 : x0;
    y1 = wy1 ? y1 : y0;
    uint32_t base = mipOff[mip];
    o.i0 = base + (uint32_t)y0 * mw + (uint32_t)x0;
    o.i1 = base + (uint32_t)y1 * mw + (uint32_t)x0;
    o.pair = x1 != x0;                                               // then x1 == x0 + 1
    o.depthSphere = div_(nearPlane, c.z - r);                           // :79
    return o;
}

__device__ __forceinline__ bool occlusionResolve(const OccSample& o, uint32_t row0, uint32_t row1)
{
    float d00 = texelLo(row0), d10 = texelLo(row1);
    float d01 = o.pair ? texelHi(row0) : d00, d11 = o.pair ? texelHi(row1) : d10;
    float depth = min_(min_(min_(d00, d01), d10), d11);
    return o.accept | (o.depthSphere >= depth);                      // :81
}

// Footprint-min table of an HZB (built by k_hzb.hip next to the mip chain): entry (x0+1, y0+1) of mip k holds
// the min over the 2x2 edge-clamped footprint whose origin floor(uv*dim - 0.5) is (x0, y0), x0 in [-1, w-1].
// Entries are stored in 8 x 8 blocks (one cache line each), ((w_k >> 3) + 1) blocks per block row.
struct HzbQuad
{
    const _Float16* base;
    uint32_t total;
    uint32_t offset[16];    // first entry of mip k
};
__device__ __forceinline__ uint32_t quadIndex(uint32_t offset, uint32_t blockRowStride /* 64 * blocks per row */, uint32_t X, uint32_t Y)
{
    // offset + ((Y >> 3) * blocksPerRow + (X >> 3)) * 64 + (Y & 7) * 8 + (X & 7)
    return offset + (__umul24(Y >> 3, blockRowStride) + ((Y << 3) & 56u)) + __umul24(X >> 3, 56u) + X;
}

// culling.hlsli:36-82 for the hot kernel (occTailQuad): same arithmetic as occlusionPrepare up to the footprint origin,
// then ONE table entry instead of up to four texels.  The table entry equals the footprint minimum exactly when both
// bilinear weights of both axes are non-zero; `slow` flags the other lookups (a fractional coordinate that is
// an exact integer), which the caller resolves with the texel path.
struct OccQuad
{
    lmask accept;             // :48-49 (lane mask)
    lmask slow;               // lanes whose lookup the table cannot serve (lane mask; 0 almost always)
    float depthSphere;        // :79
    uint32_t iq;              // table index (always in range: uv is clamped to [0,1] and NaN-free)
};

// culling.hlsli:36-82; returns true = visible
__device__ __forceinline__ bool occlusionVisible(F3 c, float r, float nearPlane, float P00, float P11, const Hzb& h)
{
    if ((c.z - nearPlane) < r) return true;                          // :48-49
    OccSample o = occlusionPrepare(c, r, nearPlane, P00, P11, h, h.mipOffset);
    return occlusionResolve(o, loadTexelPair(h.base, o.i0), loadTexelPair(h.base, o.i1));
}

// x / 255.0f for x in [0,255], correctly rounded (== IEEE division; verified exhaustively by
// tests/test_gpu_primitives.py): one Newton correction of x * RN(1/255).
__device__ __forceinline__ float u8Unorm(uint32_t x)
{
    const float r = 0x1.010102p-8f;          // RN(1/255)
    float xf = (float)x;
    float q = xf * r;
    float rem = fma_(-q, 255.0f, xf);
    return fma_(rem, r, q);
}

// basepass.hlsl:92-108; adj0..2 = MakeAdjugateMatrix(world) rows (toyrenderer_common.hlsli:124-132).
// Returns true = back-facing (ConeCull, culling.hlsli:84-87).
__device__ __forceinline__ bool coneBackfacing(uint32_t packed, F3 cv, float r, F3 adj0, F3 adj1, F3 adj2, const M43& view)
{
    float q0 = u8Unorm(packed & 0xFFu), q1 = u8Unorm((packed >> 8) & 0xFFu);
    float q2 = u8Unorm((packed >> 16) & 0xFFu), cutoff = u8Unorm(packed >> 24);
    F3 a = { fma_(q0, 2.0f, -1.0f), fma_(q1, 2.0f, -1.0f), fma_(q2, 2.0f, -1.0f) };
    F3 t = mulVec(a, adj0, adj1, adj2);
    float len = sqrt_(dot3(t, t));
    t = { div_(t.x, len), div_(t.y, len), div_(t.z, len) };                         // normalize = v / length
    F3 axis = mulVec(t, view.r0, view.r1, view.r2);
    axis.z = -axis.z;
    return dot3(cv, axis) >= fma_(cutoff, sqrt_(dot3(cv, cv)), r);
}

// == coneBackfacing, with the byte decode, the two 3x3 products and the normalisation divisions packed.  The third
// normalisation division shares its instruction slots with one unrelated division of the caller: extraQuotient =
// extraNum / extraDen (the occlusion test's nearPlane / (c.z - r), culling.hlsli:79).
__device__ __forceinline__ bool coneBackfacingP(uint32_t packed, F3 cv, float r, const M33P& adj, const M33P& viewRot,
                                                float extraNum, float extraDen, float* extraQuotient)
{
    const float rc = 0x1.010102p-8f;         // RN(1/255), see u8Unorm
    const v2f x01 = { (float)(packed & 0xFFu), (float)((packed >> 8) & 0xFFu) };
    const v2f x23 = { (float)((packed >> 16) & 0xFFu), (float)(packed >> 24) };
    const v2f q01i = x01 * rc, q23i = x23 * rc;
    const v2f q01 = fma2(fma2(-q01i, splat2(255.0f), x01), splat2(rc), q01i);      // (q0, q1)
    const v2f q23 = fma2(fma2(-q23i, splat2(255.0f), x23), splat2(rc), q23i);      // (q2, cutoff)
    const v2f a01 = fma2(q01, splat2(2.0f), splat2(-1.0f));
    const F3 a = { a01.x, a01.y, fma_(q23.x, 2.0f, -1.0f) };
    F3 t = mulVecP(a, adj);
    const v2f lens = sqrt2(v2f{ dot3(t, t), dot3(cv, cv) });                       // length(t), length(cv)
    const float len = lens.x;
    const v2f txy = div2(v2f{ t.x, t.y }, splat2(len));                            // normalize = v / length
    const v2f tzq = div2(v2f{ t.z, extraNum }, v2f{ len, extraDen });
    *extraQuotient = tzq.y;
    t = { txy.x, txy.y, tzq.x };
    F3 axis = mulVecP(t, viewRot);
    axis.z = -axis.z;
    return dot3(cv, axis) >= fma_(q23.y, lens.y, r);
}

// ------------------------------------------------------------------------------------------------------------------
// The square roots and divisions of ONE cull step (culling.hlsli:56-62,79 and the normalize of basepass.hlsl:103),
// evaluated together.  The arithmetic convention wants every '/' and sqrt correctly rounded (IEEE); the compiler's
// expansions of them (v_div_scale / v_div_fmas / v_div_fixup around a Newton iteration, a rescaling square root) spend
// about half of their instructions on operands near the ends of the exponent range.  When EVERY lane of the wave has
// all its operands in a comfortable middle range those instructions are no-ops by definition:
//   * v_div_scale returns its input and clears VCC unless an operand is zero / denormal / tiny (exponent <= 23), the
//     denominator's reciprocal or the quotient would be denormal, or the exponents are >= 96 apart (ISA manual);
//     with VCC = 0 v_div_fmas is a plain fma, and v_div_fixup only replaces the result for zero / inf / NaN operands or
//     an exponent difference beyond the format, otherwise it copies it;
//   * the 8-operation square root of sqrt2() is exact on [2^-96, FLT_MAX].
// So the FAST path below runs the same Newton iterations without the glue -- bit for bit the compiler's results -- and
// shares the refined reciprocal of `len` between the three components of the normalisation (it depends on the
// denominator only).  One wave-uniform branch sends a wave with any operand outside the range through the EXACT path,
// which is the code the kernel ran before (sqrt2 / div2: the compiler's full sequences).  Without the VCC traffic of
// v_div_scale / v_div_fmas the independent chains also interleave, which removes the wait states dependent packed-fp32
// instructions need on gfx950 (one s_nop each).
//
// Range argument (SAFE): the four radicands in [2^-96, 2^60], |r| <= 2^30 (with c.x^2 + c.z^2 - r^2 <= 2^60 that bounds
// |c.x|, |c.y|, |c.z| by 2^30.5; without the occlusion test length(c)^2 <= 2^60 does), every numerator and denominator
// at least 2^-30 in magnitude, nearPlane in [2^-20, 2^20].  Then all numerators / denominators are below 2^63, exponent
// differences stay under 96, no reciprocal or quotient is denormal and no numerator has an exponent <= 23.
struct StepQuot
{
    v2f mn, mx;               // (minx, miny), (maxx, maxy)            culling.hlsli:57-62
    F3 tn;                    // normalize(mul(coneAxis, adjugate))     basepass.hlsl:103
    float depthSphere;        // nearPlane / (c.z - r)                  culling.hlsli:79
    float lenC;               // length(c)                              culling.hlsli:86
    float cutoff;             // cone cutoff byte / 255                 basepass.hlsl:105
    bool coneExact;           // wave-uniform: tn and lenC are the correctly rounded values (else: within 5 ulp, see coneBack)
};

__device__ __forceinline__ v2f rcpRefined2(v2f d)                       // Fma1 of the fdiv expansion: depends on d only
{
    const v2f r = { __builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y) };
    const v2f e0 = fma2(-d, r, splat2(1.0f));
    return fma2(e0, r, r);
}
__device__ __forceinline__ v2f quotient2(v2f n, v2f d, v2f r1)          // the rest of the expansion, unscaled operands
{
    const v2f q = n * r1;
    const v2f e1 = fma2(-d, q, n);
    const v2f q1 = fma2(e1, r1, q);
    const v2f e2 = fma2(-d, q1, n);
    return fma2(e2, r1, q1);                                            // v_div_fmas with VCC = 0
}
__device__ __forceinline__ float rcpRefined1(float d)
{
    const float r = __builtin_amdgcn_rcpf(d);
    return fma_(fma_(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float quotient1(float n, float d, float r1)
{
    const float q = n * r1;
    const float q1 = fma_(fma_(-d, q, n), r1, q);
    return fma_(fma_(-d, q1, n), r1, q1);
}
__device__ __forceinline__ v2f sqrtSeq2(v2f x)                          // sqrt2()'s in-range sequence, unconditionally
{
    const v2f y = { __builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y) };
    const v2f g0 = x * y, h0 = y * splat2(0.5f);
    const v2f r0 = fma2(-h0, g0, splat2(0.5f));
    const v2f g1 = fma2(g0, r0, g0), h1 = fma2(h0, r0, h0);
    const v2f d1 = fma2(-g1, g1, x);
    return fma2(d1, h1, g1);
}
__device__ __forceinline__ float sqrtSeq1(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g0 = x * y, h0 = y * 0.5f;
    const float r0 = fma_(-h0, g0, 0.5f);
    const float g1 = fma_(g0, r0, g0), h1 = fma_(h0, r0, h0);
    return fma_(fma_(-g1, g1, x), h1, g1);
}
__device__ __forceinline__ float minAbs3(float a, float b, float c) { return min_(min_(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c)); }

// basepass.hlsl:92-99: cone bytes -> axis (xyz * 2 - 1) and cutoff; x / 255 exactly (see u8Unorm)
__device__ __forceinline__ F3 coneAxisCutoff(uint32_t packed, float* cutoff)
{
    const float rc = 0x1.010102p-8f;         // RN(1/255)
    const v2f x01 = { (float)(packed & 0xFFu), (float)((packed >> 8) & 0xFFu) };
    const v2f x23 = { (float)((packed >> 16) & 0xFFu), (float)(packed >> 24) };
    const v2f q01i = x01 * rc, q23i = x23 * rc;
    const v2f q01 = fma2(fma2(-q01i, splat2(255.0f), x01), splat2(rc), q01i);      // (q0, q1)
    const v2f q23 = fma2(fma2(-q23i, splat2(255.0f), x23), splat2(rc), q23i);      // (q2, cutoff)
    const v2f a01 = fma2(q01, splat2(2.0f), splat2(-1.0f));
    *cutoff = q23.y;
    return { a01.x, a01.y, fma_(q23.x, 2.0f, -1.0f) };
}

// The same through a 256-entry table in LDS, tab[x] = x / 255 computed with the arithmetic above; the axis components take their
// fma(q, 2, -1) from there (three plain fma: the same operation on the same operand as in coneAxisCutoff).  Four byte-indexed
// reads instead of 4 conversions + 6 packed operations per meshlet.  One dense float table: the byte-indexed ds_read_b32 of
// 64 lanes spread over all 32 LDS banks (round 3 kept {axis, cutoff} pairs: every read of a half used 16 banks --
// SQ_LDS_BANK_CONFLICT 28.0 M cycles per launch, 17.9 M with dense tables; profiles/r4/experiments.md), and 1 KB instead of 2
// (the deferred list needs the LDS: 5 workgroups per CU want <= 32 000 bytes each).
constexpr uint32_t kConeTabEntries = 256;
__device__ __forceinline__ float coneTableEntry(uint32_t i) { return u8Unorm(i & 0xFFu); }
__device__ __forceinline__ F3 coneAxisCutoffLds(uint32_t packed, float* cutoff, const float* tab)
{
    *cutoff = tab[packed >> 24];
    return { fma_(tab[packed & 0xFFu], 2.0f, -1.0f), fma_(tab[(packed >> 8) & 0xFFu], 2.0f, -1.0f), fma_(tab[(packed >> 16) & 0xFFu], 2.0f, -1.0f) };
}

#ifdef TR_COUNT_PATHS
#define TR_PATH_COUNT(exact) do { if ((threadIdx.x & 63u) == 0) atomicAdd(&g_pathCount[(exact) ? 1 : 0], 1ull); } while (0)
#else
#define TR_PATH_COUNT(exact) do {} while (0)
#endif

template <bool OCC, bool CONE, bool CONETAB = false>
__device__ __forceinline__ void stepQuotients(lmask active, F3 c, float r, uint32_t packed, const M33P& adj, float nearPlane, bool nearInRange, StepQuot& o,
                                              const float* coneTab = nullptr /* CONETAB: LDS, coneTableEntry(i), i < kConeTabEntries */)
{
    // active: the lane tests a meshlet.  The others (past the end of a record, records past the end of the list) run along
    // on whatever operands they hold, never reach an output and must not send the wave down the EXACT path.
    // ---- everything before the square roots: exact by construction, shared by both paths --------------------------
    const v2f cxy = { c.x, c.y };
    const float dz = c.z - r;                                          // culling.hlsli:79 denominator
    F3 t = { 1.0f, 1.0f, 1.0f };
    float st = 1.0f, sc = 1.0f;
    o.cutoff = 0.0f;
    if (CONE) {
        const F3 a = CONETAB ? coneAxisCutoffLds(packed, &o.cutoff, coneTab) : coneAxisCutoff(packed, &o.cutoff);
        t = mulVecP(a, adj);                                           // basepass.hlsl:103 mul(axis, adjugate)
        st = dot3(t, t);
        sc = dot3(c, c);                                               // culling.hlsli:86 length(center)^2
    }
    v2f cr = { 1.0f, 1.0f }, vArg = { 1.0f, 1.0f };
    float crz = 1.0f;
    const v2f czz = splat2(c.z);
    if (OCC) {
        cr = cxy * r;                                                  // :53 cr.xy
        crz = c.z * r;                                                 // :53 cr.z
        const float czr2 = fma_(c.z, c.z, -(r * r));                   // :54
        vArg = fma2(cxy, cxy, splat2(czr2));                           // :56, :60 radicands
    }
    // ---- optimistic: the fast square roots and what hangs on them ---------------------------------------------------
    v2f vv = { 1.0f, 1.0f }, lens = { 1.0f, 1.0f };
    if (OCC) vv = sqrtSeq2(vArg);                                      // :56, :60  vx, vy
    // 1 / length(t), 1 / length(c) to 1 ulp: all the FAST path needs of the two lengths (coneBack)
    const v2f rlen = CONE ? v2f{ __builtin_amdgcn_rsqf(st), __builtin_amdgcn_rsqf(sc) } : v2f{ 1.0f, 1.0f };
    v2f n1 = cxy, d1 = cxy, n2 = cxy, d2 = cxy;
    if (OCC) {
        n1 = fma2(vv, cxy, splat2(-crz)); d1 = fma2(vv, czz, cr);      // :57, :61
        n2 = fma2(vv, cxy, splat2(crz));  d2 = fma2(vv, czz, -cr);     // :58, :62
    }
    // ---- is every operand of this lane in the comfortable range? ---------------------------------------------------
    const uint32_t kLo = 0x0F800000u, kHi = 0x5D800000u;               // 2^-96, 2^60
    uint32_t uMin = 0x3F800000u, uMax = 0x3F800000u;                   // radicands as bit patterns: negative / NaN read as huge
    float mag = 1.0f;
    lmask safe = ~0ull;                                                // lane masks straight from the compares (no bool round trip through a VGPR)
    if (OCC) {
        uMin = min(__float_as_uint(vArg.x), __float_as_uint(vArg.y));
        uMax = max(max(__float_as_uint(vArg.x), __float_as_uint(vArg.y)), uMax);
        mag = min_(min_(minAbs3(n1.x, n1.y, n2.x), minAbs3(n2.y, d1.x, d1.y)), minAbs3(d2.x, d2.y, dz));
        safe = mGe(mag, 0x1p-30f) & mAbsLe(r, 0x1p30f);                // both false for NaN
    }
    if (CONE) {
        uMin = min(min(__float_as_uint(st), __float_as_uint(sc)), uMin);
        uMax = max(max(__float_as_uint(st), __float_as_uint(sc)), uMax);
    }
    safe &= mGeU(uMin, kLo) & mLeU(uMax, kHi);
    o.coneExact = false;
    if (__builtin_expect((active & ~safe) != 0ull || !nearInRange, 0)) {
        // ---- EXACT path (rare): the compiler's full square root / division sequences, as before ---------------------
        TR_PATH_COUNT(true);
        o.coneExact = true;
        if (OCC) {
            vv = sqrt2(vArg);
            o.mn = div2(fma2(vv, cxy, splat2(-crz)), fma2(vv, czz, cr));
            o.mx = div2(fma2(vv, cxy, splat2(crz)), fma2(vv, czz, -cr));
        }
        if (CONE) {
            lens = sqrt2(v2f{ st, sc });
            const v2f txy = div2(v2f{ t.x, t.y }, splat2(lens.x));      // normalize = v / length
            const v2f tzq = div2(v2f{ t.z, nearPlane }, v2f{ lens.x, dz });
            o.tn = { txy.x, txy.y, tzq.x };
            o.depthSphere = tzq.y;
            o.lenC = lens.y;
        } else {
            o.depthSphere = OCC ? div_(nearPlane, dz) : 0.0f;
        }
    } else {
        // ---- FAST path: the same Newton iterations on unscaled operands, the independent chains issued side by side.
        // (Stage by stage with scheduling barriers: left alone, the compiler emits one chain after the other and every
        // dependent packed instruction then costs a wait state.)
        TR_PATH_COUNT(false);
#define TR_STAGE() __builtin_amdgcn_sched_barrier(0x0094)              /* SALU, VMEM and DS may cross; VALU may not */
        if (OCC && CONE) {
            // the four projection quotients as two packed chains, near / (z - r) as a plain one; the cone's normalisation and
            // length(c) are NOT divided / rooted exactly here: t * (1 / length(t)) and c.c * (1 / length(c)) to a few ulp,
            // which decides the cone test for every lane that is not within 2^-18 of its boundary (coneBack)
            const v2f ra = { __builtin_amdgcn_rcpf(d1.x), __builtin_amdgcn_rcpf(d1.y) };
            const v2f rb = { __builtin_amdgcn_rcpf(d2.x), __builtin_amdgcn_rcpf(d2.y) };
            const float rc = __builtin_amdgcn_rcpf(dz);
            TR_STAGE();
            const v2f ea = fma2(-d1, ra, splat2(1.0f)), eb = fma2(-d2, rb, splat2(1.0f));
            const float ec = fma_(-dz, rc, 1.0f);
            TR_STAGE();
            const v2f r1 = fma2(ea, ra, ra), r2 = fma2(eb, rb, rb);
            const float r3 = fma_(ec, rc, rc);
            TR_STAGE();
            const v2f qa = n1 * r1, qb = n2 * r2;
            const float qc = nearPlane * r3;
            const v2f txy = v2f{ t.x, t.y } * splat2(rlen.x);
            TR_STAGE();
            const v2f fa = fma2(-d1, qa, n1), fb = fma2(-d2, qb, n2);
            const float fc = fma_(-dz, qc, nearPlane);
            const v2f tzl = v2f{ t.z, sc } * rlen;                     // t.z / length(t), length(c) = c.c / length(c)
            TR_STAGE();
            const v2f ga = fma2(fa, r1, qa), gb = fma2(fb, r2, qb);
            const float gc = fma_(fc, r3, qc);
            TR_STAGE();
            const v2f ha = fma2(-d1, ga, n1), hb = fma2(-d2, gb, n2);
            const float hc = fma_(-dz, gc, nearPlane);
            TR_STAGE();
            o.mn = fma2(ha, r1, ga); o.mx = fma2(hb, r2, gb);
            o.depthSphere = fma_(hc, r3, gc);
            TR_STAGE();
            o.tn = { txy.x, txy.y, tzl.x };
            o.lenC = tzl.y;
        } else if (OCC) {
            const v2f ra = { __builtin_amdgcn_rcpf(d1.x), __builtin_amdgcn_rcpf(d1.y) };
            const v2f rb = { __builtin_amdgcn_rcpf(d2.x), __builtin_amdgcn_rcpf(d2.y) };
            const float rc = __builtin_amdgcn_rcpf(dz);
            TR_STAGE();
            const v2f ea = fma2(-d1, ra, splat2(1.0f)), eb = fma2(-d2, rb, splat2(1.0f));
            const float ec = fma_(-dz, rc, 1.0f);
            TR_STAGE();
            const v2f r1 = fma2(ea, ra, ra), r2 = fma2(eb, rb, rb);
            const float r3 = fma_(ec, rc, rc);
            TR_STAGE();
            const v2f qa = n1 * r1, qb = n2 * r2;
            const float qc = nearPlane * r3;
            TR_STAGE();
            const v2f fa = fma2(-d1, qa, n1), fb = fma2(-d2, qb, n2);
            const float fc = fma_(-dz, qc, nearPlane);
            TR_STAGE();
            const v2f ga = fma2(fa, r1, qa), gb = fma2(fb, r2, qb);
            const float gc = fma_(fc, r3, qc);
            TR_STAGE();
            const v2f ha = fma2(-d1, ga, n1), hb = fma2(-d2, gb, n2);
            const float hc = fma_(-dz, gc, nearPlane);
            TR_STAGE();
            o.mn = fma2(ha, r1, ga); o.mx = fma2(hb, r2, gb);
            o.depthSphere = fma_(hc, r3, gc);
        } else if (CONE) {
            const v2f txy = v2f{ t.x, t.y } * splat2(rlen.x), tzl = v2f{ t.z, sc } * rlen;
            o.tn = { txy.x, txy.y, tzl.x };
            o.depthSphere = 0.0f;
            o.lenC = tzl.y;
        } else {
            o.depthSphere = 0.0f;
        }
#undef TR_STAGE
    }
}

// ------------------------------------------------------------------------------------------------------------------
// FILTERED PROJECTION (round 4; the footprint-table kernel of large passes).  culling.hlsli:53-78 turns the view-space sphere
// into the mip level and the footprint origin of ONE table lookup through two exact square roots, four exact quotients, a
// floor(log2) and two floors.  Everything the meshlet's visibility takes from that chain are three INTEGERS (level, x0, y0;
// the zero-weight flags are "fraction == 0").  The fast path below computes the chain approximately -- closed form, one
// v_rsq_f32 per axis, ONE v_rcp_f32 for both -- together with a proven bound on how far its real-valued intermediates can be
// from the reference's; a lane is SURE when no integer can differ inside that bound.  A lane that is not sure, and whose
// lookup still matters, is written to a list and re-evaluated with the exact sequences by a fix-up kernel behind the cull
// (k_basepass_as.hip: deferred list); the bits the cull stored for it are overwritten.  Nothing is decided approximately.
//
// Closed form.  With Z = c.z^2 - r^2, v = sqrt(c.x^2 + Z):   (v c.x -+ c.z r) / (v c.z +- c.x r) = (c.x c.z -+ v r) / Z
// (multiply numerator and denominator by (v c.z -+ c.x r); v^2 c.z^2 - c.x^2 r^2 = Z (c.x^2 + c.z^2)).  Same for y.
//
// Preconditions of a sure lane (projSure): 8 |r| <= c.z, |c.x| <= Bx c.z, |c.y| <= By c.z, c.z <= 2^30 -- and, wave-uniform,
// nearPlane in [2^-20, 2^20] (a lane that still matters has c.z - r >= nearPlane, so c.z >= 0.88 * 2^-20: no intermediate
// leaves the normal range).  Bx = 1.125 / P00 + 0.25 holds every sphere with 8 r <= c.z that touches the frustum.
//
// Error bound (u = 2^-24; rho = |r| / c.z <= 1/8; beta = |c.x| / c.z <= B; lengths in units of c.z; q = the quotient's real
// value; v_rsq_f32 / v_rcp_f32 within 1 ulp = 2 u):
//   Z  = fma(cz, cz, -RN(r r)) = Z* (1 + eZ), |eZ| <= (rho^2 / (1 - rho^2) + 1) u <= 1.016 u; Z* >= 63/64
//   X  = fma(cx, cx, Z)        = X* (1 + eX), |eX| <= 2.016 u                  (both are the REFERENCE's values: shared)
//   reference:  v = RN(sqrt X) = sqrt(X) (1 + d),  |d| <= u;     fast:  v' = RN(X rsq(X)) = sqrt(X) (1 + d'), |d'| <= 3 u
//   the reference's quotient, as a real function of its v, IS (cx cz - v r + t cx cz) / (Z* + t cz^2) with
//     t = (v^2 - X*) / (cx^2 + cz^2), |t| <= 2 (|eX| / 2 + u) = 4.02 u
//   so  q_ref - q_fast  =  - v r (d - d') / Z*  +  t cx cz / Z*  -  q t cz^2 / Z*  +  q eZ   (first order), i.e.
//     <= u (4 sqrt(B^2 + 1) rho + 4.02 B + (4.02 + 1.016) |q|) / (63/64)
//   rounding of the reference's remaining operations (two products, two fma, one division; d1 >= 0.85 v >= 0.85 beta):
//     <= u (rho / d1 + |q| beta rho / d1 + 3 |q|) <= u (0.15 + 3.15 |q|)
//   rounding of the fast path's (product, fma, v_rcp_f32, product):  <= u (beta / Z* + 4 |q|) <= u (1.016 B + 4 |q|)
//   |q_ref - q_fast| <= Cq u,  Cq = 0.51 sqrt(B^2 + 1) + 5.11 B + 0.15 + 12.3 |q|,  |q| <= 1 / P + (slack) wherever the
//   clamp to [-1, 1] (:64-67) lets a difference through.  kProjMargin (1.0625) covers the second-order terms (tools/proj_filter_check.c: the largest difference seen is 0.25 of the bound).
//   Downstream (every operation 1-Lipschitz or a product with an exact constant; RN(a) - RN(b) <= |a - b| + u |a| + u |b|):
//     clamp(q P): E1 = P Cq u + 2 u;  ClipXYToUV: E1 / 2 + 2 u;  lo + hi: E1 + 8 u;  f = fma(lo + hi, dim / 2, -0.5):
//     |f - f'| <= (dim / 2) (E1 + 12 u) = (dim / 2) K u,   K = P Cq + 14
//     width = (hi - lo) dimension:  |w - w'| <= W (E1 + 6 u) + 2 u w
//   floor(f) and the zero-weight flag (f == floor f) agree if  K u dim / 2 < frac(f') < 1 - K u dim / 2;
//   floor(log2(max(w, h, 1))) agrees if no power of two >= 2 lies within W (E1 + 6 u) + 2 u m of m' = max(w', h', 1): in
//   mantissa units of m' in [2^k, 2^(k+1)) that is (mipDelta >> k) + 4 (projMipDelta).
// tools/proj_filter_check.c replays both chains on the CPU with v_rsq / v_rcp modelled as ANY value within 1 ulp and checks
// the bounds and the decisions on 10^9 random and adversarial spheres; tests/test_gpu_parity.py puts projections ON texel
// edges and ON level boundaries +- 0 ... 10^5 ulp (the -DTR_EXP_PROJ_NOBAND build must fail it).
constexpr float kProjMargin = 1.0625f;
struct ProjBands               // wave-uniform (scalar registers)
{
    v2f invB;                  // (1 / Bx, 1 / By)
    v2f K;                     // (Kx, Ky) u: |f - f'| <= K u * (dim / 2), margin included
    uint32_t mipDelta;         // mantissa units (2^-23) a power of two must stay away from m' = max(w', h') in [1, 2); in [2^k, 2^(k+1)): (mipDelta >> k) + 4
    float mFloor;              // 1 + (mipDelta + 4) 2^-23: the floor of max(w', h', .) -- below 2 everything is level 0
};
__host__ __device__ inline ProjBands projBands(float P00, float P11, uint32_t hzbWidth, uint32_t hzbHeight)
{
    const float u = 0x1p-24f;
    ProjBands b;
    float K[2], invB[2];
    const float P[2] = { __builtin_fabsf(P00), __builtin_fabsf(P11) }, dim[2] = { (float)hzbWidth, (float)hzbHeight };
    float wBand = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float B = 1.125f / P[i] + 0.25f;
        const float qmax = 1.0f / P[i] + 0x1p-10f;
        const float Cq = kProjMargin * (0.51f * __builtin_sqrtf(B * B + 1.0f) + 5.11f * B + 0.15f + 12.3f * qmax);
        const float E1 = P[i] * Cq + 2.0f;                           // in units of u
        K[i] = (E1 + 12.0f) * u;
        invB[i] = 1.0f / B;
        wBand = __builtin_fmaxf(wBand, dim[i] * (E1 + 6.0f) * u);
    }
    b.invB = v2f{ invB[0], invB[1] };
    b.K = v2f{ K[0], K[1] };
#ifdef TR_EXP_PROJ_NOBAND     /* mutation test (results WRONG by design): the boundary test must fail without the bands */
    b.K = v2f{ 0.f, 0.f };
    wBand = 0.f;
#endif
    // |m - m'| <= wBand + 2 u m;  for m' in [2^k, 2^(k+1)) the distance to either end is (mantissa units) 2^(k-23) >= wBand + 2 u m
    // <=> mantissa units >= wBand 2^(23-k) + 2: projMipDelta(b, e), e = k + 1 (what v_frexp_exp_i32_f32 returns)
    const float d = wBand * 0x1p23f;
    b.mipDelta = d < 0x1p21f ? (uint32_t)d + 1u : 0x200000u;         // (2^21: every lane unsure -- a P or an HZB beyond any use)
    b.mFloor = 1.0f + (float)(b.mipDelta + 4u) * 0x1p-23f;
#ifdef TR_EXP_PROJ_NOBAND
    b.mFloor = 1.0f;
#endif
    return b;
}

__host__ __device__ inline uint32_t projMipDelta(const ProjBands& b, uint32_t e /* >= 1 */)
{
#ifdef TR_EXP_PROJ_NOBAND
    return 0u;
#else
    return (b.mipDelta >> (e - 1u)) + 4u;
#endif
}

// culling.hlsli:56-62 in closed form, approximately (see above): (minx, miny), (maxx, maxy); `sure` = the preconditions.
template <bool CZ_BOUNDED /* the caller's own checks already bound c.z (the cone's: c.c <= 2^60) */>
__device__ __forceinline__ void projectFiltered(F3 c, float r, const ProjBands& b, v2f& mn, v2f& mx, lmask& sure)
{
    const v2f cxy = { c.x, c.y };
    const float Z = fma_(c.z, c.z, -(r * r));                         // :54
    const v2f X = fma2(cxy, cxy, splat2(Z));                          // :56, :60 radicands
    const v2f y = { __builtin_amdgcn_rsqf(X.x), __builtin_amdgcn_rsqf(X.y) };
    const float rD = __builtin_amdgcn_rcpf(Z);
    const v2f vv = X * y;                                             // vx, vy
    const v2f a = cxy * splat2(c.z);
    const v2f mnN = fma2(-vv, splat2(r), a), mxN = fma2(vv, splat2(r), a);
    mn = mnN * splat2(rD);
    mx = mxN * splat2(rD);
    const v2f h = cxy * b.invB;
    const float m3 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(h.x), __builtin_fabsf(h.y)), __builtin_fabsf(r) * 8.0f);
    sure = mLe(m3, c.z);                                              // false for NaN
    if (!CZ_BOUNDED) sure &= mLe(c.z, 0x1p30f);
}

// culling.hlsli:64-78 from the four quotients on: clamp, ClipXYToUV, mip level, footprint origin.
struct OccUv { v2f uv; int mip; };
__device__ __forceinline__ v2f occClampUv(v2f m, v2f P)
{
    const v2f s = m * P;                                               // :64-67
    const v2f cl = { clamp_(s.x, -1.0f, 1.0f), clamp_(s.y, -1.0f, 1.0f) };
    return fma2(cl, v2f{ 0.5f, -0.5f }, splat2(0.5f));                 // :70-71 ClipXYToUV
}

// Footprint-min table path.  mipTab[e], e = floor(log2(max(w, h, 1))) + 1 clamped to `mips` (what v_frexp_exp returns):
// { first table entry of mip e-1, its block-row stride (64 * blocks per row), 0.5 * width, 0.5 * height }.
__device__ __forceinline__ OccQuad occTailQuad(const StepQuot& q, F3 c, float r, float nearPlane, float P00, float P11, const Hzb& h,
                                               const uint4* mipTab, uint32_t quadTotal)
{
    OccQuad o;
    o.accept = mLt(c.z - nearPlane, r);                                // :48-49
    const v2f P = { P00, P11 };
    const v2f lo = occClampUv(q.mn, P), hi = occClampUv(q.mx, P);      // (ax, ay), (az, aw)
    const v2f wh = (hi - lo) * v2f{ (float)h.width, (float)h.height }; // :73-74
    // :75 floor(log2(max(w, h))), clamped by SampleLevel to [0, mips-1]; anything below 1 (and NaN) -> mip 0 (Q6)
    const float m = max_(max_(wh.x, wh.y), 1.0f);
    int e = __builtin_amdgcn_frexp_expf(m);                            // exponent + 1 of a finite m >= 1
    e = e < (int)h.mips ? e : (int)h.mips;
    e = e > 1 ? e : 1;
    const uint4 tab = mipTab[e];
    // :78 uv = (lo + hi) * 0.5 (exact: lo + hi is 0 or >= 2^-25), then fma(uv, dim, -0.5) = RN(uv * dim - 0.5).  dim / 2 is
    // a float too, and (lo + hi) * (dim / 2) is the same real number as uv * dim: fma(lo + hi, dim / 2, -0.5) rounds the same
    // exact value -- the table holds dim / 2
    const v2f f = fma2(lo + hi, v2f{ __uint_as_float(tab.z), __uint_as_float(tab.w) }, splat2(-0.5f));
    const float flx = __builtin_floorf(f.x), fly = __builtin_floorf(f.y);
    const int x0 = (int)flx, y0 = (int)fly;                            // in [-1, mw-1] x [-1, mh-1]
    o.iq = min(quadIndex(tab.x, tab.y, (uint32_t)(x0 + 1), (uint32_t)(y0 + 1)), quadTotal - 1u);
    // A zero weight drops the second column (row) from the footprint; that only changes the set of texels when the
    // second column (row) is a different texel after edge clamping, i.e. 0 <= x0 and x0 + 1 <= mw - 1.  Rare: only a
    // wave that has an exactly integral coordinate somewhere looks at the rest of the condition.
    const lmask zx = mNotGt(f.x, flx), zy = mNotGt(f.y, fly);
    o.slow = 0ull;
    if (__builtin_expect((zx | zy) != 0ull, 0)) {
        const int mw = (int)(2.0f * __uint_as_float(tab.z)), mh = (int)(2.0f * __uint_as_float(tab.w));
        const bool bx = !(f.x > flx), by = !(f.y > fly);
        o.slow = __builtin_amdgcn_ballot_w64((bx & (x0 >= 0) & (x0 + 1 < mw)) | (by & (y0 >= 0) & (y0 + 1 < mh)));
    }
    o.depthSphere = q.depthSphere;
    return o;
}

// One cull step of the DEFERRED mode (footprint-table kernel): nothing here is exact-or-fallback, every lane gets the fast
// values and a verdict on whether they decide what the reference decides:
//   projection  projectFiltered (above), sureOcc = its preconditions (the bands are added by occTailQuadFiltered);
//   depthSphere nearPlane / (c.z - r), the compiler's Newton iteration without v_div_scale / v_div_fmas / v_div_fixup: exact
//               under the same preconditions (see stepQuotients: operands >= 2^-30, nearPlane in [2^-20, 2^20]);
//   cone        t * rsq(t.t), c.c * rsq(c.c) as in stepQuotients' fast path; sureCone = the radicands in [2^-96, 2^60] and
//               |r| <= 2^30 (coneBack's analysis needs that); its 2^-18 band is checked by coneBackSure.
template <bool CONE>
__device__ __forceinline__ void stepDeferred(F3 c, float r, uint32_t packed, const M33P& adj, float nearPlane, const ProjBands& bands, const float* coneTab,
                                             StepQuot& o, lmask& sureOcc, lmask& sureCone)
{
    projectFiltered<CONE>(c, r, bands, o.mn, o.mx, sureOcc);
    const float dz = c.z - r;
    const float rc = __builtin_amdgcn_rcpf(dz);
    const float r3 = fma_(fma_(-dz, rc, 1.0f), rc, rc);
    o.depthSphere = quotient1(nearPlane, dz, r3);                      // culling.hlsli:79
    o.coneExact = false;
    o.cutoff = 0.0f; o.lenC = 0.0f; o.tn = { 0.0f, 0.0f, 0.0f };
    sureCone = ~0ull;
    if (CONE) {
        const F3 a = coneAxisCutoffLds(packed, &o.cutoff, coneTab);
        const F3 t = mulVecP(a, adj);                                  // basepass.hlsl:103 mul(axis, adjugate)
        const float st = dot3(t, t), sc = dot3(c, c);
        const v2f rlen = { __builtin_amdgcn_rsqf(st), __builtin_amdgcn_rsqf(sc) };
        const v2f txy = v2f{ t.x, t.y } * splat2(rlen.x), tzl = v2f{ t.z, sc } * rlen;
        o.tn = { txy.x, txy.y, tzl.x };
        o.lenC = tzl.y;
        // radicands in [2^-96, 2^60] (negative / NaN read as huge) and |r| <= 2^30 <=> r r <= 2^60: one compare for the three upper bounds
        const uint32_t uMin = min(__float_as_uint(st), __float_as_uint(sc));
        const uint32_t uMax = max(max(__float_as_uint(st), __float_as_uint(sc)), __float_as_uint(r * r));
        sureCone = mGeU(uMin, 0x0F800000u) & mLeU(uMax, 0x5D800000u);
        sureOcc &= sureCone;                                               // (c.c <= 2^60 is projectFiltered's bound on c.z)
    }
}

// coneBack for the deferred mode: the fast decision and whether it is certain (outside the 2^-18 band, see coneBack).
__device__ __forceinline__ lmask coneBackSure(const StepQuot& q, F3 cv, float r, const M33P& viewRot, float kV, lmask& sure)
{
    F3 axis = mulVecP(q.tn, viewRot);
    axis.z = -axis.z;
    const float D = dot3(cv, axis), R = fma_(q.cutoff, q.lenC, r);
#ifndef TR_EXP_CONE_NOBAND
    const float E = fma_(q.lenC, kV, __builtin_fabsf(r)) * 0x1p-18f;
    sure &= mAbsGt(D - R, E);                                          // false for NaN
#endif
    return mGe(D, R);
}

// The same from the FILTERED quotients (projectFiltered): the table index of the lookup and, per lane, whether level, footprint
// origin and zero-weight flags are certain to be the reference's (`sure`, ANDed into the caller's).  A sure lane's lookup is
// never one the table cannot serve (its fractions are > 0), so there is no `slow` here.
__device__ __forceinline__ OccQuad occTailQuadFiltered(v2f mn, v2f mx, float depthSphere, F3 c, float r, float nearPlane, float P00, float P11, const Hzb& h,
                                                       const uint4* mipTab, const uint2* mipBand /* [e] = { projMipDelta(e), twice that } */, uint32_t quadTotal,
                                                       const ProjBands& b, lmask& sure)
{
    OccQuad o;
    o.accept = mLt(c.z - nearPlane, r);                                // :48-49 (exact)
    const v2f P = { P00, P11 };
    const v2f lo = occClampUv(mn, P), hi = occClampUv(mx, P);
    const v2f wh = (hi - lo) * v2f{ (float)h.width, (float)h.height }; // :73-74
    const float m = max_(max_(wh.x, wh.y), b.mFloor);                  // :75; NaN -> mFloor
    int e = __builtin_amdgcn_frexp_expf(m);
    e = e < (int)h.mips ? e : (int)h.mips;
    e = e > 1 ? e : 1;
    const uint4 tab = mipTab[e];
    const v2f half = { __uint_as_float(tab.z), __uint_as_float(tab.w) };
    const v2f f = fma2(lo + hi, half, splat2(-0.5f));                  // :78 + the sampler's uv * dim - 0.5 (see occTailQuad)
    const v2f fl = { __builtin_floorf(f.x), __builtin_floorf(f.y) };
    const int x0 = (int)fl.x, y0 = (int)fl.y;
    o.iq = min(quadIndex(tab.x, tab.y, (uint32_t)(x0 + 1), (uint32_t)(y0 + 1)), quadTotal - 1u);
    // bands: |frac - 1/2| <= 1/2 - K u dim / 2 on both axes (false for NaN); no power of two within mipDelta mantissa units
    const v2f t = f - (fl + splat2(0.5f));
    const v2f cap = fma2(half, -b.K, splat2(0.5f));
    const uint2 band = mipBand[e];
    const uint32_t mb = __float_as_uint(m) + band.x;
    sure &= mAbsLe(t.x, cap.x) & mAbsLe(t.y, cap.y) & mGeU(mb & 0x7FFFFFu, band.y);
    o.slow = 0ull;
    o.depthSphere = depthSphere;
    return o;
}

// Texel path: the same, ending in the two texel-pair indices (see occlusionPrepare).
__device__ __forceinline__ OccSample occTailTexel(const StepQuot& q, F3 c, float r, float nearPlane, float P00, float P11, const Hzb& h, const uint32_t* mipOff)
{
    OccSample o;
    o.accept = (c.z - nearPlane) < r;                                  // :48-49
    const v2f P = { P00, P11 };
    const v2f lo = occClampUv(q.mn, P), hi = occClampUv(q.mx, P);
    const v2f wh = (hi - lo) * v2f{ (float)h.width, (float)h.height };
    const int mip = hzbLevel(wh.x, wh.y, h.mips);                      // :75
    const v2f uv = (lo + hi) * splat2(0.5f);                           // :78
    const uint32_t mw = (h.width >> mip) ? (h.width >> mip) : 1u;
    const uint32_t mh = (h.height >> mip) ? (h.height >> mip) : 1u;
    const v2f f = fma2(uv, v2f{ (float)mw, (float)mh }, splat2(-0.5f));
    const float flx = __builtin_floorf(f.x), fly = __builtin_floorf(f.y);
    int x0 = (int)flx, y0 = (int)fly;
    const bool wx1 = (f.x - flx) > 0.0f, wy1 = (f.y - fly) > 0.0f;
    const int xm = (int)mw - 1, ym = (int)mh - 1;
    int x1 = min(max(x0 + 1, 0), xm), y1 = min(max(y0 + 1, 0), ym);
    x0 = min(max(x0, 0), xm);
    y0 = min(max(y0, 0), ym);
    x1 = wx1 ? x1 : x0;
    y1 = wy1 ? y1 : y0;
    const uint32_t base = mipOff[mip];
    o.i0 = base + (uint32_t)y0 * mw + (uint32_t)x0;
    o.i1 = base + (uint32_t)y1 * mw + (uint32_t)x0;
    o.pair = x1 != x0;                                                 // then x1 == x0 + 1
    o.depthSphere = q.depthSphere;
    return o;
}

// basepass.hlsl:104-107 + ConeCull (culling.hlsli:84-87) from the normalised axis on; lane mask of the back-facing meshlets.
//
// FILTERED PREDICATE.  The reference's test is  D >= R  with  D = dot(c, mul(normalize(t), V)),  R = cutoff * length(c) + r,
// every '/' and sqrt correctly rounded.  On the fast path stepQuotients hands over t * rsq(t.t) and c.c * rsq(c.c) instead
// (one v_rsq_f32 each, 1 ulp; no Newton steps, no division): D', R'.  With u = 2^-24, a = t / |t| in real arithmetic and
// S_j = sum_i |a_i| |V_ij| <= |V_.j|:
//   normalize, exact: length 2.5 u (dot 1.5 u, sqrt u), quotient u -> 3.5 u per component; fast: rsq 2 u + 1.5 u, product u
//     -> 4.5 u: the two differ by <= 8 u |a_i|;
//   axis_j (a 3-term fma chain in both): |axis_j - axis'_j| <= (8 + 2 * 3) u S_j;  D (another chain): |D - D'| <=
//     (14 + 6) u sum_j |c_j| S_j <= 20 u |c| |V|_F;
//   length(c): exact sqrt(c.c) (1 + u), fast (1 + 3 u): |R - R'| <= 4 u cutoff |c| + 2 u (cutoff |c| + |r|), cutoff <= 1;
//   |(D - R) - (D' - R')| <= u ((20 |V|_F + 6) |c| + 2 |r|)  <  E = 2^-18 ((|V|_F + 1) length(c) + |r|) = 64 u (...).
// So |D' - R'| > E decides the reference's comparison; a lane inside the band (or NaN: st = 0, ...) that still matters
// sends its wave through the exact sequences (coneBackfacingP).  kV = (|V|_F + 1)(1 + 2^-10), wave-uniform.
// Operands are in the range stepQuotients' `safe` check established (radicands in [2^-96, 2^60], |r| <= 2^30), so the
// relative bounds above hold (no denormal intermediate that matters: E >= 2^-18 * 2^-48).
__device__ __forceinline__ lmask coneBack(const StepQuot& q, F3 cv, float r, const M33P& viewRot, float kV, lmask relevant,
                                          uint32_t packed, const M33P& adj)
{
    F3 axis = mulVecP(q.tn, viewRot);
    axis.z = -axis.z;
    const float D = dot3(cv, axis), R = fma_(q.cutoff, q.lenC, r);
    lmask back = mGe(D, R);
#ifndef TR_EXP_CONE_NOBAND   /* mutation test (results WRONG by design): tests/test_gpu_parity.py::test_cone_test_at_its_decision_boundary must fail without the band */
    if (!q.coneExact) {
        const float E = fma_(q.lenC, kV, __builtin_fabsf(r)) * 0x1p-18f;
        const lmask certain = mAbsGt(D - R, E);
        if (__builtin_expect((~certain & relevant) != 0ull, 0)) {
            float unused;
            back = __builtin_amdgcn_ballot_w64(coneBackfacingP(packed, cv, r, adj, viewRot, 1.0f, 1.0f, &unused));
        }
    }
#endif
    return back;
}
// |V|_F + 1 of the view rotation, inflated: coneBack's kV
__device__ __forceinline__ float coneSlackFactor(const M43& v)
{
    const float f2 = dot3(v.r0, v.r0) + dot3(v.r1, v.r1) + dot3(v.r2, v.r2);
    return (__builtin_sqrtf(f2) + 1.0f) * (1.0f + 0x1p-10f);
}

__device__ __forceinline__ M43 loadM43(const interop::Matrix& m)
{
    return { { m.m[0][0], m.m[0][1], m.m[0][2] }, { m.m[1][0], m.m[1][1], m.m[1][2] },
             { m.m[2][0], m.m[2][1], m.m[2][2] }, { m.m[3][0], m.m[3][1], m.m[3][2] } };
}

} // namespace cm
