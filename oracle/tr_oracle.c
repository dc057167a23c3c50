/*
 * tr_oracle.c -- CPU ORACLE (test infrastructure, PARITY UNPINNED -- see tr_oracle.h).
 *
 * Plain-C restatement of the reference's GPU-driven visibility path.  Each function cites
 * the reference file:line it follows (paths relative to /root/reference/source).
 * Build: gcc -O2 -ffp-contract=off -mfma (oracle/Makefile).  Every fmaf() below is part of
 * the arithmetic convention; nothing else may be fused.
 */
#include "tr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* small helpers                                                                        */
/* ------------------------------------------------------------------------------------ */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* toyrenderer_common.hlsli:119-122 */
static inline uint32_t div_round_up(uint32_t x, uint32_t y) { return (x + y - 1) / y; }

/* dot(float3,float3) as an FMA chain (convention). */
static inline float dot3(const float a[3], const float b[3])
{
    return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
}

/* cross(a,b) per the HLSL definition; each component a*b - c*d = fmaf(a,b,-(c*d)). */
static inline void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}

/* mul(float4(p,1), M).xyz : ((p.x*M0j (+) p.y*M1j) (+) p.z*M2j) + M3j */
static inline void mul_point(const float p[3], const OrcMatrix* M, float o[3])
{
    for (int j = 0; j < 3; ++j)
        o[j] = fmaf(p[2], M->m[2][j], fmaf(p[1], M->m[1][j], p[0] * M->m[0][j])) + M->m[3][j];
}

/* mul(float3, float3x3) with rows r0,r1,r2 */
static inline void mul_vec3_rows(const float v[3], const float r0[3], const float r1[3], const float r2[3], float o[3])
{
    for (int j = 0; j < 3; ++j)
        o[j] = fmaf(v[2], r2[j], fmaf(v[1], r1[j], v[0] * r0[j]));
}

static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* ------------------------------------------------------------------------------------ */
/* fp16 <-> fp32 (HZB texel format R16_FLOAT, GraphicConstants.h:28; Q10)               */
/* ------------------------------------------------------------------------------------ */

uint16_t orc_f32_to_f16(float f)
{
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | (ax > 0x7F800000u ? (0x0200u | ((ax >> 13) & 0x3FFu)) : 0u));
    if (ax >= 0x477FF000u) /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7C00u);
    if (ax >= 0x38800000u) { /* normal half */
        uint32_t mant = ax & 0x7FFFFFu;
        uint32_t exp = (ax >> 23) - 112u;
        uint32_t h = (exp << 10) | (mant >> 13);
        uint32_t rem = mant & 0x1FFFu;
        if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h += 1u; /* may carry into exponent: correct */
        return (uint16_t)(sign | h);
    }
    if (ax < 0x33000000u) /* < 2^-25 -> 0 (ties at exactly 2^-25 go to even = 0) */
        return (uint16_t)(sign | (ax > 0x33000000u ? 1u : 0u));
    { /* subnormal half: value = mant24 * 2^(e-150); half ulp = 2^-24 */
        uint32_t e = ax >> 23;                      /* 102..112 */
        uint32_t mant = (ax & 0x7FFFFFu) | 0x800000u;
        uint32_t shift = 126u - e;                  /* 14..24 */
        uint32_t h = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (h & 1u))) h += 1u;
        return (uint16_t)(sign | h);
    }
}

float orc_f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t mant = h & 0x3FFu;
    if (exp == 0x1Fu) return u2f(sign | 0x7F800000u | (mant << 13));
    if (exp != 0) return u2f(sign | ((exp + 112u) << 23) | (mant << 13));
    if (mant == 0) return u2f(sign);
    /* subnormal: mant * 2^-24 exactly */
    float v = (float)mant * 5.9604644775390625e-08f;
    return sign ? -v : v;
}

/* BasePassRenderers.cpp:600-606 + Graphic.h:227-231: mips = bit_width(max(w,h)). */
uint32_t orc_hzb_layout(uint32_t w, uint32_t h, uint64_t* mipOffset, uint64_t* totalTexels)
{
    uint32_t res = w > h ? w : h;
    uint32_t mips = 0;
    while (res) { ++mips; res >>= 1; }
    uint64_t off = 0;
    for (uint32_t k = 0; k < mips && k < ORC_MAX_MIPS; ++k) {
        uint32_t mw = (w >> k) ? (w >> k) : 1u, mh = (h >> k) ? (h >> k) : 1u;
        if (mipOffset) mipOffset[k] = off;
        off += (uint64_t)mw * mh;
    }
    if (totalTexels) *totalTexels = off;
    return mips;
}

/* ------------------------------------------------------------------------------------ */
/* culling.hlsli                                                                        */
/* ------------------------------------------------------------------------------------ */

/* culling.hlsli:6-21 (Q7: near-plane test compiled out, no far plane). */
int orc_frustum_cull(const float c[3], float r, const float f[4])
{
    int visible = 1;
    visible &= fmaf(c[2], f[1], fabsf(c[0]) * f[0]) < r;
    visible &= fmaf(c[2], f[3], fabsf(c[1]) * f[2]) < r;
    return visible;
}

/* level = floor(log2(max(width,height))) (culling.hlsli:75) clamped by SampleLevel to
 * [0, mips-1]; log2 by exponent extraction; non-positive / NaN -> mip 0 (Q6). */
int orc_hzb_level(float width, float height, uint32_t mips)
{
    float m = fmaxf(width, height);
    if (!(m >= 1.0f)) return 0;
    int e = (int)((f2u(m) >> 23) & 0xFFu) - 127;
    int last = (int)mips - 1;
    return e > last ? last : e;
}

/* HZB.SampleLevel(linear-clamp-MIN sampler, uv, level) (culling.hlsli:78; sampler
 * CommonResources.cpp:276-287,298).  Convention (D3D min-reduction semantics): footprint =
 * texels floor(uv*dim-0.5)+{0,1} clamped to the edge; a texel takes part iff its bilinear
 * weight is non-zero. */
static float sample_hzb_min_mip(const OrcHZB* hzb, float u, float v, int mip)
{
    uint32_t mw = (hzb->width >> mip) ? (hzb->width >> mip) : 1u;
    uint32_t mh = (hzb->height >> mip) ? (hzb->height >> mip) : 1u;
    const uint16_t* t = hzb->texels + hzb->mipOffset[mip];
    float fx = fmaf(u, (float)mw, -0.5f);
    float fy = fmaf(v, (float)mh, -0.5f);
    float flx = floorf(fx), fly = floorf(fy);
    int x0 = (int)flx, y0 = (int)fly;
    int wx1 = (fx - flx) > 0.0f, wy1 = (fy - fly) > 0.0f;
    int x1 = x0 + 1, y1 = y0 + 1;
    int xm = (int)mw - 1, ym = (int)mh - 1;
    x0 = x0 < 0 ? 0 : (x0 > xm ? xm : x0);
    x1 = x1 < 0 ? 0 : (x1 > xm ? xm : x1);
    y0 = y0 < 0 ? 0 : (y0 > ym ? ym : y0);
    y1 = y1 < 0 ? 0 : (y1 > ym ? ym : y1);
    float d = orc_f16_to_f32(t[(uint64_t)y0 * mw + x0]);
    if (wx1) d = fminf(d, orc_f16_to_f32(t[(uint64_t)y0 * mw + x1]));
    if (wy1) {
        d = fminf(d, orc_f16_to_f32(t[(uint64_t)y1 * mw + x0]));
        if (wx1) d = fminf(d, orc_f16_to_f32(t[(uint64_t)y1 * mw + x1]));
    }
    return d;
}

float orc_sample_hzb_min(const OrcHZB* hzb, float u, float v, float level)
{
    int mip = (int)level;
    if (mip < 0) mip = 0;
    if (mip > (int)hzb->mips - 1) mip = (int)hzb->mips - 1;
    return sample_hzb_min_mip(hzb, u, v, mip);
}

/* culling.hlsli:53-78: screen-space bounds of the sphere -> HZB level and sample position. */
static void occlusion_sample_position(const float c[3], float r, float P00, float P11,
                                      uint32_t hzbW, uint32_t hzbH, uint32_t mips, float* u, float* v, int* level)
{
    float cr[3] = { c[0] * r, c[1] * r, c[2] * r };                 /* :53 */
    float czr2 = fmaf(c[2], c[2], -(r * r));                        /* :54 */

    float vx = sqrtf(fmaf(c[0], c[0], czr2));                       /* :56 */
    float minx = fmaf(vx, c[0], -cr[2]) / fmaf(vx, c[2], cr[0]);    /* :57 */
    float maxx = fmaf(vx, c[0], cr[2]) / fmaf(vx, c[2], -cr[0]);    /* :58 */

    float vy = sqrtf(fmaf(c[1], c[1], czr2));                       /* :60 */
    float miny = fmaf(vy, c[1], -cr[2]) / fmaf(vy, c[2], cr[1]);    /* :61 */
    float maxy = fmaf(vy, c[1], cr[2]) / fmaf(vy, c[2], -cr[1]);    /* :62 */

    float ax = clampf(minx * P00, -1.0f, 1.0f);                     /* :64-67 */
    float ay = clampf(miny * P11, -1.0f, 1.0f);
    float az = clampf(maxx * P00, -1.0f, 1.0f);
    float aw = clampf(maxy * P11, -1.0f, 1.0f);

    /* :70-71 ClipXYToUV (toyrenderer_common.hlsli:79-82): xy*(0.5,-0.5)+(0.5,0.5) */
    ax = fmaf(ax, 0.5f, 0.5f);
    ay = fmaf(ay, -0.5f, 0.5f);
    az = fmaf(az, 0.5f, 0.5f);
    aw = fmaf(aw, -0.5f, 0.5f);

    float width = (az - ax) * (float)hzbW;                          /* :73 */
    float height = (aw - ay) * (float)hzbH;                         /* :74 */
    *level = orc_hzb_level(width, height, mips);                    /* :75 */
    *u = (ax + az) * 0.5f;                                          /* :78 */
    *v = (ay + aw) * 0.5f;
}

/* culling.hlsli:36-82 (Mara & McGuire 2013 sphere bounds; Q6 y flip kept). */
int orc_occlusion_cull(const float c[3], float radius, float nearPlane, float P00, float P11, const OrcHZB* hzb)
{
    /* :48-49 trivially accept if the sphere intersects the near plane */
    if ((c[2] - nearPlane) < radius)
        return 1;

    float r = radius, u, v;
    int level;
    occlusion_sample_position(c, r, P00, P11, hzb->width, hzb->height, hzb->mips, &u, &v, &level);
    float depth = sample_hzb_min_mip(hzb, u, v, level);
    float depthSphere = nearPlane / (c[2] - r);                     /* :79 */
    return depthSphere >= depth;                                    /* :81 */
}

/* Test helper (not part of the path): where the occlusion test of each sphere samples the HZB.  For sphere i
 * (world-space centre, radius; identity instance transform) out[5i..] = { level, x0, y0, fracX == 0, fracY == 0 }
 * with (x0, y0) = floor(uv * mipDim - 0.5), the origin of the bilinear footprint (sample_hzb_min_mip).  Used to
 * construct scenes whose lookups have zero bilinear weights (tests/test_gpu_parity.py). */
void orc_occlusion_footprints(const float* centres, const float* radii, uint32_t n, const OrcMatrix* worldToView,
                              float P00, float P11, uint32_t hzbW, uint32_t hzbH, uint32_t mips, int32_t* out)
{
    for (uint32_t i = 0; i < n; ++i) {
        float cv[3], u, v;
        int level;
        orc_to_view(&centres[3 * i], worldToView, cv);
        occlusion_sample_position(cv, radii[i], P00, P11, hzbW, hzbH, mips, &u, &v, &level);
        uint32_t mw = (hzbW >> level) ? (hzbW >> level) : 1u, mh = (hzbH >> level) ? (hzbH >> level) : 1u;
        float fx = fmaf(u, (float)mw, -0.5f), fy = fmaf(v, (float)mh, -0.5f);
        float flx = floorf(fx), fly = floorf(fy);
        out[5 * i + 0] = level;
        out[5 * i + 1] = (int32_t)flx;
        out[5 * i + 2] = (int32_t)fly;
        out[5 * i + 3] = !((fx - flx) > 0.0f);
        out[5 * i + 4] = !((fy - fly) > 0.0f);
    }
}

/* culling.hlsli:84-87 */
int orc_cone_cull(const float c[3], float r, const float axis[3], float cutoff)
{
    return dot3(c, axis) >= fmaf(cutoff, sqrtf(dot3(c, c)), r);
}

/* ------------------------------------------------------------------------------------ */
/* toyrenderer_common.hlsli                                                             */
/* ------------------------------------------------------------------------------------ */

/* toyrenderer_common.hlsli:134-140 */
float orc_max_scale(const OrcMatrix* W)
{
    float dx = dot3(W->m[0], W->m[0]);
    float dy = dot3(W->m[1], W->m[1]);
    float dz = dot3(W->m[2], W->m[2]);
    return sqrtf(fmaxf(fmaxf(dx, dy), dz));
}

/* toyrenderer_common.hlsli:205-223 */
void orc_sphere_to_world(const OrcMatrix* W, const float s[4], float out[4])
{
    mul_point(s, W, out);
    out[3] = s[3] * orc_max_scale(W);
}

/* gpuculling.hlsl:118-119, basepass.hlsl:68-69: view transform then z *= -1 */
void orc_to_view(const float p[3], const OrcMatrix* V, float out[3])
{
    mul_point(p, V, out);
    out[2] = -out[2];
}

/* basepass.hlsl:92-105 + toyrenderer_common.hlsli:124-132 (adjugate) */
void orc_unpack_cone_view(uint32_t packed, const OrcMatrix* W, const OrcMatrix* V, float axisOut[3], float* cutoffOut)
{
    float q[4];
    for (int i = 0; i < 4; ++i)
        q[i] = (float)((packed >> (8 * i)) & 0xFFu) / 255.0f;       /* :92-98 */
    float a[3] = { fmaf(q[0], 2.0f, -1.0f), fmaf(q[1], 2.0f, -1.0f), fmaf(q[2], 2.0f, -1.0f) }; /* :101 */

    float adj0[3], adj1[3], adj2[3];                                 /* MakeAdjugateMatrix */
    cross3(W->m[1], W->m[2], adj0);
    cross3(W->m[2], W->m[0], adj1);
    cross3(W->m[0], W->m[1], adj2);

    float t[3];
    mul_vec3_rows(a, adj0, adj1, adj2, t);                           /* :103 mul(axis, adj) */
    float len = sqrtf(dot3(t, t));                                   /* normalize = v / length */
    t[0] = t[0] / len; t[1] = t[1] / len; t[2] = t[2] / len;
    mul_vec3_rows(t, V->m[0], V->m[1], V->m[2], axisOut);            /* :104 */
    axisOut[2] = -axisOut[2];                                        /* :105 */
    *cutoffOut = q[3];
}

/* gpuculling.hlsl:39-57 (SubmitInstance LOD choice: the LAST i whose error is below the threshold) */
uint32_t orc_select_lod(const OrcMeshData* mesh, const float cv[3], float r, const OrcMatrix* W,
                        uint32_t forcedLOD, float lodTarget)
{
    uint32_t lod = 0;
    if (forcedLOD != ORC_INVALID_LOD) {
        uint32_t last = mesh->m_NumLODs - 1u;
        lod = forcedLOD < last ? forcedLOD : last;
    } else {
        float distance = fmaxf(sqrtf(dot3(cv, cv)) - r, 0.0f);
        float threshold = distance * lodTarget / orc_max_scale(W);
        for (uint32_t i = 1; i < mesh->m_NumLODs; ++i)
            if (mesh->m_MeshLODDatas[i].m_Error < threshold)
                lod = i;
    }
    return lod;
}

/* toyrenderer_common.hlsli:151-203: MakeWorldMatrix = (R * S) * T, general 4x4 products. */
static void matmul4(const OrcMatrix* A, const OrcMatrix* B, OrcMatrix* C)
{
    OrcMatrix r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = fmaf(A->m[i][3], B->m[3][j], fmaf(A->m[i][2], B->m[2][j], fmaf(A->m[i][1], B->m[1][j], A->m[i][0] * B->m[0][j])));
    *C = r;
}

void orc_make_world_matrix(const float p[3], const float q[4], const float s[3], OrcMatrix* out)
{
    float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz;
    OrcMatrix R, S, T, RS;
    memset(&R, 0, sizeof R); memset(&S, 0, sizeof S); memset(&T, 0, sizeof T);
    R.m[0][0] = (1.f - 2.f * qyy) - 2.f * qzz;
    R.m[0][1] = (2.f * qx) * qy + (2.f * qz) * qw;
    R.m[0][2] = (2.f * qx) * qz - (2.f * qy) * qw;
    R.m[1][0] = (2.f * qx) * qy - (2.f * qz) * qw;
    R.m[1][1] = (1.f - 2.f * qxx) - 2.f * qzz;
    R.m[1][2] = (2.f * qy) * qz + (2.f * qx) * qw;
    R.m[2][0] = (2.f * qx) * qz + (2.f * qy) * qw;
    R.m[2][1] = (2.f * qy) * qz - (2.f * qx) * qw;
    R.m[2][2] = (1.f - 2.f * qxx) - 2.f * qyy;
    R.m[3][3] = 1.f;
    S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2]; S.m[3][3] = 1.f;
    T.m[0][0] = 1.f; T.m[1][1] = 1.f; T.m[2][2] = 1.f; T.m[3][3] = 1.f;
    T.m[3][0] = p[0]; T.m[3][1] = p[1]; T.m[3][2] = p[2];
    matmul4(&R, &S, &RS);
    matmul4(&RS, &T, out);
}

/* BasePassRenderers.cpp:557-563: frustum = (nX.x, nX.z, nY.y, nY.z) with
 * nX = normalize4(Pt[3]+Pt[0]), nY = normalize4(Pt[3]+Pt[1]); Pt = transpose(ViewToClip).
 * (DirectXMath XMVector4Normalize is absent -> v / sqrt(dot4) convention; host-side only.) */
void orc_culling_frustum(const OrcMatrix* P, float out[4])
{
    float fx[4], fy[4];
    for (int j = 0; j < 4; ++j) {
        fx[j] = P->m[j][3] + P->m[j][0];
        fy[j] = P->m[j][3] + P->m[j][1];
    }
    float lx = sqrtf(fmaf(fx[3], fx[3], fmaf(fx[2], fx[2], fmaf(fx[1], fx[1], fx[0] * fx[0]))));
    float ly = sqrtf(fmaf(fy[3], fy[3], fmaf(fy[2], fy[2], fmaf(fy[1], fy[1], fy[0] * fy[0]))));
    out[0] = fx[0] / lx; out[1] = fx[2] / lx;
    out[2] = fy[1] / ly; out[3] = fy[2] / ly;
}

/* ------------------------------------------------------------------------------------ */
/* gpuculling.hlsl : CS_GPUCulling + SubmitInstance                                     */
/* ------------------------------------------------------------------------------------ */

enum { ST_DROP = 0, ST_SUBMIT = 1, ST_LATE = 2 };

/* One dispatch thread of CS_GPUCulling up to (not including) the ordered side effects.
 * gpuculling.hlsl:105-178.  Returns ST_*, and for ST_SUBMIT the LOD and group count. */
static int classify_instance(const OrcGPUCullingPassConstants* k, int lateCull, uint32_t id,
                             const OrcBasePassInstanceConstants* instances, const OrcMeshData* meshData,
                             const OrcHZB* hzb, uint32_t* lodOut, uint32_t* groupsOut)
{
    const int doFrustum = (k->m_CullingFlags & ORC_CULL_FRUSTUM) != 0;
    const int doOcclusion = (k->m_CullingFlags & ORC_CULL_OCCLUSION) != 0;

    const OrcBasePassInstanceConstants* inst = &instances[id];                  /* :114 */
    const OrcMeshData* mesh = &meshData[inst->m_MeshDataIdx];
    float ws[4];
    orc_sphere_to_world(&inst->m_WorldMatrix, mesh->m_BoundingSphere, ws);      /* :116 */
    float cv[3];
    orc_to_view(ws, &k->m_WorldToView, cv);                                     /* :118-119 */
    float r = ws[3];

    int visible = 1;
    if (!lateCull)                                                              /* :124-129 */
        visible = !doFrustum || orc_frustum_cull(cv, r, k->m_Frustum);
    if (!visible) return ST_DROP;                                               /* :131-134 */

    int status = ST_SUBMIT;
    if (doOcclusion) {
        if (!lateCull)
            orc_to_view(ws, &k->m_PrevWorldToView, cv);                         /* :143-146 (Q3) */
        int occVisible = orc_occlusion_cull(cv, r, k->m_NearPlane, k->m_P00, k->m_P11, hzb); /* :148-158 */
        if (!occVisible) return lateCull ? ST_DROP : ST_LATE;                   /* :162-178 */
    }
    /* SubmitInstance :35-62 */
    uint32_t lod = orc_select_lod(mesh, cv, r, &inst->m_WorldMatrix, k->m_ForcedMeshLOD, k->m_MeshLODTarget);
    *lodOut = lod;
    *groupsOut = div_round_up(mesh->m_MeshLODDatas[lod].m_NumMeshlets, ORC_NUM_THREADS_PER_WAVE);
    return status;
}

typedef struct {
    const OrcGPUCullingPassConstants* k; int lateCull;
    const OrcBasePassInstanceConstants* instances; const uint32_t* ids;
    const OrcMeshData* meshData; const OrcHZB* hzb;
    uint32_t begin, end;
    uint8_t* status; uint32_t* lod; uint32_t* groups;
} ClassifyJob;

static void* classify_worker(void* p)
{
    ClassifyJob* j = (ClassifyJob*)p;
    for (uint32_t t = j->begin; t < j->end; ++t)
        j->status[t] = (uint8_t)classify_instance(j->k, j->lateCull, j->ids[t], j->instances, j->meshData, j->hzb, &j->lod[t], &j->groups[t]);
    return 0;
}

static void instance_cull_mt(const OrcGPUCullingPassConstants* k, int lateCull,
                             const OrcBasePassInstanceConstants* instances, const uint32_t* primitiveIds,
                             const OrcMeshData* meshData, const OrcHZB* hzb,
                             OrcMeshletAmplificationData* records, uint32_t dispatchArgs[3],
                             uint32_t* lateCount, uint32_t* lateIds, uint32_t lateDispatchArgsX,
                             uint32_t maxGroups, uint32_t* validRecords, uint32_t threads,
                             const uint32_t* shardLate /* NULL or {base, total} (multi-GPU checker) */)
{
    /* :94-103 thread range */
    uint32_t nbInstances = lateCull ? *lateCount : k->m_NbInstances;
    uint32_t nThreads;
    const uint32_t* ids;
    if (lateCull) {
        uint64_t launched = (uint64_t)lateDispatchArgsX * ORC_NUM_THREADS_PER_WAVE;   /* Q1 */
        if (shardLate) {   /* the same rule on the rank-major concatenation of all shards' late lists */
            uint64_t all = (((uint64_t)shardLate[1] + 63u) / 64u) * ORC_NUM_THREADS_PER_WAVE;
            launched = all > shardLate[0] ? all - shardLate[0] : 0u;
        }
        nThreads = (uint32_t)(launched < nbInstances ? launched : nbInstances);
        ids = lateIds;
    } else {
        nThreads = nbInstances; /* ceil(n/32) groups, bounds check :100 */
        ids = primitiveIds;
    }

    uint8_t* status = (uint8_t*)malloc(nThreads ? nThreads : 1);
    uint32_t* lod = (uint32_t*)malloc(sizeof(uint32_t) * (nThreads ? nThreads : 1));
    uint32_t* groups = (uint32_t*)malloc(sizeof(uint32_t) * (nThreads ? nThreads : 1));

    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (threads == 1 || nThreads < 4096) {
        ClassifyJob j = { k, lateCull, instances, ids, meshData, hzb, 0, nThreads, status, lod, groups };
        classify_worker(&j);
    } else {
        pthread_t th[256]; ClassifyJob jobs[256];
        for (uint32_t i = 0; i < threads; ++i) {
            uint32_t b = (uint32_t)((uint64_t)nThreads * i / threads), e = (uint32_t)((uint64_t)nThreads * (i + 1) / threads);
            ClassifyJob j = { k, lateCull, instances, ids, meshData, hzb, b, e, status, lod, groups };
            jobs[i] = j;
            pthread_create(&th[i], 0, classify_worker, &jobs[i]);
        }
        for (uint32_t i = 0; i < threads; ++i) pthread_join(th[i], 0);
    }

    /* ordered side effects, ascending thread id (canonical order) */
    uint32_t valid = 0xFFFFFFFFu;
    for (uint32_t t = 0; t < nThreads; ++t) {
        if (status[t] == ST_LATE) {                                              /* :162-167 */
            uint32_t idx = (*lateCount)++;
            lateIds[idx] = ids[t];
        } else if (status[t] == ST_SUBMIT) {                                     /* :64-84 */
            uint32_t n = groups[t];
            uint32_t off = dispatchArgs[0];
            dispatchArgs[0] = off + n;
            dispatchArgs[1] = 1; dispatchArgs[2] = 1;
            if (off + n >= maxGroups) {                                          /* Q2 */
                if (valid == 0xFFFFFFFFu && n) valid = off;
                continue;
            }
            for (uint32_t i = 0; i < n; ++i) {
                records[off + i].m_InstanceConstIdx = ids[t];
                records[off + i].m_MeshLOD = lod[t];
                records[off + i].m_MeshletGroupOffset = i * ORC_NUM_THREADS_PER_WAVE;
            }
        }
    }
    if (validRecords) *validRecords = (valid == 0xFFFFFFFFu) ? dispatchArgs[0] : valid;
    free(status); free(lod); free(groups);
}

void orc_instance_cull(const OrcGPUCullingPassConstants* k, int lateCull,
                       const OrcBasePassInstanceConstants* instances, const uint32_t* primitiveIds,
                       const OrcMeshData* meshData, const OrcHZB* hzb,
                       OrcMeshletAmplificationData* records, uint32_t dispatchArgs[3],
                       uint32_t* lateCount, uint32_t* lateIds, uint32_t lateDispatchArgsX,
                       uint32_t maxGroups, uint32_t* validRecords)
{
    instance_cull_mt(k, lateCull, instances, primitiveIds, meshData, hzb, records, dispatchArgs,
                     lateCount, lateIds, lateDispatchArgsX, maxGroups, validRecords, 1, NULL);
}

/* gpuculling.hlsl:182-195 (Q1: 64, not kNumThreadsPerWave) */
/* giprobevisualization.hlsl:16-69 */
uint32_t orc_gi_probe_cull(const OrcGIProbeVisualizationUpdateConsts* k, const float* probePositions, const float* probeStates,
                           const OrcHZB* hzb, float* outPositions, uint32_t* drawArgs, uint32_t* outInstanceToProbe)
{
    uint32_t appended = 0;
    for (uint32_t probeIndex = 0; probeIndex < k->m_NumProbes; ++probeIndex) {      /* :18-23 */
        if (k->m_bHideInactiveProbes && probeStates[probeIndex] == 1.0f) continue;   /* :29-34, RTXGI_DDGI_PROBE_STATE_INACTIVE */
        const float* wp = probePositions + 3u * probeIndex;                          /* :36-37 (input instead of the DDGI volume) */
        float v[3];
        orc_to_view(wp, &k->m_WorldToView, v);                                        /* :39-40 */
        if (!orc_frustum_cull(v, k->m_ProbeRadius, k->m_Frustum)) continue;           /* :42-45 */
        if (!orc_occlusion_cull(v, k->m_ProbeRadius, k->m_NearPlane, k->m_P00, k->m_P11, hzb)) continue;   /* :47-60 */
        const uint32_t outInstanceIndex = drawArgs[1]++;                              /* :62-63 InterlockedAdd(m_InstanceCount, 1) */
        outPositions[3u * outInstanceIndex + 0] = wp[0];                              /* :66 */
        outPositions[3u * outInstanceIndex + 1] = wp[1];
        outPositions[3u * outInstanceIndex + 2] = wp[2];
        outInstanceToProbe[outInstanceIndex] = probeIndex;                            /* :67 */
        ++appended;
    }
    return appended;
}

void orc_build_late_args(uint32_t lateCount, uint32_t out[3])
{
    out[0] = div_round_up(lateCount, 64);
    out[1] = 1;
    out[2] = 1;
}

/* ------------------------------------------------------------------------------------ */
/* basepass.hlsl : AS_Main                                                              */
/* ------------------------------------------------------------------------------------ */

/* basepass.hlsl:52-121 for one group; returns the 32-bit lane-visibility mask. */
static uint32_t meshlet_group(const OrcBasePassConstants* k,
                              const OrcBasePassInstanceConstants* instances, const OrcMeshData* meshData,
                              const OrcMeshletData* meshlets, const OrcMeshletAmplificationData* rec,
                              const OrcHZB* hzb, uint32_t* testedOut)
{
    const int doFrustum = (k->m_CullingFlags & ORC_CULL_FRUSTUM) != 0;
    const int doOcclusion = (k->m_CullingFlags & ORC_CULL_OCCLUSION) != 0;
    const int doCone = (k->m_CullingFlags & ORC_CULL_CONE) != 0;

    const OrcBasePassInstanceConstants* inst = &instances[rec->m_InstanceConstIdx];     /* :56 */
    const OrcMeshData* mesh = &meshData[inst->m_MeshDataIdx];                           /* :57 */
    const OrcMeshLODData* lod = &mesh->m_MeshLODDatas[rec->m_MeshLOD];                  /* :58 */

    uint32_t mask = 0, tested = 0;
    for (uint32_t lane = 0; lane < ORC_NUM_THREADS_PER_WAVE; ++lane) {
        uint32_t meshletIdx = rec->m_MeshletGroupOffset + lane;                         /* :62 */
        if (!(meshletIdx < lod->m_NumMeshlets)) continue;                               /* :63 */
        ++tested;
        const OrcMeshletData* md = &meshlets[lod->m_MeshletDataBufferIdx + meshletIdx]; /* :65 */

        float cw[3], cv[3];
        mul_point(md->m_BoundingSphere, &inst->m_WorldMatrix, cw);                      /* :67 */
        orc_to_view(cw, &k->m_WorldToView, cv);                                         /* :68-69 */
        float r = md->m_BoundingSphere[3] * orc_max_scale(&inst->m_WorldMatrix);        /* :71 */

        int vis = !doFrustum || orc_frustum_cull(cv, r, k->m_Frustum);                  /* :73 */
        if (vis && doOcclusion)                                                         /* :75-88 (Q4) */
            vis = orc_occlusion_cull(cv, r, k->m_NearPlane, k->m_P00, k->m_P11, hzb);
        if (vis && doCone) {                                                            /* :90-108 */
            float axis[3], cutoff;
            orc_unpack_cone_view(md->m_ConeAxisAndCutoff, &inst->m_WorldMatrix, &k->m_WorldToView, axis, &cutoff);
            vis = !orc_cone_cull(cv, r, axis, cutoff);
        }
        if (vis) mask |= 1u << lane;                                                    /* :111-118 */
    }
    *testedOut = tested;
    return mask;
}

uint64_t orc_meshlet_cull(const OrcBasePassConstants* k,
                          const OrcBasePassInstanceConstants* instances, const OrcMeshData* meshData,
                          const OrcMeshletData* meshlets, const OrcMeshletAmplificationData* records,
                          uint32_t groupBegin, uint32_t groupEnd, const OrcHZB* hzb,
                          uint32_t* visMask, uint32_t* visibleList, uint64_t* listCursor)
{
    uint64_t tested = 0;
    for (uint32_t g = groupBegin; g < groupEnd; ++g) {
        uint32_t t;
        uint32_t mask = meshlet_group(k, instances, meshData, meshlets, &records[g], hzb, &t);
        tested += t;
        if (visMask) visMask[g] = mask;
        if (visibleList) {
            /* WavePrefixCountBits order (:116-117): ascending lane */
            for (uint32_t lane = 0; lane < 32; ++lane)
                if (mask & (1u << lane))
                    visibleList[(*listCursor)++] = (g << 5) | lane;
        }
    }
    return tested;
}

/* ------------------------------------------------------------------------------------ */
/* Depth of the visible meshlets (convention; see tr_oracle.h)                          */
/* ------------------------------------------------------------------------------------ */

/* mul(float4(p,1), M) incl. the w column: same chain as mul_point per component */
static inline void mul_point4(const float p[3], const OrcMatrix* M, float o[4])
{
    for (int j = 0; j < 4; ++j)
        o[j] = fmaf(p[2], M->m[2][j], fmaf(p[1], M->m[1][j], p[0] * M->m[0][j])) + M->m[3][j];
}

/* edge function of (a, b) at p */
static inline float edge_fn(const float a[2], const float b[2], float px, float py)
{
    return fmaf(b[0] - a[0], py - a[1], -((b[1] - a[1]) * (px - a[0])));
}

void orc_raster_depth(const OrcBasePassConstants* k,
                      const OrcBasePassInstanceConstants* instances, const OrcMeshData* meshData,
                      const OrcMeshletData* meshlets, const OrcRawVertexFormat* vertices,
                      const uint32_t* meshletVertexIds, const uint32_t* meshletTriangles,
                      const OrcMeshletAmplificationData* records, const uint32_t* visibleList, uint32_t numVisible,
                      float* depth)
{
    const uint32_t W = k->m_OutputResolution[0], H = k->m_OutputResolution[1];
    const float halfW = 0.5f * (float)W, halfH = 0.5f * (float)H;
    for (uint32_t v = 0; v < numVisible; ++v) {
        const uint32_t g = visibleList[v] >> 5, lane = visibleList[v] & 31u;
        const OrcMeshletAmplificationData* rec = &records[g];                       /* basepass.hlsl:138-142 */
        const OrcBasePassInstanceConstants* inst = &instances[rec->m_InstanceConstIdx];
        const uint32_t lodIdx = rec->m_MeshLOD < ORC_MAX_LODS ? rec->m_MeshLOD : ORC_MAX_LODS - 1;
        const OrcMeshLODData* lod = &meshData[inst->m_MeshDataIdx].m_MeshLODDatas[lodIdx];
        const OrcMeshletData* ml = &meshlets[lod->m_MeshletDataBufferIdx + rec->m_MeshletGroupOffset + lane];
        uint32_t nv = ml->m_VertexAndTriangleCount & 0xFFu;                          /* :144-145 */
        const uint32_t nt = (ml->m_VertexAndTriangleCount >> 8) & 0xFFu;
        if (nv > 64u) nv = 64u;                                                     /* kMaxMeshletVertices (ShaderInterop.h:19) */
        float sx[256], sy[256], sd[256];
        int ok[256];
        for (uint32_t i = 0; i < nv; ++i) {                                         /* :149-158 */
            const OrcRawVertexFormat* vin = &vertices[meshletVertexIds[ml->m_MeshletVertexIDsBufferIdx + i]];
            float wp[3], clip[4];
            mul_point(vin->m_Position, &inst->m_WorldMatrix, wp);
            mul_point4(wp, &k->m_WorldToClip, clip);
            ok[i] = clip[3] > k->m_NearPlane;
            const float x = clip[0] / clip[3], y = clip[1] / clip[3];
            sd[i] = clip[2] / clip[3];
            sx[i] = fmaf(x, halfW, halfW);
            sy[i] = fmaf(-y, halfH, halfH);
        }
        for (uint32_t t = 0; t < nt; ++t) {                                         /* :178-187 */
            const uint32_t packed = meshletTriangles[ml->m_MeshletIndexIDsBufferIdx + t];
            const uint32_t ia = packed & 0xFFu, ib = (packed >> 8) & 0xFFu, ic = (packed >> 16) & 0xFFu;
            if (ia >= nv || ib >= nv || ic >= nv) continue;
            if (!(ok[ia] && ok[ib] && ok[ic])) continue;
            const float v0[2] = { sx[ia], sy[ia] }, v1[2] = { sx[ib], sy[ib] }, v2[2] = { sx[ic], sy[ic] };
            const float d0 = sd[ia], d1 = sd[ib], d2 = sd[ic];
            const float area = edge_fn(v0, v1, v2[0], v2[1]);
            if (!(area != 0.0f)) continue;                                          /* degenerate or NaN */
            const float sgn = area < 0.0f ? -1.0f : 1.0f;
            const float fminx = fminf(fminf(v0[0], v1[0]), v2[0]), fmaxx = fmaxf(fmaxf(v0[0], v1[0]), v2[0]);
            const float fminy = fminf(fminf(v0[1], v1[1]), v2[1]), fmaxy = fmaxf(fmaxf(v0[1], v1[1]), v2[1]);
            if (!(fmaxx >= 0.0f && fmaxy >= 0.0f && fminx <= (float)W && fminy <= (float)H)) continue;   /* off screen or NaN */
            const int x0 = (int)fmaxf(floorf(fminx), 0.0f), x1 = (int)fminf(ceilf(fmaxx), (float)(W - 1));
            const int y0 = (int)fmaxf(floorf(fminy), 0.0f), y1 = (int)fminf(ceilf(fmaxy), (float)(H - 1));
            for (int py = y0; py <= y1; ++py)
                for (int px = x0; px <= x1; ++px) {
                    const float cx = (float)px + 0.5f, cy = (float)py + 0.5f;
                    const float e0 = sgn * edge_fn(v1, v2, cx, cy), e1 = sgn * edge_fn(v2, v0, cx, cy), e2 = sgn * edge_fn(v0, v1, cx, cy);
                    if (!(e0 >= 0.0f && e1 >= 0.0f && e2 >= 0.0f)) continue;
                    const float den = (e0 + e1) + e2;
                    if (!(den > 0.0f)) continue;
                    const float d = fmaf(e2, d2, fmaf(e1, d1, e0 * d0)) / den;
                    float* dst = &depth[(uint64_t)py * W + px];
                    if (d > *dst) *dst = d;                                         /* GREATER test; NaN never passes */
                }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* HZB build                                                                            */
/* ------------------------------------------------------------------------------------ */

typedef struct { const float* depth; uint32_t W, H; uint16_t* m0; uint32_t hw, hh, y0, y1; } HzbMip0Job;

static void* hzb_mip0_rows(void* p)
{
    /* minmaxdownsample.hlsl:15-34 with m_bDownsampleMax = 0 (BasePassRenderers.cpp:517):
     * uv = (tid + 0.5) / outDim; Gather = the 2x2 quad floor(uv*dim - 0.5) + {0,1}, clamped (Q11). */
    const HzbMip0Job* j = (const HzbMip0Job*)p;
    const float* depth = j->depth;
    const uint32_t W = j->W, H = j->H, hw = j->hw, hh = j->hh;
    uint16_t* m0 = j->m0;
    for (uint32_t y = j->y0; y < j->y1; ++y) {
        float v = ((float)y + 0.5f) / (float)hh;
        float fy = fmaf(v, (float)H, -0.5f);
        int y0 = (int)floorf(fy), y1 = y0 + 1;
        y0 = y0 < 0 ? 0 : (y0 > (int)H - 1 ? (int)H - 1 : y0);
        y1 = y1 < 0 ? 0 : (y1 > (int)H - 1 ? (int)H - 1 : y1);
        for (uint32_t x = 0; x < hw; ++x) {
            float u = ((float)x + 0.5f) / (float)hw;
            float fx = fmaf(u, (float)W, -0.5f);
            int x0 = (int)floorf(fx), x1 = x0 + 1;
            x0 = x0 < 0 ? 0 : (x0 > (int)W - 1 ? (int)W - 1 : x0);
            x1 = x1 < 0 ? 0 : (x1 > (int)W - 1 ? (int)W - 1 : x1);
            float a = depth[(uint64_t)y0 * W + x0], b = depth[(uint64_t)y0 * W + x1];
            float c = depth[(uint64_t)y1 * W + x0], d = depth[(uint64_t)y1 * W + x1];
            float mn = fminf(fminf(a, b), fminf(c, d));              /* Min4, toyrenderer_common.hlsli:60-63 */
            m0[(uint64_t)y * hw + x] = orc_f32_to_f16(mn);
        }
    }
    return 0;
}

static void hzb_build_mt(const float* depth, uint32_t W, uint32_t H,
                         uint16_t* texels, uint32_t hw, uint32_t hh, uint32_t mips, const uint64_t* mipOffset, uint32_t threads)
{
    uint16_t* m0 = texels + mipOffset[0];
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (threads > hh) threads = hh ? hh : 1;
    if (threads == 1) {
        HzbMip0Job j = { depth, W, H, m0, hw, hh, 0, hh };
        hzb_mip0_rows(&j);
    } else {
        pthread_t th[256]; HzbMip0Job jobs[256];
        for (uint32_t i = 0; i < threads; ++i) {
            HzbMip0Job j = { depth, W, H, m0, hw, hh, (uint32_t)((uint64_t)hh * i / threads), (uint32_t)((uint64_t)hh * (i + 1) / threads) };
            jobs[i] = j;
            pthread_create(&th[i], 0, hzb_mip0_rows, &jobs[i]);
        }
        for (uint32_t i = 0; i < threads; ++i) pthread_join(th[i], 0);
    }
    /* SPD min filter (FFXHelpers.cpp:108): mip k+1 texel = min of the 2x2 block of mip k. */
    for (uint32_t k = 1; k < mips; ++k) {
        uint32_t pw = (hw >> (k - 1)) ? (hw >> (k - 1)) : 1u, ph = (hh >> (k - 1)) ? (hh >> (k - 1)) : 1u;
        uint32_t mw = (hw >> k) ? (hw >> k) : 1u, mh = (hh >> k) ? (hh >> k) : 1u;
        const uint16_t* src = texels + mipOffset[k - 1];
        uint16_t* dst = texels + mipOffset[k];
        for (uint32_t y = 0; y < mh; ++y)
            for (uint32_t x = 0; x < mw; ++x) {
                uint32_t x0 = 2 * x < pw - 1 ? 2 * x : pw - 1, x1 = 2 * x + 1 < pw - 1 ? 2 * x + 1 : pw - 1;
                uint32_t y0 = 2 * y < ph - 1 ? 2 * y : ph - 1, y1 = 2 * y + 1 < ph - 1 ? 2 * y + 1 : ph - 1;
                float a = orc_f16_to_f32(src[(uint64_t)y0 * pw + x0]), b = orc_f16_to_f32(src[(uint64_t)y0 * pw + x1]);
                float c = orc_f16_to_f32(src[(uint64_t)y1 * pw + x0]), d = orc_f16_to_f32(src[(uint64_t)y1 * pw + x1]);
                dst[(uint64_t)y * mw + x] = orc_f32_to_f16(fminf(fminf(a, b), fminf(c, d)));
            }
    }
}

void orc_hzb_build(const float* depth, uint32_t W, uint32_t H,
                   uint16_t* texels, uint32_t hw, uint32_t hh, uint32_t mips, const uint64_t* mipOffset)
{
    hzb_build_mt(depth, W, H, texels, hw, hh, mips, mipOffset, 1);
}

/* ------------------------------------------------------------------------------------ */
/* updateinstanceconsts.hlsl                                                            */
/* ------------------------------------------------------------------------------------ */

void orc_update_instance_consts(const OrcNodeLocalTransform* nodes, const uint32_t* primToNode,
                                OrcBasePassInstanceConstants* instances, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) {                                              /* :13-16 */
        uint32_t nodeID = primToNode[i];                                            /* :19 */
        const OrcNodeLocalTransform* lt = &nodes[nodeID];                           /* :20 */
        OrcMatrix world;
        orc_make_world_matrix(lt->m_Position, lt->m_Rotation, lt->m_Scale, &world); /* :22 */
        uint32_t parent = lt->m_ParentNodeIdx;                                      /* :24 */
        while (parent != 0xFFFFFFFFu) {                                             /* :25-32 */
            const OrcNodeLocalTransform* pt = &nodes[parent];
            OrcMatrix pm;
            orc_make_world_matrix(pt->m_Position, pt->m_Rotation, pt->m_Scale, &pm);
            matmul4(&world, &pm, &world);
            parent = pt->m_ParentNodeIdx;
        }
        instances[i].m_PrevWorldMatrix = instances[i].m_WorldMatrix;                /* :35 */
        instances[i].m_WorldMatrix = world;                                         /* :36 */
    }
}

/* ------------------------------------------------------------------------------------ */
/* whole frame                                                                          */
/* ------------------------------------------------------------------------------------ */

typedef struct {
    const OrcBasePassConstants* k; const OrcFrameDesc* d; const OrcMeshletAmplificationData* records;
    const OrcHZB* hzb; uint32_t begin, end; uint32_t* visMask; uint64_t tested;
} MeshletJob;

static void* meshlet_worker(void* p)
{
    MeshletJob* j = (MeshletJob*)p;
    j->tested = orc_meshlet_cull(j->k, j->d->instances, j->d->meshData, j->d->meshlets, j->records,
                                 j->begin, j->end, j->hzb, j->visMask, 0, 0);
    return 0;
}

/* GPUCulling + RenderInstances(cull part) for one list / phase. BasePassRenderers.cpp:298-503. */
static void run_pass(OrcFrameDesc* d, OrcFrameOut* o, int slot, int late, int alphaMask,
                     uint32_t flags, const float frustum[4], const OrcHZB* hzb)
{
    const uint32_t* ids = alphaMask ? d->alphaMaskIds : d->opaqueIds;
    uint32_t nb = alphaMask ? d->numAlphaMask : d->numOpaque;
    int li = alphaMask ? 1 : 0;
    o->passRan[slot] = 0;
    if (nb == 0) return;                                                   /* :311-314, :420-423 */
    const int occlusion = (flags & ORC_CULL_OCCLUSION) != 0;
    if (late && !occlusion) return;                                        /* :392-402 */
    o->passRan[slot] = 1;

    memset(o->dispatchArgs[slot], 0, sizeof o->dispatchArgs[slot]);      /* :325 */
    if (!late && occlusion) {                                              /* :327-331 */
        o->lateCount[li] = 0;
        memset(o->lateIds[li], 0, sizeof(uint32_t) * nb);
    }

    OrcGPUCullingPassConstants k;                                          /* :336-347 */
    memset(&k, 0, sizeof k);
    k.m_NbInstances = nb;
    k.m_CullingFlags = flags;
    memcpy(k.m_Frustum, frustum, sizeof k.m_Frustum);
    k.m_HZBDimensions[0] = occlusion ? d->hzbW : 1; k.m_HZBDimensions[1] = occlusion ? d->hzbH : 1;
    k.m_WorldToView = d->worldToView;
    k.m_PrevWorldToView = d->prevWorldToView;
    k.m_NearPlane = d->nearPlane;
    k.m_P00 = d->viewToClip.m[0][0];
    k.m_P11 = d->viewToClip.m[1][1];
    k.m_ForcedMeshLOD = d->forceMeshLOD >= 0 ? (uint32_t)d->forceMeshLOD : ORC_INVALID_LOD;
    k.m_MeshLODTarget = (2.0f / d->viewToClip.m[1][1]) * (1.0f / (float)d->renderHeight);

    const uint32_t shard[2] = { d->shardLateBase[li], d->shardLateTotal[li] };
    instance_cull_mt(&k, late, d->instances, ids, d->meshData, hzb, o->records[slot], o->dispatchArgs[slot],
                     &o->lateCount[li], o->lateIds[li], o->lateArgs[li][0], d->maxGroups, &o->validRecords[slot], d->threads,
                     (late && d->shardLate) ? shard : NULL);
    if (!late && occlusion)
        orc_build_late_args(o->lateCount[li], o->lateArgs[li]);           /* :377-389 */

    /* RenderInstances :406-503 -> AS_Main per group */
    OrcBasePassConstants bk;
    memset(&bk, 0, sizeof bk);
    bk.m_WorldToView = d->worldToView;                                     /* :448 culling view */
    memcpy(bk.m_Frustum, frustum, sizeof bk.m_Frustum);
    bk.m_CullingFlags = alphaMask ? (flags & ~(uint32_t)ORC_CULL_CONE) : flags;   /* :436-442 (Q8) */
    bk.m_HZBDimensions[0] = k.m_HZBDimensions[0]; bk.m_HZBDimensions[1] = k.m_HZBDimensions[1];
    bk.m_P00 = k.m_P00; bk.m_P11 = k.m_P11; bk.m_NearPlane = k.m_NearPlane;

    uint32_t G = o->dispatchArgs[slot][0] < o->validRecords[slot] ? o->dispatchArgs[slot][0] : o->validRecords[slot];
    uint32_t threads = d->threads < 1 ? 1 : (d->threads > 256 ? 256 : d->threads);
    uint64_t tested = 0;
    if (threads == 1 || G < 1024) {
        tested = orc_meshlet_cull(&bk, d->instances, d->meshData, d->meshlets, o->records[slot], 0, G, hzb, o->visMask[slot], 0, 0);
    } else {
        pthread_t th[256]; MeshletJob jobs[256];
        for (uint32_t i = 0; i < threads; ++i) {
            MeshletJob j = { &bk, d, o->records[slot], hzb, (uint32_t)((uint64_t)G * i / threads), (uint32_t)((uint64_t)G * (i + 1) / threads), o->visMask[slot], 0 };
            jobs[i] = j;
            pthread_create(&th[i], 0, meshlet_worker, &jobs[i]);
        }
        for (uint32_t i = 0; i < threads; ++i) { pthread_join(th[i], 0); tested += jobs[i].tested; }
    }
    o->meshletsTested[slot] = tested;
    /* ordered compaction (basepass.hlsl:111-121): groups ascending, lanes ascending */
    uint64_t cur = 0;
    for (uint32_t g = 0; g < G; ++g) {
        uint32_t m = o->visMask[slot][g];
        while (m) {
            uint32_t lane = (uint32_t)__builtin_ctz(m);
            if (cur < o->listCapacity) o->visibleList[slot][cur] = (g << 5) | lane;
            ++cur;
            m &= m - 1;
        }
    }
    o->drawArgs[slot][0] = (uint32_t)cur; o->drawArgs[slot][1] = 1; o->drawArgs[slot][2] = 1;

    if (d->rasterDepth) {                                                  /* MS_Main + depth test of the same DispatchMeshIndirect */
        bk.m_WorldToClip = d->worldToClip;
        bk.m_OutputResolution[0] = d->depthW; bk.m_OutputResolution[1] = d->depthH;
        const uint32_t n = cur < o->listCapacity ? (uint32_t)cur : (uint32_t)o->listCapacity;
        orc_raster_depth(&bk, d->instances, d->meshData, d->meshlets, d->vertices, d->meshletVertexIds, d->meshletTriangles,
                         o->records[slot], o->visibleList[slot], n, (float*)d->depth);
    }
}

void orc_frame(OrcFrameDesc* d, OrcFrameOut* o)
{
    /* BasePassRenderers.cpp:551-563 */
    uint32_t flags = d->cullingFlags & 7u;
    const int occlusion = (flags & ORC_CULL_OCCLUSION) != 0;
    float frustum[4];
    orc_culling_frustum(&d->viewToClip, frustum);

    OrcHZB hzb;
    memset(&hzb, 0, sizeof hzb);
    hzb.width = d->hzbW; hzb.height = d->hzbH; hzb.mips = d->hzbMips; hzb.texels = d->hzbTexels;
    memcpy(hzb.mipOffset, d->hzbMipOffset, sizeof hzb.mipOffset);

    for (int s = 0; s < 4; ++s) { o->passRan[s] = 0; o->meshletsTested[s] = 0; memset(o->drawArgs[s], 0, 12); memset(o->dispatchArgs[s], 0, 12); o->validRecords[s] = 0; }

    if (d->rasterDepth)
        memset((float*)d->depth, 0, sizeof(float) * (size_t)d->depthW * d->depthH);

    run_pass(d, o, 0, 0, 0, flags, frustum, &hzb);                          /* :565-566 */
    if (occlusion) {
        if (!d->freezeCullingCamera)                                        /* :507-510, :570 */
            hzb_build_mt(d->depth, d->depthW, d->depthH, d->hzbTexels, d->hzbW, d->hzbH, d->hzbMips, d->hzbMipOffset, d->threads);
        run_pass(d, o, 1, 1, 0, flags, frustum, &hzb);                      /* :572-573 */
        run_pass(d, o, 2, 0, 1, flags, frustum, &hzb);                      /* :575-576 (Q12) */
        run_pass(d, o, 3, 1, 1, flags, frustum, &hzb);                      /* :577-578 */
        if (!d->freezeCullingCamera)                                        /* :580 */
            hzb_build_mt(d->depth, d->depthW, d->depthH, d->hzbTexels, d->hzbW, d->hzbH, d->hzbMips, d->hzbMipOffset, d->threads);
    } else {
        run_pass(d, o, 2, 0, 1, flags, frustum, &hzb);                      /* :585-586 */
    }
}
