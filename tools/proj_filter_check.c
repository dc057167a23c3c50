/* proj_filter_check.c -- CPU check of the FILTERED PROJECTION's error bound and decisions (cull_math.hip.h: projBands,
 * projectFiltered, occTailQuadFiltered).  Diagnostic tool, no GPU:
 *   gcc -O2 -fopenmp -ffp-contract=off tools/proj_filter_check.c -lm -o /tmp/pfc && /tmp/pfc [millions of samples]
 *
 * For random and adversarial view-space spheres it evaluates
 *   (a) the reference chain (culling.hlsli:53-78 under the build's arithmetic convention: fmaf where the convention says so,
 *       correctly rounded sqrt and division) -> level, footprint origin (x0, y0), zero-weight flags;
 *   (b) the fast chain exactly as the kernel issues it, with v_rsq_f32 / v_rcp_f32 modelled as ANY float within one ulp of
 *       the correctly rounded value (picked at random per call: -1, 0, +1 ulp -- up to 1.5 ulp from the real value, more
 *       than the hardware's 1 ulp);
 * and checks: every lane the kernel would call SURE has the same five integers; the real-valued differences stay inside
 * the proven bounds (reports the largest observed fraction of each bound); how many lanes are not sure. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

typedef struct { uint64_t s; } Rng;
static inline uint64_t rnd(Rng* r) { r->s ^= r->s << 13; r->s ^= r->s >> 7; r->s ^= r->s << 17; return r->s; }
static inline double uni(Rng* r) { return (double)(rnd(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline float ulp_step(float x, int k) { return u2f(f2u(x) + (uint32_t)k); }      /* positive finite x */
/* perturbation of the modelled v_rsq / v_rcp: per call random in {-1, 0, +1} ulp, or (bias != 0) the same direction in every call of a sample */
static __thread int g_bias[3];
static inline int pert(Rng* r, int which) { return g_bias[which] ? g_bias[which] : (int)(rnd(r) % 3) - 1; }
static inline float hw_rsq(Rng* r, float x, int which) { return ulp_step((float)(1.0 / sqrt((double)x)), pert(r, which)); }
static inline float hw_rcp(Rng* r, float x) { float v = (float)(1.0 / (double)x); return x > 0 ? ulp_step(v, pert(r, 2)) : -ulp_step(-v, pert(r, 2)); }

typedef struct { float invB[2], K[2]; uint32_t mipDelta; float mFloor; } Bands;
#define PROJ_MARGIN 1.0625f
static Bands proj_bands(float P00, float P11, uint32_t W, uint32_t H)        /* == cm::projBands */
{
    const float u = 0x1p-24f;
    Bands b;
    const float P[2] = { fabsf(P00), fabsf(P11) }, dim[2] = { (float)W, (float)H };
    float wBand = 0.f;
    for (int i = 0; i < 2; ++i) {
        const float B = 1.125f / P[i] + 0.25f;
        const float qmax = 1.0f / P[i] + 0x1p-10f;
        const float Cq = PROJ_MARGIN * (0.51f * sqrtf(B * B + 1.0f) + 5.11f * B + 0.15f + 12.3f * qmax);
        const float E1 = P[i] * Cq + 2.0f;
        b.K[i] = (E1 + 12.0f) * u;
        b.invB[i] = 1.0f / B;
        wBand = fmaxf(wBand, dim[i] * (E1 + 6.0f) * u);
    }
    const float d = wBand * 0x1p23f;
    b.mipDelta = d < 0x1p21f ? (uint32_t)d + 1u : 0x200000u;
    b.mFloor = 1.0f + (float)(b.mipDelta + 4u) * 0x1p-23f;
    return b;
}

typedef struct { int level, x0, y0, zx, zy; double f[2], m; } Foot;

/* (a) reference: oracle/tr_oracle.c occlusion_sample_position + sample_hzb_min_mip's footprint */
static Foot reference(const float c[3], float r, float P00, float P11, uint32_t W, uint32_t H, uint32_t mips)
{
    float cr[3] = { c[0] * r, c[1] * r, c[2] * r };
    float czr2 = fmaf(c[2], c[2], -(r * r));
    float vx = sqrtf(fmaf(c[0], c[0], czr2));
    float minx = fmaf(vx, c[0], -cr[2]) / fmaf(vx, c[2], cr[0]);
    float maxx = fmaf(vx, c[0], cr[2]) / fmaf(vx, c[2], -cr[0]);
    float vy = sqrtf(fmaf(c[1], c[1], czr2));
    float miny = fmaf(vy, c[1], -cr[2]) / fmaf(vy, c[2], cr[1]);
    float maxy = fmaf(vy, c[1], cr[2]) / fmaf(vy, c[2], -cr[1]);
    float ax = fmaf(clampf(minx * P00, -1.f, 1.f), 0.5f, 0.5f), ay = fmaf(clampf(miny * P11, -1.f, 1.f), -0.5f, 0.5f);
    float az = fmaf(clampf(maxx * P00, -1.f, 1.f), 0.5f, 0.5f), aw = fmaf(clampf(maxy * P11, -1.f, 1.f), -0.5f, 0.5f);
    float width = (az - ax) * (float)W, height = (aw - ay) * (float)H;
    Foot o;
    float m = fmaxf(width, height);
    o.m = m;
    if (!(m >= 1.0f)) o.level = 0;
    else { int e = (int)((f2u(m) >> 23) & 0xFFu) - 127; o.level = e > (int)mips - 1 ? (int)mips - 1 : e; }
    float u = (ax + az) * 0.5f, v = (ay + aw) * 0.5f;
    uint32_t mw = (W >> o.level) ? (W >> o.level) : 1u, mh = (H >> o.level) ? (H >> o.level) : 1u;
    float fx = fmaf(u, (float)mw, -0.5f), fy = fmaf(v, (float)mh, -0.5f);
    float flx = floorf(fx), fly = floorf(fy);
    o.x0 = (int)flx; o.y0 = (int)fly;
    o.zx = !((fx - flx) > 0.0f); o.zy = !((fy - fly) > 0.0f);
    o.f[0] = fx; o.f[1] = fy;
    return o;
}

/* (b) the kernel's fast chain: cm::projectFiltered + cm::occTailQuadFiltered */
static Foot fast(Rng* g, const float c[3], float r, float P00, float P11, uint32_t W, uint32_t H, uint32_t mips, const Bands* b, int* sure,
                 double* mnmx /* minx, miny, maxx, maxy */)
{
    const float Z = fmaf(c[2], c[2], -(r * r));
    const float X[2] = { fmaf(c[0], c[0], Z), fmaf(c[1], c[1], Z) };
    const float rD = hw_rcp(g, Z);
    float mn[2], mx[2];
    for (int i = 0; i < 2; ++i) {
        const float y = hw_rsq(g, X[i], i);
        const float vv = X[i] * y;
        const float a = c[i] * c[2];
        const float mnN = fmaf(-vv, r, a), mxN = fmaf(vv, r, a);
        mn[i] = mnN * rD; mx[i] = mxN * rD;
    }
    mnmx[0] = mn[0]; mnmx[1] = mn[1]; mnmx[2] = mx[0]; mnmx[3] = mx[1];
    const float h0 = c[0] * b->invB[0], h1 = c[1] * b->invB[1];
    const float m3 = fmaxf(fmaxf(fabsf(h0), fabsf(h1)), fabsf(r) * 8.0f);
    *sure = (m3 <= c[2]) && (c[2] <= 0x1p30f);                             /* (the kernel bounds c.z through the cone's c.c <= 2^60 when the cone test is on) */
    const float P[2] = { P00, P11 }, sgn[2] = { 0.5f, -0.5f }, dim[2] = { (float)W, (float)H };
    float lo[2], hi[2], wh[2];
    for (int i = 0; i < 2; ++i) {
        lo[i] = fmaf(clampf(mn[i] * P[i], -1.f, 1.f), sgn[i], 0.5f);
        hi[i] = fmaf(clampf(mx[i] * P[i], -1.f, 1.f), sgn[i], 0.5f);
        wh[i] = (hi[i] - lo[i]) * dim[i];
    }
    const float m = fmaxf(fmaxf(wh[0], wh[1]), b->mFloor);
    int e; frexpf(m, &e);
    e = e < (int)mips ? e : (int)mips;
    e = e > 1 ? e : 1;
    Foot o;
    o.level = e - 1; o.m = m;
    const uint32_t mw = (W >> o.level) ? (W >> o.level) : 1u, mh = (H >> o.level) ? (H >> o.level) : 1u;
    const float half[2] = { 0.5f * (float)mw, 0.5f * (float)mh };
    float f[2], fl[2];
    for (int i = 0; i < 2; ++i) {
        f[i] = fmaf(lo[i] + hi[i], half[i], -0.5f);
        fl[i] = floorf(f[i]);
        const float t = f[i] - (fl[i] + 0.5f);
        const float cap = fmaf(half[i], -b->K[i], 0.5f);
        *sure &= fabsf(t) <= cap;
        o.f[i] = f[i];
    }
    o.x0 = (int)fl[0]; o.y0 = (int)fl[1];
    o.zx = o.zy = 0;                                                        /* a sure lane has fractions > 0 */
    const uint32_t de = (b->mipDelta >> (e - 1)) + 4u;                       /* == cm::projMipDelta */
    const uint32_t mb = f2u(m) + de;
    *sure &= (mb & 0x7FFFFFu) >= 2u * de;
    return o;
}

int main(int argc, char** argv)
{
    const long millions = argc > 1 ? atol(argv[1]) : 200;
    const long total = millions * 1000000L;
    long bad = 0, sureN = 0, relevant = 0, anyLevel = 0, anyFoot = 0;
    double worstF = 0, worstM = 0, worstQ = 0;
    static const uint32_t dims[][2] = { { 2048, 2048 }, { 2048, 1024 }, { 4096, 2048 }, { 512, 256 }, { 1024, 1024 } };
#pragma omp parallel reduction(+ : bad, sureN, relevant, anyLevel, anyFoot) reduction(max : worstF, worstM, worstQ)
    {
        Rng g;
#ifdef _OPENMP
        g.s = 0x9E3779B97F4A7C15ull * (uint64_t)(omp_get_thread_num() + 1);
#else
        g.s = 0x9E3779B97F4A7C15ull;
#endif
#pragma omp for schedule(static)
        for (long it = 0; it < total; ++it) {
            const int cfg = (int)(rnd(&g) % 5);
            const uint32_t W = dims[cfg][0], H = dims[cfg][1];
            uint32_t mips = 1; while ((W >> mips) || (H >> mips)) ++mips;
            const float P11 = (float)(0.5 + 3.0 * uni(&g)), P00 = P11 * (float)(0.4 + 0.8 * uni(&g));
            const Bands b = proj_bands(P00, P11, W, H);
            const double cz = exp(log(0.05) + uni(&g) * log(1e5 / 0.05));
            const int mode = (int)(rnd(&g) % 8);
            { const uint64_t k = rnd(&g); for (int j = 0; j < 3; ++j) g_bias[j] = (k >> (8 * j) & 1) ? 0 : ((k >> (8 * j + 1) & 1) ? 1 : -1); }   /* half of the calls biased */
            double rho = uni(&g) * 0.14;                                   /* around the 1/8 precondition */
            if (mode == 1) rho = 0.125 * (1.0 - 1e-6 * uni(&g));
            if (mode == 2) rho = uni(&g) * 0.01;
            double bx = (2.0 * uni(&g) - 1.0) * (1.2 / P00 + 0.3), by = (2.0 * uni(&g) - 1.0) * (1.2 / P11 + 0.3);
            if (mode == 3) bx = (rnd(&g) & 1 ? 1 : -1) * (1.0 / P00) * (1.0 + 1e-5 * (uni(&g) - 0.5));     /* minx / maxx at the clamp */
            if (mode == 4) { bx *= 1e-3; by *= 1e-3; }                      /* the centre of the screen: cancellation in cx cz - v r */
            float c[3] = { (float)(bx * cz), (float)(by * cz), (float)cz };
            float r = (float)(rho * cz);
            if (mode == 5) {                                                /* put f.x next to an integer: nudge cx by bisection on the reference */
                float lo_ = c[0] * 0.999f - 1e-6f * c[2], hi_ = c[0] * 1.001f + 1e-6f * c[2];
                if (lo_ > hi_) { float t = lo_; lo_ = hi_; hi_ = t; }
                float cc[3] = { lo_, c[1], c[2] };
                Foot a0 = reference(cc, r, P00, P11, W, H, mips);
                for (int k = 0; k < 40; ++k) {
                    float mid = 0.5f * (lo_ + hi_);
                    cc[0] = mid;
                    Foot am = reference(cc, r, P00, P11, W, H, mips);
                    if (am.x0 == a0.x0 && am.level == a0.level) lo_ = mid; else hi_ = mid;
                }
                c[0] = ulp_step(fabsf(lo_) > 0 ? fabsf(lo_) : 1e-30f, (int)(rnd(&g) % 41) - 20) * (lo_ < 0 ? -1.f : 1.f);
            }
            if (mode == 6) {                                                /* max(w, h) next to a power of two: nudge r */
                float lo_ = r * 0.5f, hi_ = r * 1.5f;
                Foot a0 = reference(c, lo_, P00, P11, W, H, mips);
                for (int k = 0; k < 40; ++k) {
                    float mid = 0.5f * (lo_ + hi_);
                    Foot am = reference(c, mid, P00, P11, W, H, mips);
                    if (am.level == a0.level) lo_ = mid; else hi_ = mid;
                }
                r = ulp_step(lo_ > 0 ? lo_ : 1e-30f, (int)(rnd(&g) % 41) - 20);
            }
            if (!(c[2] - r > 1e-3f * c[2])) continue;
            ++relevant;
            int sure;
            double q[4];
            const Foot R = reference(c, r, P00, P11, W, H, mips);
            const Foot F = fast(&g, c, r, P00, P11, W, H, mips, &b, &sure, q);
            if (R.level != F.level) ++anyLevel; else if (R.x0 != F.x0 || R.y0 != F.y0) ++anyFoot;   /* what a build without the bands would get wrong */
            if (!sure) continue;
            ++sureN;
            if (R.level != F.level || R.x0 != F.x0 || R.y0 != F.y0 || R.zx || R.zy) {
#pragma omp critical
                if (bad < 20) printf("MISMATCH c = (%a, %a, %a) r = %a P = (%a, %a) %ux%u: ref level %d (%d, %d) z %d%d f (%.9g, %.9g) m %.9g | fast level %d (%d, %d) f (%.9g, %.9g) m %.9g\n",
                                     c[0], c[1], c[2], r, P00, P11, W, H, R.level, R.x0, R.y0, R.zx, R.zy, R.f[0], R.f[1], R.m, F.level, F.x0, F.y0, F.f[0], F.f[1], F.m);
                ++bad;
                continue;
            }
            /* how much of each bound was used (same level, so f and m are comparable) */
            const uint32_t mw = (W >> R.level) ? (W >> R.level) : 1u, mh = (H >> R.level) ? (H >> R.level) : 1u;
            const double bf0 = b.K[0] * 0.5 * mw, bf1 = b.K[1] * 0.5 * mh;
            worstF = fmax(worstF, fmax(fabs(R.f[0] - F.f[0]) / bf0, fabs(R.f[1] - F.f[1]) / bf1));
            if (R.m >= 2.0) worstM = fmax(worstM, fabs(R.m - F.m) / (((double)(b.mipDelta >> R.level) + 4.0) * ldexp(1.0, R.level - 23)));
            (void)worstQ;
        }
    }
    printf("%ld samples, %ld with the sphere in front of the near plane, %ld sure (%.3f %% not sure)\n", total, relevant, sureN, 100.0 * (double)(relevant - sureN) / (double)relevant);
    printf("mismatches among the sure lanes: %ld\n", bad);
    printf("(without the bands: %ld spheres at another level, %ld at another footprint origin)\n", anyLevel, anyFoot);
    printf("largest observed |f - f'| / bound: %.3f;  largest |m - m'| / bound: %.3f\n", worstF, worstM);
    return bad != 0;
}
