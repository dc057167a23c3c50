#!/usr/bin/env python3
"""Generates tests/golden/kat_primitives.json: known-answer vectors for the primitives of the
visibility path, computed with EXACT rational arithmetic (fractions.Fraction) and one explicit
round-to-nearest-even to binary32 per operation of the arithmetic convention.

This is a third, scalar restatement of the formulas in SURVEY.md section 10 -- independent of
oracle/tr_oracle.c (C, hardware floats) and oracle/np_oracle.py (numpy).  The reference ships no
vectors of its own (PARITY UNPINNED), so these pin the oracle's arithmetic, not the reference's.

Run:  python tests/golden/make_kat.py   (rewrites the JSON; deterministic)
"""
import json
import math
import os
import random
import struct
from fractions import Fraction as Fr

HERE = os.path.dirname(os.path.abspath(__file__))


# ---- exact binary32 model ---------------------------------------------------------------------
def bits(f: float) -> int:
    return struct.unpack("<I", struct.pack("<f", f))[0]


def from_bits(u: int) -> float:
    return struct.unpack("<f", struct.pack("<I", u & 0xFFFFFFFF))[0]


def rn32(x: Fr) -> float:
    """Correctly rounded (RNE) binary32 of an exact rational."""
    if x == 0:
        return 0.0
    s = -1 if x < 0 else 1
    a = abs(x)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fr(2) ** e > a:
        e -= 1
    if Fr(2) ** (e + 1) <= a:
        e += 1
    e = max(e, -126)                     # subnormals share the exponent of the smallest normal
    q = a / Fr(2) ** (e - 23)            # in [2^23, 2^24) for normals
    m = q.numerator // q.denominator
    r = q - m
    if r > Fr(1, 2) or (r == Fr(1, 2) and (m & 1)):
        m += 1
    v = Fr(m) * Fr(2) ** (e - 23)
    if v >= Fr(2) ** 128:
        return s * math.inf
    return s * float(v)                  # exactly representable in binary64


def F(x: float) -> Fr:
    return Fr(x)


def fadd(a, b): return rn32(F(a) + F(b))
def fsub(a, b): return rn32(F(a) - F(b))
def fmul(a, b): return rn32(F(a) * F(b))
def ffma(a, b, c): return rn32(F(a) * F(b) + F(c))
def fdiv(a, b): return rn32(F(a) / F(b))


def fsqrt(a: float) -> float:
    if a == 0:
        return 0.0
    x = F(a)
    g = from_bits(bits(math.sqrt(a)))               # starting guess, then exact neighbour search
    g = rn32(Fr(g))
    lo = g
    while F(lo) * F(lo) > x:
        lo = from_bits(bits(lo) - 1)
    while True:
        up = from_bits(bits(lo) + 1)
        if F(up) * F(up) <= x:
            lo = up
        else:
            break
    up = from_bits(bits(lo) + 1)
    mid = (F(lo) + F(up)) / 2
    if x > mid * mid:
        return up
    if x < mid * mid:
        return lo
    return lo if (bits(lo) & 1) == 0 else up


def fmin(a, b): return a if a < b else b
def fmax(a, b): return a if a > b else b
def clamp(x, lo, hi): return fmin(fmax(x, lo), hi)


def dot3(a, b): return ffma(a[2], b[2], ffma(a[1], b[1], fmul(a[0], b[0])))
def cross3(a, b):
    return [ffma(a[1], b[2], -fmul(a[2], b[1])), ffma(a[2], b[0], -fmul(a[0], b[2])), ffma(a[0], b[1], -fmul(a[1], b[0]))]
def mul_point(p, M): return [fadd(ffma(p[2], M[2][j], ffma(p[1], M[1][j], fmul(p[0], M[0][j]))), M[3][j]) for j in range(3)]
def mul_vec(v, R): return [ffma(v[2], R[2][j], ffma(v[1], R[1][j], fmul(v[0], R[0][j]))) for j in range(3)]
def max_scale(W): return fsqrt(fmax(fmax(dot3(W[0][:3], W[0][:3]), dot3(W[1][:3], W[1][:3])), dot3(W[2][:3], W[2][:3])))
def to_view(p, V):
    o = mul_point(p, V)
    o[2] = -o[2]
    return o


def frustum_visible(c, r, f):
    return (ffma(c[2], f[1], fmul(abs(c[0]), f[0])) < r) and (ffma(c[2], f[3], fmul(abs(c[1]), f[2])) < r)


def f16_to_f32(h: int) -> float:
    return struct.unpack("<e", struct.pack("<H", h))[0]


def hzb_level(w, h, mips):
    m = fmax(w, h)
    if not (m >= 1.0):
        return 0
    return min(((bits(m) >> 23) & 0xFF) - 127, mips - 1)


def sample_min(hzb, u, v, mip):
    mw, mh = max(hzb["w"] >> mip, 1), max(hzb["h"] >> mip, 1)
    off = hzb["offsets"][mip]
    fx, fy = ffma(u, float(mw), -0.5), ffma(v, float(mh), -0.5)
    flx, fly = math.floor(fx), math.floor(fy)
    wx1, wy1 = fsub(fx, float(flx)) > 0, fsub(fy, float(fly)) > 0
    cl = lambda v_, m_: min(max(v_, 0), m_)
    x0, x1, y0, y1 = cl(flx, mw - 1), cl(flx + 1, mw - 1), cl(fly, mh - 1), cl(fly + 1, mh - 1)
    t = lambda y, x: f16_to_f32(hzb["texels"][off + y * mw + x])
    d = t(y0, x0)
    if wx1: d = fmin(d, t(y0, x1))
    if wy1:
        d = fmin(d, t(y1, x0))
        if wx1: d = fmin(d, t(y1, x1))
    return d


def occlusion_visible(c, r, near, P00, P11, hzb):
    if fsub(c[2], near) < r:
        return True
    cr = [fmul(c[i], r) for i in range(3)]
    czr2 = ffma(c[2], c[2], -fmul(r, r))
    vx = fsqrt(ffma(c[0], c[0], czr2))
    minx = fdiv(ffma(vx, c[0], -cr[2]), ffma(vx, c[2], cr[0]))
    maxx = fdiv(ffma(vx, c[0], cr[2]), ffma(vx, c[2], -cr[0]))
    vy = fsqrt(ffma(c[1], c[1], czr2))
    miny = fdiv(ffma(vy, c[1], -cr[2]), ffma(vy, c[2], cr[1]))
    maxy = fdiv(ffma(vy, c[1], cr[2]), ffma(vy, c[2], -cr[1]))
    ax = ffma(clamp(fmul(minx, P00), -1.0, 1.0), 0.5, 0.5)
    ay = ffma(clamp(fmul(miny, P11), -1.0, 1.0), -0.5, 0.5)
    az = ffma(clamp(fmul(maxx, P00), -1.0, 1.0), 0.5, 0.5)
    aw = ffma(clamp(fmul(maxy, P11), -1.0, 1.0), -0.5, 0.5)
    width = fmul(fsub(az, ax), float(hzb["w"]))
    height = fmul(fsub(aw, ay), float(hzb["h"]))
    level = hzb_level(width, height, hzb["mips"])
    depth = sample_min(hzb, fmul(fadd(ax, az), 0.5), fmul(fadd(ay, aw), 0.5), level)
    return fdiv(near, fsub(c[2], r)) >= depth


def cone_axis_view(packed, W, V):
    q = [fdiv(float((packed >> (8 * i)) & 0xFF), 255.0) for i in range(4)]
    a = [ffma(q[i], 2.0, -1.0) for i in range(3)]
    adj = [cross3(W[1][:3], W[2][:3]), cross3(W[2][:3], W[0][:3]), cross3(W[0][:3], W[1][:3])]
    t = mul_vec(a, adj)
    ln = fsqrt(dot3(t, t))
    t = [fdiv(t[i], ln) for i in range(3)]
    ax = mul_vec(t, [V[0][:3], V[1][:3], V[2][:3]])
    ax[2] = -ax[2]
    return ax, q[3]


def cone_backfacing(c, r, axis, cutoff):
    return dot3(c, axis) >= ffma(cutoff, fsqrt(dot3(c, c)), r)


def make_world(p, q, s):
    qx, qy, qz, qw = q
    qxx, qyy, qzz = fmul(qx, qx), fmul(qy, qy), fmul(qz, qz)
    two = lambda a: fmul(2.0, a)
    R = [[0.0] * 4 for _ in range(4)]
    R[0][0] = fsub(fsub(1.0, two(qyy)), two(qzz)); R[0][1] = fadd(fmul(two(qx), qy), fmul(two(qz), qw)); R[0][2] = fsub(fmul(two(qx), qz), fmul(two(qy), qw))
    R[1][0] = fsub(fmul(two(qx), qy), fmul(two(qz), qw)); R[1][1] = fsub(fsub(1.0, two(qxx)), two(qzz)); R[1][2] = fadd(fmul(two(qy), qz), fmul(two(qx), qw))
    R[2][0] = fadd(fmul(two(qx), qz), fmul(two(qy), qw)); R[2][1] = fsub(fmul(two(qy), qz), fmul(two(qx), qw)); R[2][2] = fsub(fsub(1.0, two(qxx)), two(qyy))
    R[3][3] = 1.0
    S = [[0.0] * 4 for _ in range(4)]
    S[0][0], S[1][1], S[2][2], S[3][3] = s[0], s[1], s[2], 1.0
    T = [[1.0 if i == j else 0.0 for j in range(4)] for i in range(4)]
    T[3][0], T[3][1], T[3][2] = p
    mm = lambda A, B: [[ffma(A[i][3], B[3][j], ffma(A[i][2], B[2][j], ffma(A[i][1], B[1][j], fmul(A[i][0], B[0][j])))) for j in range(4)] for i in range(4)]
    return mm(mm(R, S), T)


# ---- vector generation ----------------------------------------------------------------------------
def f32(x): return from_bits(bits(x))
def hexv(v): return [bits(x) for x in v]
def hexm(M): return [bits(x) for row in M for x in row]


def main():
    rnd = random.Random(0x5EED0001)
    U = lambda a, b: f32(rnd.uniform(a, b))
    out = {"about": "exact-rational known answers for the arithmetic convention (see make_kat.py); floats are IEEE-754 binary32 bit patterns"}

    def rand_matrix(scale_lo=0.5, scale_hi=2.0, trans=50.0):
        q = [rnd.gauss(0, 1) for _ in range(4)]
        n = math.sqrt(sum(x * x for x in q))
        q = [f32(x / n) for x in q]
        return make_world([U(-trans, trans), U(-trans, trans), U(-trans, -1.0)], q, [U(scale_lo, scale_hi) for _ in range(3)])

    V = make_world([U(-1, 1), U(-1, 1), U(-1, 1)], [0.0, f32(math.sin(0.01)), 0.0, f32(math.cos(0.01))], [1.0, 1.0, 1.0])
    P00, P11, near = f32(1.357995), f32(2.4142134), f32(0.1)
    fr = [f32(0.8052), f32(-0.5930), f32(0.9239), f32(-0.3827)]

    # fp16 conversion (expected from struct 'e' = IEEE RNE, an implementation independent of the oracle's)
    conv = [0.0, -0.0, 1.0, 65504.0, 65519.996, 65520.0, 1e-8, 5.9604645e-8, 2.9802322e-8, 2.98023259e-8, 6.1035156e-5, 6.097555e-5, 0.1, 0.333333, 0.99951172, 0.99975586, 1e-5, 3.14159, 1e6, -2.5]
    conv += [U(0, 1) for _ in range(40)] + [f32(rnd.uniform(0, 1) ** 8) for _ in range(40)]
    def to_h(x):
        try:
            return struct.unpack("<H", struct.pack("<e", x))[0]
        except OverflowError:
            return 0x7C00 | (0x8000 if x < 0 else 0)
    out["f32_to_f16"] = [{"in": bits(f32(x)), "out": to_h(f32(x))} for x in conv]

    out["hzb_level"] = [{"w": bits(f32(w)), "h": bits(f32(h)), "mips": m, "out": hzb_level(f32(w), f32(h), m)}
                        for w, h, m in [(0, 0, 12), (-3, -1, 12), (0.99, -5, 12), (1, 0, 12), (1.99, 0, 12), (2, -1, 12), (3.5, 7.9, 12), (2047.9, 0, 12),
                                        (2048, 0, 12), (5000, 1, 12), (1e30, 0, 12), (17, 0, 3), (float("inf"), 0, 12), (0.5, 900, 10)]]

    out["late_args"] = [{"count": c, "out": [(c + 63) // 64, 1, 1]} for c in (0, 1, 63, 64, 65, 127, 128, 129, 1000)]

    fcases = []
    for _ in range(60):
        c = [U(-30, 30), U(-20, 20), U(0.5, 80)]
        r = U(0.01, 3)
        fcases.append({"c": hexv(c), "r": bits(r), "f": hexv(fr), "out": int(frustum_visible(c, r, fr))})
    out["frustum"] = fcases

    mcases = []
    for _ in range(12):
        W = rand_matrix()
        sph = [U(-1, 1), U(-1, 1), U(-1, 1), U(0.1, 2)]
        wc = mul_point(sph[:3], W)
        mcases.append({"W": hexm(W), "sphere": hexv(sph), "maxScale": bits(max_scale(W)), "world": hexv(wc + [fmul(sph[3], max_scale(W))]),
                       "view": hexv(to_view(wc, V)), "V": hexm(V)})
    out["transform"] = mcases

    wcases = []
    for _ in range(8):
        p = [U(-10, 10) for _ in range(3)]
        q = [rnd.gauss(0, 1) for _ in range(4)]
        n = math.sqrt(sum(x * x for x in q))
        q = [f32(x / n) for x in q]
        s = [U(0.3, 3) for _ in range(3)]
        wcases.append({"p": hexv(p), "q": hexv(q), "s": hexv(s), "out": hexm(make_world(p, q, s))})
    out["make_world"] = wcases

    ccases = []
    for i in range(40):
        W = rand_matrix()
        packed = rnd.getrandbits(32) & 0xFEFFFFFF
        if i < 4:
            packed = [0x00000000, 0xFEFFFFFF, 0x007F7F7F, 0x80FF0080][i]
        ax, cut = cone_axis_view(packed, W, V)
        c = [U(-5, 5), U(-5, 5), U(1, 40)]
        r = U(0.01, 1.0)
        ccases.append({"packed": packed, "W": hexm(W), "V": hexm(V), "axis": hexv(ax), "cutoff": bits(cut), "c": hexv(c), "r": bits(r),
                       "backfacing": int(cone_backfacing(c, r, ax, cut))})
    out["cone"] = ccases

    # occlusion against a small random HZB (16 x 8, 5 mips built as 2x2 min)
    w, h = 16, 8
    mips = max(w, h).bit_length()
    offsets, tex = [], []
    prev = None
    for k in range(mips):
        mw, mh = max(w >> k, 1), max(h >> k, 1)
        offsets.append(len(tex))
        if k == 0:
            cur = [[to_h(f32(rnd.choice([0.0, rnd.uniform(0, 0.05), rnd.uniform(0, 0.01)]))) for _ in range(mw)] for _ in range(mh)]
        else:
            pw, ph = len(prev[0]), len(prev)
            cur = [[min((prev[min(2 * y + dy, ph - 1)][min(2 * x + dx, pw - 1)] for dy in (0, 1) for dx in (0, 1)), key=f16_to_f32)
                    for x in range(mw)] for y in range(mh)]
        tex += [t for row in cur for t in row]
        prev = cur
    hzb = {"w": w, "h": h, "mips": mips, "offsets": offsets, "texels": tex}
    ocases = []
    for i in range(80):
        c = [U(-3, 3), U(-2, 2), U(0.3, 30)]
        r = U(0.005, 0.8)
        if i < 6:   # near-plane accept, and spheres projecting to < 1 texel (Q6 -> level 0)
            c, r = [[U(-0.1, 0.1), U(-0.1, 0.1), f32(0.3)], f32(0.25)] if i < 3 else [[U(-1, 1), U(-1, 1), f32(25.0)], f32(0.004)]
        ocases.append({"c": hexv(c), "r": bits(r), "out": int(occlusion_visible(c, r, near, P00, P11, hzb))})
    out["occlusion"] = {"near": bits(near), "P00": bits(P00), "P11": bits(P11), "hzb": hzb, "cases": ocases}

    scases = []
    for i in range(40):
        u, v = U(-0.1, 1.1), U(-0.1, 1.1)
        if i < 8:
            u, v = f32((i % 4 + 0.5) / 16 * (2 ** (i // 4))), f32(0.5 / 8)      # exactly on texel centres: single-texel footprints
        mip = rnd.randrange(mips)
        scases.append({"u": bits(clamp(u, 0.0, 1.0)), "v": bits(clamp(v, 0.0, 1.0)), "mip": mip, "out": bits(sample_min(hzb, clamp(u, 0.0, 1.0), clamp(v, 0.0, 1.0), mip))})
    out["sample"] = scases

    with open(os.path.join(HERE, "kat_primitives.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", os.path.join(HERE, "kat_primitives.json"), {k: (len(v) if isinstance(v, list) else "...") for k, v in out.items()})


if __name__ == "__main__":
    main()
