"""Independent numpy restatement of the visibility path, written from the formula-level
specification in SURVEY.md section 10 (NOT from oracle/tr_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED.  Its single job is to catch transcription errors in
the C oracle: tests/test_oracle_crosscheck.py demands bit-identical results from both on random
scenes.  Vectorised float32 numpy; fmaf is emulated exactly (double product + round-to-odd sum +
one rounding to float32), so the arithmetic convention of tr_oracle.h holds here too.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def fma(a, b, c):
    """Exact float32 fma: a*b is exact in float64; the float64 sum is rounded to odd (sticky bit)
    so that the final float64->float32 rounding is the single correct rounding."""
    a = np.asarray(a, np.float32).astype(np.float64)
    b = np.asarray(b, np.float32).astype(np.float64)
    c = np.asarray(c, np.float32).astype(np.float64)
    p = a * b
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)          # TwoSum residual, exact
    s, err = np.broadcast_arrays(s, err)
    s = s.copy()
    inexact = (err != 0) & np.isfinite(s)
    bits = s.view(np.int64)
    even = (bits & 1) == 0
    adj = inexact & even
    toward = np.where(err > 0, np.inf, -np.inf)
    s_adj = np.nextafter(s, toward)
    s = np.where(adj, s_adj, s)
    return s.astype(np.float32)


def dot3(a, b):
    return fma(a[..., 2], b[..., 2], fma(a[..., 1], b[..., 1], a[..., 0] * b[..., 0]))


def cross3(a, b):
    return np.stack([fma(a[..., 1], b[..., 2], -(a[..., 2] * b[..., 1])),
                     fma(a[..., 2], b[..., 0], -(a[..., 0] * b[..., 2])),
                     fma(a[..., 0], b[..., 1], -(a[..., 1] * b[..., 0]))], axis=-1)


def mul_point(p, M):
    """(p,1) * M -> xyz.  p [...,3], M [...,4,4] (broadcast)."""
    return np.stack([fma(p[..., 2], M[..., 2, j], fma(p[..., 1], M[..., 1, j], p[..., 0] * M[..., 0, j])) + M[..., 3, j]
                     for j in range(3)], axis=-1).astype(F)


def mul_vec3(v, R):
    """v * R3x3 with R [...,3,3]"""
    return np.stack([fma(v[..., 2], R[..., 2, j], fma(v[..., 1], R[..., 1, j], v[..., 0] * R[..., 0, j]))
                     for j in range(3)], axis=-1).astype(F)


def max_scale(W):
    d = np.stack([dot3(W[..., i, :3], W[..., i, :3]) for i in range(3)], axis=-1)
    return np.sqrt(np.max(d, axis=-1)).astype(F)


def to_view(p, V):
    o = mul_point(p, V)
    o[..., 2] = -o[..., 2]
    return o


def frustum_visible(c, r, f):
    a = fma(c[..., 2], f[1], np.abs(c[..., 0]) * f[0]) < r
    b = fma(c[..., 2], f[3], np.abs(c[..., 1]) * f[2]) < r
    return a & b


def f16_to_f32(h):
    return np.asarray(h, np.uint16).view(np.float16).astype(np.float32)


def hzb_level(width, height, mips):
    m = np.maximum(width, height)          # no NaNs reach here (uv are clamped)
    ok = m >= 1
    e = ((m.view(np.uint32) >> 23) & 0xFF).astype(np.int64) - 127
    return np.where(ok, np.minimum(e, mips - 1), 0).astype(np.int64)


def sample_min(hzb, u, v, level):
    """hzb: object with .w .h .mips .offsets .texels (uint16)."""
    u = np.asarray(u, F); v = np.asarray(v, F)
    out = np.empty(u.shape, F)
    for mip in np.unique(level):
        sel = level == mip
        mw, mh = max(hzb.w >> int(mip), 1), max(hzb.h >> int(mip), 1)
        t = f16_to_f32(hzb.texels[hzb.offsets[int(mip)]:hzb.offsets[int(mip)] + mw * mh]).reshape(mh, mw)
        fx = fma(u[sel], F(mw), F(-0.5)); fy = fma(v[sel], F(mh), F(-0.5))
        flx = np.floor(fx); fly = np.floor(fy)
        wx1 = (fx - flx) > 0; wy1 = (fy - fly) > 0
        x0 = flx.astype(np.int64); y0 = fly.astype(np.int64)
        x1 = np.clip(x0 + 1, 0, mw - 1); y1 = np.clip(y0 + 1, 0, mh - 1)
        x0 = np.clip(x0, 0, mw - 1); y0 = np.clip(y0, 0, mh - 1)
        inf = F(np.inf)
        d = t[y0, x0]
        d = np.minimum(d, np.where(wx1, t[y0, x1], inf))
        d = np.minimum(d, np.where(wy1, t[y1, x0], inf))
        d = np.minimum(d, np.where(wx1 & wy1, t[y1, x1], inf))
        out[sel] = d
    return out


def occlusion_visible(c, r, near, P00, P11, hzb):
    c = np.asarray(c, F); r = np.asarray(r, F)
    near, P00, P11 = F(near), F(P00), F(P11)
    vis = np.ones(r.shape, bool)
    test = ~((c[..., 2] - near) < r)
    if not test.any():
        return vis
    cc = c[test]; rr = r[test]
    with np.errstate(all="ignore"):
        cr = cc * rr[:, None]
        czr2 = fma(cc[:, 2], cc[:, 2], -(rr * rr))
        vx = np.sqrt(fma(cc[:, 0], cc[:, 0], czr2)).astype(F)
        minx = fma(vx, cc[:, 0], -cr[:, 2]) / fma(vx, cc[:, 2], cr[:, 0])
        maxx = fma(vx, cc[:, 0], cr[:, 2]) / fma(vx, cc[:, 2], -cr[:, 0])
        vy = np.sqrt(fma(cc[:, 1], cc[:, 1], czr2)).astype(F)
        miny = fma(vy, cc[:, 1], -cr[:, 2]) / fma(vy, cc[:, 2], cr[:, 1])
        maxy = fma(vy, cc[:, 1], cr[:, 2]) / fma(vy, cc[:, 2], -cr[:, 1])

        def clamp(x):
            return np.fmin(np.fmax(x, F(-1)), F(1))
        ax = fma(clamp(minx * P00), F(0.5), F(0.5)); ay = fma(clamp(miny * P11), F(-0.5), F(0.5))
        az = fma(clamp(maxx * P00), F(0.5), F(0.5)); aw = fma(clamp(maxy * P11), F(-0.5), F(0.5))
        width = (az - ax) * F(hzb.w); height = (aw - ay) * F(hzb.h)
        level = hzb_level(width.astype(F), height.astype(F), hzb.mips)
        depth = sample_min(hzb, ((ax + az) * F(0.5)).astype(F), ((ay + aw) * F(0.5)).astype(F), level)
        depth_sphere = near / (cc[:, 2] - rr)
        vis[test] = depth_sphere >= depth
    return vis


def cone_axis_view(packed, W, V):
    packed = np.asarray(packed, np.uint32)
    q = np.stack([((packed >> (8 * i)) & 0xFF).astype(F) / F(255.0) for i in range(4)], axis=-1).astype(F)
    a = fma(q[..., :3], F(2.0), F(-1.0))
    adj = np.stack([cross3(W[..., 1, :3], W[..., 2, :3]), cross3(W[..., 2, :3], W[..., 0, :3]),
                    cross3(W[..., 0, :3], W[..., 1, :3])], axis=-2)
    t = mul_vec3(a, adj)
    with np.errstate(all="ignore"):
        ln = np.sqrt(dot3(t, t)).astype(F)
        t = (t / ln[..., None]).astype(F)
    ax = mul_vec3(t, V[..., :3, :3])
    ax[..., 2] = -ax[..., 2]
    return ax, q[..., 3]


def cone_backfacing(c, r, axis, cutoff):
    return dot3(c, axis) >= fma(cutoff, np.sqrt(dot3(c, c)).astype(F), r)


# --------------------------------------------------------------------------------------- passes
def instance_pass(k, late, instances, ids, meshData, hzb, dispatchArgsX0=0, lateCount=0, lateIds=None,
                  lateArgsX=0, maxGroups=65535):
    """SURVEY 10.3.  k: dict of GPUCullingPassConstants fields.  Returns dict(records[G,3], argsX,
    valid, lateIds (appended), lateCount)."""
    flags = int(k["cullingFlags"])
    if late:
        n = min(int(lateCount), int(lateArgsX) * 32)
        tid_ids = np.asarray(lateIds[:n], np.uint32)
    else:
        tid_ids = np.asarray(ids[:int(k["nbInstances"])], np.uint32)
    inst = instances[tid_ids]
    mesh = meshData[inst[inst.dtype.names[2]]]
    W = inst[inst.dtype.names[0]]
    sph = mesh[mesh.dtype.names[0]]
    lods = mesh[mesh.dtype.names[1]]
    numLODs = mesh[mesh.dtype.names[2]]
    ms = max_scale(W)
    wc = mul_point(sph[:, :3], W)
    wr = (sph[:, 3] * ms).astype(F)
    V, PV = np.asarray(k["worldToView"], F), np.asarray(k["prevWorldToView"], F)
    cv = to_view(wc, V)
    alive = np.ones(len(tid_ids), bool)
    if not late and (flags & 1):
        alive &= frustum_visible(cv, wr, np.asarray(k["frustum"], F))
    submit = alive.copy()
    to_late = np.zeros(len(tid_ids), bool)
    if flags & 2:
        if not late:
            cv = to_view(wc, PV)
        occ = occlusion_visible(cv, wr, k["nearPlane"], k["P00"], k["P11"], hzb)
        submit = alive & occ
        if not late:
            to_late = alive & ~occ
    # LOD
    forced = int(k["forcedMeshLOD"])
    if forced != 0xFF:
        lod = np.minimum(np.uint32(forced), numLODs - 1).astype(np.int64)
    else:
        with np.errstate(all="ignore"):
            dist = np.maximum(np.sqrt(dot3(cv, cv)).astype(F) - wr, F(0))
            thr = ((dist * F(k["meshLODTarget"])).astype(F) / ms).astype(F)
        lod = np.zeros(len(tid_ids), np.int64)
        err = lods[lods.dtype.names[2]]
        for i in range(1, 8):
            hit = (i < numLODs) & (err[:, i] < thr)
            lod = np.where(hit, i, lod)
    nm = lods[lods.dtype.names[1]][np.arange(len(tid_ids)), lod]
    groups = ((nm.astype(np.uint64) + 31) // 32).astype(np.int64)
    # ordered emission
    records = []
    X = int(dispatchArgsX0)
    valid = None
    out_late = list(np.asarray(lateIds[:int(lateCount)], np.uint32)) if (lateIds is not None and not late) else []
    lc = int(lateCount)
    for t in range(len(tid_ids)):
        if to_late[t]:
            out_late.append(int(tid_ids[t])); lc += 1
        elif submit[t]:
            g = int(groups[t]); off = X; X += g
            if off + g >= maxGroups:
                if valid is None and g:
                    valid = off
                continue
            for i in range(g):
                records.append((int(tid_ids[t]), int(lod[t]), i * 32))
    return dict(records=np.array(records, np.uint32).reshape(-1, 3), argsX=X, valid=X if valid is None else valid,
                lateIds=np.array(out_late, np.uint32), lateCount=lc)


def meshlet_pass(k, instances, meshData, meshlets, records, hzb):
    """SURVEY 10.4.  records [G,3] uint32.  Returns (visMask[G], visibleList)."""
    flags = int(k["cullingFlags"])
    G = len(records)
    if G == 0:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32)
    rec_inst = records[:, 0]; rec_lod = records[:, 1].astype(np.int64); rec_off = records[:, 2]
    inst = instances[rec_inst]
    W = inst[inst.dtype.names[0]]
    mesh = meshData[inst[inst.dtype.names[2]]]
    lods = mesh[mesh.dtype.names[1]]
    base = lods[lods.dtype.names[0]][np.arange(G), rec_lod].astype(np.int64)
    nm = lods[lods.dtype.names[1]][np.arange(G), rec_lod].astype(np.int64)
    lane = np.arange(32)[None, :]
    m = rec_off[:, None].astype(np.int64) + lane
    active = m < nm[:, None]
    idx = np.where(active, base[:, None] + m, 0)
    md = meshlets[idx]
    sph = md[md.dtype.names[0]]
    Wb = np.broadcast_to(W[:, None], (G, 32, 4, 4))
    V = np.asarray(k["worldToView"], F)
    cw = mul_point(sph[..., :3], Wb)
    cv = to_view(cw, V)
    r = (sph[..., 3] * max_scale(W)[:, None]).astype(F)
    vis = active.copy()
    if flags & 1:
        vis &= frustum_visible(cv, r, np.asarray(k["frustum"], F))
    if flags & 2:
        sel = vis.copy()
        o = occlusion_visible(cv[sel], r[sel], k["nearPlane"], k["P00"], k["P11"], hzb)
        vis[sel] = o
    if flags & 4:
        axis, cutoff = cone_axis_view(md[md.dtype.names[1]], Wb, V)
        with np.errstate(all="ignore"):
            vis &= ~cone_backfacing(cv, r, axis, cutoff)
    mask = (vis.astype(np.uint64) << np.arange(32, dtype=np.uint64)[None, :]).sum(axis=1).astype(np.uint32)
    g_idx, l_idx = np.nonzero(vis)
    lst = ((g_idx.astype(np.uint32) << 5) | l_idx.astype(np.uint32)).astype(np.uint32)
    return mask, lst


def hzb_build(depth, hw, hh, mips, offsets):
    """SURVEY 10.5."""
    H, W = depth.shape
    total = offsets[-1] + max(hw >> (mips - 1), 1) * max(hh >> (mips - 1), 1)
    tex = np.zeros(total, np.uint16)
    xs = np.arange(hw); ys = np.arange(hh)
    u = ((xs.astype(F) + F(0.5)) / F(hw)).astype(F); v = ((ys.astype(F) + F(0.5)) / F(hh)).astype(F)
    fx = fma(u, F(W), F(-0.5)); fy = fma(v, F(H), F(-0.5))
    x0 = np.floor(fx).astype(np.int64); y0 = np.floor(fy).astype(np.int64)
    x1 = np.clip(x0 + 1, 0, W - 1); y1 = np.clip(y0 + 1, 0, H - 1)
    x0 = np.clip(x0, 0, W - 1); y0 = np.clip(y0, 0, H - 1)
    m = np.minimum(np.minimum(depth[y0][:, x0], depth[y0][:, x1]), np.minimum(depth[y1][:, x0], depth[y1][:, x1]))
    cur = m.astype(np.float16)
    tex[offsets[0]:offsets[0] + hw * hh] = cur.view(np.uint16).ravel()
    for k in range(1, mips):
        ph, pw = cur.shape
        mw, mh = max(hw >> k, 1), max(hh >> k, 1)
        xa = np.minimum(2 * np.arange(mw), pw - 1); xb = np.minimum(2 * np.arange(mw) + 1, pw - 1)
        ya = np.minimum(2 * np.arange(mh), ph - 1); yb = np.minimum(2 * np.arange(mh) + 1, ph - 1)
        nxt = np.minimum(np.minimum(cur[ya][:, xa], cur[ya][:, xb]), np.minimum(cur[yb][:, xa], cur[yb][:, xb]))
        tex[offsets[k]:offsets[k] + mw * mh] = nxt.view(np.uint16).ravel()
        cur = nxt
    return tex
