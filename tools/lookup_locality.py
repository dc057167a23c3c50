#!/usr/bin/env python3
"""CPU statistics (no GPU): where the HZB table lookups of the early meshlet cull land, per window of the tile-ordered list.

Question behind it (profiles/r3/experiments.md section 8): how much LDS would a workgroup need to serve the
lookups of ONE window of records (128 records = 32 instances of C3) from a staged copy of the footprint-min table, per mip level
and per binning granularity of the instance pass?

For config C3 (bench.py's scene, camera and depth): instances that survive the early instance cull (frustum + previous-frame
HZB, float32 numpy -- statistics only, not the bit-exact oracle), binned by screen tile as k_gpuculling.hip::screenTile does
(stable, list order inside a tile); a sample of windows; for every meshlet of a window the lookup's mip level and footprint
origin (culling.hlsli:56-78).  Prints the distribution of levels, and for each {tiles per axis, lowest staged mip} the LDS bytes
of the window's bounding rectangle of 8 x 8 blocks per mip (median / 90th percentile) and the share of lookups a given budget
would serve.

  python tools/lookup_locality.py [--windows 60] [--records 128]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from toyrenderer_amd import synth, interop as I   # noqa: E402
from toyrenderer_amd import frame as F_            # noqa: E402
from oracle import np_oracle as O                 # noqa: E402  (statistics tool: test infrastructure, never shipped)

F = np.float32


def project(c, r, near, P00, P11, hw, hh, mips):
    """Lookup (level, X, Y) of spheres c (view space), r: culling.hlsli:56-78 in float32 (statistics)."""
    with np.errstate(all="ignore"):
        cr = c * r[:, None]
        czr2 = c[:, 2] * c[:, 2] - r * r
        vx = np.sqrt(c[:, 0] * c[:, 0] + czr2)
        minx = (vx * c[:, 0] - cr[:, 2]) / (vx * c[:, 2] + cr[:, 0])
        maxx = (vx * c[:, 0] + cr[:, 2]) / (vx * c[:, 2] - cr[:, 0])
        vy = np.sqrt(c[:, 1] * c[:, 1] + czr2)
        miny = (vy * c[:, 1] - cr[:, 2]) / (vy * c[:, 2] + cr[:, 1])
        maxy = (vy * c[:, 1] + cr[:, 2]) / (vy * c[:, 2] - cr[:, 1])
        cl = lambda x: np.clip(x, -1, 1)
        ax = cl(minx * P00) * 0.5 + 0.5; ay = cl(miny * P11) * -0.5 + 0.5
        az = cl(maxx * P00) * 0.5 + 0.5; aw = cl(maxy * P11) * -0.5 + 0.5
        w = (az - ax) * hw; h = (aw - ay) * hh
        m = np.maximum(np.maximum(w, h), 1.0)
        lvl = np.minimum(np.floor(np.log2(m)).astype(np.int64), mips - 1)
        u = (ax + az) * 0.5; v = (ay + aw) * 0.5
        mw = np.maximum(hw >> lvl, 1); mh = np.maximum(hh >> lvl, 1)
        X = np.floor(u * mw - 0.5).astype(np.int64) + 1
        Y = np.floor(v * mh - 0.5).astype(np.int64) + 1
    return lvl, X, Y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=60)
    ap.add_argument("--records", type=int, default=128)
    args = ap.parse_args()
    spec = synth.config_spec("C3")
    view = synth.make_view(eye=(0.0, 0.0, 0.0), prev_eye=(0.05, 0.0, 0.1), prev_yaw=0.002)
    depth = synth.gen_depth(view, 200)
    hw, hh = view.hzb_dims
    mips, offs, total = I.hzb_layout(hw, hh)

    class Hzb:
        pass
    hzb = Hzb(); hzb.w, hzb.h, hzb.mips, hzb.offsets = hw, hh, mips, offs
    hzb.texels = O.hzb_build(depth, hw, hh, mips, offs)
    md, _total = synth.gen_mesh_table(spec)
    inst = synth.gen_instances(spec)
    P = view.viewToClip
    P00, P11 = F(P[0, 0]), F(P[1, 1])
    near = F(view.nearPlane)
    W = inst["m_WorldMatrix"]
    sph = md["m_BoundingSphere"][inst["m_MeshDataIdx"]]
    ms = O.max_scale(W)
    wc = O.mul_point(sph[:, :3], W)
    wr = (sph[:, 3] * ms).astype(F)
    cv = O.to_view(wc, view.worldToView.astype(F))
    fr = F_.culling_frustum(view.viewToClip)
    alive = O.frustum_visible(cv, wr, fr.astype(F))
    cvp = O.to_view(wc, view.prevWorldToView.astype(F))
    occ = O.occlusion_visible(cvp, wr, near, P00, P11, hzb)
    keep = np.nonzero(alive & occ)[0]
    print(f"C3: {len(inst)} instances, {len(keep)} submitted by the early instance cull ({100.0 * len(keep) / len(inst):.1f} %)")

    groups_per_inst = spec.meshlets_lod0 // 32
    inst_per_window = args.records // groups_per_inst
    rng = np.random.default_rng(1)
    ckv = cv[keep]
    iz = 1.0 / np.maximum(ckv[:, 2], 1e-6)
    # NOTE screenTile works on +z = distance in front of the camera? (k_gpuculling.hip:127) -- view space looks down -Z here:
    zf = -ckv[:, 2] if np.median(ckv[:, 2]) < 0 else ckv[:, 2]
    iz = 1.0 / np.maximum(zf, 1e-6)
    u = ckv[:, 0] * iz * P00 * 0.5 + 0.5
    v = ckv[:, 1] * iz * P11 * -0.5 + 0.5

    chunk = spec.chunk_meshes
    ml_cache = {}

    def meshlets_of(mesh):
        b = (mesh // chunk) * chunk
        if b not in ml_cache:
            if len(ml_cache) > 2:
                ml_cache.clear()
            ml_cache[b] = synth.gen_meshlets_for_meshes(spec, md, b, min(b + chunk, spec.num_meshes))
        first = int(md["m_MeshLODDatas"]["m_MeshletDataBufferIdx"][mesh, 0]) - int(md["m_MeshLODDatas"]["m_MeshletDataBufferIdx"][b, 0])
        return ml_cache[b][first:first + spec.meshlets_lod0]

    levels_all = []
    for tiles in (16, 32, 64, 128):
        tx = np.clip((u * tiles).astype(np.int64), 0, tiles - 1)
        ty = np.clip((v * tiles).astype(np.int64), 0, tiles - 1)
        order = np.argsort(ty * tiles + tx, kind="stable")
        nwin = len(order) // inst_per_window
        pick = np.sort(rng.choice(nwin, size=min(args.windows, nwin), replace=False))
        need = {m0: [] for m0 in range(0, 4)}      # LDS bytes of the window's block rectangles, mips >= m0
        cover = {m0: [] for m0 in range(0, 4)}     # share of the window's lookups at mips >= m0
        rows = []
        for wdx in pick:
            ids = keep[order[wdx * inst_per_window:(wdx + 1) * inst_per_window]]
            ids_sorted = np.sort(ids)               # (mesh cache locality only)
            L = []; Xs = []; Ys = []
            for i in ids_sorted:
                ml = meshlets_of(int(inst["m_MeshDataIdx"][i]))
                s = ml["m_BoundingSphere"]
                c = O.to_view(O.mul_point(s[:, :3], W[i][None].repeat(len(s), 0)), view.worldToView.astype(F))
                r = (s[:, 3] * ms[i]).astype(F)
                # the lookup is only issued for meshlets that pass the near-plane accept test; frustum / cone ignored here
                lv, X, Y = project(c.astype(np.float64), r.astype(np.float64), float(near), float(P00), float(P11), hw, hh, mips)
                okz = np.isfinite(X) & np.isfinite(Y)
                L.append(lv[okz]); Xs.append(X[okz]); Ys.append(Y[okz])
            L = np.concatenate(L); Xs = np.concatenate(Xs); Ys = np.concatenate(Ys)
            if tiles == 16:
                levels_all.append(L)
            per_mip = {}
            for m in np.unique(L):
                sel = L == m
                bx0, bx1 = Xs[sel].min() >> 3, Xs[sel].max() >> 3
                by0, by1 = Ys[sel].min() >> 3, Ys[sel].max() >> 3
                per_mip[int(m)] = (int((bx1 - bx0 + 1) * (by1 - by0 + 1) * 128), int(sel.sum()))
            for m0 in need:
                need[m0].append(sum(b for m, (b, n) in per_mip.items() if m >= m0))
                cover[m0].append(sum(n for m, (b, n) in per_mip.items() if m >= m0) / len(L))
            rows.append(per_mip)
        print(f"\n{tiles} x {tiles} tiles, windows of {args.records} records = {inst_per_window} instances, {len(pick)} windows sampled")
        for m0 in need:
            a = np.array(need[m0]); c = np.array(cover[m0])
            print(f"   staging mips >= {m0}: LDS bytes median {int(np.median(a)):7d}  p90 {int(np.percentile(a, 90)):7d}  max {a.max():7d};"
                  f" lookups served {100 * c.mean():5.1f} %")
        # per-mip medians
        ms_ = sorted({m for r_ in rows for m in r_})
        print("   per mip (median bytes of the block rectangle, share of lookups): " +
              "  ".join(f"m{m}: {int(np.median([r_.get(m, (0, 0))[0] for r_ in rows]))} B {100 * np.mean([r_.get(m, (0, 0))[1] for r_ in rows]) / (inst_per_window * spec.meshlets_lod0):.0f}%" for m in ms_))
    L = np.concatenate(levels_all)
    print("\nlookup levels (share of meshlets): " + "  ".join(f"mip{m}: {100 * np.mean(L == m):.1f} %" for m in range(int(L.max()) + 1)))


if __name__ == "__main__":
    main()
