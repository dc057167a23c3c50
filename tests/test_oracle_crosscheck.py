"""C oracle (oracle/tr_oracle.c) vs the independent numpy restatement (oracle/np_oracle.py).

PARITY UNPINNED: the reference has no fixtures.  Two restatements written from
different texts (the HLSL vs SURVEY section 10) agreeing bit for bit is the guard against
transcription errors.  CPU only.
"""
import numpy as np
import pytest

from oracle import np_oracle as NP
from toyrenderer_amd import interop as I
from toyrenderer_amd import synth


def _consts(view, flags, nb, hzb_dims, forced=0xFF, oracle=None):
    k = np.zeros(1, I.GPUCullingPassConstants)
    k["m_NbInstances"] = nb
    k["m_CullingFlags"] = flags
    k["m_HZBDimensions"] = hzb_dims if flags & 2 else (1, 1)
    k["m_Frustum"] = oracle.culling_frustum(view.viewToClip)
    k["m_WorldToView"] = view.worldToView
    k["m_PrevWorldToView"] = view.prevWorldToView
    k["m_NearPlane"] = view.nearPlane
    k["m_P00"] = view.viewToClip[0, 0]
    k["m_P11"] = view.viewToClip[1, 1]
    k["m_ForcedMeshLOD"] = forced
    k["m_MeshLODTarget"] = np.float32(np.float32(2.0) / view.viewToClip[1, 1]) * np.float32(np.float32(1.0) / np.float32(view.renderH))
    return k


def _kdict(k):
    return dict(nbInstances=int(k["m_NbInstances"][0]), cullingFlags=int(k["m_CullingFlags"][0]),
                frustum=k["m_Frustum"][0], worldToView=k["m_WorldToView"][0], prevWorldToView=k["m_PrevWorldToView"][0],
                nearPlane=k["m_NearPlane"][0], P00=k["m_P00"][0], P11=k["m_P11"][0],
                forcedMeshLOD=int(k["m_ForcedMeshLOD"][0]), meshLODTarget=k["m_MeshLODTarget"][0])


@pytest.fixture(scope="module")
def small_world(oracle):
    spec = synth.SceneSpec(num_meshes=24, num_instances=300, meshlets_lod0=70, jitter_meshlets=True, max_lods=5,
                           alpha_mask_fraction=0.15, seed=1234)
    scene = synth.make_scene(spec)
    view = synth.make_view(eye=(0.5, 0.2, 1.0), yaw=0.03, prev_eye=(0.0, 0.0, 0.0), prev_yaw=0.0, render=(640, 360))
    depth = synth.gen_depth(view, num_occluders=60, scale=3.0)
    hw, hh = view.hzb_dims
    hzb = oracle.HzbTexture(hw, hh)
    hzb.build_from_depth(depth)
    return scene, view, depth, hzb


def test_fma_emulation_exact():
    rng = np.random.default_rng(5)
    a = rng.standard_normal(20000).astype(np.float32) * np.float32(1e3)
    b = rng.standard_normal(20000).astype(np.float32)
    c = (-(a.astype(np.float64) * b.astype(np.float64))).astype(np.float32) + rng.standard_normal(20000).astype(np.float32) * np.float32(1e-4)
    got = NP.fma(a, b, c)
    from fractions import Fraction
    for i in range(0, 20000, 97):
        exact = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        # correctly rounded float32 of an exact rational: float64 of a Fraction is correctly rounded,
        # and candidates are its two float32 neighbours
        d = float(exact)
        lo = np.float32(d)
        cands = [lo, np.nextafter(lo, np.float32(np.inf)), np.nextafter(lo, np.float32(-np.inf))]
        best = min(cands, key=lambda x: (abs(Fraction(float(x)) - exact), int(np.float32(x).view(np.uint32)) & 1))
        assert np.float32(got[i]).view(np.uint32) == np.float32(best).view(np.uint32), i


def test_hzb_build_matches_numpy(oracle, small_world):
    _, view, depth, hzb = small_world
    tex = NP.hzb_build(depth, hzb.w, hzb.h, hzb.mips, hzb.offsets)
    assert np.array_equal(tex, hzb.texels)


@pytest.mark.parametrize("flags", range(8))
@pytest.mark.parametrize("forced", [0xFF, 0, 2, 7])
def test_instance_and_meshlet_pass(oracle, small_world, flags, forced):
    scene, view, depth, hzb = small_world
    ids = scene.opaqueIds
    k = _consts(view, flags, len(ids), (hzb.w, hzb.h), forced, oracle)
    cap = 65535
    records = np.zeros(cap, I.MeshletAmplificationData)
    args = np.zeros(3, np.uint32); lateCount = np.zeros(1, np.uint32); lateIds = np.zeros(len(ids), np.uint32)
    valid = oracle.instance_cull(k, False, scene.instances, ids, scene.meshData, hzb, records, args, lateCount, lateIds, 0)
    ref = NP.instance_pass(_kdict(k), False, scene.instances, ids, scene.meshData, hzb)
    assert int(args[0]) == ref["argsX"] and valid == ref["valid"]
    G = min(int(args[0]), valid)
    got = records[:G].view(np.uint32).reshape(-1, 3)
    assert np.array_equal(got, ref["records"])
    assert int(lateCount[0]) == ref["lateCount"] and np.array_equal(lateIds[:ref["lateCount"]], ref["lateIds"])
    if flags & 2:
        assert int(lateCount[0]) > 0, "scene must exercise the late list"

    # meshlet pass on those records
    bk = np.zeros(1, I.BasePassConstants)
    bk["m_WorldToView"] = view.worldToView; bk["m_Frustum"] = k["m_Frustum"]; bk["m_HZBDimensions"] = k["m_HZBDimensions"]
    bk["m_P00"] = k["m_P00"]; bk["m_P11"] = k["m_P11"]; bk["m_NearPlane"] = k["m_NearPlane"]; bk["m_CullingFlags"] = flags
    mask, lst, tested = oracle.meshlet_cull(bk, scene.instances, scene.meshData, scene.meshlets, records, 0, G, hzb)
    kd = dict(cullingFlags=flags, frustum=k["m_Frustum"][0], worldToView=view.worldToView, nearPlane=k["m_NearPlane"][0],
              P00=k["m_P00"][0], P11=k["m_P11"][0])
    rmask, rlst = NP.meshlet_pass(kd, scene.instances, scene.meshData, scene.meshlets, got, hzb)
    assert np.array_equal(mask, rmask)
    assert np.array_equal(lst, rlst)
    if flags == 7 and forced == 0xFF:
        assert 0 < len(lst) < tested

    # late pass (Q1: only ceil(count/64)*32 threads exist)
    if flags & 2:
        largs = oracle.build_late_args(int(lateCount[0]))
        assert int(largs[0]) == (int(lateCount[0]) + 63) // 64
        args2 = np.zeros(3, np.uint32); rec2 = np.zeros(cap, I.MeshletAmplificationData)
        lc = lateCount.copy()
        valid2 = oracle.instance_cull(k, True, scene.instances, ids, scene.meshData, hzb, rec2, args2, lc, lateIds, int(largs[0]))
        ref2 = NP.instance_pass(_kdict(k), True, scene.instances, ids, scene.meshData, hzb, lateCount=int(lateCount[0]),
                                lateIds=lateIds, lateArgsX=int(largs[0]))
        assert int(args2[0]) == ref2["argsX"] and valid2 == ref2["valid"]
        G2 = min(int(args2[0]), valid2)
        assert np.array_equal(rec2[:G2].view(np.uint32).reshape(-1, 3), ref2["records"])
        assert int(lc[0]) == int(lateCount[0])
