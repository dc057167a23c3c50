"""Known-answer tests: the C oracle against tests/golden/kat_primitives.json, vectors computed with
exact rational arithmetic by tests/golden/make_kat.py (a third restatement of the formulas).
PARITY UNPINNED w.r.t. the reference (it ships no vectors); these pin the oracle's arithmetic.  CPU only."""
import json
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def f(u):
    return struct.unpack("<f", struct.pack("<I", u))[0]


def vec(us):
    return np.array([f(u) for u in us], np.float32)


def mat(us):
    return vec(us).reshape(4, 4)


def b(x):
    return int(np.float32(x).view(np.uint32))


@pytest.fixture(scope="module")
def kat():
    return json.load(open(os.path.join(HERE, "golden", "kat_primitives.json")))


def test_f32_to_f16(oracle, kat):
    for c in kat["f32_to_f16"]:
        assert oracle.f32_to_f16(f(c["in"])) == c["out"], hex(c["in"])
    # and the inverse on every half value that is not a NaN
    for h in range(0, 0x10000, 7):
        if (h & 0x7C00) == 0x7C00 and (h & 0x3FF):
            continue
        assert b(oracle.f16_to_f32(h)) == b(np.array([h], np.uint16).view(np.float16).astype(np.float32)[0])


def test_f32_to_f16_exhaustive_sample_vs_numpy(oracle):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.random(20000, np.float32), (rng.random(20000, np.float32) ** 12), np.float32(2.0) ** rng.integers(-30, 17, 2000).astype(np.float32)])
    want = xs.astype(np.float16).view(np.uint16)
    got = np.array([oracle.f32_to_f16(float(x)) for x in xs], np.uint16)
    assert np.array_equal(got, want)


def test_hzb_level(oracle, kat):
    for c in kat["hzb_level"]:
        assert oracle.hzb_level(f(c["w"]), f(c["h"]), c["mips"]) == c["out"], c


def test_late_args_q1(oracle, kat):
    for c in kat["late_args"]:
        assert list(oracle.build_late_args(c["count"])) == c["out"]


def test_frustum(oracle, kat):
    seen = set()
    for c in kat["frustum"]:
        got = oracle.frustum_cull(vec(c["c"]), f(c["r"]), vec(c["f"]))
        assert int(got) == c["out"], c
        seen.add(c["out"])
    assert seen == {0, 1}


def test_transforms(oracle, kat):
    import ctypes as C
    L = oracle.lib()
    for c in kat["transform"]:
        W, V, sph = mat(c["W"]), mat(c["V"]), vec(c["sphere"])
        assert b(oracle.max_scale(W)) == c["maxScale"]
        out = np.zeros(4, np.float32)
        L.orc_sphere_to_world(W.ctypes.data, sph.ctypes.data, out.ctypes.data)
        assert [b(x) for x in out] == c["world"]
        view = np.zeros(3, np.float32)
        L.orc_to_view(out.ctypes.data, V.ctypes.data, view.ctypes.data)
        assert [b(x) for x in view] == c["view"]


def test_make_world_matrix(oracle, kat):
    for c in kat["make_world"]:
        got = oracle.make_world_matrix(vec(c["p"]), vec(c["q"]), vec(c["s"]))
        assert [b(x) for x in got.ravel()] == c["out"]


def test_cone(oracle, kat):
    seen = set()
    for c in kat["cone"]:
        axis, cutoff = oracle.unpack_cone_view(c["packed"], mat(c["W"]), mat(c["V"]))
        assert [b(x) for x in axis] == c["axis"], hex(c["packed"])
        assert b(cutoff) == c["cutoff"]
        got = oracle.cone_cull(vec(c["c"]), f(c["r"]), axis, cutoff)
        assert int(got) == c["backfacing"]
        seen.add(c["backfacing"])
    assert seen == {0, 1}


def _hzb(oracle, d):
    h = oracle.HzbTexture(d["w"], d["h"], np.array(d["texels"], np.uint16))
    assert h.offsets == d["offsets"] and h.mips == d["mips"]
    return h


def test_sample_min_reduction(oracle, kat):
    h = _hzb(oracle, kat["occlusion"]["hzb"])
    for c in kat["sample"]:
        assert b(oracle.sample_hzb_min(h, f(c["u"]), f(c["v"]), c["mip"])) == c["out"], c


def test_occlusion(oracle, kat):
    o = kat["occlusion"]
    h = _hzb(oracle, o["hzb"])
    seen = set()
    for c in o["cases"]:
        got = oracle.occlusion_cull(vec(c["c"]), f(c["r"]), f(o["near"]), f(o["P00"]), f(o["P11"]), h)
        assert int(got) == c["out"], c
        seen.add(c["out"])
    assert seen == {0, 1}


def test_hand_derived_cases(oracle):
    """A few answers that can be checked by hand."""
    # symmetric frustum with 90 degree FOV: planes x = +-z.  f = (1/sqrt2, -1/sqrt2, ...)
    s = np.float32(1.0) / np.sqrt(np.float32(2.0))
    fr = np.array([s, -s, s, -s], np.float32)
    assert oracle.frustum_cull([0, 0, 5], 1.0, fr)          # on the axis
    assert oracle.frustum_cull([5.9, 0, 5], 1.0, fr)        # 0.9/sqrt2 = 0.64 < r: touches the plane
    assert not oracle.frustum_cull([7, 0, 5], 1.0, fr)      # 2/sqrt2 = 1.41 > r: outside
    # cone: axis = +z, cutoff 0.5, centre straight ahead -> dot = |c| >= 0.5|c| + r  when r <= 0.5|c|
    assert oracle.cone_cull([0, 0, 10], 4.0, [0, 0, 1], 0.5)
    assert not oracle.cone_cull([0, 0, 10], 6.0, [0, 0, 1], 0.5)
    # occlusion: HZB all far (0) -> everything visible; all near (1) -> hidden unless it touches the near plane
    far = oracle.HzbTexture(8, 8)
    assert oracle.occlusion_cull([0, 0, 10], 1.0, 0.1, 1.0, 1.0, far)
    near = oracle.HzbTexture(8, 8, np.full(far.total, 0x3C00, np.uint16))
    assert not oracle.occlusion_cull([0, 0, 10], 1.0, 0.1, 1.0, 1.0, near)
    assert oracle.occlusion_cull([0, 0, 1.0], 0.95, 0.1, 1.0, 1.0, near)   # c.z - near < r
    # Q1
    assert list(oracle.build_late_args(65)) == [2, 1, 1]
