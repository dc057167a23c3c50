"""`<scene>_CachedData.bin` v3 (toyrenderer_amd/cached_scene.py; layout of SceneLoading.cpp:57-79,706-781,1090-1145).
No file of this format ships with the reference (it is generated on first load), so the tests pin the layout by its
byte offsets, the round trip, and the cull results of a scene whose geometry went through the file."""
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from scene_gen import write_city_gltf  # noqa: E402
from toyrenderer_amd import cached_scene, gltf_lite  # noqa: E402
from toyrenderer_amd import interop as I  # noqa: E402


def test_layout_and_round_trip(tmp_path):
    s = gltf_lite.load(write_city_gltf(tmp_path))
    c = cached_scene.from_scene(s)
    path = str(tmp_path / "city_CachedData.bin")
    cached_scene.write(path, c)
    raw = open(path, "rb").read()
    hdr = struct.unpack("<8I", raw[:32])
    assert hdr == (3, 0, len(s.vertices), len(c.indices), len(s.meshData), len(s.meshletVertexIds), len(s.meshletTriangles), len(s.meshlets))
    # arrays back to back in the order the reference freads them
    off = 32
    assert raw[off:off + 20 * hdr[2]] == s.vertices.tobytes(); off += 20 * hdr[2]
    off += 4 * hdr[3]
    assert len(c.meshData.tobytes()) == 156 * hdr[4]; off += 156 * hdr[4]
    assert raw[off:off + 4 * hdr[5]] == s.meshletVertexIds.tobytes(); off += 4 * hdr[5]
    assert raw[off:off + 4 * hdr[6]] == s.meshletTriangles.tobytes(); off += 4 * hdr[6]
    assert raw[off:off + 32 * hdr[7]] == s.meshlets.tobytes(); off += 32 * hdr[7]
    assert len(raw) == off + 32 * hdr[4], "MeshSpecificData (32 B per mesh) ends the file when there are no animations"
    back = cached_scene.read(path)
    for name in ("vertices", "indices", "meshData", "meshletVertexIds", "meshletTriangles", "meshlets", "meshSpecific"):
        assert getattr(back, name).tobytes() == getattr(c, name).tobytes(), name
    # the rebuilt index buffer addresses each mesh's own vertices and covers its triangles
    for i, m in enumerate(c.meshData):
        n = int(c.meshSpecific["m_NumIndices"][i])
        idx = c.indices[int(m["m_GlobalIndexBufferIdx"]):][:n]
        assert n % 3 == 0 and n > 0 and int(idx.max()) < int(c.meshSpecific["m_NumVertices"][i])
        pos = s.vertices["m_Position"][int(m["m_GlobalVertexBufferIdx"]):][:int(c.meshSpecific["m_NumVertices"][i])]
        assert np.allclose(c.meshSpecific["m_AABBCenter"][i] - c.meshSpecific["m_AABBExtents"][i], pos.min(0), atol=1e-6)
    # an animation tail is carried through untouched
    c.tail = b"\x01\x02\x03\x04" * 5
    cached_scene.write(path, c)
    assert cached_scene.read(path).tail == c.tail


def test_reader_rejects_what_the_reference_checks(tmp_path):
    s = gltf_lite.load(write_city_gltf(tmp_path))
    c = cached_scene.from_scene(s)
    path = str(tmp_path / "x_CachedData.bin")
    cached_scene.write(path, c)
    raw = bytearray(open(path, "rb").read())
    bad = bytearray(raw); bad[0] = 2
    open(path, "wb").write(bad)
    with pytest.raises(ValueError, match="version 2"):
        cached_scene.read(path)
    open(path, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(ValueError, match="truncated"):
        cached_scene.read(path)
    c.meshlets = c.meshlets.copy(); c.meshlets["m_MeshletVertexIDsBufferIdx"][-1] = len(c.meshletVertexIds)
    with pytest.raises(ValueError, match="vertex-id range"):
        cached_scene.write(path, c)


def test_scene_through_the_cache_culls_the_same(tmp_path, oracle):
    """glTF -> (cache written, read back, applied to a fresh glTF load) -> cull: identical outputs."""
    gltf = write_city_gltf(tmp_path)
    a = gltf_lite.load(gltf)
    path = str(tmp_path / "city_CachedData.bin")
    cached_scene.write(path, cached_scene.from_scene(a))
    b = cached_scene.apply(gltf_lite.load(gltf), cached_scene.read(path))
    view = gltf_lite.view_of(a.cameras[0], (640, 360))
    outs = []
    for s in (a, b):
        inst = s.instances.copy()
        oracle.update_instance_consts(s.nodes, s.primToNode, inst)
        sc = dict(s.as_oracle()); sc["instances"] = inst
        outs.append(oracle.frame(sc, view.as_dict(), oracle.HzbTexture(*view.hzb_dims), None, cullingFlags=5, record_capacity=4096))
    for slot in (0, 2):
        assert np.array_equal(outs[0].visibleList[slot], outs[1].visibleList[slot]) and len(outs[0].visibleList[slot]) > 0
        assert np.array_equal(outs[0].records[slot], outs[1].records[slot])
    with pytest.raises(ValueError, match="meshes"):
        c = cached_scene.read(path); c.meshData = c.meshData[:-1]; cached_scene.apply(a, c)
