"""Scene ingestion (toyrenderer_amd/gltf_lite.py) and BASELINE.json configs[0]: "cornell.gltf (~1k meshlets) scalar
C++ frustum+cone cull on CPU -- plumbing + bit-exact ref, no GPU".  The reference's asset cannot travel, so the test
runs on tests/golden/cornell_scene.npz (the scene DERIVED from it by tests/golden/make_cornell.py, in the path's wire
formats, plus the oracle's outputs) and, where /root/reference is present (the build container), re-derives the
fixture from the asset.  The meshlet partition / bounds are this build's own (meshoptimizer absent): parity with the
reference itself is unpinned."""
import json
import os
import struct

import numpy as np
import pytest

from toyrenderer_amd import gltf_lite
from toyrenderer_amd import interop as I

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_scene.npz")
REF_ASSET = "/root/reference/resources/cornell.gltf"


def _fixture():
    z = np.load(GOLDEN)                                  # allow_pickle=False (default): plain arrays only
    cam = z["camera"]
    camera = gltf_lite.Camera("fixture", tuple(cam[0:3]), tuple(cam[3:7]), float(cam[7]), float(cam[8]), float(cam[9]))
    scene = gltf_lite.LoadedScene(z["instances"].view(I.BasePassInstanceConstants).reshape(-1), z["meshData"].view(I.MeshData).reshape(-1),
                                  z["meshlets"].view(I.MeshletData).reshape(-1), z["opaqueIds"], z["alphaMaskIds"],
                                  z["nodes"].view(I.NodeLocalTransform).reshape(-1), z["primToNode"], [camera],
                                  z["vertices"].view(I.RawVertexFormat).reshape(-1) if "vertices" in z.files else np.zeros(0, I.RawVertexFormat),
                                  z["meshletVertexIds"] if "meshletVertexIds" in z.files else np.zeros(0, np.uint32),
                                  z["meshletTriangles"] if "meshletTriangles" in z.files else np.zeros(0, np.uint32))
    return z, scene, camera


def _cull(oracle, scene, view, flags):
    inst = scene.instances.copy()
    oracle.update_instance_consts(scene.nodes, scene.primToNode, inst)
    sc = dict(scene.as_oracle()); sc["instances"] = inst
    hzb = oracle.HzbTexture(*view.hzb_dims)
    depth = np.zeros((view.renderH, view.renderW), np.float32) if flags & 2 else None
    return inst, oracle.frame(sc, view.as_dict(), hzb, depth, cullingFlags=flags, maxGroups=65535, record_capacity=65535)


@pytest.mark.parametrize("flags", [5, 7])
def test_cornell_cpu_cull_matches_fixture(oracle, flags):
    z, scene, camera = _fixture()
    assert len(scene.instances) == 3 and 3 <= len(scene.meshlets) <= 4, "cornell: 3 primitives, 3-4 meshlets"
    view = gltf_lite.view_of(camera, (1920, 1080))
    inst, ref = _cull(oracle, scene, view, flags)
    assert np.array_equal(inst["m_WorldMatrix"], z[f"world_{flags}"]), "node transforms -> world matrices"
    # node 0 of the asset is rotated by 90 degrees about x (quaternion (0.7071069, 0, 0, 0.7071066))
    assert abs(float(inst["m_WorldMatrix"][0][1][2]) - 1.0) < 1e-5 and abs(float(inst["m_WorldMatrix"][0][2][1]) + 1.0) < 1e-5
    for s in (0, 1):
        if f"f{flags}_s{s}_records" not in z.files:
            assert not ref.passRan[s] or len(ref.records[s]) == 0
            continue
        assert np.array_equal(ref.records[s].view(np.uint32).reshape(-1, 3), z[f"f{flags}_s{s}_records"])
        assert np.array_equal(ref.visMask[s], z[f"f{flags}_s{s}_visMask"])
        assert np.array_equal(ref.visibleList[s], z[f"f{flags}_s{s}_visibleList"])
    # the open side of the box faces the camera at (0, 1, 5): every primitive is submitted, nothing is cone-culled away entirely
    assert int(ref.dispatchArgs[0][0]) == 3 and int(ref.drawArgs[0][0]) >= 3


@pytest.mark.skipif(not os.path.exists(REF_ASSET), reason="the reference tree is only present in the build container")
def test_fixture_is_what_the_loader_derives_from_the_reference_asset():
    z, scene, camera = _fixture()
    fresh = gltf_lite.load(REF_ASSET)
    for name in ("instances", "meshData", "meshlets", "nodes"):
        assert getattr(fresh, name).tobytes() == getattr(scene, name).tobytes(), name
    assert np.array_equal(fresh.opaqueIds, scene.opaqueIds) and np.array_equal(fresh.primToNode, scene.primToNode)
    c = fresh.cameras[0]
    assert c.position == (0.0, 1.0, 5.0) and abs(c.yfov - 0.3995965) < 1e-6 and abs(c.znear - 0.1) < 1e-7


def _write_gltf(tmp_path):
    """A generated two-node scene: a 12x12-vertex wavy grid (144 vertices -> several meshlets) under a rotated,
    scaled parent, and a masked single triangle on a child node."""
    n = 12
    xs, ys = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    grid = np.stack([xs, ys, 0.1 * np.sin(3 * xs) * np.cos(2 * ys)], -1).reshape(-1, 3).astype(np.float32)
    idx = []
    for y in range(n - 1):
        for x in range(n - 1):
            a = y * n + x
            idx += [a, a + 1, a + n, a + 1, a + n + 1, a + n]
    idx = np.array(idx, np.uint16)
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    blob = grid.tobytes() + idx.tobytes() + tri.tobytes()
    o1, o2 = len(grid.tobytes()), len(grid.tobytes()) + len(idx.tobytes())
    g = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
         "nodes": [{"mesh": 0, "children": [1], "rotation": [0, 0.38268343, 0, 0.92387953], "scale": [2, 2, 2], "translation": [0, 0, -10]},
                   {"mesh": 1, "translation": [1, 0, 0]}, {"camera": 0, "translation": [0, 0, 3]}],
         "cameras": [{"type": "perspective", "perspective": {"yfov": 0.8, "znear": 0.1, "aspectRatio": 1.5}}],
         "materials": [{"name": "opaque"}, {"name": "cutout", "alphaMode": "MASK"}],
         "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "material": 0}]},
                    {"primitives": [{"attributes": {"POSITION": 2}, "material": 1}]}],
         "accessors": [{"bufferView": 0, "componentType": 5126, "count": len(grid), "type": "VEC3"},
                       {"bufferView": 1, "componentType": 5123, "count": len(idx), "type": "SCALAR"},
                       {"bufferView": 2, "componentType": 5126, "count": 3, "type": "VEC3"}],
         "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": o1}, {"buffer": 0, "byteOffset": o1, "byteLength": o2 - o1},
                         {"buffer": 0, "byteOffset": o2, "byteLength": len(tri.tobytes())}],
         "buffers": [{"byteLength": len(blob), "uri": "scene.bin"}]}
    (tmp_path / "scene.bin").write_bytes(blob)
    (tmp_path / "scene.gltf").write_text(json.dumps(g))
    return str(tmp_path / "scene.gltf"), grid, idx.reshape(-1, 3), tri


def test_loader_invariants_on_a_generated_file(tmp_path, oracle):
    path, grid, tris, tri = _write_gltf(tmp_path)
    s = gltf_lite.load(path)
    assert len(s.instances) == 2 and s.opaqueIds.tolist() == [0] and s.alphaMaskIds.tolist() == [1]
    assert s.primToNode.tolist() == [0, 1] and int(s.nodes[1]["m_ParentNodeIdx"]) == 0 and int(s.nodes[0]["m_ParentNodeIdx"]) == 0xFFFFFFFF
    md = s.meshData[0]
    n0 = int(md["m_MeshLODDatas"]["m_NumMeshlets"][0])
    assert int(md["m_NumLODs"]) == 1 and n0 >= 3 and int(s.meshData[1]["m_MeshLODDatas"]["m_NumMeshlets"][0]) == 1
    covered = []
    for m in s.meshlets[:n0]:
        nv, nt = int(m["m_VertexAndTriangleCount"]) & 0xFF, int(m["m_VertexAndTriangleCount"]) >> 8
        assert 1 <= nv <= gltf_lite.MAX_MESHLET_VERTICES and 1 <= nt <= gltf_lite.MAX_MESHLET_TRIANGLES
        vids = s.meshletVertexIds[int(m["m_MeshletVertexIDsBufferIdx"]):][:nv]
        packed = s.meshletTriangles[int(m["m_MeshletIndexIDsBufferIdx"]):][:nt]
        local = np.stack([packed & 0xFF, (packed >> 8) & 0xFF, (packed >> 16) & 0xFF], 1)
        assert local.max() < nv
        covered.append(vids[local])
        c, r = m["m_BoundingSphere"][:3], float(m["m_BoundingSphere"][3])
        assert np.all(np.linalg.norm(grid[vids] - c, axis=1) <= r * (1 + 1e-5) + 1e-6), "meshlet vertices inside the bounding sphere"
        # the packed cone contains every triangle normal (decode as the shader does: basepass.hlsl:92-105)
        pk = int(m["m_ConeAxisAndCutoff"])
        axis = np.array([(pk & 0xFF), (pk >> 8) & 0xFF, (pk >> 16) & 0xFF], np.float64) / 255.0 * 2 - 1
        cutoff = (pk >> 24) / 255.0
        if cutoff < 1.0 and np.linalg.norm(axis) > 0:
            p = grid[vids].astype(np.float64)
            nrm = np.cross(p[local[:, 1]] - p[local[:, 0]], p[local[:, 2]] - p[local[:, 0]])
            nrm /= np.linalg.norm(nrm, axis=1)[:, None]
            ax = axis / np.linalg.norm(axis)
            assert np.all(nrm @ ax >= np.sqrt(max(0.0, 1 - cutoff * cutoff)) - 0.03), "triangle normals inside the cone"
    got = np.sort(np.sort(np.concatenate(covered), axis=1), axis=0)
    assert np.array_equal(got, np.sort(np.sort(tris.astype(np.uint32), axis=1), axis=0)), "every triangle in exactly one meshlet"
    assert np.all(np.linalg.norm(grid - md["m_BoundingSphere"][:3], axis=1) <= float(md["m_BoundingSphere"][3]) * (1 + 1e-5))
    # the camera sits on node 2 and the whole thing culls through the oracle (world matrices from the node chain)
    cam = s.cameras[0]
    assert cam.position == (0.0, 0.0, 3.0) and abs(cam.aspect - 1.5) < 1e-9
    view = gltf_lite.view_of(cam, (1200, 800))
    inst, ref = _cull(oracle, s, view, 5)
    assert np.allclose(inst["m_WorldMatrix"][1][3][:3], [np.sqrt(2.0), 0.0, -10 - np.sqrt(2.0)], atol=1e-5), "child = parent TRS applied to (1,0,0)"
    assert ref.passRan[0] and ref.passRan[2] and int(ref.dispatchArgs[0][0]) == (n0 + 31) // 32 and int(ref.dispatchArgs[2][0]) == 1
    assert 0 < int(ref.drawArgs[0][0]) <= n0
