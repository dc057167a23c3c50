"""SURVEY.md 8(f) rank 1 on the GPU: "basepass_MS_Main_depth" (csrc/k_raster.hip) against orc_raster_depth, and the
two-phase frame that builds its HZB from the depth it rasterised itself against pyoracle.frame(raster=...).
Bit-exact: the depth is a maximum over per-pixel values computed with the same operations, so it does not depend on
the order the GPU draws the triangles in."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from scene_gen import all_meshlets_visible, write_city_gltf  # noqa: E402
from toyrenderer_amd import gltf_lite, synth  # noqa: E402
from toyrenderer_amd import interop as I  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from toyrenderer_amd import rhi
    d = rhi.Device(0)
    yield d
    d.destroy()


def _world(oracle, s):
    inst = s.instances.copy()
    oracle.update_instance_consts(s.nodes, s.primToNode, inst)
    sc = dict(s.as_oracle()); sc["instances"] = inst
    return inst, sc


def _gpu_scene(dev, s, inst):
    from toyrenderer_amd.frame import GpuScene
    gs = GpuScene(dev, inst, s.meshData, s.meshlets, s.opaqueIds, s.alphaMaskIds)
    gs.set_geometry(s.vertices, s.meshletVertexIds, s.meshletTriangles)
    return gs


def _consts(view):
    k = np.zeros(1, I.BasePassConstants)
    k["m_WorldToClip"] = I.world_to_clip(view.worldToView, view.viewToClip)
    k["m_NearPlane"] = view.nearPlane
    k["m_OutputResolution"] = (view.renderW, view.renderH)
    return k


def _raster_everything(dev, oracle, s, view, eye=None):
    from toyrenderer_amd import rhi
    from toyrenderer_amd.rhi import CB, SRV, TEX_UAV
    inst, sc = _world(oracle, s)
    gs = _gpu_scene(dev, s, inst)
    rec, lst = all_meshlets_visible(s)
    k = _consts(view)
    ref = np.zeros((view.renderH, view.renderW), np.float32)
    oracle.raster_depth(k, sc, s.vertices, s.meshletVertexIds, s.meshletTriangles, rec, lst, ref)
    records = dev.buffer_from(rec, "records", min_bytes=12)
    visible = dev.buffer_from(lst, "visible")
    args = dev.create_buffer(12, "drawArgs", stride=12, indirect=True)
    args.upload(np.array([len(lst), 1, 1], np.uint32))
    depth = dev.create_texture(view.renderW, view.renderH, 1, rhi.FORMAT_R32_FLOAT, "Depth Buffer")
    cl = dev.create_command_list()
    try:
        cl.open()
        cl.clear_texture_f32(depth, 0.0)
        cb = cl.constant_buffer(k, "BasePassConstants")
        cl.dispatch_indirect("basepass_MS_Main_depth",
                             [CB(0, cb), SRV(0, gs.instances), SRV(1, gs.vertices), SRV(2, gs.meshData), SRV(4, gs.meshlets), SRV(5, gs.meshletVertexIds),
                              SRV(6, gs.meshletTriangles), SRV(7, records), SRV(9, visible), TEX_UAV(0, depth, 0)], args)
        cl.close()
        dev.execute(cl); dev.wait_idle()
        got = depth.download_mip(0)
    finally:
        cl.release(); depth.release(); args.release(); visible.release(); records.release(); gs.release()
    return got, ref


@pytest.mark.parametrize("render", [(1920, 1080), (257, 131)])
def test_cornell_depth_matches_the_oracle(dev, oracle, render):
    from test_gltf_cornell import _fixture
    z, s, camera = _fixture()
    assert len(s.vertices) and len(s.meshletTriangles), "fixture carries the geometry (tests/golden/make_cornell.py)"
    view = gltf_lite.view_of(camera, render)
    got, ref = _raster_everything(dev, oracle, s, view)
    assert np.count_nonzero(ref) > 0.5 * ref.size, "the box fills most of the screen"
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_generated_scene_depth_matches_the_oracle(dev, oracle, tmp_path):
    s = gltf_lite.load(write_city_gltf(tmp_path))
    view = gltf_lite.view_of(s.cameras[0], (1280, 720))
    got, ref = _raster_everything(dev, oracle, s, view)
    assert 0.2 < np.count_nonzero(ref) / ref.size < 0.95
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_camera_inside_the_geometry(dev, oracle, tmp_path):
    """Triangles crossing the near plane and far off-screen vertices: the drop / clamp rules, not a crash."""
    s = gltf_lite.load(write_city_gltf(tmp_path))
    cam = s.cameras[0]
    P = synth.perspective_rh_reverse_z_infinite(cam.yfov, 16 / 9, cam.znear)
    V = synth.world_to_view((0.3, 0.0, -8.02), (0.0, float(np.sin(0.4)), 0.0, float(np.cos(0.4))))     # inside the wall, turned
    view = synth.View(V, V.copy(), P, float(np.float32(cam.znear)), 640, 360)
    got, ref = _raster_everything(dev, oracle, s, view)
    assert np.count_nonzero(ref) > 0
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("flags", [7, 3])
def test_two_phase_frames_on_own_depth(dev, oracle, tmp_path, flags):
    """Four frames of a moving camera, HZB built from the depth the frame rasterised: every list, the depth buffer and
    the HZB chain equal the oracle's after every frame."""
    from test_gpu_parity import _compare_frame
    from toyrenderer_amd.frame import FrameDriver
    s = gltf_lite.load(write_city_gltf(tmp_path))
    inst, sc = _world(oracle, s)
    gs = _gpu_scene(dev, s, inst)
    cam = s.cameras[0]
    render = (1280, 720)
    P = synth.perspective_rh_reverse_z_infinite(cam.yfov, render[0] / render[1], cam.znear)
    V0 = synth.world_to_view((0.0, 0.0, 0.0), cam.orientation)
    view = synth.View(V0, V0.copy(), P, float(np.float32(cam.znear)), *render)
    drv = FrameDriver(dev, gs, view, record_capacity=4096, culling_flags=flags, raster_depth=True)
    hzb = oracle.HzbTexture(*view.hzb_dims)
    depth = np.zeros((render[1], render[0]), np.float32)
    prevV = V0
    try:
        late_seen = culled = False
        for f, eye in enumerate([(0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.4, 0.1, -0.3), (0.9, 0.1, -0.5)]):
            V = synth.world_to_view(eye, cam.orientation)
            drv.view = view = synth.View(V, prevV, P, float(np.float32(cam.znear)), *render)
            prevV = V
            drv.record(); drv.run()
            got = drv.results()
            geo = (I.world_to_clip(V, P), s.vertices, s.meshletVertexIds, s.meshletTriangles)
            ref = oracle.frame(sc, view.as_dict(), hzb, depth, cullingFlags=flags, record_capacity=4096, raster=geo)
            _compare_frame(got, ref)
            assert np.array_equal(drv.depth.download_mip(0).view(np.uint32), depth.view(np.uint32)), f"frame {f}: depth differs"
            assert np.array_equal(drv.hzb.download_chain(), hzb.texels), f"frame {f}: HZB chain differs"
            late_seen |= bool(ref.lateCount[0] > 0 and ref.drawArgs[1][0] > 0)
            culled |= bool(f > 0 and ref.drawArgs[0][0] + ref.drawArgs[1][0] < ref.meshletsTested[0] + ref.meshletsTested[1])
        assert late_seen and culled, "the case must exercise the late pass and cull something"
    finally:
        drv.release(); gs.release()


def test_drop_in_path_renders_its_own_depth(oracle, tmp_path):
    """The same loop through the C++ host mirror (RenderGraph / GBufferRenderer over the C ABI): node transforms on the
    GPU, cull, depth of the visible meshlets, HZB -- four frames, moving camera, everything equal to the oracle."""
    from test_gpu_parity import _compare_frame
    from toyrenderer_amd import host
    s = gltf_lite.load(write_city_gltf(tmp_path))
    inst, sc = _world(oracle, s)
    cam = s.cameras[0]
    render = (1280, 720)
    P = synth.perspective_rh_reverse_z_infinite(cam.yfov, render[0] / render[1], cam.znear)
    hzb = oracle.HzbTexture(*I.hzb_dims(*render))
    depth = np.zeros((render[1], render[0]), np.float32)
    r = host.Renderer(render=render, max_groups=4096)
    try:
        r.load_scene(s.instances, s.meshData, s.meshlets, s.opaqueIds, s.alphaMaskIds)
        r.load_nodes(s.nodes, s.primToNode)
        r.load_geometry(s.vertices, s.meshletVertexIds, s.meshletTriangles)
        r.set_raster_depth(True)
        r.set_culling(7)
        prevV = synth.world_to_view((0.0, 0.0, 0.0), cam.orientation)
        for f, eye in enumerate([(0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.4, 0.1, -0.3), (0.9, 0.1, -0.5)]):
            V = synth.world_to_view(eye, cam.orientation)
            view = synth.View(V, prevV, P, float(np.float32(cam.znear)), *render)
            prevV = V
            r.set_node_transforms(s.nodes)
            r.set_camera(view)
            r.frame()
            got = r.results()
            geo = (I.world_to_clip(V, P), s.vertices, s.meshletVertexIds, s.meshletTriangles)
            ref = oracle.frame(sc, view.as_dict(), hzb, depth, cullingFlags=7, record_capacity=4096, maxGroups=4096, raster=geo)
            _compare_frame(got, ref)
            if f == 0:   # the ingested scene carries its own LOD chain (gltf_lite.build_lod_chain): LOD selection picks from it
                assert len(np.unique(got[0]["records"]["m_MeshLOD"])) >= 2 and int(s.meshData["m_NumLODs"].max()) >= 3
            assert np.array_equal(r.download_depth().view(np.uint32), depth.view(np.uint32)), f"frame {f}: depth differs"
            assert np.array_equal(r.download_hzb(), hzb.texels), f"frame {f}: HZB chain differs"
        assert np.count_nonzero(depth) > 0.2 * depth.size
    finally:
        r.shutdown()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_triangle_soup_with_hostile_vertices(dev, oracle, seed):
    """Random meshlets whose vertices include points behind the camera, on the near plane, far off screen, at 1e30,
    infinities and NaNs, degenerate and sub-pixel triangles, triangle indices past the meshlet's vertex count: the
    rules of the convention (drop / clamp / never pass the depth test), bit-exact, and no out-of-range access."""
    from toyrenderer_amd import rhi
    rng = np.random.default_rng(seed)
    n_meshlets = 40
    view = synth.make_view(render=(320, 200))
    verts, vids, tris, meshlets = [], [], [], np.zeros(n_meshlets, I.MeshletData)
    for m in range(n_meshlets):
        nv, nt = int(rng.integers(3, 65)), int(rng.integers(1, 97))
        base = len(verts)
        p = np.stack([rng.uniform(-6, 6, nv), rng.uniform(-4, 4, nv), rng.uniform(-30, 2, nv)], 1)
        kind = rng.integers(0, 12, nv)
        p[kind == 0] *= 1e30
        p[kind == 1, 2] = -0.1                       # exactly on the near plane
        p[kind == 2, 0] = np.inf
        p[kind == 3, 1] = np.nan
        p[kind == 4] = p[0]                          # coincident vertices -> degenerate triangles
        verts += [tuple(x) for x in p.astype(np.float32)]
        meshlets[m]["m_MeshletVertexIDsBufferIdx"] = len(vids)
        vids += list(range(base, base + nv))
        meshlets[m]["m_MeshletIndexIDsBufferIdx"] = len(tris)
        hi = nv + (4 if m % 5 == 0 else 0)           # some indices point past the vertex count
        idx = rng.integers(0, hi, (nt, 3))
        tris += [int(a | (b << 8) | (c << 16)) for a, b, c in idx]
        meshlets[m]["m_VertexAndTriangleCount"] = nv | (nt << 8)
    v = np.zeros(len(verts), I.RawVertexFormat); v["m_Position"] = np.array(verts, np.float32)
    inst = np.zeros(2, I.BasePassInstanceConstants)
    inst["m_WorldMatrix"][0] = np.eye(4, dtype=np.float32)
    inst["m_WorldMatrix"][1] = np.diag([0.5, 2.0, 1.0, 1.0]).astype(np.float32); inst["m_WorldMatrix"][1][3, :3] = (1.0, -0.5, -3.0)
    md = np.zeros(1, I.MeshData); md["m_NumLODs"] = 1; md["m_MeshLODDatas"]["m_NumMeshlets"][0][0] = n_meshlets
    sc = dict(instances=inst, meshData=md, meshlets=meshlets)
    rec = np.zeros(4, I.MeshletAmplificationData)
    rec["m_InstanceConstIdx"] = [0, 0, 1, 1]; rec["m_MeshletGroupOffset"] = [0, 32, 0, 32]
    lst = np.array([(g << 5) | l for g in range(4) for l in range(32 if g % 2 == 0 else n_meshlets - 32)], np.uint32)
    lst = rng.permutation(lst)
    k = _consts(view)
    ref = np.zeros((200, 320), np.float32)
    oracle.raster_depth(k, sc, v, np.array(vids, np.uint32), np.array(tris, np.uint32), rec, lst, ref)
    assert np.count_nonzero(ref) > 1000 and np.all(np.isfinite(ref[ref > 0]) | np.isinf(ref[ref > 0]))
    from toyrenderer_amd.rhi import CB, SRV, TEX_UAV
    bufs = [dev.buffer_from(inst, "inst", uav=False), dev.buffer_from(v, "v", uav=False), dev.buffer_from(md, "md", uav=False),
            dev.buffer_from(meshlets, "ml", uav=False), dev.buffer_from(np.array(vids, np.uint32), "vid", uav=False),
            dev.buffer_from(np.array(tris, np.uint32), "tri", uav=False), dev.buffer_from(rec, "rec"), dev.buffer_from(lst, "lst")]
    args = dev.create_buffer(12, "drawArgs", stride=12, indirect=True)
    args.upload(np.array([len(lst), 1, 1], np.uint32))
    depth = dev.create_texture(320, 200, 1, rhi.FORMAT_R32_FLOAT, "Depth Buffer")
    cl = dev.create_command_list()
    try:
        cl.open()
        cl.clear_texture_f32(depth, 0.0)
        cb = cl.constant_buffer(k, "BasePassConstants")
        cl.dispatch_indirect("basepass_MS_Main_depth", [CB(0, cb), SRV(0, bufs[0]), SRV(1, bufs[1]), SRV(2, bufs[2]), SRV(4, bufs[3]), SRV(5, bufs[4]),
                                                        SRV(6, bufs[5]), SRV(7, bufs[6]), SRV(9, bufs[7]), TEX_UAV(0, depth, 0)], args)
        cl.close()
        dev.execute(cl); dev.wait_idle()
        got = depth.download_mip(0)
    finally:
        cl.release(); depth.release(); args.release()
        for b in bufs:
            b.release()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_scene_from_the_cached_data_file_through_the_native_reader(oracle, tmp_path):
    """glTF -> `<scene>_CachedData.bin` v3 (cached_scene.write) -> the host library's own reader
    (trhost_load_scene_cached: meshes, meshlets, vertices, meshlet index buffers from the file) -> two frames with
    self-rendered depth: equal to the oracle on the in-memory scene.  A wrong version is refused."""
    from test_gpu_parity import _compare_frame
    from toyrenderer_amd import cached_scene, host
    s = gltf_lite.load(write_city_gltf(tmp_path))
    path = str(tmp_path / "city_CachedData.bin")
    cached_scene.write(path, cached_scene.from_scene(s))
    inst, sc = _world(oracle, s)
    cam = s.cameras[0]
    render = (960, 540)
    view = gltf_lite.view_of(cam, render)
    hzb = oracle.HzbTexture(*I.hzb_dims(*render))
    depth = np.zeros((render[1], render[0]), np.float32)
    r = host.Renderer(render=render, max_groups=4096)
    try:
        r.load_scene_cached(path, s.instances, s.opaqueIds, s.alphaMaskIds)
        r.load_nodes(s.nodes, s.primToNode)
        r.set_raster_depth(True)
        r.set_culling(7)
        geo = (I.world_to_clip(view.worldToView, view.viewToClip), s.vertices, s.meshletVertexIds, s.meshletTriangles)
        for f in range(2):
            r.set_node_transforms(s.nodes)
            r.set_camera(view)
            r.frame()
            ref = oracle.frame(sc, view.as_dict(), hzb, depth, cullingFlags=7, record_capacity=4096, maxGroups=4096, raster=geo)
            _compare_frame(r.results(), ref)
            assert np.array_equal(r.download_depth().view(np.uint32), depth.view(np.uint32))
        raw = bytearray(open(path, "rb").read()); raw[0] = 2
        open(path, "wb").write(raw)
        with pytest.raises(host.HostError, match="version 2"):
            r.load_scene_cached(path, s.instances, s.opaqueIds, s.alphaMaskIds)
    finally:
        r.shutdown()
