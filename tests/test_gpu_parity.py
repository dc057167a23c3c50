"""Parity of the HIP path (through the C ABI, include/trhip.h) against the CPU oracle.

Bar: bit-exact (integer / index / fp16-texel outputs).  PARITY UNPINNED with respect to the
reference itself (no reference fixtures exist): the oracle is the scalar restatement
of the HLSL in oracle/tr_oracle.c.  Everything here needs a real MI355X.
"""
import numpy as np
import pytest

from toyrenderer_amd import interop as I
from toyrenderer_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from toyrenderer_amd import rhi
    d = rhi.Device(0)
    yield d
    d.destroy()


def _oracle_hzb(oracle, view, depth):
    hw, hh = view.hzb_dims
    h = oracle.HzbTexture(hw, hh)
    if depth is not None:
        h.build_from_depth(depth)
    return h


def _upload_hzb(driver, hzb):
    driver.hzb.upload_chain(hzb.texels, hzb.offsets)


def _compare_frame(got, ref, slots=(0, 1, 2, 3)):
    for s in slots:
        if not ref.passRan[s]:
            assert got[s] is None, f"slot {s} ran on the GPU but not in the oracle"
            continue
        g = got[s]
        assert g is not None, f"slot {s} did not run on the GPU"
        assert np.array_equal(g["dispatchArgs"], ref.dispatchArgs[s]), (s, g["dispatchArgs"], ref.dispatchArgs[s])
        assert g["validRecords"] == int(ref.validRecords[s]), s
        assert np.array_equal(g["records"].view(np.uint32), ref.records[s].view(np.uint32)), f"slot {s}: records differ"
        assert np.array_equal(g["visMask"], ref.visMask[s]), f"slot {s}: visibility masks differ"
        assert np.array_equal(g["drawArgs"], ref.drawArgs[s]), (s, g["drawArgs"], ref.drawArgs[s])
        assert np.array_equal(g["visibleList"], ref.visibleList[s]), f"slot {s}: visible lists differ"


def _run_case(dev, oracle, spec, view, *, flags=7, forced=-1, max_groups=65535, depth_prev=None, depth_cur=None,
              freeze=False, frames=1):
    from toyrenderer_amd.frame import FrameDriver, GpuScene
    scene = synth.make_scene(spec)
    gs = GpuScene(dev, scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
    drv = FrameDriver(dev, gs, view, record_capacity=max_groups, culling_flags=flags, force_mesh_lod=forced,
                      freeze_culling_camera=freeze)
    hzb = _oracle_hzb(oracle, view, depth_prev)
    if depth_prev is not None:
        _upload_hzb(drv, hzb)
    if depth_cur is not None:
        drv.depth.upload_mip(0, depth_cur)
    try:
        for _ in range(frames):
            drv.record()
            drv.run()
            got = drv.results()
            ref = oracle.frame(scene.as_oracle(), view.as_dict(), hzb, depth_cur, cullingFlags=flags, forceMeshLOD=forced,
                               freeze=freeze, maxGroups=max_groups, record_capacity=max_groups)
            _compare_frame(got, ref)
            if flags & 2:
                assert got["lateCount"] == int(ref.lateCount[1] if scene.alphaMaskIds.size else ref.lateCount[0])
            # the HZB the frame leaves behind (consumed by the next frame's early pass)
            if flags & 2 and depth_cur is not None and not freeze:
                assert np.array_equal(drv.hzb.download_chain(), hzb.texels), "HZB chain differs"
        return got, ref
    finally:
        drv.release()
        gs.release()


SMALL = synth.SceneSpec(num_meshes=24, num_instances=300, meshlets_lod0=70, jitter_meshlets=True, max_lods=5,
                        alpha_mask_fraction=0.15, seed=1234)


@pytest.mark.parametrize("flags", range(8))
def test_frame_all_flag_combinations(dev, oracle, flags):
    view = synth.make_view(eye=(0.5, 0.2, 1.0), yaw=0.03, prev_eye=(0.0, 0.0, 0.0), prev_yaw=0.0, render=(640, 360))
    d_prev = synth.gen_depth(view, num_occluders=60, seed=11, scale=3.0)
    d_cur = synth.gen_depth(view, num_occluders=40, seed=12, scale=3.0)
    got, ref = _run_case(dev, oracle, SMALL, view, flags=flags, depth_prev=d_prev, depth_cur=d_cur)
    if flags == 7:
        assert ref.lateCount[0] > 0 and ref.dispatchArgs[1][0] > 0, "case must exercise the late pass"
        assert 0 < ref.drawArgs[0][0] < ref.meshletsTested[0]


@pytest.mark.parametrize("forced", [0, 1, 3, 7, 200])
def test_forced_lod(dev, oracle, forced):
    view = synth.make_view(render=(640, 360))
    d = synth.gen_depth(view, num_occluders=30, seed=3, scale=3.0)
    _run_case(dev, oracle, SMALL, view, flags=7, forced=forced, depth_prev=d, depth_cur=d)


def test_c1_sponza_scale_frustum_cone(dev, oracle):
    """BASELINE configs[1]: ~50 k meshlets... (C1 = 400 instances), phase-1 frustum + cone only."""
    view = synth.make_view()
    _run_case(dev, oracle, synth.config_spec("C1"), view, flags=5)


def test_c0_tiny(dev, oracle):
    view = synth.make_view(eye=(0, 0, 0), render=(1280, 720))
    d = synth.gen_depth(view, 5, seed=9, scale=3.0)
    _run_case(dev, oracle, synth.config_spec("C0"), view, flags=7, depth_prev=d, depth_cur=d)


def test_group_cap_overflow_q2(dev, oracle):
    """Q2: the counter keeps counting, everything from the first dropped instance on is undefined."""
    view = synth.make_view(render=(640, 360))
    spec = synth.SceneSpec(num_meshes=10, num_instances=500, meshlets_lod0=90, jitter_meshlets=True, max_lods=1, seed=77)
    got, ref = _run_case(dev, oracle, spec, view, flags=1, max_groups=257)
    assert ref.dispatchArgs[0][0] > 257 and ref.validRecords[0] < 257


def test_late_list_q1_odd_sizes(dev, oracle):
    """Q1: only ceil(count/64)*32 late entries are re-tested.  Everything occluded in phase 1
    (HZB = near everywhere), nothing occluded in phase 2 (depth = far)."""
    view = synth.make_view(render=(640, 360))
    hw, hh = view.hzb_dims
    near_everywhere = np.ones((view.renderH, view.renderW), np.float32)
    far = np.zeros((view.renderH, view.renderW), np.float32)
    for n in (1, 31, 33, 64, 65, 97, 200):
        spec = synth.SceneSpec(num_meshes=4, num_instances=n, meshlets_lod0=40, max_lods=1, seed=n, z_near=20.0, z_far=60.0,
                               box_x=4.0, box_y=3.0)
        got, ref = _run_case(dev, oracle, spec, view, flags=2, depth_prev=near_everywhere, depth_cur=far)
        assert ref.lateCount[0] == n, "all instances must be deferred to the late pass"
        expect = min(n, ((n + 63) // 64) * 32)
        assert len(np.unique(ref.records[1]["instanceConstIdx"])) == expect


def test_two_frames_hzb_feedback(dev, oracle):
    view = synth.make_view(eye=(0.2, 0.0, 0.5), yaw=0.01, render=(1280, 720))
    d = synth.gen_depth(view, num_occluders=80, seed=21, scale=3.0)
    _run_case(dev, oracle, SMALL, view, flags=7, depth_prev=None, depth_cur=d, frames=2)


def test_freeze_culling_camera_skips_hzb(dev, oracle):
    view = synth.make_view(render=(640, 360))
    d_prev = synth.gen_depth(view, num_occluders=60, seed=11, scale=3.0)
    d_cur = synth.gen_depth(view, num_occluders=40, seed=12, scale=3.0)
    _run_case(dev, oracle, SMALL, view, flags=7, depth_prev=d_prev, depth_cur=d_cur, freeze=True)


def test_empty_lists(dev, oracle):
    view = synth.make_view(render=(640, 360))
    spec = synth.SceneSpec(num_meshes=3, num_instances=40, meshlets_lod0=10, max_lods=2, alpha_mask_fraction=0.0, seed=5)
    got, ref = _run_case(dev, oracle, spec, view, flags=5)
    assert got[2] is None and got[1] is None


def test_cone_unorm_all_byte_values(dev, oracle):
    """x/255 for every byte value (cull_math.hip.h u8Unorm) through the cone test."""
    from toyrenderer_amd.frame import FrameDriver, GpuScene
    view = synth.make_view(render=(640, 360))
    spec = synth.SceneSpec(num_meshes=1, num_instances=64, meshlets_lod0=256, max_lods=1, seed=42, z_near=8, z_far=30, box_x=3, box_y=2)
    scene = synth.make_scene(spec)
    b = np.arange(256, dtype=np.uint32)
    rng = np.random.default_rng(1)
    scene.meshlets["m_ConeAxisAndCutoff"] = b | (rng.permutation(256).astype(np.uint32) << 8) | (rng.permutation(256).astype(np.uint32) << 16) | (b[::-1] << 24)
    gs = GpuScene(dev, scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
    drv = FrameDriver(dev, gs, view, record_capacity=4096, culling_flags=4)
    try:
        drv.record(); drv.run()
        got = drv.results()
        hzb = _oracle_hzb(oracle, view, None)
        ref = oracle.frame(scene.as_oracle(), view.as_dict(), hzb, None, cullingFlags=4, maxGroups=4096, record_capacity=4096)
        _compare_frame(got, ref, slots=(0,))
        assert 0 < ref.drawArgs[0][0] < ref.meshletsTested[0]
    finally:
        drv.release(); gs.release()


@pytest.mark.parametrize("render", [(100, 40), (640, 360), (1920, 1080), (3840, 2160), (2560, 1440), (2048, 64), (48, 3000)])
def test_hzb_build(dev, oracle, render):
    """minmaxdownsample + SPD replacement vs orc_hzb_build, incl. a non-tiled (<64) and non-square chain, and chains the
    64x64 tiling does not fit (1024x32, 32x2048: one launch per mip until the tail kernel takes over)."""
    from toyrenderer_amd.frame import FrameDriver, GpuScene
    view = synth.make_view(render=render)
    rng = np.random.default_rng(render[0])
    depth = rng.random((render[1], render[0]), np.float32) ** 6
    depth[rng.random(depth.shape) < 0.2] = 0
    spec = synth.SceneSpec(num_meshes=1, num_instances=1, meshlets_lod0=1)
    scene = synth.make_scene(spec)
    gs = GpuScene(dev, scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
    drv = FrameDriver(dev, gs, view, record_capacity=64)
    try:
        drv.depth.upload_mip(0, depth)
        cl = drv.cl
        cl.open(); drv._generate_hzb(cl); cl.close()
        dev.execute(cl)
        dev.wait_idle()
        ref = _oracle_hzb(oracle, view, depth)
        assert np.array_equal(drv.hzb.download_chain(), ref.texels)
    finally:
        drv.release(); gs.release()


def test_update_instance_consts(dev, oracle):
    from toyrenderer_amd import rhi
    rng = np.random.default_rng(8)
    n_nodes, n_inst = 500, 2000
    nodes = np.zeros(n_nodes, I.NodeLocalTransform)
    nodes["m_Position"] = rng.standard_normal((n_nodes, 3)).astype(np.float32) * 5
    q = rng.standard_normal((n_nodes, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    nodes["m_Rotation"] = q.astype(np.float32)
    nodes["m_Scale"] = rng.uniform(0.5, 2.0, (n_nodes, 3)).astype(np.float32)
    parent = np.full(n_nodes, 0xFFFFFFFF, np.uint32)
    for i in range(1, n_nodes):
        if rng.random() < 0.8:
            parent[i] = rng.integers(0, i)          # parents precede children: chains up to ~10 deep
    nodes["m_ParentNodeIdx"] = parent
    prim = rng.integers(0, n_nodes, n_inst).astype(np.uint32)
    inst = np.zeros(n_inst, I.BasePassInstanceConstants)
    inst["m_WorldMatrix"] = rng.standard_normal((n_inst, 4, 4)).astype(np.float32)
    b_nodes = dev.buffer_from(nodes, "nodes", uav=False); b_prim = dev.buffer_from(prim, "prim", uav=False)
    b_inst = dev.buffer_from(inst, "inst")
    cl = dev.create_command_list()
    k = np.array([n_inst], np.uint32)
    cl.open()
    for _ in range(2):
        cl.dispatch("updateinstanceconsts_CS_UpdateInstanceConstsAndBuildTLAS",
                    [rhi.PUSH(0), rhi.SRV(0, b_nodes), rhi.SRV(1, b_prim), rhi.UAV(0, b_inst)], ((n_inst + 31) // 32, 1, 1), push=k)
    cl.close()
    dev.execute(cl)
    got = b_inst.download(I.BasePassInstanceConstants)
    ref = inst.copy()
    oracle.update_instance_consts(nodes, prim, ref); oracle.update_instance_consts(nodes, prim, ref)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # the same pass over a RANGE of the table (this build's second push-constant word, { count, first }: what a rank of a sharded
    # scene dispatches -- trhost_set_instance_update_range): the range follows the oracle, everything else keeps its bytes
    for first, count in ((0, 1), (1, 255), (257, 256), (700, 1300), (1999, 1)):
        b_inst.upload(inst)
        cl2 = dev.create_command_list()
        cl2.open()
        cl2.dispatch("updateinstanceconsts_CS_UpdateInstanceConstsAndBuildTLAS",
                     [rhi.PUSH(0), rhi.SRV(0, b_nodes), rhi.SRV(1, b_prim), rhi.UAV(0, b_inst)], ((count + 31) // 32, 1, 1), push=np.array([count, first], np.uint32))
        cl2.close()
        dev.execute(cl2)
        got = b_inst.download(I.BasePassInstanceConstants)
        ref = inst.copy()
        part = inst[first:first + count].copy()
        oracle.update_instance_consts(nodes, prim[first:first + count], part)
        ref[first:first + count] = part
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (first, count)
        cl2.release()
    cl.release(); b_nodes.release(); b_prim.release(); b_inst.release()


def test_errors_are_reported_not_fatal(dev):
    from toyrenderer_amd import rhi
    cl = dev.create_command_list()
    cl.open()
    with pytest.raises(rhi.TrhipError, match="unknown shader"):
        cl.dispatch("no_such_shader", [], (1, 1, 1))
    with pytest.raises(rhi.TrhipError, match="constant buffer b0"):
        cl.dispatch("gpuculling_CS_GPUCulling LATE_CULL=0", [], (1, 1, 1))
    cl.close()
    cl.release()


def test_deterministic_across_runs(dev, oracle):
    from toyrenderer_amd.frame import FrameDriver, GpuScene
    view = synth.make_view(render=(1280, 720))
    scene = synth.make_scene(synth.config_spec("C1"))
    d = synth.gen_depth(view, 50, seed=2, scale=3.0)
    gs = GpuScene(dev, scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
    drv = FrameDriver(dev, gs, view, record_capacity=65535, culling_flags=7)
    drv.depth.upload_mip(0, d)
    try:
        outs = []
        for _ in range(3):
            drv.record(); drv.run()
            r = drv.results()
            outs.append((r[0]["visibleList"].copy(), r[1]["visibleList"].copy() if r[1] else None, r[0]["records"].copy()))
        # frame 0 sees the cleared HZB, frames 1 and 2 the same rebuilt HZB -> identical bytes
        assert np.array_equal(outs[1][0], outs[2][0]) and np.array_equal(outs[1][2], outs[2][2])
    finally:
        drv.release(); gs.release()


def _zero_weight_lookup_scene(oracle, view, n_dense_meshes=64, n_sparse_meshes=128, seed=11):
    """Meshlets whose HZB lookup has a ZERO bilinear weight (fractional texel coordinate exactly 0 with a second
    column / row that survives edge clamping): the case the footprint-min table of the meshlet cull kernel cannot
    serve (DESIGN.md "HZB lookup = one 2-byte load").  They are about 1 in 10^4 of random spheres, so candidates
    are screened with the oracle (orc_occlusion_footprints).  Scene: `n_dense_meshes` meshes of 32 such meshlets
    (every lane of every step defers -> the per-wave list overflows) + `n_sparse_meshes` meshes with one such
    meshlet among 31 ordinary ones (deferred lookups patched one by one).  Identity instance transforms."""
    rng = np.random.default_rng(seed)
    vd = view.as_dict()
    hw, hh = view.hzb_dims
    need = 32 * n_dense_meshes + n_sparse_meshes
    slow_c, slow_r = [], []
    ordinary = None
    have = 0
    for _ in range(40):
        n = 4_000_000
        z = rng.uniform(8, 120, n).astype(np.float32)
        c = np.stack([rng.uniform(-0.5, 0.5, n) * z, rng.uniform(-0.28, 0.28, n) * z, -z], 1).astype(np.float32)   # the camera looks down -z
        r = (rng.uniform(0.004, 0.05, n) * (z / 20)).astype(np.float32)
        fp = oracle.occlusion_footprints(c, r, vd, (hw, hh))
        mw, mh = np.maximum(hw >> fp[:, 0], 1), np.maximum(hh >> fp[:, 0], 1)
        slow = ((fp[:, 3] == 1) & (fp[:, 1] >= 0) & (fp[:, 1] + 1 < mw)) | ((fp[:, 4] == 1) & (fp[:, 2] >= 0) & (fp[:, 2] + 1 < mh))
        slow_c.append(c[slow]); slow_r.append(r[slow])
        have += int(slow.sum())
        if ordinary is None:
            keep = np.nonzero(~slow)[0][:31 * n_sparse_meshes]
            ordinary = (c[keep], r[keep])
        if have >= need:
            break
    assert have >= need, f"only {have} zero-weight lookups found"
    sc, sr = np.concatenate(slow_c)[:need], np.concatenate(slow_r)[:need]
    num_meshes = n_dense_meshes + n_sparse_meshes
    ml = np.zeros(32 * num_meshes, I.MeshletData)
    centres = np.zeros((32 * num_meshes, 3), np.float32)
    radii = np.zeros(32 * num_meshes, np.float32)
    centres[:32 * n_dense_meshes] = sc[:32 * n_dense_meshes]
    radii[:32 * n_dense_meshes] = sr[:32 * n_dense_meshes]
    for k in range(n_sparse_meshes):
        base = 32 * (n_dense_meshes + k)
        pos = int(rng.integers(0, 32))
        others = [j for j in range(32) if j != pos]
        centres[base + pos], radii[base + pos] = sc[32 * n_dense_meshes + k], sr[32 * n_dense_meshes + k]
        centres[[base + j for j in others]] = ordinary[0][31 * k:31 * (k + 1)]
        radii[[base + j for j in others]] = ordinary[1][31 * k:31 * (k + 1)]
    ml["m_BoundingSphere"][:, :3] = centres
    ml["m_BoundingSphere"][:, 3] = radii
    ml["m_ConeAxisAndCutoff"] = rng.integers(0, 2 ** 32, len(ml), dtype=np.uint64).astype(np.uint32)
    md = np.zeros(num_meshes, I.MeshData)
    md["m_BoundingSphere"] = np.array([0, 0, 0, 1000.0], np.float32)      # around the camera: passes every instance-level test
    md["m_NumLODs"] = 1
    md["m_MeshLODDatas"]["m_MeshletDataBufferIdx"][:, 0] = np.arange(num_meshes, dtype=np.uint32) * 32
    md["m_MeshLODDatas"]["m_NumMeshlets"][:, 0] = 32
    inst = np.zeros(num_meshes, I.BasePassInstanceConstants)
    inst["m_WorldMatrix"] = np.eye(4, dtype=np.float32).reshape(inst["m_WorldMatrix"].shape[1:])
    inst["m_PrevWorldMatrix"] = inst["m_WorldMatrix"]
    inst["m_MeshDataIdx"] = np.arange(num_meshes, dtype=np.uint32)
    return inst, md, ml


def test_zero_weight_hzb_lookups_take_the_texel_path(dev, oracle):
    """The footprint-min table path of the early meshlet cull (record capacity >= 2^19) on lookups it must defer."""
    from toyrenderer_amd.frame import FrameDriver, GpuScene
    view = synth.make_view(render=(1920, 1080))
    inst, md, ml = _zero_weight_lookup_scene(oracle, view)
    ids = np.arange(len(inst), dtype=np.uint32)
    d_prev = synth.gen_depth(view, 120, seed=5, scale=2.0)
    hzb = _oracle_hzb(oracle, view, d_prev)
    cap = 1 << 19
    gs = GpuScene(dev, inst, md, ml, ids, np.zeros(0, np.uint32))
    drv = FrameDriver(dev, gs, view, record_capacity=cap, culling_flags=7)
    _upload_hzb(drv, hzb)
    drv.depth.upload_mip(0, d_prev)
    try:
        for frame in range(2):
            drv.record()
            drv.run()
            got = drv.results()
            scene = dict(instances=inst, meshData=md, meshlets=ml, opaqueIds=ids, alphaMaskIds=np.zeros(0, np.uint32))
            ref = oracle.frame(scene, view.as_dict(), hzb, d_prev, cullingFlags=7, maxGroups=cap, record_capacity=cap)
            _compare_frame(got, ref, slots=(0, 1))
            assert int(ref.dispatchArgs[0][0]) == len(inst), "every instance submitted in the early phase"
            vis = int(ref.drawArgs[0][0])
            assert 0 < vis < 32 * len(inst), "the HZB decides: some of the deferred lookups pass, some fail"
    finally:
        drv.release()
        gs.release()


@pytest.mark.parametrize("render,flags", [((640, 360), 7), ((300, 1000), 7), ((100, 100), 3), ((640, 360), 6)])
def test_table_kernel_on_small_and_non_square_hzbs(dev, oracle, render, flags):
    """The footprint-min-table variant of the early meshlet cull (chosen for record capacities >= 2^19, otherwise only
    reached by the full-size configs) on non-square and tiny HZBs (512x256, 256x512, 64x64: top mips 1 texel wide),
    two frames with HZB feedback, and with 6000 instances so that the tile-sorted processing order is on as well."""
    view = synth.make_view(eye=(0.5, 0.2, 1.0), yaw=0.03, prev_eye=(0.0, 0.0, 0.0), prev_yaw=0.0, render=render)
    d_prev = synth.gen_depth(view, num_occluders=60, seed=11, scale=3.0)
    d_cur = synth.gen_depth(view, num_occluders=40, seed=12, scale=3.0)
    spec = synth.SceneSpec(num_meshes=40, num_instances=6000, meshlets_lod0=70, jitter_meshlets=True, max_lods=4,
                           alpha_mask_fraction=0.1, seed=77)
    got, ref = _run_case(dev, oracle, spec, view, flags=flags, max_groups=1 << 19, depth_prev=d_prev, depth_cur=d_cur, frames=2)
    assert ref.dispatchArgs[0][0] > 1000 and 0 < ref.drawArgs[0][0] < ref.meshletsTested[0]


def test_clear_hoisting_and_elision_keep_command_order_semantics(dev):
    """The back end moves a clear of memory no earlier command uses into the recording's first clear launch and drops
    a clear of memory that still holds the value (trhip_internal.h); what the commands observe must not change."""
    a = dev.create_buffer(64, "a"); b = dev.create_buffer(64, "b"); c = dev.create_buffer(64, "c")
    d = dev.create_buffer(64, "d"); e = dev.create_buffer(64, "e"); f = dev.create_buffer(64, "f")
    ramp = np.arange(16, dtype=np.uint32) + 100
    for buf in (a, b, c, d, e, f):
        buf.upload(np.full(16, 0xDEADBEEF, np.uint32))
    cl = dev.create_command_list()
    try:
        for _ in range(2):                               # the second execution re-records from scratch
            cl.open()
            cl.clear_buffer_u32(a, 0)                    # first clear launch of the recording
            cl.write_buffer(d, ramp)                     # d is used ...
            cl.copy_buffer(e, d, 64)                     # ... and read before it is cleared: e must get the ramp
            cl.clear_buffer_u32(c, 5)                    # c: no earlier use -> joins the first launch
            cl.clear_buffer_u32(d, 7)                    # d: used earlier -> stays in place
            cl.clear_buffer_u32(a, 0)                    # still 0 -> dropped
            cl.write_buffer(a, ramp)
            cl.copy_buffer(f, a, 64)
            cl.clear_buffer_u32(a, 0)                    # written since -> NOT dropped
            cl.clear_buffer_u32(b, 9)
            cl.close()
            dev.execute(cl); dev.wait_idle()
            assert np.all(a.download(np.uint32, 16) == 0)
            assert np.all(b.download(np.uint32, 16) == 9)
            assert np.all(c.download(np.uint32, 16) == 5)
            assert np.all(d.download(np.uint32, 16) == 7)
            assert np.array_equal(e.download(np.uint32, 16), ramp)
            assert np.array_equal(f.download(np.uint32, 16), ramp)
    finally:
        cl.release()
        for buf in (a, b, c, d, e, f):
            buf.release()


@pytest.mark.parametrize("block", range(4))
def test_randomised_small_scene_sweep(dev, oracle, block):
    """48 random configurations (scene size from 1 instance to ~6000 groups, ragged meshlet counts, 1-8 LODs, alpha-mask
    share, culling flags, forced LOD, group capacity below / at / above what the frame needs (Q2), odd and non-square
    render sizes, 1-2 frames): every output word and the HZB chain against the oracle.  Shakes the boundaries the named
    cases do not sit on (list tiles of exactly 2048 groups, one-group passes, empty late lists, ...)."""
    renders = [(640, 360), (100, 40), (1280, 720), (333, 517), (64, 64), (1920, 1080), (2048, 64)]
    import os
    off = int(os.environ.get("TR_SWEEP_OFFSET", "0"))            # soak runs: other 48 configurations (tools/soak_sweep.sh)
    for case in range(off + block * 12, off + block * 12 + 12):
        rng = np.random.default_rng(1000 + case)
        n_inst = int(rng.choice([1, 2, 31, 32, 33, 64, 255, 500, 1024, 2048, 3000]))
        m0 = int(rng.choice([1, 31, 32, 33, 64, 70, 128, 200]))
        spec = synth.SceneSpec(num_meshes=int(rng.integers(1, 40)), num_instances=n_inst, meshlets_lod0=m0,
                               jitter_meshlets=bool(rng.integers(0, 2)), max_lods=int(rng.integers(1, 9)),
                               alpha_mask_fraction=float(rng.choice([0.0, 0.0, 0.2, 1.0])), seed=5000 + case,
                               z_near=float(rng.choice([2.0, 5.0, 20.0])), z_far=float(rng.choice([40.0, 200.0])),
                               box_x=float(rng.choice([10.0, 100.0])), box_y=float(rng.choice([6.0, 56.0])))
        render = renders[int(rng.integers(0, len(renders)))]
        view = synth.make_view(eye=tuple(rng.uniform(-0.5, 0.5, 3)), yaw=float(rng.uniform(-0.05, 0.05)),
                               prev_eye=tuple(rng.uniform(-0.5, 0.5, 3)), prev_yaw=float(rng.uniform(-0.05, 0.05)), render=render)
        flags = int(rng.integers(0, 8))
        forced = int(rng.choice([-1, -1, -1, 0, 2, 7]))
        need = n_inst * ((2 * m0 + 31) // 32) + 1
        cap = int(rng.choice([65535, need, max(1, need // 3), 2048, 4096]))
        cap = min(cap, 65535)
        if os.environ.get("TR_SWEEP_TABLE"):                       # soak runs: the footprint-table kernel (record capacity 2^19) instead of the texel-path one
            cap = 1 << 19
        d_prev = synth.gen_depth(view, num_occluders=int(rng.integers(0, 80)), seed=case, scale=3.0) if rng.integers(0, 4) else None
        d_cur = synth.gen_depth(view, num_occluders=int(rng.integers(0, 80)), seed=case + 500, scale=3.0)
        try:
            _run_case(dev, oracle, spec, view, flags=flags, forced=forced, max_groups=cap, depth_prev=d_prev, depth_cur=d_cur,
                      frames=int(rng.integers(1, 3)))
        except AssertionError as e:
            raise AssertionError(f"case {case}: {spec} render={render} flags={flags} forced={forced} cap={cap}: {e}") from e


@pytest.mark.parametrize("table", [False, True])
def test_meshlet_buffer_rewritten_between_frames(dev, oracle, table):
    """The cull kernel reads a derived copy of the 20 bytes per meshlet it needs (the meshlet cull stream, k_basepass_as.hip),
    rebuilt when the meshlet buffer's version moves.  Frame 1 on the scene as loaded; then the meshlet buffer is rewritten
    -- an upload, and a write_buffer command inside the frame's own recording for a part of it -- and frame 2 must see the
    new spheres and cones (every output word against the oracle on the new data)."""
    from toyrenderer_amd.frame import FrameDriver, GpuScene
    view = synth.make_view(eye=(0.2, 0.1, 0.4), yaw=0.01, render=(640, 360))
    spec = synth.SceneSpec(num_meshes=30, num_instances=500, meshlets_lod0=70, jitter_meshlets=True, max_lods=2, seed=313)
    scene = synth.make_scene(spec)
    d = synth.gen_depth(view, num_occluders=50, seed=9, scale=3.0)
    hzb = _oracle_hzb(oracle, view, d)
    gs = GpuScene(dev, scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
    drv = FrameDriver(dev, gs, view, record_capacity=(1 << 19) if table else 65535, culling_flags=7)
    _upload_hzb(drv, hzb)
    drv.depth.upload_mip(0, d)
    cap = (1 << 19) if table else 65535
    try:
        drv.record(); drv.run()
        _compare_frame(drv.results(), oracle.frame(scene.as_oracle(), view.as_dict(), hzb, d, cullingFlags=7, maxGroups=cap, record_capacity=cap))
        # new meshlets: spheres moved and resized, cone words permuted
        rng = np.random.default_rng(5)
        ml2 = scene.meshlets.copy()
        ml2["m_BoundingSphere"][:, :3] += rng.uniform(-0.3, 0.3, (len(ml2), 3)).astype(np.float32)
        ml2["m_BoundingSphere"][:, 3] *= rng.uniform(0.5, 2.0, len(ml2)).astype(np.float32)
        ml2["m_ConeAxisAndCutoff"] = rng.permutation(ml2["m_ConeAxisAndCutoff"])
        half = (len(ml2) // 2)
        gs.meshlets.upload(ml2[:half])                                        # first half: upload
        pre = dev.create_command_list()                                       # second half: a write_buffer command executed in front of the frame
        pre.open(); pre.write_buffer(gs.meshlets, ml2[half:], offset=half * ml2.dtype.itemsize); pre.close()
        dev.execute(pre); pre.release()
        scene2 = scene.as_oracle(); scene2["meshlets"] = ml2
        hzb2 = _oracle_hzb(oracle, view, d)                                   # the HZB frame 1 left behind = built from d
        drv.record(); drv.run()
        _compare_frame(drv.results(), oracle.frame(scene2, view.as_dict(), hzb2, d, cullingFlags=7, maxGroups=cap, record_capacity=cap))
    finally:
        drv.release()
        gs.release()


@pytest.mark.parametrize("table", [False, True])
def test_meshlet_and_instance_buffers_written_out_of_band_then_marked(dev, oracle, table):
    """a write the back end cannot see -- here through a SECOND wrap of the same device memory, as a torch kernel or
    a hipMemcpy on the raw pointer would -- leaves the derived copies (meshlet cull stream, instance cull cache) stale until
    trhip_buffer_mark_written is called on the bound buffer (include/trhip.h).  After the call the next frame follows the
    new contents word for word."""
    from toyrenderer_amd.frame import FrameDriver, GpuScene
    view = synth.make_view(eye=(0.1, 0.2, 0.3), yaw=-0.02, render=(640, 360))
    spec = synth.SceneSpec(num_meshes=24, num_instances=400, meshlets_lod0=66, jitter_meshlets=True, max_lods=2, seed=717)
    scene = synth.make_scene(spec)
    d = synth.gen_depth(view, num_occluders=40, seed=3, scale=3.0)
    hzb = _oracle_hzb(oracle, view, d)
    gs = GpuScene(dev, scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
    cap = (1 << 19) if table else 65535
    drv = FrameDriver(dev, gs, view, record_capacity=cap, culling_flags=7)
    _upload_hzb(drv, hzb)
    drv.depth.upload_mip(0, d)
    alias_m = alias_i = None
    try:
        drv.record(); drv.run()
        _compare_frame(drv.results(), oracle.frame(scene.as_oracle(), view.as_dict(), hzb, d, cullingFlags=7, maxGroups=cap, record_capacity=cap))
        rng = np.random.default_rng(11)
        ml2 = scene.meshlets.copy()
        ml2["m_BoundingSphere"][:, :3] += rng.uniform(-0.4, 0.4, (len(ml2), 3)).astype(np.float32)
        ml2["m_BoundingSphere"][:, 3] *= rng.uniform(0.5, 1.5, len(ml2)).astype(np.float32)
        ml2["m_ConeAxisAndCutoff"] = rng.permutation(ml2["m_ConeAxisAndCutoff"])
        inst2 = scene.instances.copy()
        inst2["m_WorldMatrix"][:, 3, :3] += rng.uniform(-0.5, 0.5, (len(inst2), 3)).astype(np.float32)
        # out of band: another handle on the same memory
        alias_m = dev.wrap_buffer(gs.meshlets.ptr, ml2.nbytes, name="meshlets alias", stride=ml2.dtype.itemsize)
        alias_i = dev.wrap_buffer(gs.instances.ptr, inst2.nbytes, name="instances alias", stride=inst2.dtype.itemsize)
        alias_m.upload(ml2)
        alias_i.upload(inst2)
        gs.meshlets.mark_written()
        gs.instances.mark_written()
        scene2 = scene.as_oracle(); scene2["meshlets"] = ml2; scene2["instances"] = inst2
        hzb2 = _oracle_hzb(oracle, view, d)
        drv.record(); drv.run()
        _compare_frame(drv.results(), oracle.frame(scene2, view.as_dict(), hzb2, d, cullingFlags=7, maxGroups=cap, record_capacity=cap))
    finally:
        for b in (alias_m, alias_i):
            if b is not None:
                b.release()
        drv.release()
        gs.release()


@pytest.mark.parametrize("flags,mid_cap", [(7, False), (3, False), (7, True)])
def test_instance_pass_continues_from_nonzero_counters(dev, oracle, flags, mid_cap):
    """gpuculling.hlsl:64-66, 165: the pass ADDS to whatever the group counter and the late counter hold.  Two dispatches
    (the opaque list, then the alpha-mask list) into the SAME record / counter / late-list buffers with no clear in between:
    the second pass starts from the first one's counters.  The single-launch pass reads those starting values while its last
    tile rewrites them (k_gpuculling.hip instanceFusedKernel); test_three_kernel_instance_pass_on_small_scenes
    runs this case through the three-kernel path as well.  mid_cap: a group capacity between the two passes' counters -- the
    second pass starts below it and drops at it (Q2)."""
    from toyrenderer_amd.frame import FrameDriver, GpuScene
    from toyrenderer_amd.rhi import CB, SRV, UAV, TEX_SRV, SAMPLER
    view = synth.make_view(eye=(0.3, 0.1, 0.6), yaw=0.02, render=(640, 360))
    spec = synth.SceneSpec(num_meshes=20, num_instances=700, meshlets_lod0=60, jitter_meshlets=True, max_lods=3,
                           alpha_mask_fraction=0.4, seed=91)
    scene = synth.make_scene(spec)
    d_prev = synth.gen_depth(view, num_occluders=70, seed=5, scale=3.0)
    hzb = _oracle_hzb(oracle, view, d_prev)
    gs = GpuScene(dev, scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
    lists = [(gs.opaqueIds, gs.numOpaque, scene.opaqueIds), (gs.alphaMaskIds, gs.numAlphaMask, scene.alphaMaskIds)]
    assert all(n > 0 for _, n, _ in lists)
    probe = FrameDriver(dev, gs, view, record_capacity=16, culling_flags=flags)       # only for its constants

    def oracle_passes(cap):
        recs = np.zeros(cap, oracle.RECORD_DT)
        o_args, o_late, o_ids = np.zeros(3, np.uint32), np.zeros(1, np.uint32), np.zeros(len(scene.instances), np.uint32)
        xs, valid = [], 0
        for _, nb, ids in lists:
            valid = oracle.instance_cull(probe._cull_consts(nb), False, scene.instances, ids, scene.meshData, hzb, recs, o_args, o_late, o_ids, 0, cap)
            xs.append(int(o_args[0]))
        return recs, o_args, int(o_late[0]), o_ids, xs, valid
    _, _, _, _, xs, _ = oracle_passes(65535)
    assert 0 < xs[0] < xs[1], "both passes must submit groups"
    cap = (xs[0] + xs[1]) // 2 if mid_cap else 65535
    recs, o_args, o_late, o_ids, xs, valid = oracle_passes(cap)
    probe.release()
    drv = FrameDriver(dev, gs, view, record_capacity=cap, culling_flags=flags)
    _upload_hzb(drv, hzb)
    try:
        cl = drv.cl
        cl.open()
        cl.clear_buffer_u32(drv.dispatchArgs[0], 0)
        cl.clear_buffer_u32(drv.lateCount, 0)
        cl.clear_buffer_u32(drv.lateIds, 0)
        for ids_buf, nb, _ in lists:
            cb = cl.constant_buffer(drv._cull_consts(nb), "GPUCullingPassConstants")
            b = [CB(0, cb), SRV(0, gs.instances), SRV(1, ids_buf), SRV(2, gs.meshData), UAV(0, drv.records[0]),
                 UAV(1, drv.dispatchArgs[0]), UAV(2, drv.lateCount), UAV(3, drv.lateIds), SAMPLER(0), TEX_SRV(3, drv.hzb)]
            cl.dispatch("gpuculling_CS_GPUCulling LATE_CULL=0", b, ((nb + 31) // 32, 1, 1))
        cl.close()
        dev.execute(cl)
        dev.wait_idle()
        args = drv.dispatchArgs[0].download(np.uint32, 4)
        late_n = int(drv.lateCount.download(np.uint32, 1)[0])
        assert int(args[0]) == xs[1] and late_n == o_late, (args, xs, late_n, o_late)
        if flags & 2:
            assert late_n > 0
            assert np.array_equal(drv.lateIds.download(np.uint32, late_n), o_ids[:late_n])
        assert int(args[3]) == valid, (args, valid)
        G = min(int(args[0]), int(args[3]), cap)
        if mid_cap:
            assert xs[0] < G < cap < int(args[0]), "the second pass must start below the capacity and drop at it"
        assert np.array_equal(drv.records[0].download(I.MeshletAmplificationData, G).view(np.uint32), recs[:G].view(np.uint32))
    finally:
        drv.release()
        gs.release()


def test_three_kernel_instance_pass_on_small_scenes():
    """Small passes run classify + scan + emit as ONE launch (instanceFusedKernel); the three-kernel path then only sees
    the full-size configs.  TRHIP_NO_FUSED_INSTANCE=1 sends the small cases of this file through it as well -- incl. the
    table-kernel cases on small and non-square HZBs: the scan launch's extra workgroups build their footprint tables."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TRHIP_NO_FUSED_INSTANCE="1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", "all_flag or late_dispatch or group_cap or two_frames or sweep or empty or forced or continues_from or table_kernel_on_small"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.parametrize("percent", ["0", "37", "100"])
def test_footprint_table_strips_split_between_scan_and_emit(percent):
    """A large early instance pass builds the HZB's footprint table with extra workgroups of its scan and emit launches (half of
    the strips each); TRHIP_QUAD_SCAN_PERCENT moves the split.  The table-kernel cases on small and non-square HZBs through the
    three-kernel instance pass with all strips in emit, an odd split, all in scan."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TRHIP_NO_FUSED_INSTANCE="1", TRHIP_QUAD_SCAN_PERCENT=percent)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", "table_kernel_on_small"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0 and "4 passed" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_texel_kernel_batch_path_on_small_passes():
    """A pass with at most two records per half-wave of the texel kernel's grid (every late pass, most small early passes of this
    file: 65 535 records of capacity = 4096 half-waves) skips the kernel's batch machinery: every half-wave evaluates its records
    exactly from global memory.  TRHIP_NO_SHORT_PASS=1 sends the small cases of this file through the batches as well."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TRHIP_NO_SHORT_PASS="1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", "all_flag or forced or two_frames or sweep or hostile or cone_test_at or zero_weight or cone_unorm or q1_odd"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.parametrize("flags,table", [(7, False), (7, True), (5, False), (3, True), (6, False)])
def test_hostile_operands_take_the_exact_arithmetic_path(dev, oracle, flags, table):
    """The cull kernel runs the square roots and divisions of a step on a fast path when every lane of the wave has its
    operands in a comfortable exponent range and on the compiler's full IEEE sequences otherwise (cm::stepQuotients).
    This scene makes sure the second path and the switch between them are exercised: world matrices with scales from
    2^-45 to 2^40, a zero scale on one axis (zero adjugate rows: normalize(0 / 0)), meshlet radii of 0, 1e-38, 1e30 and
    NaN, centres on the camera plane and behind it, at 1e20, infinities, cone bytes 0 / 127 / 128 / 255 -- mixed at
    random with ordinary instances so that waves contain both kinds.  table: record capacity 2^19 (footprint-table
    kernel) instead of the texel-path kernel.  Bit-exact against the oracle like every other case."""
    from toyrenderer_amd.frame import FrameDriver, GpuScene
    rng = np.random.default_rng(77 + flags)
    spec = synth.SceneSpec(num_meshes=40, num_instances=900, meshlets_lod0=50, jitter_meshlets=True, max_lods=3, seed=4242)
    scene = synth.make_scene(spec)
    inst, ml = scene.instances, scene.meshlets
    n = len(inst)
    kind = rng.integers(0, 8, n)
    W = inst["m_WorldMatrix"]
    for i in range(n):
        k = kind[i]
        if k == 1: W[i, :3, :3] *= np.float32(2.0 ** rng.integers(-45, -30))
        elif k == 2: W[i, :3, :3] *= np.float32(2.0 ** rng.integers(25, 40))
        elif k == 3: W[i, int(rng.integers(0, 3)), :3] = 0.0                    # zero scale on one axis
        elif k == 4: W[i, 3, :3] = rng.choice([0.0, 1e20, -1e20, 3e-39], 3).astype(np.float32)
        elif k == 5: W[i, 3, 2] = np.float32(rng.choice([0.05, -0.05, 0.0, 5.0]))  # near / on / behind the camera plane
    inst["m_PrevWorldMatrix"] = W
    m = len(ml)
    mk = rng.integers(0, 12, m)
    sph = ml["m_BoundingSphere"]
    sph[mk == 1, 3] = 0.0
    sph[mk == 2, 3] = np.float32(1e-38)
    sph[mk == 3, 3] = np.float32(1e30)
    sph[mk == 4, 3] = np.nan
    sph[mk == 5, :3] = np.float32(1e20)
    sph[mk == 6, 0] = np.inf
    sph[mk == 7, :3] = 0.0
    cone = ml["m_ConeAxisAndCutoff"]
    cone[mk == 8] = 0x00000000
    cone[mk == 9] = 0xFF7F7F7F
    cone[mk == 10] = 0x00808080
    cone[mk == 11] = 0xFFFFFFFF
    view = synth.make_view(eye=(0.2, 0.1, 0.4), yaw=0.02, render=(1280, 720))
    d_prev = synth.gen_depth(view, num_occluders=50, seed=5, scale=3.0)
    d_cur = synth.gen_depth(view, num_occluders=50, seed=6, scale=3.0)
    cap = 1 << 19 if table else 65535
    gs = GpuScene(dev, inst, scene.meshData, ml, scene.opaqueIds, scene.alphaMaskIds)
    drv = FrameDriver(dev, gs, view, record_capacity=cap, culling_flags=flags)
    hzb = _oracle_hzb(oracle, view, d_prev)
    _upload_hzb(drv, hzb)
    drv.depth.upload_mip(0, d_cur)
    try:
        with np.errstate(all="ignore"):
            for _ in range(2):
                drv.record()
                drv.run()
                got = drv.results()
                ref = oracle.frame(scene.as_oracle(), view.as_dict(), hzb, d_cur, cullingFlags=flags, maxGroups=cap, record_capacity=cap)
                _compare_frame(got, ref)
        assert int(ref.meshletsTested[0]) > 5000 and 0 < int(ref.drawArgs[0][0]) < int(ref.meshletsTested[0])
    finally:
        drv.release()
        gs.release()


@pytest.mark.parametrize("flags,table", [(4, False), (5, False), (7, False), (7, True), (6, True)])
def test_cone_test_at_its_decision_boundary(dev, oracle, flags, table):
    """The cull kernel decides the cone test from t * rsq(t.t) and c.c * rsq(c.c) (a few ulp, no division) unless a lane
    is within 2^-18 of the boundary dot(c, axis) = cutoff * |c| + r, in which case its wave redoes the test with the exact
    square roots and divisions (cm::coneBack).  Here the radii of most meshlets are PUT on that boundary: the radius
    that makes the two sides equal (evaluated in float64 from the float32 inputs), moved by 0, +-1, +-2, ... +-10^5 ulp,
    so that lanes sit exactly on it, a few ulp beside it, at the edge of the 2^-18 band and outside -- in the same waves
    as ordinary meshlets.  Bit-exact against the oracle like every other case."""
    from toyrenderer_amd.frame import FrameDriver, GpuScene
    rng = np.random.default_rng(900 + flags)
    spec = synth.SceneSpec(num_meshes=600, num_instances=600, meshlets_lod0=64, max_lods=1, seed=777, nonuniform_scale_fraction=0.3,
                           unique=True)                          # every instance its own meshlets: the radii below are per instance
    scene = synth.make_scene(spec)
    inst, ml, md = scene.instances, scene.meshlets, scene.meshData
    view = synth.make_view(eye=(0.3, -0.2, 0.5), yaw=0.03, render=(1280, 720))
    V = view.worldToView.astype(np.float64)
    moved = 0
    steps = np.array([0, 1, -1, 2, -2, 3, -3, 5, -5, 16, -16, 50, -50, 200, -200, 1000, -1000, 10 ** 4, -10 ** 4, 10 ** 5, -10 ** 5], np.int64)
    for i in range(len(inst)):
        W = inst["m_WorldMatrix"][i].astype(np.float64)
        lod = md["m_MeshLODDatas"][inst["m_MeshDataIdx"][i]][0]
        b, n = int(lod["m_MeshletDataBufferIdx"]), int(lod["m_NumMeshlets"])
        sph = ml["m_BoundingSphere"][b:b + n].astype(np.float64)
        packed = ml["m_ConeAxisAndCutoff"][b:b + n]
        c = np.concatenate([sph[:, :3], np.ones((n, 1))], axis=1) @ W @ V
        c = c[:, :3] * np.array([1.0, 1.0, -1.0])                                               # basepass.hlsl:68-69
        by = np.stack([(packed >> s) & 0xFF for s in (0, 8, 16, 24)], axis=1).astype(np.float64) / 255.0
        a = by[:, :3] * 2.0 - 1.0                                                                # :92-99
        R3 = W[:3, :3]
        adj = np.stack([np.cross(R3[1], R3[2]), np.cross(R3[2], R3[0]), np.cross(R3[0], R3[1])])  # toyrenderer_common.hlsli:124-132
        t = a @ adj
        tl = np.linalg.norm(t, axis=1)
        ok = tl > 1e-6
        axis = (t / np.where(ok, tl, 1.0)[:, None]) @ V[:3, :3] * np.array([1.0, 1.0, -1.0])    # :103-104
        scale = np.sqrt(max(R3[0] @ R3[0], R3[1] @ R3[1], R3[2] @ R3[2]))                        # :134-140
        r_star = (np.einsum("ij,ij->i", c, axis) - by[:, 3] * np.linalg.norm(c, axis=1)) / scale  # sphere.w that puts the meshlet ON the boundary
        take = ok & (r_star > 1e-3) & (r_star < 50.0) & (rng.random(n) < 0.8)
        w = r_star.astype(np.float32)
        bits = w.view(np.int32).astype(np.int64) + rng.choice(steps, n)
        w = bits.astype(np.int32).view(np.float32)
        ml["m_BoundingSphere"][b:b + n, 3] = np.where(take, w, ml["m_BoundingSphere"][b:b + n, 3])
        moved += int(take.sum())
    assert moved > 5000, moved
    d_prev = synth.gen_depth(view, num_occluders=40, seed=15, scale=3.0)
    d_cur = synth.gen_depth(view, num_occluders=40, seed=16, scale=3.0)
    cap = 1 << 19 if table else 65535
    gs = GpuScene(dev, inst, md, ml, scene.opaqueIds, scene.alphaMaskIds)
    drv = FrameDriver(dev, gs, view, record_capacity=cap, culling_flags=flags)
    hzb = _oracle_hzb(oracle, view, d_prev)
    _upload_hzb(drv, hzb)
    drv.depth.upload_mip(0, d_cur)
    try:
        drv.record()
        drv.run()
        got = drv.results()
        ref = oracle.frame(scene.as_oracle(), view.as_dict(), hzb, d_cur, cullingFlags=flags, maxGroups=cap, record_capacity=cap)
        _compare_frame(got, ref)
        assert int(ref.meshletsTested[0]) > 5000 and 0 < int(ref.drawArgs[0][0]) < int(ref.meshletsTested[0])
    finally:
        drv.release()
        gs.release()


@pytest.mark.parametrize("kind,flags", [("texel_x", 3), ("texel_y", 7), ("level", 3), ("level", 7), ("depth", 3)])
def test_projection_filter_at_its_decision_boundaries(dev, oracle, kind, flags):
    """The footprint-table kernel (deferred mode, k_basepass_as.hip) takes level and footprint origin of a meshlet's HZB lookup
    from a closed-form projection evaluated with v_rsq_f32 / v_rcp_f32 and decides on them only outside proven bands
    (cm::projectFiltered); anything inside a band is re-evaluated with the reference's exact square roots and divisions.  Here
    most meshlets are PUT on the boundaries those bands guard, in float64 from the float32 inputs, and then moved by 0, +-1,
    +-2 ... +-10^5 ulp of the moved parameter:
      texel_x / texel_y   the sphere centre is shifted along view-space x / y until uv * dim - 0.5 is an integer: the footprint
                          origin floor(...) flips there, and the bilinear weight of the second column / row is zero ON it;
      level               the radius is scaled until max(width, height) of the projected bounds is a power of two >= 2:
                          floor(log2(...)) flips there;
      depth               the radius is set so that nearPlane / (c.z - r) EQUALS the (constant) HZB depth.
    The HZB holds independent random depths in every texel of every level (except `depth`), so a lookup one texel or one level
    off decides differently about every second time.  Bit-exact against the oracle, like every other case; a build without the
    bands (-DTR_EXP_PROJ_NOBAND) fails the first four (profiles/r4/experiments.md)."""
    from toyrenderer_amd.frame import FrameDriver, GpuScene
    rng = np.random.default_rng(4200 + flags + len(kind))
    spec = synth.SceneSpec(num_meshes=700, num_instances=700, meshlets_lod0=64, max_lods=1, seed=4242, nonuniform_scale_fraction=0.3,
                           unique=True, z_near=6.0, z_far=120.0)
    scene = synth.make_scene(spec)
    inst, ml, md = scene.instances, scene.meshlets, scene.meshData
    view = synth.make_view(eye=(0.1, -0.1, 0.3), yaw=0.02, render=(1920, 1080))
    V = view.worldToView.astype(np.float64)
    P00, P11 = float(view.viewToClip[0, 0]), float(view.viewToClip[1, 1])
    near = float(view.nearPlane)
    hw, hh = view.hzb_dims
    hzb = oracle.HzbTexture(hw, hh)
    d0 = np.float16(near / 60.0)                                                                  # `depth`: the sphere depth equals it at c.z - r = near / d0 (~60)
    if kind == "depth":
        hzb.texels[:] = np.uint16(d0.view(np.uint16))
    else:
        hzb.texels[:] = np.exp(rng.uniform(np.log(near / 150.0), np.log(near / 5.0), hzb.total)).astype(np.float16).view(np.uint16)
    sparse = np.array([0, 0, 1, -1, 2, -2, 3, -3, 5, -5, 16, -16, 50, -50, 200, -200, 1000, -1000, 10 ** 4, -10 ** 4, 10 ** 5, -10 ** 5], np.int64)

    def pick_steps(n):
        # the float64 boundary is a few hundred ulp from where the float32 chains flip (their own rounding noise): half of the
        # meshlets are spread densely over +-600 ulp -- some land BETWEEN the reference's flip and the fast path's -- the others
        # take the sparse ladder out to +-10^5
        dense = rng.integers(-600, 601, n)
        return np.where(rng.random(n) < 0.5, dense, rng.choice(sparse, n))

    def ulps(x32, k):
        b = x32.view(np.int32).astype(np.int64)
        return (b + np.where(b >= 0, k, -k)).astype(np.int32).view(np.float32)

    moved = 0
    for i in range(len(inst)):
        W = inst["m_WorldMatrix"][i].astype(np.float64)
        lod = md["m_MeshLODDatas"][inst["m_MeshDataIdx"][i]][0]
        b, n = int(lod["m_MeshletDataBufferIdx"]), int(lod["m_NumMeshlets"])
        sph = ml["m_BoundingSphere"][b:b + n].astype(np.float64)
        R3 = W[:3, :3]
        scale = np.sqrt(max(R3[0] @ R3[0], R3[1] @ R3[1], R3[2] @ R3[2]))                          # toyrenderer_common.hlsli:134-140
        M = (W @ V)[:3, :3]                                                                         # view = [p, 1] W V, then z negated
        c = np.concatenate([sph[:, :3], np.ones((n, 1))], axis=1) @ W @ V
        cx, cy, cz = c[:, 0], c[:, 1], -c[:, 2]
        r = sph[:, 3] * scale
        Z = cz * cz - r * r
        ok = (cz > near + r) & (Z > 0) & (rng.random(n) < 0.85)
        with np.errstate(all="ignore"):
            vx, vy = np.sqrt(cx * cx + Z), np.sqrt(cy * cy + Z)
            w = P00 * hw * vx * r / Z                                                               # (maxx - minx) * 0.5 P00 * W, unclamped
            h = P11 * hh * vy * r / Z
            m = np.maximum(np.maximum(w, h), 1.0)
            level = np.minimum(np.floor(np.log2(m)), hzb.mips - 1).astype(np.int64)
            inside = (np.abs((cx * cz + vx * r) / Z * P00) < 0.98) & (np.abs((cx * cz - vx * r) / Z * P00) < 0.98) & \
                     (np.abs((cy * cz + vy * r) / Z * P11) < 0.98) & (np.abs((cy * cz - vy * r) / Z * P11) < 0.98)    # no clamp at the screen edge
        if kind in ("texel_x", "texel_y"):
            ax = 0 if kind == "texel_x" else 1
            dim = (np.maximum(hw >> level, 1) if ax == 0 else np.maximum(hh >> level, 1)).astype(np.float64)
            Pa, ca = (P00, cx) if ax == 0 else (-P11, cy)
            f = dim * (0.5 + 0.5 * Pa * ca * cz / Z) - 0.5                                          # uv * dim - 0.5 (the two bounds average to c c.z / Z)
            tgt = np.round(f)
            c_star = ((tgt + 0.5) / dim - 0.5) * Z / (0.5 * Pa * cz)
            delta = np.zeros((n, 3)); delta[:, ax] = c_star - ca
            dp = delta @ np.linalg.inv(M)                                                           # object-space shift with that view-space image
            take = ok & inside & (np.abs(c_star - ca) < 0.6 * r + 0.05) & np.isfinite(dp).all(axis=1)
            p_new = (sph[:, :3] + dp).astype(np.float32)
            j = np.argmax(np.abs(np.linalg.inv(M)[ax]))                                             # the component that moves it most
            p_new[:, j] = ulps(np.ascontiguousarray(p_new[:, j]), pick_steps(n))
            ml["m_BoundingSphere"][b:b + n, :3] = np.where(take[:, None], p_new, ml["m_BoundingSphere"][b:b + n, :3])
        elif kind == "level":
            # max(w, h) is monotone in r: bisect r (float64) onto the nearest power of two >= 2
            tgt = np.exp2(np.maximum(np.round(np.log2(np.maximum(m, 1.0))), 1.0))
            lo_, hi_ = r * 0.25, np.minimum(r * 4.0, cz * 0.45)
            for _ in range(60):
                mid = 0.5 * (lo_ + hi_)
                Zm = cz * cz - mid * mid
                mm = np.maximum(P00 * hw * np.sqrt(cx * cx + Zm) * mid / Zm, P11 * hh * np.sqrt(cy * cy + Zm) * mid / Zm)
                big = mm > tgt
                hi_ = np.where(big, mid, hi_); lo_ = np.where(big, lo_, mid)
            r_star = 0.5 * (lo_ + hi_)
            Zs = cz * cz - r_star * r_star
            reached = np.abs(np.maximum(P00 * hw * np.sqrt(cx * cx + Zs) * r_star / Zs, P11 * hh * np.sqrt(cy * cy + Zs) * r_star / Zs) / tgt - 1.0) < 1e-9
            take = ok & inside & reached & (tgt <= 2.0 ** (hzb.mips - 1))
            w_new = ulps((r_star / scale).astype(np.float32), pick_steps(n))
            ml["m_BoundingSphere"][b:b + n, 3] = np.where(take, w_new, ml["m_BoundingSphere"][b:b + n, 3])
        else:
            r_star = cz - near / float(d0)                                                          # nearPlane / (c.z - r) == d0
            take = ok & (r_star > 0.02) & (r_star < cz / 9.0)
            w_new = ulps((r_star / scale).astype(np.float32), pick_steps(n))
            ml["m_BoundingSphere"][b:b + n, 3] = np.where(take, w_new, ml["m_BoundingSphere"][b:b + n, 3])
        moved += int(take.sum())
    assert moved > (200 if kind == "depth" else 5000), moved
    d_cur = synth.gen_depth(view, num_occluders=40, seed=16, scale=3.0)
    cap = 1 << 19                                                                                   # the footprint-table kernel
    gs = GpuScene(dev, inst, md, ml, scene.opaqueIds, scene.alphaMaskIds)
    drv = FrameDriver(dev, gs, view, record_capacity=cap, culling_flags=flags)
    _upload_hzb(drv, hzb)
    drv.depth.upload_mip(0, d_cur)
    try:
        drv.record()
        drv.run()
        got = drv.results()
        ref = oracle.frame(scene.as_oracle(), view.as_dict(), hzb, d_cur, cullingFlags=flags, maxGroups=cap, record_capacity=cap)
        _compare_frame(got, ref)
        assert int(ref.meshletsTested[0]) > (1000 if kind == "depth" else 5000) and 0 < int(ref.drawArgs[0][0]) < int(ref.meshletsTested[0])
    finally:
        drv.release()
        gs.release()
