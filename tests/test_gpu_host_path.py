"""Parity through the DROP-IN path: C++ host mirror (Graphic / Scene / RenderGraph / IRenderer /
BasePassRenderers / AddComputePass, toyrenderer_amd/csrc/host) -> C ABI -> HIP kernels, against the
CPU oracle.  Bit-exact.  PARITY UNPINNED with respect to the reference itself.
Needs a real MI355X."""
import numpy as np
import pytest

from toyrenderer_amd import interop as I
from toyrenderer_amd import synth

pytestmark = pytest.mark.gpu

SMALL = synth.SceneSpec(num_meshes=24, num_instances=300, meshlets_lod0=70, jitter_meshlets=True, max_lods=5,
                        alpha_mask_fraction=0.15, seed=1234)


def _compare(got, ref):
    for s in range(4):
        if not ref.passRan[s]:
            assert got[s] is None, f"slot {s} ran on the GPU but not in the oracle"
            continue
        g = got[s]
        assert g is not None, f"slot {s} did not run"
        assert np.array_equal(g["dispatchArgs"], ref.dispatchArgs[s]), (s, g["dispatchArgs"], ref.dispatchArgs[s])
        assert g["validRecords"] == int(ref.validRecords[s])
        assert np.array_equal(g["records"].view(np.uint32), ref.records[s].view(np.uint32)), f"slot {s}: records"
        assert np.array_equal(g["visMask"], ref.visMask[s]), f"slot {s}: masks"
        assert np.array_equal(g["drawArgs"], ref.drawArgs[s]), f"slot {s}: draw args"
        assert np.array_equal(g["visibleList"], ref.visibleList[s]), f"slot {s}: visible list"


class _Ctx:
    def __init__(self, render, **kw):
        from toyrenderer_amd import host
        self.r = host.Renderer(render=render, **kw)

    def __enter__(self):
        return self.r

    def __exit__(self, *a):
        self.r.shutdown()


@pytest.mark.parametrize("flags", [7, 5, 2, 0])
def test_frames_through_rendergraph(oracle, flags):
    view = synth.make_view(eye=(0.5, 0.2, 1.0), yaw=0.03, prev_eye=(0.0, 0.0, 0.0), prev_yaw=0.0, render=(640, 360))
    scene = synth.make_scene(SMALL)
    d_prev = synth.gen_depth(view, num_occluders=60, seed=11, scale=3.0)
    d_cur = synth.gen_depth(view, num_occluders=40, seed=12, scale=3.0)
    hzb = oracle.HzbTexture(*view.hzb_dims)
    hzb.build_from_depth(d_prev)
    with _Ctx((640, 360)) as r:
        assert (r.hzb_w, r.hzb_h, r.hzb_mips) == (hzb.w, hzb.h, hzb.mips)
        r.load_scene(scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
        r.set_culling(flags)
        r.upload_hzb(hzb.texels, hzb.offsets)
        r.upload_depth(d_cur)
        for frame in range(3):          # frame 2 also exercises transient-resource reuse across frames
            r.set_camera(view)
            r.frame()
            got = r.results()
            ref = oracle.frame(scene.as_oracle(), view.as_dict(), hzb, d_cur, cullingFlags=flags, maxGroups=65535, record_capacity=65535)
            _compare(got, ref)
            if flags & 2:
                assert np.array_equal(r.download_hzb(), hzb.texels)
                assert got["lateCount"] == int(ref.lateCount[1])
        stats = r.render_graph_stats()
        assert stats["passes"] == 1 and stats["heaps"] >= 1 and stats["used"] > 0
        cpu_ms, gpu_ms = r.renderer_times("GBufferRenderer")
        assert cpu_ms > 0 and gpu_ms > 0


def test_group_cap_q2_through_host(oracle):
    view = synth.make_view(render=(640, 360))
    spec = synth.SceneSpec(num_meshes=10, num_instances=500, meshlets_lod0=90, jitter_meshlets=True, max_lods=1, seed=77)
    scene = synth.make_scene(spec)
    hzb = oracle.HzbTexture(*view.hzb_dims)
    with _Ctx((640, 360), max_groups=257) as r:
        r.load_scene(scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
        r.set_culling(1)
        r.set_camera(view)
        r.frame()
        got = r.results()
        ref = oracle.frame(scene.as_oracle(), view.as_dict(), hzb, None, cullingFlags=1, maxGroups=257, record_capacity=257)
        _compare(got, ref)
        assert ref.dispatchArgs[0][0] > 257 > ref.validRecords[0]


@pytest.mark.parametrize("async_compute,rebuild_cache", [(False, False), (False, True), (True, False)])
def test_animated_transforms_then_cull(oracle, monkeypatch, async_compute, rebuild_cache):
    """configs[4] in miniature: UpdateInstanceConstsRenderer rewrites the world matrices on the GPU every
    frame from the node hierarchy, then the cull runs on them.  async_compute: the update pass records for the compute
    queue; the base pass (graphics queue) reads the instance buffer it writes, so the graph makes it wait.
    From the second frame on the update kernel also refreshes the transform-dependent entries of the instance cull cache
    in place (k_updateinstance.hip); rebuild_cache switches that off (the next cull pass then rebuilds the whole cache):
    both ways the frames equal the oracle's."""
    if rebuild_cache:
        monkeypatch.setenv("TRHIP_NO_CACHE_REFRESH", "1")
    rng = np.random.default_rng(3)
    view = synth.make_view(render=(640, 360))
    spec = synth.SceneSpec(num_meshes=12, num_instances=200, meshlets_lod0=40, jitter_meshlets=True, max_lods=3, seed=9)
    scene = synth.make_scene(spec)
    n_nodes = 260
    nodes = np.zeros(n_nodes, I.NodeLocalTransform)
    nodes["m_ParentNodeIdx"] = 0xFFFFFFFF
    for i in range(200, n_nodes):                      # 60 group nodes; instance nodes hang below some of them
        nodes["m_ParentNodeIdx"][i] = 0xFFFFFFFF if i % 3 == 0 else rng.integers(200, i) if i > 200 else 0xFFFFFFFF
    nodes["m_ParentNodeIdx"][:200] = np.where(rng.random(200) < 0.6, rng.integers(200, n_nodes, 200), 0xFFFFFFFF)
    nodes["m_Rotation"][:, 3] = 1.0
    nodes["m_Scale"] = 1.0
    prim_to_node = np.arange(200, dtype=np.uint32)
    hzb = oracle.HzbTexture(*view.hzb_dims)
    with _Ctx((640, 360)) as r:
        r.load_scene(scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
        r.load_nodes(nodes, prim_to_node)
        r.set_renderer_queue("UpdateInstanceConstsRenderer", async_compute)
        r.set_culling(5)
        inst = scene.instances.copy()
        for frame in range(3):
            nodes["m_Position"][:200] = scene.instances["m_WorldMatrix"][:, 3, :3] + rng.standard_normal((200, 3)).astype(np.float32) * 0.5
            nodes["m_Position"][200:] = rng.standard_normal((n_nodes - 200, 3)).astype(np.float32) * 0.2
            q = rng.standard_normal((n_nodes, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
            nodes["m_Rotation"] = q.astype(np.float32)
            nodes["m_Scale"] = rng.uniform(0.7, 1.5, (n_nodes, 3)).astype(np.float32)
            r.set_node_transforms(nodes)
            r.set_camera(view)
            r.frame()
            oracle.update_instance_consts(nodes, prim_to_node, inst)
            got_inst = r.instances(200)
            assert np.array_equal(got_inst.view(np.uint32), inst.view(np.uint32)), f"frame {frame}: instance matrices"
            sc = dict(scene.as_oracle()); sc["instances"] = inst
            ref = oracle.frame(sc, view.as_dict(), hzb, None, cullingFlags=5, maxGroups=65535, record_capacity=65535)
            _compare(r.results(), ref)
            fs = r.render_graph_frame_stats()
            assert fs["compute_queue_passes"] == int(async_compute) and fs["cross_queue_waits"] == int(async_compute)


def test_headline_config_subsample_spot_check(oracle):
    """BASELINE configs[3] shape (unique meshlets, 128 per instance, one LOD) at 1/64 scale through the
    host path, full 2-phase flags, bit-exact against the oracle."""
    view = synth.make_view(eye=(0.0, 0.0, 0.0), prev_eye=(0.05, 0.0, 0.1), prev_yaw=0.002)
    spec = synth.config_spec("C3s")
    scene = synth.make_scene(spec)
    depth = synth.gen_depth(view, 200)
    hzb = oracle.HzbTexture(*view.hzb_dims)
    cap = spec.num_instances * 4 + 1
    with _Ctx((3840, 2160), max_groups=cap) as r:
        r.load_scene(scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
        r.set_culling(7)
        r.upload_depth(depth)
        for frame in range(2):
            r.set_camera(view)
            r.frame()
            ref = oracle.frame(scene.as_oracle(), view.as_dict(), hzb, depth, cullingFlags=7, maxGroups=cap, record_capacity=cap, threads=8)
            _compare(r.results(), ref)
        assert ref.meshletsTested[0] > 500_000 and 0 < ref.drawArgs[0][0] < ref.meshletsTested[0]


@pytest.mark.parametrize("direct", [True, False])
def test_bench_rccl_gather_path_single_rank(direct):
    """bench.py end to end on a small unique-meshlet config with the RCCL gather forced on a 1-rank
    group: exercises device-pointer wrapping, the pack / all-gather / unpack pipeline on two
    streams with double buffering, and checks the gathered whole-scene lists against the local ones.
    direct=False: the fallback that calls the process group's all_gather_into_tensor (TR_NO_DIRECT_RCCL)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TR_FORCE_GATHER="1")
    if not direct:
        env["TR_NO_DIRECT_RCCL"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "C3s", "--steps", "3", "--warmup", "2",
                        "--cpu-sample-instances", "4096"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["gather_checked"] is True
    assert ("using torch.distributed all_gather" in p.stderr) == (not direct)
    assert out["value"] > 0 and out["roofline"]["achieved"] > 0 and out["cpu_baseline"]["value"] > 0
    assert out["config"]["meshlets_tested_per_frame"] > 100_000


@pytest.mark.parametrize("flags", [5, 7])
def test_cornell_scene_through_the_drop_in_path(oracle, flags):
    """BASELINE.json configs[0] on the GPU: the scene derived from the reference's cornell.gltf (fixture
    tests/golden/cornell_scene.npz, tests/golden/make_cornell.py) -- node transforms -> world matrices on the GPU
    (UpdateInstanceConstsRenderer), then the cull; outputs equal the fixture's (= the oracle's) bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gltf_cornell import _cull, _fixture
    from toyrenderer_amd import gltf_lite
    z, scene, camera = _fixture()
    view = gltf_lite.view_of(camera, (1920, 1080))
    with _Ctx((1920, 1080)) as r:
        r.load_scene(scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
        r.load_nodes(scene.nodes, scene.primToNode)
        r.set_culling(flags)
        if flags & 2:
            r.upload_depth(np.zeros((1080, 1920), np.float32))
        for frame in range(2):
            r.set_node_transforms(scene.nodes)
            r.set_camera(view)
            r.frame()
        got = r.results()
        assert np.array_equal(r.instances(3)["m_WorldMatrix"], z[f"world_{flags}"])
    inst, ref = _cull(oracle, scene, view, flags)
    for s in (0, 1):
        if f"f{flags}_s{s}_records" not in z.files:
            continue
        assert np.array_equal(got[s]["records"].view(np.uint32).reshape(-1, 3), z[f"f{flags}_s{s}_records"]), f"slot {s}: records"
        assert np.array_equal(got[s]["visMask"], z[f"f{flags}_s{s}_visMask"]), f"slot {s}: masks"
        assert np.array_equal(got[s]["visibleList"], z[f"f{flags}_s{s}_visibleList"]), f"slot {s}: visible list"
    assert int(got[0]["drawArgs"][0]) == int(ref.drawArgs[0][0]) >= 3


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_equal_one_rank(world):
    """bench.py with 2 and 3 ranks (torch.distributed.run, gloo with host-staged collectives because RCCL needs a GPU per
    rank) sharing this GPU: instance sharding, the in-frame late-count exchange between real ranks, shard-slot
    all-gather, unpack -- the whole-scene records and visible lists (sha1 digest) must equal a single-rank run's."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--config", "C3s", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-profile"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *common], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    # `python bench.py --gpus N` with no launcher around it (the form the driver uses): bench.py starts its own ranks
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TR_DIST_BACKEND"] = "gloo"
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), *common],
                         env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    d2 = json.loads([l for l in two.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d2["n_gpus"] == world and d2["config"]["meshlets_tested_per_frame"] == d1["config"]["meshlets_tested_per_frame"]
    assert d2["config"]["visible_per_frame"] == d1["config"]["visible_per_frame"]
    assert d2["lists_digest"] == d1["lists_digest"], "multi-rank whole-scene lists differ from the 1-rank lists"
    assert d2["collective"] == "host-staged" and d1["collective"] is None


def _run_ranks(world, mode, tmp_path, extra_env=None):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import free_rendezvous_port
    port = free_rendezvous_port()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tests", "mr_host_ranks.py"), mode, str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    return p.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_with_different_kinds_of_lists(world, tmp_path):
    """the last rank holds every alpha-mask id and no opaque id, the others only opaque ids.  All ranks post the
    same in-frame late-count collectives (a rank without a list contributes 0); whole-scene results == the oracle's frame."""
    out = _run_ranks(world, "uneven", tmp_path)
    assert out.count("uneven: ok") == world


def test_an_overflowed_shard_slot_is_reported_by_the_exchange(tmp_path):
    """trhost_exchange_wait reads the unpacked status words; a slot too small for a rank's groups is an error."""
    out = _run_ranks(2, "overflow", tmp_path)
    assert out.count("overflow: ok") == 2


def test_sharded_raster_depth_equals_single_gpu(tmp_path):
    """with self-rasterised depth every rank draws only its shard; the depth buffers are MAX-combined across the
    ranks before each HZB build, so lists, depth and HZB equal the single-GPU frame -- and without the reduction the
    combination is rejected."""
    out = _run_ranks(2, "raster", tmp_path)
    assert out.count("raster: ok") == 2
    out = _run_ranks(2, "raster", tmp_path, {"TR_TEST_NO_DEPTH_REDUCE": "1"})
    assert out.count("rejected as expected") == 2


@pytest.mark.parametrize("block", range(2))
def test_randomised_sweep_through_the_host_mirror(oracle, block):
    """24 random configurations through Graphic / RenderGraph / GBufferRenderer (one Renderer per case: initialise,
    load, 1-3 frames, shut down): scene sizes from 1 instance up, alpha-mask-only and opaque-only scenes, all flag
    combinations, forced LODs, group capacities below what the frame needs, odd render sizes, freeze-culling-camera."""
    renders = [(640, 360), (100, 40), (1280, 720), (333, 517), (64, 64), (2048, 64)]
    for case in range(block * 12, block * 12 + 12):
        rng = np.random.default_rng(7000 + case)
        n_inst = int(rng.choice([1, 2, 33, 64, 255, 1024, 3000]))
        m0 = int(rng.choice([1, 31, 33, 64, 128, 200]))
        spec = synth.SceneSpec(num_meshes=int(rng.integers(1, 30)), num_instances=n_inst, meshlets_lod0=m0,
                               jitter_meshlets=bool(rng.integers(0, 2)), max_lods=int(rng.integers(1, 9)),
                               alpha_mask_fraction=float(rng.choice([0.0, 0.2, 1.0])), seed=9000 + case,
                               z_near=float(rng.choice([2.0, 5.0, 20.0])), z_far=float(rng.choice([40.0, 200.0])),
                               box_x=float(rng.choice([10.0, 100.0])), box_y=float(rng.choice([6.0, 56.0])))
        render = renders[int(rng.integers(0, len(renders)))]
        view = synth.make_view(eye=tuple(rng.uniform(-0.5, 0.5, 3)), yaw=float(rng.uniform(-0.05, 0.05)),
                               prev_eye=tuple(rng.uniform(-0.5, 0.5, 3)), prev_yaw=float(rng.uniform(-0.05, 0.05)), render=render)
        flags = int(rng.integers(0, 8))
        forced = int(rng.choice([-1, -1, 0, 3]))
        freeze = bool(rng.integers(0, 5) == 0)
        need = n_inst * ((2 * m0 + 31) // 32) + 1
        cap = min(int(rng.choice([65535, need, max(1, need // 3), 2048])), 65535)
        scene = synth.make_scene(spec)
        d_cur = synth.gen_depth(view, num_occluders=int(rng.integers(0, 80)), seed=case + 500, scale=3.0)
        hzb = oracle.HzbTexture(*view.hzb_dims)
        hzb.build_from_depth(synth.gen_depth(view, num_occluders=int(rng.integers(0, 80)), seed=case, scale=3.0))
        try:
            with _Ctx(render, max_groups=cap) as r:
                r.load_scene(scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
                r.upload_hzb(hzb.texels, hzb.offsets)
                r.upload_depth(d_cur)
                for frame in range(int(rng.integers(1, 4))):
                    fz = freeze and frame > 0         # the culling camera is frozen at the view of the last unfrozen frame (Scene.cpp:139-144)
                    r.set_culling(flags, freeze=fz, force_mesh_lod=forced)
                    r.set_camera(view)
                    r.frame()
                    got = r.results()
                    ref = oracle.frame(scene.as_oracle(), view.as_dict(), hzb, d_cur, cullingFlags=flags, forceMeshLOD=forced, freeze=fz,
                                       maxGroups=cap, record_capacity=cap)
                    _compare(got, ref)
                    if flags & 2:
                        assert np.array_equal(r.download_hzb(), hzb.texels), "HZB chain"
        except AssertionError as e:
            raise AssertionError(f"case {case}: {spec} render={render} flags={flags} forced={forced} freeze={freeze} cap={cap}: {e}") from e


@pytest.mark.parametrize("world", [2, 3])
def test_group_capacity_drop_is_global_with_real_ranks(world, tmp_path):
    """Q2 (gpuculling.hlsl:64-74) under sharding through the C++ host path: real ranks (sharing the one GPU, gloo), every
    rank at the single-GPU group capacity, which bites; the exchange (`global_group_cap`) delivers the 1-rank oracle frame."""
    out = _run_ranks(world, "q2", tmp_path)
    assert out.count("q2: ok") == world
