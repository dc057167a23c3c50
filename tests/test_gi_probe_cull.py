"""CS_VisualizeGIProbesCulling (giprobevisualization.hlsl:16-69, host GIRenderer.cpp:690-735), the
second consumer of FrustumCull / OcclusionCull + compaction into DrawIndexedIndirectArguments.  Probe world positions
and states are input buffers (the reference derives them from the RTXGI-DDGI volume, an absent SDK); everything after
`probeWorldPosition` (:39-67) is restated.  PARITY UNPINNED like the rest of the path (no reference fixtures).

CPU: C oracle == numpy restatement (written from the HLSL text, vectorised).  GPU: HIP kernel == C oracle, bit for bit,
in ascending probe order, with and without "hide inactive", appended behind a non-zero m_InstanceCount, for probe
counts around the tile size, and through the C++ host mirror's GIDebugRenderer."""
import numpy as np
import pytest

from oracle import np_oracle as NP
from toyrenderer_amd import interop as I
from toyrenderer_amd import synth


def _setup(oracle, n, seed=3, render=(1280, 720), radius=0.35):
    rng = np.random.default_rng(seed)
    view = synth.make_view(eye=(0.4, 0.2, 0.8), yaw=0.04, render=render)
    depth = synth.gen_depth(view, num_occluders=80, seed=21, scale=3.0)
    hzb = oracle.HzbTexture(*view.hzb_dims)
    hzb.build_from_depth(depth)
    # a regular probe grid (what a DDGI volume is) jittered by the per-probe offsets, in front of and around the camera
    g = int(np.ceil(n ** (1 / 3)))
    ix = np.arange(g ** 3)[:n]
    grid = np.stack([ix % g, (ix // g) % g, ix // (g * g)], 1).astype(np.float32)
    pos = (grid / max(g - 1, 1) - 0.5) * np.array([160.0, 90.0, 180.0], np.float32) + np.array([0.0, 0.0, -95.0], np.float32)
    pos = (pos + rng.uniform(-0.4, 0.4, pos.shape)).astype(np.float32)
    states = (rng.random(n) < 0.3).astype(np.float32)                 # 1 = RTXGI_DDGI_PROBE_STATE_INACTIVE
    k = np.zeros(1, I.GIProbeVisualizationUpdateConsts)
    k["m_NumProbes"] = n
    k["m_Frustum"] = oracle.culling_frustum(view.viewToClip)
    k["m_WorldToView"] = view.worldToView
    k["m_HZBDimensions"] = view.hzb_dims
    k["m_P00"] = view.viewToClip[0, 0]; k["m_P11"] = view.viewToClip[1, 1]
    k["m_NearPlane"] = view.nearPlane
    k["m_ProbeRadius"] = radius
    return view, hzb, pos, states, k


@pytest.mark.parametrize("hide", [0, 1])
def test_oracle_equals_numpy_restatement(oracle, hide):
    view, hzb, pos, states, k = _setup(oracle, 5000)
    k["m_bHideInactiveProbes"] = hide
    out_pos, args, out_idx = oracle.gi_probe_cull(k, pos, states, hzb)
    # numpy restatement of giprobevisualization.hlsl:29-67
    v = NP.to_view(pos, view.worldToView)
    r = np.full(len(pos), k["m_ProbeRadius"][0], np.float32)
    vis = NP.frustum_visible(v, r, k["m_Frustum"][0])
    vis &= NP.occlusion_visible(v, r, view.nearPlane, k["m_P00"][0], k["m_P11"][0], hzb)
    if hide:
        vis &= states != 1.0
    want = np.nonzero(vis)[0].astype(np.uint32)
    assert 0 < len(want) < len(pos)
    assert np.array_equal(out_idx, want) and int(args[1]) == len(want)
    assert np.array_equal(out_pos.view(np.uint32), pos[want].view(np.uint32))


def test_appends_behind_the_incoming_instance_count(oracle):
    view, hzb, pos, states, k = _setup(oracle, 700)
    a = np.array([36, 5, 0, 0, 0], np.uint32)
    out_pos, args, out_idx = oracle.gi_probe_cull(k, pos, states, hzb, draw_args=a)
    assert int(args[0]) == 36 and int(args[1]) == 5 + len(out_idx) and len(out_idx) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,hide,base", [(1, 0, 0), (255, 1, 0), (256, 0, 3), (257, 1, 0), (5000, 0, 0), (5000, 1, 7), (200_000, 1, 0)])
def test_gpu_kernel_equals_oracle(oracle, n, hide, base):
    from toyrenderer_amd import rhi
    dev = rhi.Device(0)
    try:
        view, hzb, pos, states, k = _setup(oracle, n, seed=n)
        k["m_bHideInactiveProbes"] = hide
        a0 = np.array([36, base, 0, 0, 0], np.uint32)               # GIRenderer.cpp:687-689: {indexCount of the unit sphere, 0, ...}
        ref_pos, ref_args, ref_idx = oracle.gi_probe_cull(k, pos, states, hzb, draw_args=a0)
        tex = dev.create_texture(hzb.w, hzb.h, hzb.mips, rhi.FORMAT_R16_FLOAT, "HZB")
        tex.upload_chain(hzb.texels, hzb.offsets)
        b_pos = dev.buffer_from(pos, "probe positions", uav=False)
        b_st = dev.buffer_from(states, "probe states", uav=False)
        b_out = dev.create_buffer(12 * (n + base), "Probe Positions", stride=12)
        b_args = dev.create_buffer(20, "Probe Draw Indirect Args", stride=20, indirect=True)
        b_idx = dev.create_buffer(4 * (n + base), "Instance ID to Probe Index")
        cl = dev.create_command_list()
        cl.open()
        cl.write_buffer(b_args, a0)
        cb = cl.constant_buffer(k)
        cl.dispatch("giprobevisualization_CS_VisualizeGIProbesCulling",
                    [rhi.CB(0, cb), rhi.TEX_SRV(0, tex), rhi.SRV(10, b_pos), rhi.SRV(11, b_st), rhi.UAV(0, b_out), rhi.UAV(1, b_args), rhi.UAV(2, b_idx), rhi.SAMPLER(0)],
                    ((n + 31) // 32, 1, 1))
        cl.close()
        for _ in range(2):                                           # twice: the recording (incl. its clears) is replayable
            dev.execute(cl)
            dev.wait_idle()
            args = b_args.download(np.uint32, 5)
            assert np.array_equal(args, ref_args), (args, ref_args)
            cnt = int(args[1]) - base
            assert np.array_equal(b_idx.download(np.uint32, base + cnt)[base:], ref_idx)
            assert np.array_equal(b_out.download(np.uint32, 3 * (base + cnt))[3 * base:], ref_pos.view(np.uint32).ravel())
        if n >= 5000:
            assert 0 < cnt < n
        for r in (cl, cb, tex, b_pos, b_st, b_out, b_args, b_idx):
            r.release()
    finally:
        dev.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("hide,async_compute", [(False, False), (True, False), (True, True)])
def test_through_the_host_mirror(oracle, hide, async_compute):
    """GIDebugRenderer (csrc/host/GIRenderer.cpp) scheduled behind the base pass: culls the probes against the HZB the
    frame has just built, three frames, results == the oracle with that HZB.  async_compute: the pass records for the
    render graph's COMPUTE queue (second stream); the graph finds its read of the HZB the base pass writes and makes the
    compute queue wait for that pass (RenderGraph.cpp:251 "TODO: compute queue")."""
    from toyrenderer_amd import host
    render = (1280, 720)
    view, hzb, pos, states, k = _setup(oracle, 20_000, seed=5, render=render)
    k["m_bHideInactiveProbes"] = int(hide)
    spec = synth.SceneSpec(num_meshes=8, num_instances=100, meshlets_lod0=40, max_lods=2, seed=3)
    scene = synth.make_scene(spec)
    depth = synth.gen_depth(view, num_occluders=80, seed=21, scale=3.0)       # the image _setup built `hzb` from
    r = host.Renderer(render=render)
    try:
        r.load_scene(scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds, scene.alphaMaskIds)
        r.load_gi_probes(pos, states, float(k["m_ProbeRadius"][0]), hide)
        r.set_renderer_queue("GIDebugRenderer", async_compute)
        r.set_culling(7)
        r.upload_depth(depth)
        for _ in range(3):
            r.set_camera(view)
            r.frame()
            got_pos, got_args, got_idx = r.gi_probe_results()
            assert np.array_equal(r.download_hzb(), hzb.texels)               # the frame's GenerateHZB ran before the probe pass
            ref_pos, ref_args, ref_idx = oracle.gi_probe_cull(k, pos, states, hzb, draw_args=np.array([2880, 0, 0, 0, 0], np.uint32))
            assert np.array_equal(got_args, ref_args) and 0 < int(got_args[1]) < len(pos)
            assert np.array_equal(got_idx, ref_idx) and np.array_equal(got_pos.view(np.uint32), ref_pos.view(np.uint32))
        assert r.render_graph_stats()["passes"] == 2
        fs = r.render_graph_frame_stats()
        assert fs["compute_queue_passes"] == int(async_compute) and fs["cross_queue_waits"] == int(async_compute)
        assert 0 < fs["aliased_bytes"] <= fs["transient_bytes"]
        # the probe pass's three buffers are live in another pass than the base pass's: aliasing could share their memory
        assert fs["aliased_bytes"] < fs["transient_bytes"]
    finally:
        r.shutdown()
