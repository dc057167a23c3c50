"""Multi-rank runs of the C++ host path with ranks SHARING one GPU (torch.distributed.run, gloo, host-staged
collectives), checked against the oracle's single-GPU frame of the whole scene.  Started by test_gpu_host_path.py:

    python -m torch.distributed.run --nproc-per-node N tests/mr_host_ranks.py uneven|raster <tmpdir>

uneven: the opaque ids are spread over ranks 0..N-2 and ALL alpha-mask ids sit on the last rank, so some ranks hold no
        alpha-mask list and one holds no opaque list: every rank must still post the same in-frame late-count collectives
        (BasePassRenderers.cpp GPUCulling, `list_presence_mask`).
q2:     every rank runs with the group capacity of the single-GPU run (far below what the scene needs); the exchange is given
        that capacity (`global_group_cap`) and must deliver exactly the 1-rank oracle frame's valid prefix and
        {X, 1, 1, validRecords}, wherever the first dropped instance lies (Q2 made global, gather.py).
raster: every rank rasterises the depth of its own shard's visible meshlets; the depth buffers are MAX-combined across
        ranks before each HZB build (`depth_allreduce_max`), so lists, depth and HZB equal the single-GPU frame."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def check_slots(ex, ref, tag):
    for s in range(4):
        if not ref.passRan[s]:
            continue
        rec, lst = ex.results(s)
        assert np.array_equal(rec, ref.records[s].view(np.uint32).reshape(-1, 3)), f"{tag} slot {s}: whole-scene records differ"
        assert np.array_equal(lst, ref.visibleList[s]), f"{tag} slot {s}: whole-scene visible list differs"


def main():
    mode, tmp = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from oracle import pyoracle as oracle
    from toyrenderer_amd import host, synth
    from toyrenderer_amd import interop as I
    from toyrenderer_amd.gather import NativeShardExchange, shard_range

    if mode == "uneven":
        render = (640, 360)
        spec = synth.SceneSpec(num_meshes=24, num_instances=700, meshlets_lod0=70, jitter_meshlets=True, max_lods=4, alpha_mask_fraction=0.3, seed=4321)
        scene = synth.make_scene(spec)
        view = synth.make_view(eye=(0.5, 0.2, 1.0), yaw=0.03, prev_eye=(0.0, 0.0, 0.0), prev_yaw=0.0, render=render)
        d_prev = synth.gen_depth(view, num_occluders=60, seed=11, scale=3.0)
        d_cur = synth.gen_depth(view, num_occluders=40, seed=12, scale=3.0)
        hzb = oracle.HzbTexture(*view.hzb_dims)
        hzb.build_from_depth(d_prev)
        op, am = scene.opaqueIds, scene.alphaMaskIds
        assert len(op) and len(am)
        if rank < world - 1:
            a, b = shard_range(len(op), rank, world - 1)
            my_op, my_am = op[a:b], am[:0]
        else:
            my_op, my_am = op[:0], am
        cap = 8192
        r = host.Renderer(render=render, max_groups=cap)
        try:
            r.load_scene(scene.instances, scene.meshData, scene.meshlets, my_op, my_am)
            r.set_culling(7)
            r.upload_hzb(hzb.texels, hzb.offsets)
            r.upload_depth(d_cur)
            ex = NativeShardExchange(r, dist, world, rank, slot_groups=cap, pass_slots=(0, 1, 2, 3), group_capacity=cap * world, stage_through_host=True,
                                     slot_runs=len(op) + len(am))     # one run per submitted instance
            for f in range(3):
                r.set_camera(view)
                r.frame()
                ex.run()
                ref = oracle.frame(scene.as_oracle(), view.as_dict(), hzb, d_cur, cullingFlags=7, maxGroups=cap * world, record_capacity=cap * world)
                assert ref.passRan[2] and ref.passRan[3] and int(ref.lateCount[1]) > 0
                check_slots(ex, ref, f"frame {f}")
            ex.close()
        finally:
            r.shutdown()
    elif mode == "q2":
        render = (640, 360)
        spec = synth.SceneSpec(num_meshes=24, num_instances=900, meshlets_lod0=70, jitter_meshlets=True, max_lods=4, seed=777)
        scene = synth.make_scene(spec)
        view = synth.make_view(eye=(0.5, 0.2, 1.0), yaw=0.03, prev_eye=(0.0, 0.0, 0.0), prev_yaw=0.0, render=render)
        d_prev = synth.gen_depth(view, num_occluders=60, seed=11, scale=3.0)
        d_cur = synth.gen_depth(view, num_occluders=40, seed=12, scale=3.0)
        op = scene.opaqueIds
        a, b = shard_range(len(op), rank, world)
        h0 = oracle.HzbTexture(*view.hzb_dims); h0.build_from_depth(d_prev)
        free = oracle.frame(scene.as_oracle(), view.as_dict(), h0, d_cur, cullingFlags=7, maxGroups=1 << 20, record_capacity=1 << 16)
        GE = int(free.dispatchArgs[0][0])
        assert GE > 300
        for cap in (GE // (2 * world) + 1, GE // 2, GE - 3):          # the first dropped instance on rank 0, in the middle, on the last rank
            hzb = oracle.HzbTexture(*view.hzb_dims)
            hzb.build_from_depth(d_prev)
            r = host.Renderer(render=render, max_groups=cap)
            try:
                r.load_scene(scene.instances, scene.meshData, scene.meshlets, op[a:b], np.zeros(0, np.uint32))
                r.set_culling(7)
                r.upload_hzb(hzb.texels, hzb.offsets)
                r.upload_depth(d_cur)
                ex = NativeShardExchange(r, dist, world, rank, slot_groups=2 * cap, pass_slots=(0, 1), group_capacity=2 * cap * world,     # a slot holds the early AND the late pass
                                         stage_through_host=True, slot_runs=len(op), global_group_cap=cap)
                for f in range(2):
                    r.set_camera(view)
                    r.frame()
                    ex.run()
                    ref = oracle.frame(scene.as_oracle(), view.as_dict(), hzb, d_cur, cullingFlags=7, maxGroups=cap, record_capacity=cap)
                    assert int(ref.validRecords[0]) < int(ref.dispatchArgs[0][0]), "the capacity must bite"
                    check_slots(ex, ref, f"cap {cap} frame {f}")
                ex.close()
            finally:
                r.shutdown()
    elif mode == "raster":
        from scene_gen import write_city_gltf
        from toyrenderer_amd import gltf_lite
        import pathlib
        if rank == 0:
            write_city_gltf(pathlib.Path(tmp))
        dist.barrier()
        s = gltf_lite.load(os.path.join(tmp, "city.gltf"))
        inst = s.instances.copy()
        oracle.update_instance_consts(s.nodes, s.primToNode, inst)
        sc = dict(s.as_oracle()); sc["instances"] = inst
        cam = s.cameras[0]
        render = (1280, 720)
        P = synth.perspective_rh_reverse_z_infinite(cam.yfov, render[0] / render[1], cam.znear)
        hzb = oracle.HzbTexture(*I.hzb_dims(*render))
        depth = np.zeros((render[1], render[0]), np.float32)
        a, b = shard_range(len(s.opaqueIds), rank, world)
        c, d = shard_range(len(s.alphaMaskIds), rank, world)
        cap = 4096
        r = host.Renderer(render=render, max_groups=cap)
        try:
            r.load_scene(s.instances, s.meshData, s.meshlets, s.opaqueIds[a:b], s.alphaMaskIds[c:d])
            r.load_nodes(s.nodes, s.primToNode)
            r.load_geometry(s.vertices, s.meshletVertexIds, s.meshletTriangles)
            r.set_raster_depth(True)
            r.set_culling(7)
            if os.environ.get("TR_TEST_NO_DEPTH_REDUCE"):
                # the combination without the depth reduction must be REJECTED, not silently diverge
                ex = NativeShardExchange(r, dist, world, rank, slot_groups=cap, pass_slots=(0, 1, 2, 3), group_capacity=cap, stage_through_host=True)
                r.set_camera(synth.View(synth.world_to_view((0.0, 0.0, 0.0), cam.orientation), synth.world_to_view((0.0, 0.0, 0.0), cam.orientation), P, float(np.float32(cam.znear)), *render))
                try:
                    r.frame()
                except host.HostError as e:
                    assert "depth_allreduce_max" in str(e), e
                    print(f"[rank {rank}] rejected as expected")
                else:
                    raise AssertionError("raster depth + exchange without a depth reduction was accepted")
                return
            ex = NativeShardExchange(r, dist, world, rank, slot_groups=cap, pass_slots=(0, 1, 2, 3), group_capacity=cap, stage_through_host=True, raster_depth=True)
            prevV = synth.world_to_view((0.0, 0.0, 0.0), cam.orientation)
            for f, eye in enumerate([(0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.4, 0.1, -0.3), (0.9, 0.1, -0.5)]):
                V = synth.world_to_view(eye, cam.orientation)
                view = synth.View(V, prevV, P, float(np.float32(cam.znear)), *render)
                prevV = V
                r.set_node_transforms(s.nodes)
                r.set_camera(view)
                r.frame()
                ex.run()
                geo = (I.world_to_clip(V, P), s.vertices, s.meshletVertexIds, s.meshletTriangles)
                ref = oracle.frame(sc, view.as_dict(), hzb, depth, cullingFlags=7, record_capacity=cap, maxGroups=cap, raster=geo)
                check_slots(ex, ref, f"frame {f}")
                assert np.array_equal(r.download_depth().view(np.uint32), depth.view(np.uint32)), f"frame {f}: depth differs from the single-GPU frame"
                assert np.array_equal(r.download_hzb(), hzb.texels), f"frame {f}: HZB chain differs from the single-GPU frame"
            assert np.count_nonzero(depth) > 0.2 * depth.size
            ex.close()
        finally:
            r.shutdown()
    elif mode == "overflow":
        # a shard slot too small for a rank's groups poisons the frame (status bit 1 in the unpacked arguments):
        # trhost_exchange_wait must report it instead of handing out a silently wrong frame.
        render = (640, 360)
        spec = synth.SceneSpec(num_meshes=24, num_instances=600, meshlets_lod0=70, jitter_meshlets=True, max_lods=2, seed=99)
        scene = synth.make_scene(spec)
        view = synth.make_view(eye=(0.5, 0.2, 1.0), yaw=0.03, render=render)
        op = scene.opaqueIds
        a, b = shard_range(len(op), rank, world)
        cap = 8192
        r = host.Renderer(render=render, max_groups=cap)
        try:
            r.load_scene(scene.instances, scene.meshData, scene.meshlets, op[a:b], np.zeros(0, np.uint32))
            r.set_culling(5)
            ex = NativeShardExchange(r, dist, world, rank, slot_groups=64, pass_slots=(0,), group_capacity=cap, stage_through_host=True, slot_runs=len(op))
            r.set_camera(view)
            r.frame()
            ex.run()
            try:
                ex.wait()
            except host.HostError as e:
                assert "status" in str(e) and "slot_groups" in str(e), e
            else:
                raise AssertionError("an overflowed shard slot went unreported")
            ex.close()
        finally:
            r.shutdown()
    else:
        raise SystemExit(f"unknown mode {mode}")
    dist.barrier()
    print(f"[rank {rank}] {mode}: ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
