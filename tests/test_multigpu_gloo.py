"""The N > 1 path on CPU: world_size-2 (and 3) gloo processes shard the instance list by contiguous
ranges, each cull their shard (with the oracle standing in for the GPU), and the variable-length
all-gather + group rebase of toyrenderer_amd/gather.py must reproduce the single-process lists bit for
bit (rank-major concatenation == single-GPU canonical order)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle
        from toyrenderer_amd import gather, synth
        spec = synth.SceneSpec(num_meshes=30, num_instances=257, meshlets_lod0=50, jitter_meshlets=True, max_lods=4, seed=99)
        scene = synth.make_scene(spec)
        view = synth.make_view(eye=(0.3, 0.1, 0.4), yaw=0.02, prev_eye=(0, 0, 0), prev_yaw=0.0, render=(640, 360))
        d_prev = synth.gen_depth(view, 50, seed=5, scale=3.0)
        d_cur = synth.gen_depth(view, 40, seed=6, scale=3.0)

        def run(ids):
            hzb = pyoracle.HzbTexture(*view.hzb_dims)
            hzb.build_from_depth(d_prev)
            sc = dict(scene.as_oracle()); sc["opaqueIds"] = ids; sc["alphaMaskIds"] = np.zeros(0, np.uint32)
            return pyoracle.frame(sc, view.as_dict(), hzb, d_cur, cullingFlags=7, maxGroups=1 << 20, record_capacity=4096)

        all_ids = np.arange(spec.num_instances, dtype=np.uint32)
        i0, i1 = gather.shard_range(spec.num_instances, rank, world)
        local = run(all_ids[i0:i1])
        full = run(all_ids) if rank == 0 else None

        for slot in (0, 1):
            rec = torch.from_numpy(local.records[slot].view(np.uint32).astype(np.int64).astype(np.int32).reshape(-1).copy())
            lst = torch.from_numpy(local.visibleList[slot].astype(np.int64).astype(np.int32).copy())
            # pad like the fixed-capacity device buffers
            rec_buf = torch.zeros(3 * 4096, dtype=torch.int32); rec_buf[:rec.numel()] = rec
            lst_buf = torch.zeros(32 * 4096, dtype=torch.int32); lst_buf[:lst.numel()] = lst
            counts = gather.exchange_counts(dist, torch, torch.tensor([len(local.records[slot]), len(local.visibleList[slot])], dtype=torch.int32), world)
            G, V = counts[:, 0], counts[:, 1]
            out_list = torch.zeros(32 * 4096 * world, dtype=torch.int32)
            out_rec = torch.zeros(3 * 4096 * world, dtype=torch.int32)

            def rebase(add, lst_buf=lst_buf, n=int(V[rank])):
                lst_buf[:n] += add
            g_tot, v_tot = gather.gather_slot(dist, rank, world, out_list, out_rec, lst_buf, rec_buf, G, V, rebase, uneven_ok=False)
            if rank == 0:
                got_rec = out_rec[:3 * g_tot].numpy().view(np.uint32).reshape(-1, 3)
                got_lst = out_list[:v_tot].numpy().view(np.uint32)
                assert np.array_equal(got_rec, full.records[slot].view(np.uint32).reshape(-1, 3)), f"slot {slot}: records"
                assert np.array_equal(got_lst, full.visibleList[slot]), f"slot {slot}: visible list"
                assert g_tot == len(full.records[slot]) and v_tot == int(full.drawArgs[slot][0])
        if rank == 0:
            assert full.dispatchArgs[0][0] > 0 and full.drawArgs[0][0] > 0
            open(os.path.join(out_dir, f"ok_{world}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_cull_plus_gather_equals_single_process(tmp_path, world):
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / f"ok_{world}").exists()


def test_shard_ranges_partition_the_list():
    from toyrenderer_amd import gather
    for n in (0, 1, 7, 257, 781250):
        for world in (1, 2, 3, 8):
            r = [gather.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
