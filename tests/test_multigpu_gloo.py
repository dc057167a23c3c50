"""The N > 1 path on CPU: world_size-2 (and 3) gloo processes shard the instance list by contiguous
ranges, each cull their shard (with the oracle standing in for the GPU), and the shard-slot exchange of
toyrenderer_amd/gather.py (one equal-size all-gather of compact records + lane masks, rank-major
concatenation, list rebuild) must reproduce the single-process records and visible lists bit for bit
(rank-major concatenation == single-GPU canonical order).  The host logic under test is
gather.ShardExchange; the data movers are the numpy statement of the protocol (tests/exchange_ref.py),
which the -m gpu tests hold the HIP kernels to."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle
        from toyrenderer_amd import gather, synth
        from exchange_ref import NumpyShardExchange
        spec = synth.SceneSpec(num_meshes=30, num_instances=3001, meshlets_lod0=50, jitter_meshlets=True, max_lods=4, seed=99)
        scene = synth.make_scene(spec)
        view = synth.make_view(eye=(0.3, 0.1, 0.4), yaw=0.02, prev_eye=(0, 0, 0), prev_yaw=0.0, render=(640, 360))
        d_prev = synth.gen_depth(view, 50, seed=5, scale=3.0)
        d_cur = synth.gen_depth(view, 40, seed=6, scale=3.0)

        def run(ids, shard_late=None):
            hzb = pyoracle.HzbTexture(*view.hzb_dims)
            hzb.build_from_depth(d_prev)
            sc = dict(scene.as_oracle()); sc["opaqueIds"] = ids; sc["alphaMaskIds"] = np.zeros(0, np.uint32)
            return pyoracle.frame(sc, view.as_dict(), hzb, d_cur, cullingFlags=7, maxGroups=1 << 20, record_capacity=1 << 15, shard_late=shard_late)

        all_ids = np.arange(spec.num_instances, dtype=np.uint32)
        i0, i1 = gather.shard_range(spec.num_instances, rank, world)
        # the in-frame exchange of the late-list lengths (gather.py docstring): early phase -> all-gather of
        # the late counts -> late phase with {entries of the lower ranks, of all ranks}
        mine = torch.tensor([int(run(all_ids[i0:i1]).lateCount[0])], dtype=torch.int32)
        counts = torch.zeros(world, dtype=torch.int32)
        dist.all_gather_into_tensor(counts, mine)
        local = run(all_ids[i0:i1], shard_late=(gather.shard_late_info(counts.tolist(), rank), (0, 0)))
        full = run(all_ids) if rank == 0 else None
        if rank == 0:
            assert int(counts.sum()) == int(full.lateCount[0]) > 64, "late list too short for the dispatch-size rule (Q1) to matter"
            assert len(full.records[1]) > 0

        md = scene.as_oracle()["meshData"]
        inst = scene.as_oracle()["instances"]
        cap_local = gather.shard_group_capacity(md["m_MeshLODDatas"]["m_NumMeshlets"], inst["m_MeshDataIdx"][all_ids[i0:i1]])
        slot_groups = gather.agree_slot_groups(dist, torch, cap_local)
        assert slot_groups >= cap_local
        ex = NumpyShardExchange(dist, torch, world, rank, slot_groups, pass_slots=(0, 1))
        for frame in range(3):                      # three frames: both buffer indices, and reuse of one
            ex.set_local({s: (local.records[s].view(np.uint32).reshape(-1, 3), local.visMask[s], int(local.drawArgs[s][0])) for s in (0, 1)})
            ex.run()
            for slot in (0, 1):
                got_rec, got_lst = ex.results(slot)
                if rank == 0:
                    assert np.array_equal(got_rec, full.records[slot].view(np.uint32).reshape(-1, 3)), f"slot {slot}: records"
                    assert np.array_equal(got_lst, full.visibleList[slot]), f"slot {slot}: visible list"
                    assert len(got_lst) == int(full.drawArgs[slot][0])
        # every rank holds the same whole-scene result
        chk = torch.tensor([int(np.bitwise_xor.reduce(ex.results(0)[1])) & 0x7FFFFFFF, len(ex.results(1)[0])], dtype=torch.int64)
        parts = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(parts, chk)
        assert all(torch.equal(p_, parts[0]) for p_ in parts)
        if rank == 0:
            assert full.dispatchArgs[0][0] > 0 and full.drawArgs[0][0] > 0
            open(os.path.join(out_dir, f"ok_{world}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_cull_plus_gather_equals_single_process(tmp_path, world):
    from bench import free_rendezvous_port
    port = free_rendezvous_port()
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


def test_slot_protocol_numpy_roundtrip_and_overflow():
    """pack -> (concatenated slots) -> unpack on one process: ordering, absent pass slots, overflow flags."""
    from exchange_ref import expand_masks_np, pack_shard_np, unpack_shards_np
    from toyrenderer_amd.gather import slot_words
    rng = np.random.default_rng(3)
    world, S = 3, 40
    shards, slots = [], []
    for p in range(world):
        loc = {}
        for s in (0, 1, 3):
            g = int(rng.integers(0, 12))
            rec = rng.integers(0, 2 ** 32, (g, 3), dtype=np.uint64).astype(np.uint32)
            m = rng.integers(0, 2 ** 32, g, dtype=np.uint64).astype(np.uint32)
            loc[s] = (rec, m, int(sum(bin(int(x)).count("1") for x in m)))
        shards.append(loc)
        slots.append(pack_shard_np(loc, S))
        assert len(slots[-1]) == slot_words(S) and slots[-1][8] == 0
    out = unpack_shards_np(np.concatenate(slots), world, S, (0, 1, 3), world * S)
    for s in (0, 1, 3):
        rec = np.concatenate([sh[s][0] for sh in shards])
        m = np.concatenate([sh[s][1] for sh in shards])
        assert np.array_equal(out[s]["records"], rec) and np.array_equal(out[s]["masks"], m)
        assert np.array_equal(out[s]["list"], expand_masks_np(m)) and out[s]["status"] == 0
        assert out[s]["V"] == sum(sh[s][2] for sh in shards)
    lst = expand_masks_np(np.array([0b101, 0, 1 << 31], np.uint32))
    assert lst.tolist() == [0, 2, (2 << 5) | 31]
    # a shard that emits more groups than the slot holds is flagged, never silently truncated
    big = {0: (np.zeros((30, 3), np.uint32), np.ones(30, np.uint32), 30), 1: (np.zeros((20, 3), np.uint32), np.ones(20, np.uint32), 20)}
    sl = pack_shard_np(big, S)
    assert sl[8] == 1 and sl[0] == 30 and sl[2] == 10
    assert unpack_shards_np(sl, 1, S, (0, 1), S)[0]["status"] & 1
    # whole-scene capacity too small
    assert unpack_shards_np(np.concatenate(slots), world, S, (0, 1, 3), 5)[0]["status"] & 2


def test_run_encoding_is_lossless_and_one_entry_per_instance():
    """The records travel as runs (gather.py): {instance, lod, first offset, first record} per maximal run of records that
    continue each other.  Round trip on records the way the instance pass emits them, on adversarial ones, and through a
    slot whose run capacity is far below its group capacity."""
    from exchange_ref import pack_shard_np, records_of_runs_np, runs_of_records_np, unpack_shards_np
    from toyrenderer_amd.gather import slot_words
    rng = np.random.default_rng(11)
    groups = rng.integers(1, 9, 500)
    rec = np.concatenate([np.stack([np.full(g, i, np.uint32), np.full(g, i % 5, np.uint32), 32 * np.arange(g, dtype=np.uint32)], 1) for i, g in enumerate(groups)])
    runs = runs_of_records_np(rec)
    assert len(runs) == 500 and np.array_equal(runs[:, 3], np.concatenate([[0], np.cumsum(groups)[:-1]]))
    assert np.array_equal(records_of_runs_np(runs, len(rec)), rec)
    # the same instance twice in a row stays two runs; offsets that do not start at 0 or wrap around 2^32 survive
    odd = np.array([[7, 1, 0], [7, 1, 32], [7, 1, 0], [7, 2, 32], [9, 2, 0xFFFFFFF0], [9, 2, 0x10], [9, 2, 0x30]], np.uint32)
    r = runs_of_records_np(odd)
    assert r[:, 3].tolist() == [0, 2, 3, 4] and np.array_equal(records_of_runs_np(r, len(odd)), odd)
    S, R = len(rec) + 10, 520
    m = rng.integers(0, 2 ** 32, len(rec), dtype=np.uint64).astype(np.uint32)
    half = int(np.cumsum(groups)[249])
    loc = {0: (rec[:half], m[:half], 0), 1: (rec[half:], m[half:], 0)}
    sl = pack_shard_np(loc, S, R)
    assert len(sl) == slot_words(S, R) < slot_words(S) / 2 and sl[8] == 0 and sl[10] == 250 and sl[11] == sl[13] == 500
    out = unpack_shards_np(np.concatenate([sl, sl]), 2, S, (0, 1), 2 * S, R)
    assert np.array_equal(out[0]["records"], np.concatenate([rec[:half]] * 2)) and np.array_equal(out[1]["records"], np.concatenate([rec[half:]] * 2))
    assert out[0]["status"] == 0
    assert pack_shard_np(loc, S, 499)[8] == 1 and unpack_shards_np(pack_shard_np(loc, S, 499), 1, S, (0, 1), S, 499)[0]["status"] != 0


def test_global_q2_cut_numpy():
    """The unpack's Q2 rule (gather.py): three ranks of emitted-style records, capacities chosen so that the first dropped
    instance lies on rank 0, 1, 2, is a rank's own dropped instance, or does not exist -- against a direct statement of the
    single-GPU rule on the concatenated instance list."""
    from exchange_ref import pack_shard_np, unpack_shards_np
    rng = np.random.default_rng(5)
    world, S = 3, 400
    groups = [rng.integers(1, 7, int(rng.integers(20, 40))) for _ in range(world)]          # groups per submitted instance, per rank
    all_groups = np.concatenate(groups)
    total = int(all_groups.sum())

    def single_gpu(cap):                                                                    # gpuculling.hlsl:64-74 in canonical order
        off = 0
        for g in all_groups:
            if off + g >= cap:
                return off                                                                  # validRecords = offset of the first dropped instance
            off += g
        return off

    def rank_local(p, cap):                                                                 # what rank p's own pass produces at the same capacity
        rec, off, valid = [], 0, None
        for i, g in enumerate(groups[p]):
            if off + g >= cap and valid is None:
                valid = off
            rec += [[1000 * p + i, i % 3, 32 * j] for j in range(g)]
            off += g
        rec = np.array(rec, np.uint32)
        valid = off if valid is None else valid
        return rec[:valid], off                                                             # valid records, counter

    for cap in [1, 2, int(groups[0].sum()) - 1, int(groups[0].sum()), int(groups[0].sum()) + 1, int(groups[0].sum() + groups[1].sum()) + 3,
                total - 1, total, total + 1, total + 50] + [int(x) for x in rng.integers(1, total + 5, 40)]:
        slots, cat = [], []
        for p in range(world):
            rec, counted = rank_local(p, cap)
            m = (np.arange(len(rec), dtype=np.uint32) * np.uint32(2654435761)) ^ np.uint32(p)
            slots.append(pack_shard_np({0: (rec, m, 0, counted)}, S))
            cat.append((rec, m))
        out = unpack_shards_np(np.concatenate(slots), world, S, (0,), world * S, global_cap=cap)[0]
        want = single_gpu(cap)
        assert out["status"] == 0 and out["G"] == want and out["X"] == total, (cap, out["G"], want)
        assert np.array_equal(out["records"], np.concatenate([r for r, _ in cat])[:want])
        assert np.array_equal(out["masks"], np.concatenate([m for _, m in cat])[:want])
        # without the global capacity a drop is flagged, never silently accepted
        flagged = unpack_shards_np(np.concatenate(slots), world, S, (0,), world * S)[0]["status"] & 8
        assert bool(flagged) == any(rank_local(p, cap)[1] != len(rank_local(p, cap)[0]) for p in range(world))


def _worker_q2(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle
        from toyrenderer_amd import gather, synth
        from exchange_ref import NumpyShardExchange
        spec = synth.SceneSpec(num_meshes=30, num_instances=2000, meshlets_lod0=50, jitter_meshlets=True, max_lods=4, seed=321)
        scene = synth.make_scene(spec)
        view = synth.make_view(eye=(0.3, 0.1, 0.4), yaw=0.02, prev_eye=(0, 0, 0), prev_yaw=0.0, render=(640, 360))
        d_prev = synth.gen_depth(view, 50, seed=5, scale=3.0)
        d_cur = synth.gen_depth(view, 40, seed=6, scale=3.0)
        all_ids = np.arange(spec.num_instances, dtype=np.uint32)
        i0, i1 = gather.shard_range(spec.num_instances, rank, world)

        def run(ids, cap, shard_late=None):
            hzb = pyoracle.HzbTexture(*view.hzb_dims)
            hzb.build_from_depth(d_prev)
            sc = dict(scene.as_oracle()); sc["opaqueIds"] = ids; sc["alphaMaskIds"] = np.zeros(0, np.uint32)
            return pyoracle.frame(sc, view.as_dict(), hzb, d_cur, cullingFlags=7, maxGroups=cap, record_capacity=1 << 15, shard_late=shard_late)

        uncapped = run(all_ids, 1 << 20)
        GE, GL = int(uncapped.dispatchArgs[0][0]), int(uncapped.dispatchArgs[1][0])
        assert GE > 600 and GL > 40
        slot_groups = gather.agree_slot_groups(dist, torch, 1 << 14)
        # capacities that cut the early pass on rank 0, in the middle, on the last rank; one that only bites the early pass
        for cap in (GE // (2 * world), GE // 2, GE - 5, GL + 1, GE + 10):
            mine = torch.tensor([int(run(all_ids[i0:i1], cap).lateCount[0])], dtype=torch.int32)
            counts = torch.zeros(world, dtype=torch.int32)
            dist.all_gather_into_tensor(counts, mine)
            local = run(all_ids[i0:i1], cap, shard_late=(gather.shard_late_info(counts.tolist(), rank), (0, 0)))
            full = run(all_ids, cap)
            ex = NumpyShardExchange(dist, torch, world, rank, slot_groups, pass_slots=(0, 1), global_group_cap=cap)
            ex.set_local({s: (local.records[s].view(np.uint32).reshape(-1, 3), local.visMask[s], int(local.drawArgs[s][0]), int(local.dispatchArgs[s][0]))
                          for s in (0, 1)})
            ex.run()
            dropped = False
            for slot in (0, 1):
                got_rec, got_lst = ex.results(slot)
                assert np.array_equal(got_rec, full.records[slot].view(np.uint32).reshape(-1, 3)), f"cap {cap} slot {slot}: records"
                assert np.array_equal(got_lst, full.visibleList[slot]), f"cap {cap} slot {slot}: visible list"
                args = ex.out[slot]["args"].numpy().view(np.uint32)
                assert int(args[0]) == int(full.dispatchArgs[slot][0]) and int(args[3]) == int(full.validRecords[slot]), (cap, slot, args)
                dropped |= int(full.validRecords[slot]) < int(full.dispatchArgs[slot][0])
            assert dropped == (cap < GE + 10), cap
        if rank == 0:
            open(os.path.join(out_dir, f"q2_{world}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_group_capacity_drop_is_global_under_sharding(tmp_path, world):
    """Q2 (gpuculling.hlsl:64-74) under sharding: every rank runs with the single-GPU capacity; one (or several) of them
    hit it; the gathered whole-scene records, masks -> visible lists and {X, 1, 1, validRecords} equal the 1-rank oracle
    frame at that capacity, wherever the first dropped instance lies."""
    from bench import free_rendezvous_port
    port = free_rendezvous_port()
    mp.spawn(_worker_q2, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / f"q2_{world}").exists()
