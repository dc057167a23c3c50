"""The multi-GPU exchange kernels ("visibility_CS_PackShard" / "visibility_CS_UnpackShards") on one
MI355X: R ranks are simulated by culling R instance shards one after the other, packing each into its
shard slot and laying the slots out the way the all-gather would; the unpacked whole-scene records and
visible lists must equal a single full-scene frame of the oracle, and every word of the slots and
outputs must equal the numpy statement of the protocol (tests/exchange_ref.py).  Bar: bit-exact."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from exchange_ref import expand_masks_np, pack_shard_np, unpack_shards_np  # noqa: E402

from toyrenderer_amd import gather, synth  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from toyrenderer_amd import rhi
    d = rhi.Device(0)
    yield d
    d.destroy()


def _pack_state(dev, slot_groups):
    """u1 of visibility_CS_PackShard: two halves of 8-byte status words, zero before the first dispatch; every launch
    uses one half and zeroes the other for the next launch on the same buffer."""
    n = 4 * (slot_groups // 1024 + 5)
    st = dev.create_buffer(4 * n, "packstate")
    st.upload(np.zeros(n, np.uint32))
    return st, n


def _pack(dev, local, slot_groups, caps=None, slot_runs=None, repeat=1, state=None):
    """local: pass slot -> (records u32[G,3], masks u32[G], V).  Runs the HIP pack kernel, returns the slot words."""
    from toyrenderer_amd import rhi
    R = slot_groups if slot_runs is None else slot_runs
    W = gather.slot_words(slot_groups, R)
    bufs, binds = [], [rhi.PUSH(0)]
    slot = dev.create_buffer(4 * W, "slot")
    slot.upload(np.full(W, 0xDEADBEEF, np.uint32))
    bufs += [slot]
    if state is None:                                   # else: a state buffer shared by consecutive packs, as in the host path
        state, _ = _pack_state(dev, slot_groups)
        bufs += [state]
    binds += [rhi.UAV(0, slot), rhi.UAV(1, state)]
    for s, item in local.items():
        rec, masks, V = item[:3]
        G = len(rec)
        X = item[3] if len(item) > 3 else G                                             # the counter also counts dropped groups (Q2)
        cap = max(G, 1) if caps is None else caps[s]
        r = dev.buffer_from(np.resize(np.asarray(rec, np.uint32).reshape(-1), 3 * cap) if G else np.zeros(3 * cap, np.uint32), f"rec{s}")
        m = dev.buffer_from(np.resize(np.asarray(masks, np.uint32), cap) if G else np.zeros(cap, np.uint32), f"mask{s}")
        a = dev.buffer_from(np.array([X, 1, 1, G], np.uint32), f"args{s}")
        d = dev.buffer_from(np.array([V, 1, 1], np.uint32), f"draw{s}")
        bufs += [r, m, a, d]
        binds += [rhi.SRV(4 * s, r), rhi.SRV(4 * s + 1, m), rhi.SRV(4 * s + 2, a), rhi.SRV(4 * s + 3, d)]
    cl = dev.create_command_list()
    cl.open()
    cl.dispatch("visibility_CS_PackShard", binds, (1, 1, 1), push=np.array([slot_groups, R], np.uint32))
    cl.close()
    for _ in range(repeat):                             # the recorded list is re-executed frame after frame
        dev.execute(cl)
    dev.wait_idle()
    out = slot.download(np.uint32, W)
    cl.release()
    for b in bufs:
        b.release()
    return out


def _same_used_words(got, ref, S, R):
    """Header, the run entries and the masks the reference wrote (the rest of the HIP slot keeps its fill pattern)."""
    H = gather.HEADER_WORDS
    runs, groups = min(int(ref[13]), R), int(ref[0] + ref[2] + ref[4] + ref[6])
    return (np.array_equal(got[:H], ref[:H]) and np.array_equal(got[H:H + 4 * runs], ref[H:H + 4 * runs])
            and np.array_equal(got[H + 4 * R:H + 4 * R + groups], ref[H + 4 * R:H + 4 * R + groups])
            and np.all(got[H + 4 * runs:H + 4 * R] == 0xDEADBEEF) and np.all(got[H + 4 * R + groups:] == 0xDEADBEEF))


def _unpack(dev, recv, world, slot_groups, pass_slots, group_cap, list_cap, slot_runs=None, global_cap=None):
    from toyrenderer_amd import rhi
    R = slot_groups if slot_runs is None else slot_runs
    rb = dev.buffer_from(np.asarray(recv, np.uint32), "recv")
    bufs, binds, outs = [rb], [rhi.PUSH(0), rhi.SRV(0, rb)], {}
    for s in pass_slots:
        o = dict(records=dev.create_buffer(12 * max(group_cap, 1), f"allrec{s}"), masks=dev.create_buffer(4 * max(group_cap, 1), f"allmask{s}"),
                 list=dev.create_buffer(4 * max(list_cap, 1), f"alllist{s}"), args=dev.create_buffer(32, f"allargs{s}"))
        outs[s] = o
        bufs += list(o.values())
        binds += [rhi.UAV(4 * s, o["records"]), rhi.UAV(4 * s + 1, o["masks"]), rhi.UAV(4 * s + 2, o["list"]), rhi.UAV(4 * s + 3, o["args"])]
    cl = dev.create_command_list()
    cl.open()
    push = [world, slot_groups, R] + ([int(global_cap)] if global_cap else [])          # the 4th word is optional (Q2 made global)
    cl.dispatch("visibility_CS_UnpackShards", binds, (1, 1, 1), push=np.array(push, np.uint32))
    cl.close()
    res = {}
    for _ in range(2):                                  # a recorded list is re-executed every other frame
        dev.execute(cl)
    dev.wait_idle()
    for s in pass_slots:
        a = outs[s]["args"].download(np.uint32, 8)
        G, V = min(int(a[0]), int(a[3])), int(a[4])
        res[s] = dict(args=a, G=G, V=V, status=int(a[7]), records=outs[s]["records"].download(np.uint32, 3 * min(G, group_cap)).reshape(-1, 3),
                      masks=outs[s]["masks"].download(np.uint32, min(G, group_cap)), list=outs[s]["list"].download(np.uint32, min(V, list_cap)))
    cl.release()
    for b in bufs:
        b.release()
    return res


def _run_records(rng, g):
    """g records the way the instance pass emits them: per instance {id, lod, 0}, {id, lod, 32}, ... (1..40 groups; now
    and then the same instance twice in a row, which must stay two runs)."""
    rec = np.zeros((g, 3), np.uint32)
    i = 0
    inst = 0
    while i < g:
        n = min(int(rng.integers(1, 41)) if rng.random() < 0.3 else int(rng.integers(1, 6)), g - i)
        inst = inst if rng.random() < 0.1 else int(rng.integers(0, 2 ** 32))
        rec[i:i + n, 0], rec[i:i + n, 1], rec[i:i + n, 2] = inst, int(rng.integers(0, 8)), 32 * np.arange(n)
        i += n
    return rec


def _random_local(rng, pass_slots, max_groups, runs=False):
    loc = {}
    for s in pass_slots:
        g = int(rng.integers(0, max_groups + 1))
        rec = _run_records(rng, g) if runs else rng.integers(0, 2 ** 32, (g, 3), dtype=np.uint64).astype(np.uint32)
        m = rng.integers(0, 2 ** 32, g, dtype=np.uint64).astype(np.uint32)
        m[rng.random(g) < 0.3] = 0
        loc[s] = (rec, m, int(np.unpackbits(m.view(np.uint8)).sum()))
    return loc


@pytest.mark.parametrize("world,pass_slots,max_groups,runs", [(1, (0, 1), 300, False), (3, (0, 1), 2000, True), (8, (0, 1, 2, 3), 700, True), (64, (0,), 40, False),
                                                              (2, (1, 3), 100000, False), (2, (0, 1), 150000, True)])
def test_pack_unpack_kernels_equal_numpy_protocol(dev, world, pass_slots, max_groups, runs):
    """runs False: random record words (every record its own run: the encoding must stay lossless); True: records the way
    the instance pass emits them, with a run capacity below the group capacity."""
    from exchange_ref import runs_of_records_np
    rng = np.random.default_rng(world * 1000 + len(pass_slots))
    S = max_groups * len(pass_slots) + 3
    locals_ = [_random_local(rng, pass_slots, max_groups, runs) for _ in range(world)]
    if world > 1:
        locals_[1] = {s: (np.zeros((0, 3), np.uint32), np.zeros(0, np.uint32), 0) for s in pass_slots}    # an empty rank
    R = max(sum(len(runs_of_records_np(v[0])) for v in loc.values()) for loc in locals_) + 1 if runs else S
    assert not runs or R < 0.6 * S
    slots = []
    H = gather.HEADER_WORDS
    state, _ = _pack_state(dev, S)                       # ONE state buffer for all packs, each executed an odd number of times
    for k, loc in enumerate(locals_):
        got = _pack(dev, loc, S, slot_runs=R, repeat=1 + 2 * (k % 2), state=state)
        ref = pack_shard_np(loc, S, R)
        used = sum(len(v[0]) for v in loc.values())
        used_runs = int(ref[13])
        assert np.array_equal(got[:H], ref[:H])
        assert np.array_equal(got[H:H + 4 * used_runs], ref[H:H + 4 * used_runs]) and np.array_equal(got[H + 4 * R:H + 4 * R + used], ref[H + 4 * R:H + 4 * R + used])
        assert np.all(got[H + 4 * used_runs:H + 4 * R] == 0xDEADBEEF), "pack wrote past the packed runs"
        slots.append(got)
    state.release()
    recv = np.concatenate(slots)
    cap = world * S
    got = _unpack(dev, recv, world, S, pass_slots, cap, 32 * cap, slot_runs=R)
    ref = unpack_shards_np(recv, world, S, pass_slots, cap, R)
    for s in pass_slots:
        assert got[s]["status"] == 0 and ref[s]["status"] == 0
        assert got[s]["G"] == ref[s]["G"] and got[s]["V"] == ref[s]["V"]
        assert list(got[s]["args"][:7]) == [ref[s]["G"], 1, 1, ref[s]["G"], ref[s]["V"], 1, 1]
        assert np.array_equal(ref[s]["records"], np.concatenate([np.asarray(loc[s][0], np.uint32).reshape(-1, 3) for loc in locals_])), "the run encoding is lossless"
        assert np.array_equal(got[s]["records"], ref[s]["records"])
        assert np.array_equal(got[s]["masks"], ref[s]["masks"])
        assert np.array_equal(got[s]["list"], ref[s]["list"])


@pytest.mark.parametrize("world", [2, 5, 16])
def test_unpack_makes_the_group_capacity_drop_global(dev, world):
    """Q2 made global (gather.py): ranks whose dispatch counters counted more groups than they sent, capacities that put
    the first dropped instance on every rank in turn, on a rank's own dropped instance, nowhere: HIP unpack == numpy
    protocol word for word ({sum of the counters, 1, 1, validRecords}, records, masks, list)."""
    rng = np.random.default_rng(70 + world)
    pass_slots, G = (0, 1), 600
    S = 2 * G + 3
    locals_ = []
    for p in range(world):
        loc = {}
        for s in pass_slots:
            g = int(rng.integers(0, G + 1)) if p != 1 else 0
            rec = _run_records(rng, g)
            m = rng.integers(0, 2 ** 32, g, dtype=np.uint64).astype(np.uint32)
            counted = g + (int(rng.integers(1, 30)) if rng.random() < 0.4 else 0)                   # the rank dropped groups itself
            loc[s] = (rec, m, 0, counted)
        locals_.append(loc)
    r0 = locals_[0][0]
    locals_[0][0] = (r0[0], r0[1], 0, len(r0[0]) + 7)                                               # at least one rank dropped groups
    state, _ = _pack_state(dev, S)
    slots = []
    for loc in locals_:
        got = _pack(dev, loc, S, state=state)
        ref = pack_shard_np(loc, S)
        assert np.array_equal(got[:gather.HEADER_WORDS], ref[:gather.HEADER_WORDS])
        slots.append(got)
    state.release()
    recv = np.concatenate(slots)
    total = [sum(loc[s][3] for loc in locals_) for s in pass_slots]
    caps = sorted(set([1, 2, 3, min(total) // 2, min(total) - 1, min(total), max(total), max(total) + 1, max(total) + 100]
                      + [int(x) for x in rng.integers(1, max(total) + 2, 12)]))
    for cap in caps:
        if cap < 1:
            continue
        got = _unpack(dev, recv, world, S, pass_slots, world * S, 32 * world * S, global_cap=cap)
        ref = unpack_shards_np(recv, world, S, pass_slots, world * S, global_cap=cap)
        for i, s in enumerate(pass_slots):
            assert got[s]["status"] == 0 and ref[s]["status"] == 0
            assert list(got[s]["args"][:7]) == [ref[s]["X"], 1, 1, ref[s]["G"], ref[s]["V"], 1, 1], (cap, s, got[s]["args"], ref[s]["X"], ref[s]["G"])
            assert ref[s]["X"] == total[i] and ref[s]["G"] <= min(cap, total[i])
            assert np.array_equal(got[s]["records"], ref[s]["records"]) and np.array_equal(got[s]["masks"], ref[s]["masks"])
            assert np.array_equal(got[s]["list"], ref[s]["list"])
    # without the capacity the drop is flagged (status bit 8), as before
    got = _unpack(dev, recv, world, S, pass_slots, world * S, 32 * world * S)
    assert all(got[s]["status"] & 8 for s in pass_slots)


def test_overflow_is_flagged_not_silent(dev):
    S = 40
    big = {0: (np.arange(90, dtype=np.uint32).reshape(30, 3), np.ones(30, np.uint32), 30), 1: (np.arange(60, dtype=np.uint32).reshape(20, 3), np.full(20, 3, np.uint32), 40)}
    got = _pack(dev, big, S)
    assert _same_used_words(got, pack_shard_np(big, S), S, S) and got[8] == 1 and got[0] == 30 and got[2] == 10
    res = _unpack(dev, got, 1, S, (0, 1), S, 32 * S)
    assert res[0]["status"] & gather.STATUS_SLOT_OVERFLOW and res[1]["status"] & gather.STATUS_SLOT_OVERFLOW
    assert res[0]["G"] == 30 and res[1]["G"] == 10
    # whole-scene buffers too small: flagged, nothing written past them
    ok = {0: (np.arange(90, dtype=np.uint32).reshape(30, 3), np.ones(30, np.uint32), 30)}
    sl = _pack(dev, ok, S)
    res = _unpack(dev, np.concatenate([sl, sl]), 2, S, (0,), 45, 32 * 45)
    assert res[0]["status"] & gather.STATUS_CAPACITY and res[0]["G"] == 45
    assert np.array_equal(res[0]["records"][:30], ok[0][0]) and np.array_equal(res[0]["records"][30:], ok[0][0][:15])
    # a rank that dropped groups at its own capacity (Q2): the sharded result is not the single-GPU one -> flagged
    q2 = {0: (np.arange(90, dtype=np.uint32).reshape(30, 3), np.ones(30, np.uint32), 30, 37)}
    sl = _pack(dev, q2, S)
    assert _same_used_words(sl, pack_shard_np(q2, S), S, S) and sl[9] == 1
    assert _unpack(dev, sl, 1, S, (0,), S, 32 * S)[0]["status"] == gather.STATUS_GROUPS_DROPPED
    # more runs than the slot's run capacity: flagged, nothing written past the run array
    many = {0: (np.arange(90, dtype=np.uint32).reshape(30, 3), np.ones(30, np.uint32), 30)}
    sl = _pack(dev, many, S, slot_runs=12)
    ref = pack_shard_np(many, S, 12)
    assert sl[8] == 1 and _same_used_words(sl, ref, S, 12) and sl[10] == 30
    assert _unpack(dev, sl, 1, S, (0,), S, 32 * S, slot_runs=12)[0]["status"] & (gather.STATUS_SLOT_OVERFLOW | gather.STATUS_BAD_HEADER)


@pytest.mark.parametrize("world,instances,meshlets", [(2, 3001, 70), (5, 3001, 70), (3, 1500, 260)])
def test_sharded_frames_gather_to_the_single_gpu_result(dev, oracle, world, instances, meshlets):
    """Cull `world` contiguous instance shards (FrameDriver, one after the other on this GPU), pack, lay the
    slots out rank-major, unpack: records and visible lists of both phases == the oracle's full-scene frame."""
    from toyrenderer_amd.frame import FrameDriver, GpuScene
    # (the third case: runs of up to 11 groups per instance through the run encoding)
    spec = synth.SceneSpec(num_meshes=40, num_instances=instances, meshlets_lod0=meshlets, jitter_meshlets=True, max_lods=4, seed=7)
    scene = synth.make_scene(spec)
    view = synth.make_view(eye=(0.3, 0.1, 0.4), yaw=0.02, prev_eye=(0, 0, 0), prev_yaw=0.0, render=(1280, 720))
    d_prev = synth.gen_depth(view, 60, seed=5, scale=3.0)
    d_cur = synth.gen_depth(view, 50, seed=6, scale=3.0)
    cap = 1 << 15
    hzb = oracle.HzbTexture(*view.hzb_dims)
    hzb.build_from_depth(d_prev)
    hzb0 = (hzb.texels.copy(), hzb.offsets)
    full = oracle.frame(scene.as_oracle(), view.as_dict(), hzb, d_cur, cullingFlags=7, maxGroups=cap, record_capacity=cap)
    assert full.dispatchArgs[0][0] > 100 and full.dispatchArgs[1][0] > 0, "the case must exercise both phases"
    md, inst = scene.meshData, scene.instances
    caps = [gather.shard_group_capacity(md["m_MeshLODDatas"]["m_NumMeshlets"], inst["m_MeshDataIdx"][scene.opaqueIds[slice(*gather.shard_range(len(scene.opaqueIds), p, world))]]) for p in range(world)]
    S = max(caps)
    R = max(gather.shard_run_capacity(b - a) for a, b in (gather.shard_range(len(scene.opaqueIds), p, world) for p in range(world)))
    assert R < S / 2
    from toyrenderer_amd import rhi
    assert full.lateCount[0] > 64, "the late list must be long enough for the dispatch-size rule (Q1) to truncate it"
    L = rhi.load()
    late_counts = np.zeros(world, np.uint32)
    counts_buf = dev.create_buffer(4 * world, "GatheredLateCounts")
    slots = []
    for phase in ("count", "cull"):                 # pass 1 learns every shard's late count (what the in-frame all-gather delivers)
        counts_buf.upload(late_counts)
        for p in range(world):
            i0, i1 = gather.shard_range(len(scene.opaqueIds), p, world)
            gs = GpuScene(dev, scene.instances, scene.meshData, scene.meshlets, scene.opaqueIds[i0:i1], np.zeros(0, np.uint32))

            def hook(stream, late_count_ptr, info_ptr, bucket, phase, p=p):
                assert bucket == 0 and phase in (0, 1)
                if phase == 1:                       # (phase 0 is where a real run starts its all-gather)
                    assert L.trhip_launch_shard_late_info(stream, counts_buf.ptr, world, p, info_ptr) == 0
            drv = FrameDriver(dev, gs, view, record_capacity=cap, culling_flags=7, shard_late=hook if phase == "cull" else None)
            drv.hzb.upload_chain(*hzb0)
            drv.depth.upload_mip(0, d_cur)
            drv.record()
            drv.run()
            if phase == "count":
                late_counts[p] = drv.results()["lateCount"]
            else:
                slot = dev.create_buffer(4 * gather.slot_words(S, R), "slot")
                state, _ = _pack_state(dev, S)
                binds = [rhi.PUSH(0), rhi.UAV(0, slot), rhi.UAV(1, state)]
                for s in (0, 1):
                    binds += [rhi.SRV(4 * s, drv.records[s]), rhi.SRV(4 * s + 1, drv.visMask[s]), rhi.SRV(4 * s + 2, drv.dispatchArgs[s]), rhi.SRV(4 * s + 3, drv.drawArgs[s])]
                cl = dev.create_command_list()
                cl.open()
                cl.dispatch("visibility_CS_PackShard", binds, (1, 1, 1), push=np.array([S, R], np.uint32))
                cl.close()
                dev.execute(cl)
                dev.wait_idle()
                slots.append(slot.download(np.uint32, gather.slot_words(S, R)))
                assert slots[-1][8] == 0 and slots[-1][9] == 0 and slots[-1][0] + slots[-1][2] <= caps[p]
                assert slots[-1][11] <= i1 - i0, "one run per submitted instance"
                cl.release(); slot.release(); state.release()
            drv.release(); gs.release()
    counts_buf.release()
    assert int(late_counts.sum()) == int(full.lateCount[0]), "shards' late lists must partition the full late list"
    gcap = sum(caps)
    res = _unpack(dev, np.concatenate(slots), world, S, (0, 1), gcap, 32 * gcap, slot_runs=R)
    for s in (0, 1):
        assert res[s]["status"] == 0
        assert np.array_equal(res[s]["records"], full.records[s].view(np.uint32).reshape(-1, 3)), f"slot {s}: records"
        assert np.array_equal(res[s]["masks"], full.visMask[s]), f"slot {s}: masks"
        assert np.array_equal(res[s]["list"], full.visibleList[s]), f"slot {s}: visible list"
        assert res[s]["V"] == int(full.drawArgs[s][0])
        assert np.array_equal(res[s]["list"], expand_masks_np(full.visMask[s]))
