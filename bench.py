#!/usr/bin/env python3
"""bench.py -- Gmeshlets/s culled on the synthetic 100 M-meshlet scene (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A step = one full 2-phase visibility frame (BasePassRenderer::RenderBasePass,
source/BasePassRenderers.cpp:544-588) over the scene shard(s) resident in HBM: early instance
cull -> early meshlet cull -> HZB build -> late instance cull -> late meshlet cull -> HZB build
[-> when N > 1: RCCL all-gather of every rank's records + visibility masks (shard slots) and rebuild
of the whole-scene visible lists on every rank, overlapped with the next frame's culling].  value = meshlets tested per frame
(all ranks) / max-over-ranks frame time.  Scaling is STRONG: the 100 M-meshlet scene is fixed and
its instances are sharded over the ranks (north_star).

One JSON line on rank 0 with the driver's contract plus "roofline" (dominant kernel: the meshlet
cull, HIP-event timed on its own stream via the back end's per-shader profile) and "cpu_baseline"
(the scalar C oracle timed on the host cores on a bounded sub-scene).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from toyrenderer_amd import interop as I  # noqa: E402
from toyrenderer_amd import synth  # noqa: E402
from toyrenderer_amd.gather import shard_range  # noqa: E402

MEASURED_STREAM_GBS = 6300.0      # tools/membw.hip on MI355X: 6.1-6.4 TB/s for every access pattern (profiles/r1/membw_calibration.txt)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
DOMINANT = "basepass_AS_Main LATE_CULL=0#cull"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_shard(spec: synth.SceneSpec, rank: int, world: int, renderer, threads: int = 8):
    """Load this rank's shard through the host library: the full instance + mesh tables (replicated,
    ~230 MB), the id list of the owned contiguous instance range, and ONLY the owned meshes' meshlets
    (co-sharded), streamed in chunk by chunk."""
    assert spec.unique, "the sharded bench uses the unique-meshlet configs (C3/C4)"
    t0 = time.time()
    md, total = synth.gen_mesh_table(spec)
    i0, i1 = shard_range(spec.num_instances, rank, world)
    cm = spec.chunk_meshes
    # mesh i belongs to instance i (unique): the owned meshes are [i0, i1); generate the covering chunks
    c0, c1 = (i0 // cm) * cm, min(((i1 + cm - 1) // cm) * cm, spec.num_meshes)
    base = int(md["m_MeshLODDatas"]["m_MeshletDataBufferIdx"][i0, 0])
    end = int(md["m_MeshLODDatas"]["m_MeshletDataBufferIdx"][i1 - 1, 0]) + int(md["m_MeshLODDatas"]["m_NumMeshlets"][i1 - 1].sum())
    n_local = end - base
    inst = synth.gen_instances(spec)
    # rebase the owned meshes' meshlet indices into the local shard
    md_local = md.copy()
    lods = md_local["m_MeshLODDatas"]
    owned = np.zeros(spec.num_meshes, bool); owned[i0:i1] = True
    lods["m_MeshletDataBufferIdx"][owned] -= np.uint32(base)
    lods["m_NumMeshlets"][~owned] = 0
    ids = np.arange(i0, i1, dtype=np.uint32)
    renderer.load_scene(inst, md_local, None, ids, np.zeros(0, np.uint32), num_meshlets=max(n_local, 1))

    def gen(b):
        e = min(b + cm, spec.num_meshes)
        return int(md["m_MeshLODDatas"]["m_MeshletDataBufferIdx"][b, 0]), synth.gen_meshlets_for_meshes(spec, md, b, e)
    with ThreadPoolExecutor(max_workers=threads) as ex:
        for off, chunk in ex.map(gen, range(c0, c1, cm)):
            lo, hi = max(off, base), min(off + len(chunk), end)
            if hi > lo:
                renderer.upload_meshlets(lo - base, chunk[lo - off:hi - off])
    log(f"[rank {rank}] shard: instances [{i0},{i1}) meshlets {n_local} ({n_local * 32 / 1e9:.2f} GB) in {time.time() - t0:.1f}s")
    return (i0, i1), n_local, total


def cpu_baseline(spec: synth.SceneSpec, view, depth, sample_instances: int, threads: int):
    """Time the scalar C oracle (oracle/tr_oracle.c) on the first `sample_instances` instances of the
    same scene, same camera, same depth.  Checker code used as the reported CPU baseline only."""
    from oracle import pyoracle
    n = min(sample_instances, spec.num_instances)
    if n < spec.num_instances:
        n = (n // spec.chunk_meshes) * spec.chunk_meshes or min(spec.chunk_meshes, spec.num_instances)
    sub = synth.SceneSpec(**{**spec.__dict__, "num_meshes": n, "num_instances": n})
    full_md, _ = synth.gen_mesh_table(spec)
    md = full_md[:n].copy()
    total = int(md["m_MeshLODDatas"]["m_NumMeshlets"].sum())
    ml = np.zeros(total, I.MeshletData)
    for b in range(0, n, spec.chunk_meshes):
        off = int(md["m_MeshLODDatas"]["m_MeshletDataBufferIdx"][b, 0])
        chunk = synth.gen_meshlets_for_meshes(spec, full_md, b, min(b + spec.chunk_meshes, n))
        ml[off:off + len(chunk)] = chunk
    inst = synth.gen_instances(spec, 0, n)
    scene = dict(instances=inst, meshData=md, meshlets=ml, opaqueIds=np.arange(n, dtype=np.uint32), alphaMaskIds=np.zeros(0, np.uint32))
    hzb = pyoracle.HzbTexture(*view.hzb_dims)
    hzb.build_from_depth(depth)
    times, tested = [], 0
    cap = n * ((spec.meshlets_lod0 + 31) // 32) + 1
    for _ in range(3):
        t = time.perf_counter()
        ref = pyoracle.frame(scene, view.as_dict(), hzb, depth, cullingFlags=7, maxGroups=1 << 27, threads=threads, record_capacity=cap)
        times.append(time.perf_counter() - t)
        tested = int(ref.meshletsTested.sum())
    dt = sorted(times)[1]
    out = dict(value=tested / dt / 1e9, unit="Gmeshlets/s", cores=threads, kind="port",
               sample=f"first {n} instances ({total} meshlets in scene, {tested} tested/frame) of the same scene, full 2-phase frame incl. 2 HZB builds, median of 3, {dt * 1e3:.1f} ms/frame")
    # the same oracle on ONE thread (SURVEY 8d asks for both), on the first 65 536 instances so that it stays a few seconds
    n1 = min(n, 65536)
    if n1 == n or n1 % spec.chunk_meshes == 0:
        sc1 = dict(scene, instances=inst[:n1], opaqueIds=np.arange(n1, dtype=np.uint32))
        hzb1 = pyoracle.HzbTexture(*view.hzb_dims)
        hzb1.build_from_depth(depth)
        t = time.perf_counter()
        ref1 = pyoracle.frame(sc1, view.as_dict(), hzb1, depth, cullingFlags=7, maxGroups=1 << 27, threads=1, record_capacity=n1 * ((spec.meshlets_lod0 + 31) // 32) + 1)
        dt1 = time.perf_counter() - t
        out["single_thread"] = dict(value=int(ref1.meshletsTested.sum()) / dt1 / 1e9, unit="Gmeshlets/s",
                                    sample=f"first {n1} instances, one frame, {dt1 * 1e3:.0f} ms")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--cpu-sample-instances", type=int, default=1 << 30,
                    help="instances of the scene the CPU baseline runs on (default: all of them; C3 = 3 frames of ~0.25 s on 64 threads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-animated-leg", action="store_true", help="skip the extra 'animated' key (the same K steps with per-frame transform updates; also skipped by --no-profile)")
    ap.add_argument("--prime-steps", type=int, default=256,
                    help="setup before the warm-up steps: frames run (untimed) so that the GPU is at its working clocks (0 = off)")
    ap.add_argument("--flags", type=int, default=7, help="culling flags (7 = frustum+occlusion+cone, the headline config)")
    ap.add_argument("--animate", action="store_true",
                    help="diagnostic (BASELINE configs[4]): a node hierarchy drives the instance transforms; every frame runs "
                         "updateinstanceconsts + the instance-cache rebuild before the cull (node data resident: uploaded once)")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="diagnostic: this single process plays rank 0 of M (its shard of the scene, a 1-rank RCCL group for the exchange); "
                         "the printed line is NOT a bench result")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process has made no GPU call yet (no torch import, no
        # libtrhip), so it may start the ranks itself -- as a CHILD process (never exec: under rocprofv3 the profiler's
        # preload has already initialised the GPU), relay the children's output and leave with their exit code.
        sys.exit(self_launch(args.gpus))

    # The driver parses ONE JSON line from stdout; RCCL prints its version banner there.  Everything
    # else goes to stderr: fd 1 is pointed at fd 2 and the JSON line is written to the saved fd.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"
    # TR_DIST_BACKEND=gloo: several ranks on ONE GPU (tests of the multi-rank logic; RCCL needs a GPU per rank)
    backend = os.environ.get("TR_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    force_gather = (bool(os.environ.get("TR_FORCE_GATHER")) or args.emulate_ranks > 1) and not os.environ.get("TR_NO_GATHER")     # exercise the RCCL path with a 1-rank group (tests)
    shard_world, shard_rank = (args.emulate_ranks, 0) if args.emulate_ranks > 1 else (world, rank)
    if world > 1 or force_gather:
        import torch.distributed as dist
        if force_gather and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500), RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from toyrenderer_amd import host, rhi
    from toyrenderer_amd.gather import NativeShardExchange

    # the HIP back end's stream = torch's current stream; (N > 1) the exchange adds its own second stream
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    spec = synth.config_spec(args.config)
    view = synth.make_view(eye=(0.0, 0.0, 0.0), prev_eye=(0.05, 0.0, 0.1), prev_yaw=0.002)
    depth = synth.gen_depth(view, 200)
    i0, i1 = shard_range(spec.num_instances, shard_rank, shard_world)
    groups_per_instance = (spec.meshlets_lod0 + 31) // 32
    record_cap = (i1 - i0) * groups_per_instance + 1

    # the drop-in path: C++ host mirror (Graphic / Scene / RenderGraph / BasePassRenderers) over the C ABI
    r = host.Renderer(render=(view.renderW, view.renderH), device_index=local_rank, stream=stream,
                      max_groups=record_cap, max_transient_bytes=8 << 30)
    dev = rhi.Device(handle=r.device())
    (i0, i1), n_local, n_total = build_shard(spec, shard_rank, shard_world, r, threads=min(8, host_threads()))
    if args.animate:
        nodes, prim_to_node = synth.animated_nodes(spec, 0)
        r.load_nodes(nodes, prim_to_node)              # uploaded with the first frame, then resident
        del nodes
        if shard_world > 1:
            r.set_instance_update_range(i0, i1 - i0)   # a rank rebuilds the transforms of its own shard, not of the replicated table
    r.set_culling(args.flags)
    r.set_gpu_timers(False)          # the per-renderer timer queries are instrumentation (2 timestamp packets each)
    r.upload_depth(depth)
    gather = None
    if dist is not None:
        # one slot size for all ranks: the largest shard, every instance submitted at LOD 0
        slot_runs = max(b - a for a, b in (shard_range(spec.num_instances, p, shard_world) for p in range(shard_world)))   # one run per submitted instance
        slot_groups = slot_runs * groups_per_instance
        loopback = args.emulate_ranks > 1 and bool(os.environ.get("TR_EMULATE_LOOPBACK"))   # diagnostic: the unpack sees M shards
        gather = NativeShardExchange(r, dist, shard_world if loopback else world, shard_rank if loopback else rank, slot_groups, pass_slots=(0, 1),
                                  group_capacity=spec.num_instances * groups_per_instance + (shard_world if loopback else 0) * groups_per_instance,
                                  overlap=not os.environ.get("TR_NO_OVERLAP"), stage_through_host=backend != "nccl", loopback=loopback, slot_runs=slot_runs)

    rec_hist = []
    cpu_t = [0.0, 0.0, 0.0, 0.0]                          # host time spent submitting: frame, exchange (diagnostics, stderr only)

    def step():
        t_a = time.perf_counter()
        r.set_camera(view)
        r.frame()
        t_b = time.perf_counter()
        rec_ms, sub_ms = r.renderer_times("<frame>")
        cpu_t[2] += rec_ms
        cpu_t[3] += sub_ms
        rec_hist.append(rec_ms)
        if gather:
            gather.run()
        cpu_t[0] += t_b - t_a
        cpu_t[1] += time.perf_counter() - t_b

    def sync():
        torch.cuda.synchronize()          # all streams of the device, including the exchange's
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Setup, before the W warm-up steps: bring the device to its working state.  A GPU that has only uploaded a scene sits
    # at idle clocks; with the driver's short runs (W = 5, K = 20: 16 ms in all) the timed steps then ran 4 % slower than
    # in a 200-step run of the same binary (profiles/r3/experiments.md).  256 frames (~0.16 s), untimed, synchronised.
    for i in range(args.prime_steps):          # a fixed count: every rank takes part in the same collectives
        step()
        if i % 16 == 15:
            sync()
    sync()
    for _ in range(max(args.warmup, 1)):      # frame 0 sees the cleared HZB; steady state afterwards
        step()
    sync()
    t0 = time.perf_counter()
    cpu_t[:] = [0.0] * 4
    for _ in range(args.steps):
        step()
    t_submit = time.perf_counter() - t0
    sync()
    dt = time.perf_counter() - t0
    log(f"[rank {rank}] record ms per frame, first 12 timed frames: " + " ".join(f"{x:.3f}" for x in rec_hist[-args.steps:][:12])
        + " ... last 4: " + " ".join(f"{x:.3f}" for x in rec_hist[-4:]))
    log(f"[rank {rank}] host submission per step: frame {cpu_t[0] / args.steps * 1e3:.3f} ms (record {cpu_t[2] / args.steps:.3f}, submit {cpu_t[3] / args.steps:.3f}), exchange {cpu_t[1] / args.steps * 1e3:.3f} ms; "
        f"all steps submitted after {t_submit * 1e3:.2f} ms of {dt * 1e3:.2f} ms")
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # work done per frame (device counters of the steady-state frame)
    res = r.results()
    tested = 0
    groups = 0
    visible = 0
    for s in (0, 1):
        if res[s] is None:
            continue
        recs = res[s]["records"]
        groups += len(recs)
        visible += int(res[s]["drawArgs"][0])
        tested += gs_num_meshlets(spec, recs)
    counts = np.array([tested, groups, visible], np.int64)
    if dist is not None:
        t = torch.tensor(counts, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        counts = t.cpu().numpy()
    tested_all, groups_all, visible_all = (int(x) for x in counts)
    gather_checked = None
    if gather is not None and world == 1 and not (args.emulate_ranks > 1 and os.environ.get("TR_EMULATE_LOOPBACK")):
        # 1-rank group: the gathered whole-scene lists must equal the local ones
        gather_checked = True
        for i, s_ in enumerate((0, 1)):
            if res[s_] is None:
                continue
            g_rec, g_lst = gather.results(s_)
            gather_checked &= bool(np.array_equal(g_rec, res[s_]["records"].view(np.uint32).reshape(-1, 3)))
            gather_checked &= bool(np.array_equal(g_lst, res[s_]["visibleList"]))
    # digest of the whole-scene results (records + ordered visible lists of both phases): the same for every number of
    # ranks (tests/test_gpu_host_path.py compares a 1-rank run with a 2-rank run)
    import hashlib
    h = hashlib.sha1()
    for s_ in (0, 1):
        if gather is not None:
            if res[s_] is None and gather.results(s_)[0].size == 0:
                continue
            g_rec, g_lst = gather.results(s_)
        elif res[s_] is not None:
            g_rec, g_lst = res[s_]["records"].view(np.uint32).reshape(-1, 3), res[s_]["visibleList"]
        else:
            continue
        h.update(np.ascontiguousarray(g_rec).tobytes()); h.update(np.ascontiguousarray(g_lst).tobytes())
    lists_digest = h.hexdigest()
    ms_per_step = dt / args.steps * 1e3
    value = tested_all / (dt / args.steps) / 1e9

    # ---- roofline of the dominant kernel: HIP events on the kernel's own stream (back-end profile) ----
    # Two profiled passes of the same frame.  (1) ONLY the dominant kernel bracketed (trhip_profile_filter), K frames in
    # steady state behind 64 untimed ones: one event pair per frame, so the frame keeps the overlap and the clocks of the
    # timed region -- this is `avg_launch_ms`.  (2) every launch bracketed, 5 frames: `per_kernel_ms` (each launch then
    # runs alone behind its own event, and the first launches follow an idle gap: round 3 took the dominant kernel's figure
    # from this pass and read 8 % above the kernel trace, profiles/r4/experiments.md section 1).
    roofline = None
    if not args.no_profile:
        def run_frames(n):
            for _ in range(n):
                r.set_camera(view)
                r.frame()
        dev.profile_filter(DOMINANT)
        run_frames(64)
        dev.wait_idle()
        run_frames(64)                                     # back to back, no synchronisation from here to the last profiled frame
        dev.profile_reset()
        dev.profile_enable(True)
        run_frames(max(args.steps, 5))
        dev.wait_idle()
        n_launch, total_ms = dev.profile()[DOMINANT]
        dev.profile_enable(False)
        avg_ms = total_ms / n_launch
        dev.profile_filter(None)
        dev.profile_reset()
        dev.profile_enable(True)
        run_frames(5)
        dev.wait_idle()
        prof = dev.profile()
        dev.profile_enable(False)
        r0 = res[0]
        t0_tested = gs_num_meshlets(spec, r0["records"])
        inst_submitted = len(np.unique(r0["records"]["m_InstanceConstIdx"]))
        # ALGORITHMIC bytes of one launch of the early meshlet-cull kernel (DESIGN.md "Kernels"):
        # 32 B MeshletData per meshlet tested + 12 B record read + 4 B mask written per group
        # + 68 B (world matrix + mesh index) per submitted instance.  The kernel itself streams a derived 20-byte copy of
        # what it needs of each MeshletData (the meshlet cull stream), so `traffic` is BELOW this figure: `frac` prices the
        # reference's bytes against the kernel's time as the contract asks; `frac_by_traffic` prices the bytes that moved.
        alg_bytes = 32 * t0_tested + 16 * len(r0["records"]) + 68 * inst_submitted
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters and the kernel's average duration in the rocprofv3 kernel trace (separate
        # runs, committed summary): they belong to ONE build of the kernel on ONE workload -- the file carries the hash of the
        # kernel's sources + build flags (kernel_source_sha16) and is reported only while they are unchanged; null otherwise.
        traffic = trace_ms = valu_active = None
        for rnd in ("r4", "r3"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", rnd, "traffic.json")))
                if (tj["config"] == args.config and tj["culling_flags"] == args.flags and tj["kernel"] == DOMINANT and world == 1
                        and tj.get("kernel_source_sha16") == kernel_source_sha16()):
                    traffic = int(tj["hbm_bytes_per_launch"])
                    trace_ms = tj.get("avg_launch_ms_trace")
                    valu_active = tj.get("valu_active")
                    break
            except (OSError, KeyError, ValueError):
                pass
        # the whole frame against the same roofline (32 B per meshlet tested + 24 B per group record written
        # and read + 72 B per instance processed + 4 B per visible meshlet) / frame time
        frame_alg = 32 * tested_all + 24 * groups_all + 72 * (spec.num_instances + int(res.get("lateCount", 0))) + 4 * visible_all
        frame_frac = frame_alg / (dt / args.steps) / 1e9 / HBM_PEAK_GBS if world == 1 else None
        roofline = dict(bound="hbm", kernel=DOMINANT, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, avg_launch_ms=round(avg_ms, 4),
                        avg_launch_launches=int(n_launch), avg_launch_ms_trace=trace_ms,
                        frac_by_traffic=round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                        valu_active=valu_active,
                        note="frac = the reference's bytes (32 B MeshletData per meshlet, SURVEY 8d) / the kernel's time / 8 TB/s, as the contract "
                             "asks; the kernel streams a derived 20-byte copy (traffic, frac_by_traffic = the HBM fraction of the bytes that "
                             "moved) and is paced by its L1 misses in flight, stream and HZB-table lookups together (valu_active = SQ_ACTIVE_INST_VALU / SIMD cycles, PMC; "
                             "profiles/r4/experiments.md section 5).  avg_launch_ms: HIP events in steady state, only this kernel bracketed; "
                             "avg_launch_ms_trace: the committed rocprofv3 kernel trace",
                        frame_frac=round(frame_frac, 4) if frame_frac is not None else None, frame_algorithmic_bytes=int(frame_alg),
                        # against what a streaming-read kernel reaches on this part (tools/membw.hip, profiles/r1/membw_calibration.txt)
                        frac_of_measured_stream=round(achieved / MEASURED_STREAM_GBS, 4), measured_stream_peak=MEASURED_STREAM_GBS,
                        algorithmic_bytes_per_launch=int(alg_bytes), meshlets_per_launch=int(t0_tested),
                        per_kernel_ms={k: round(v[1] / v[0], 4) for k, v in prof.items()})

    # ---- BASELINE configs[4]'s defining feature on the same scene: the instance transforms rebuilt from a node hierarchy every
    # frame (updateinstanceconsts + cull-cache refresh in front of the cull), the same K steps, timed the same way ----
    animated = None
    if world == 1 and not args.animate and not args.no_animated_leg and not args.no_profile and args.emulate_ranks <= 1 and args.config == "C3":
        nodes, prim_to_node = synth.animated_nodes(spec, 0)
        r.load_nodes(nodes, prim_to_node)
        del nodes
        for i in range(64):
            step()
        sync()
        ta = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dta = time.perf_counter() - ta
        res_a = r.results()
        tested_a = sum(gs_num_meshlets(spec, res_a[s_]["records"]) for s_ in (0, 1) if res_a[s_] is not None)
        animated = dict(ms_per_step=round(dta / args.steps * 1e3, 4), value=round(tested_a / (dta / args.steps) / 1e9, 3), unit="Gmeshlets/s",
                        meshlets_tested_per_frame=int(tested_a), steps=args.steps,
                        what="the same scene with every instance transform rebuilt from a two-level node hierarchy each frame "
                             "(updateinstanceconsts.hlsl:11-52 + instance cull cache refresh), node data resident")

    out = None
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:           # reported at N=1 only (the host cores are shared by the ranks otherwise)
            try:
                cpu = cpu_baseline(spec, view, depth, args.cpu_sample_instances, host_threads())
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU number
                cpu = dict(error=str(e))
        out = {
            "metric": "Gmeshlets/s culled (whole node), 100M meshlets", "value": round(value, 3), "unit": "Gmeshlets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: synthetic {n_total} unique meshlets ({spec.num_instances} instances x {spec.meshlets_lod0}), "
                                   "2-phase frustum+HZB+cone cull, 3840x2160 -> 2048x2048 R16F HZB, instances sharded over ranks"
                                   + (" + RCCL all-gather of per-rank records and visibility masks, whole-scene lists rebuilt on every rank" if world > 1 else ""),
                       "meshlets_in_scene": n_total, "meshlets_tested_per_frame": tested_all, "groups_per_frame": groups_all,
                       "visible_per_frame": visible_all, "culling_flags": args.flags},
            "roofline": roofline, "cpu_baseline": cpu, "animated": animated,
            "prime_steps": args.prime_steps,       # untimed frames in front of the W warm-up steps (device at its working clocks)
        }
        out["lists_digest"] = lists_digest
        out["collective"] = gather.collective if gather is not None else None     # "rccl-direct" | "pg" | "host-staged" | "loopback"
        if args.emulate_ranks > 1:
            out["metric"] = f"DIAGNOSTIC (rank 0 of {args.emulate_ranks} emulated on one GPU) - not a bench result"
        if args.animate or args.config != "C3":
            out["metric"] = (f"DIAGNOSTIC ({args.config}{', instance transforms rebuilt from the node hierarchy every frame' if args.animate else ''}) "
                             "- Gmeshlets/s culled, not the headline config")
        if gather_checked is not None:
            out["gather_checked"] = gather_checked
    sync()
    if out is not None:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if gather is not None:
        gather.close()
    r.shutdown()
    if dist is not None:
        dist.destroy_process_group()


def free_rendezvous_port() -> int:
    """A free TCP port on 127.0.0.1 BELOW the ephemeral range (32768+): a port the kernel hands out for bind(0) can be
    taken by some process's outgoing connection before torch.distributed.run binds it (seen once: EADDRINUSE)."""
    import random
    import socket
    rng = random.Random(os.getpid() ^ int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        port = rng.randrange(20000, 32000)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    raise RuntimeError("no free port in [20000, 32000)")


def self_launch(n: int) -> int:
    """Start `n` ranks of this script through torch.distributed.run as a child process (rendezvous on 127.0.0.1, a free
    port), let its stdout (rank 0's one JSON line) and stderr pass straight through, return its exit code."""
    import subprocess
    port = free_rendezvous_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_threads() // n)))
    log(f"[bench] --gpus {n} without WORLD_SIZE: launching {n} ranks: {' '.join(cmd)}")
    return subprocess.run(cmd, env=env).returncode


def kernel_source_sha16() -> str:
    """Identity of the dominant kernel's build: sha256 over the sources of its translation unit and the build flags
    (toyrenderer_amd/csrc: k_basepass_as.hip, cull_math.hip.h, instance_cache.hip.h, meshlet_exact.hip.h, ShaderInterop.h, trhip_internal.h,
    Makefile).  profiles/r3/traffic.json stores the value of the build its counters were collected on."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "toyrenderer_amd", "csrc")
    for f in ("k_basepass_as.hip", "cull_math.hip.h", "instance_cache.hip.h", "meshlet_exact.hip.h", "ShaderInterop.h", "trhip_internal.h", "Makefile"):
        with open(os.path.join(base, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def host_threads() -> int:
    """Host cores this process may use (the GPU box gives one GPU's share of the node's cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 64))


def gs_num_meshlets(spec, recs) -> int:
    """Meshlets tested by a record list: every LOD0 group of spec.meshlets_lod0 meshlets is full but the last."""
    if len(recs) == 0:
        return 0
    off = recs["m_MeshletGroupOffset"].astype(np.int64)
    return int(np.minimum(32, spec.meshlets_lod0 - off).clip(min=0).sum())


if __name__ == "__main__":
    main()
