"""Multi-GPU exchange of the visibility results: one process per GPU, instances sharded by
contiguous ranges; every rank ends the frame with the WHOLE scene's amplification records and ordered
visible lists, bit-identical to a single-GPU frame (rank-major concatenation == single-GPU canonical
order).  Not in the reference (single GPU, GraphicRHI.cpp:165).

What crosses xGMI is the compact form of a rank's pass slots, in one fixed-capacity SHARD SLOT per rank:
  * per group of 32 meshlets the 4-byte lane mask (1 bit per tested meshlet instead of 4 bytes per visible meshlet);
  * the amplification records as RUNS: the instance pass emits, per submitted instance, consecutive records
    {instance, lod, 0}, {instance, lod, 32}, ... (gpuculling.hlsl:139-157), so a maximal run of records that continue
    each other is 16 bytes {instance, lod, first group offset, index of its first record} whatever its length -- one
    entry per submitted INSTANCE instead of 12 bytes per GROUP, lossless for any record array.
On the 100 M-meshlet config (4 groups per instance) that is 8 bytes per group: 25 MB per frame instead of the 143 MB of
the visible lists.

    words [0, 16)             header: {G_s, X_s} of pass slot s at words 2s, 2s+1 (groups sent = the rank's valid records; groups
                              its dispatch counter counted, which includes the groups of instances dropped at the capacity); word 8 = overflow flag (groups > S or
                              runs > R); word 9 = groups dropped (Q2); words 10..13 = cumulative run count after pass slot s
    words [16, 16+4R)         run entries of pass slot 0, then 1, ... back to back (R = slot_runs)
    words [16+4R, 16+4R+S)    lane masks in the same order (S = slot_groups)

Per frame:  pack (HIP kernel, compute stream)  ->  ONE equal-size all_gather_into_tensor (RCCL)  ->
unpack (HIP kernels: the whole-scene records rebuilt rank-major straight from the received run entries, the masks
concatenated, device-side offsets from the headers; then the same count/scan/expand list build the single-GPU path uses).  There is no host read-back, so nothing stalls
the submission of the next frame; the collective and the unpack run on a second stream and overlap the
next frame's culling (send/receive buffers are double-buffered and guarded by events).

One more, tiny, collective sits INSIDE the frame: the reference sizes the late instance cull from the
late-list length (gpuculling.hlsl:182-195: ceil(count/64) groups of 32 threads, i.e. only the first
ceil(count/64)*32 entries are processed).  For the sharded run to equal the single-GPU run that rule must
see the whole scene's late list, so the ranks all-gather their late counts (4 bytes each, device to
device, no host read-back) and a one-thread kernel derives {entries of the lower ranks, entries of all
ranks} for the late dispatch.  The late count is final right after the EARLY instance cull, so the
exchange is posted there, on an auxiliary stream, and has the whole early meshlet cull and HZB build to
complete; the compute stream only waits for its event before the late instance cull (`late_exchange`,
hooked into the frame through trhost_set_shard_late_exchange / FrameDriver(shard_late=...)).
Group-capacity overflow (the reference's 65 535-group cap, Q2, gpuculling.hlsl:64-74) is made GLOBAL by the unpack when the
exchange is given `global_group_cap` = the capacity of the single-GPU run it reproduces (every rank runs its passes with
that same maxGroups): in rank-major order rank p's groups start at B_p = sum of the lower ranks' COUNTED groups X_q (the
counter advances for dropped instances too).  The first instance the single-GPU pass drops is the first whose run ends at
or beyond the capacity, i.e. the run that contains global record index cap - 1; it lies on the first rank with B_p + X_p >=
cap, starts at that rank's record `first` of the run entry containing local index cap - B_p - 1 (or is the instance the
rank dropped itself), and everything behind it is undefined in the reference (stale buffer contents; this build's
convention: not part of the result).  So ranks below contribute all their groups, that rank its first `first` groups,
ranks above nothing; the whole-scene arguments are {sum X_p, 1, 1, validRecords}.  No collective inside the frame: the
ranks' own passes run on supersets, the unpack trims.  Without `global_group_cap` a rank that drops groups raises
STATUS_GROUPS_DROPPED (results() raises).
torch.distributed is plumbing here: it launches the ranks and carries the 128-byte RCCL unique ids; the collectives
themselves are direct ncclAllGather calls on this module's own communicators and streams (`RcclComm`).
"""
from __future__ import annotations

import contextlib

import numpy as np

HEADER_WORDS = 16
MAX_PASS_SLOTS = 4
STATUS_SLOT_OVERFLOW = 1      # a rank produced more groups than its shard slot holds
STATUS_CAPACITY = 2           # the whole-scene buffers are smaller than the gathered total
STATUS_BAD_HEADER = 4
STATUS_GROUPS_DROPPED = 8      # a rank hit its group capacity (Q2): the sharded result differs from single-GPU


def exclusive_offsets(counts):
    """counts[R] -> (offsets[R], total)."""
    offs, acc = [], 0
    for c in counts:
        offs.append(acc)
        acc += int(c)
    return offs, acc


def shard_range(n: int, rank: int, world: int):
    """Contiguous instance range of a rank (SURVEY.md 8(e) "Partitioning")."""
    return (rank * n) // world, ((rank + 1) * n) // world


def slot_words(slot_groups: int, slot_runs: int | None = None) -> int:
    """Words of one shard slot; slot_runs None = slot_groups (always enough: a run holds at least one group)."""
    return HEADER_WORDS + 4 * int(slot_groups if slot_runs is None else slot_runs) + int(slot_groups)


def shard_run_capacity(entries_in_shard: int) -> int:
    """Upper bound of the runs one shard can emit in a frame over ALL its pass slots: a submitted id-list entry is one
    run (its records continue each other), early and late sets are disjoint."""
    return int(entries_in_shard)


def shard_group_capacity(num_meshlets_per_lod: np.ndarray, mesh_of_entry: np.ndarray) -> int:
    """Upper bound of the groups one shard can emit in a frame over ALL its pass slots: every list entry
    submitted once (early and late sets are disjoint) at its largest LOD.
    num_meshlets_per_lod: [meshes, lods]; mesh_of_entry: mesh index of every id-list entry of the shard."""
    per_mesh = ((np.asarray(num_meshlets_per_lod, np.int64) + 31) // 32).max(axis=1)
    return int(per_mesh[np.asarray(mesh_of_entry, np.int64)].sum())


def agree_slot_groups(dist, torch, local_capacity: int, device="cpu") -> int:
    """All ranks must use one slot size: the maximum of the per-rank capacities (setup time, once)."""
    t = torch.tensor([int(local_capacity)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


class ShardExchange:
    """Host logic of the exchange (buffers, double buffering, the one collective per frame).  The data
    movers are hooks: `_pack(b)` fills send[b] from the local pass slots, `_unpack(b)` turns recv[b] into
    the whole-scene outputs.  It is the PROTOCOL stated in Python (CPU tests run it over gloo with the numpy movers of
    tests/exchange_ref.py); on the GPU the same sequence is driven natively (NativeShardExchange, ShardExchange.cpp)."""

    def __init__(self, dist, torch, world: int, rank: int, slot_groups: int, pass_slots=(0, 1),
                 group_capacity: int | None = None, list_capacity: int | None = None, device="cpu", slot_runs: int | None = None,
                 global_group_cap: int | None = None):
        assert 1 <= len(pass_slots) <= MAX_PASS_SLOTS and all(0 <= s < MAX_PASS_SLOTS for s in pass_slots)
        self.dist, self.torch, self.world, self.rank = dist, torch, int(world), int(rank)
        self.slot_groups = int(slot_groups)
        self.slot_runs = int(slot_groups if slot_runs is None else slot_runs)
        self.slot_words = slot_words(slot_groups, self.slot_runs)
        assert self.world * self.slot_words < 2 ** 32, "gathered buffer exceeds 2^32 words"
        self.pass_slots = tuple(pass_slots)
        self.group_capacity = int(group_capacity if group_capacity is not None else self.world * self.slot_groups)
        assert self.group_capacity <= 1 << 27, "more than 2^27 groups cannot be encoded as (g << 5) | lane"
        self.list_capacity = int(list_capacity if list_capacity is not None else 32 * self.group_capacity)
        self.device = device
        # Q2 made global: the group capacity (maxGroups) of the single-GPU run this exchange reproduces -- every rank runs its
        # passes with the same value; the unpack cuts the rank-major concatenation in front of the first instance that run
        # would drop.  None: a rank that drops groups makes results() raise (STATUS_GROUPS_DROPPED).
        self.global_group_cap = int(global_group_cap) if global_group_cap else None
        i32 = torch.int32
        self.send = [torch.zeros(self.slot_words, dtype=i32, device=device) for _ in range(2)]
        self.recv = [torch.zeros(self.world * self.slot_words, dtype=i32, device=device) for _ in range(2)]
        self.out = {s: dict(records=torch.zeros(3 * max(self.group_capacity, 1), dtype=i32, device=device),
                            masks=torch.zeros(max(self.group_capacity, 1), dtype=i32, device=device),
                            list=torch.zeros(max(self.list_capacity, 1), dtype=i32, device=device),
                            args=torch.zeros(8, dtype=i32, device=device)) for s in self.pass_slots}
        self.frame = 0

    # ---- hooks -------------------------------------------------------------------------------------
    def _pack(self, b: int):
        raise NotImplementedError

    def _unpack(self, b: int):
        raise NotImplementedError

    def _begin(self, b: int):
        """Before send[b] is overwritten: the exchange that last used buffers b must be done."""

    def _comm(self, b: int):
        """Context in which the collective and the unpack of buffers b are issued."""
        return contextlib.nullcontext()

    def wait(self):
        """Block until every exchange issued so far has completed."""

    # ---- per frame ---------------------------------------------------------------------------------
    def _all_gather(self, out, inp, group=None):
        """all_gather_into_tensor; with `stage_through_host` (tests that run several ranks on ONE GPU over gloo,
        where RCCL cannot be used) device tensors take the detour through host memory."""
        if getattr(self, "stage_through_host", False) and inp.is_cuda:
            o = self.torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(o, inp.cpu(), group=group)
            out.copy_(o)
        else:
            self.dist.all_gather_into_tensor(out, inp, group=group)

    def run(self):
        b = self.frame & 1
        self._begin(b)
        self._pack(b)
        with self._comm(b):
            self._exchange_slots(b)
            self._unpack(b)
        self.frame += 1

    def _exchange_slots(self, b: int):
        self._all_gather(self.recv[b], self.send[b])

    def results(self, pass_slot: int):
        """(records[G,3], visible list[V]) of the whole scene for one pass slot, as u32 (host copy).
        Raises if a slot or the whole-scene buffers overflowed."""
        self.wait()
        o = self.out[pass_slot]
        args = o["args"].cpu().numpy().view(np.uint32)
        G, V, status = min(int(args[0]), int(args[3])), int(args[4]), int(args[7])     # {X, 1, 1, validRecords}: X counts dropped groups too (Q2)
        if status:
            raise RuntimeError(f"shard exchange failed (status {status}): "
                               + ("a rank's groups exceed slot_groups; " if status & STATUS_SLOT_OVERFLOW else "")
                               + ("whole-scene capacity exceeded; " if status & STATUS_CAPACITY else "")
                               + ("corrupt slot header; " if status & STATUS_BAD_HEADER else "")
                               + ("a rank dropped groups at its capacity (Q2)" if status & STATUS_GROUPS_DROPPED else ""))
        if V > self.list_capacity:
            raise RuntimeError(f"whole-scene visible list holds {self.list_capacity} entries, frame produced {V}")
        return (o["records"][:3 * G].cpu().numpy().view(np.uint32).reshape(-1, 3),
                o["list"][:V].cpu().numpy().view(np.uint32))


class RcclComm:
    """One RCCL communicator driven directly (ctypes on the librccl torch already loaded): a collective is ONE
    ncclAllGather on the stream it belongs to -- no detour over the process group's stream, no tensor bookkeeping;
    the process group only carries the 128-byte unique id at set-up.  Every rank must construct it, in the same order."""

    _lib = None

    @classmethod
    def lib(cls, torch):
        if cls._lib is None:
            import ctypes as C
            import os
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            L = C.CDLL(path if os.path.exists(path) else "librccl.so")

            class UniqueId(C.Structure):
                _fields_ = [("internal", C.c_ubyte * 128)]
            L.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
            L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
            L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            L.ncclCommDestroy.argtypes = [C.c_void_p]
            L.ncclGetErrorString.restype = C.c_char_p
            L.ncclGetErrorString.argtypes = [C.c_int]
            cls._lib, cls._UniqueId = L, UniqueId
        return cls._lib

    def __init__(self, dist, torch, world: int, rank: int):
        import ctypes as C
        L = self.lib(torch)
        uid = self._UniqueId()
        if rank == 0:
            self._check(L.ncclGetUniqueId(C.byref(uid)))
        t = torch.frombuffer(bytearray(C.string_at(C.byref(uid), 128)), dtype=torch.uint8).cuda()   # all 128 bytes (zeros on ranks > 0)
        dist.broadcast(t, src=0)                            # the process group carries the id, nothing else
        raw = bytes(t.cpu().numpy().tobytes())
        C.memmove(C.byref(uid), raw, 128)
        self.comm = C.c_void_p()
        self._check(L.ncclCommInitRank(C.byref(self.comm), int(world), uid, int(rank)))

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("RCCL: " + self._lib.ncclGetErrorString(rc).decode(errors="replace"))

    def all_gather_i32(self, send_ptr: int, recv_ptr: int, count: int, stream_ptr: int):
        self._check(self._lib.ncclAllGather(send_ptr, recv_ptr, int(count), 2, self.comm, stream_ptr))   # 2 = ncclInt32

    def destroy(self):
        if self.comm:
            self._lib.ncclCommDestroy(self.comm)
            self.comm = None


class _DevWords:
    """Device memory owned by the back end, viewed by torch through __cuda_array_interface__."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


def shard_late_info(counts, rank: int):
    """{late entries of the lower ranks, of all ranks} -- what trhip_launch_shard_late_info computes."""
    counts = [int(c) for c in counts]
    return sum(counts[:rank]), sum(counts)


class NativeShardExchange:
    """GPU path: the exchange is driven natively by the host library (csrc/host/ShardExchange.cpp: pack on the
    renderer's stream, one all-gather, unpack + list rebuild on the exchange stream, in-frame late-count exchange on
    an auxiliary stream); per frame Python makes ONE call.  This class only sets it up: with RCCL it creates two
    communicators (`RcclComm`) and hands their ncclAllGather to the library; for tests that run several ranks on one
    GPU (`stage_through_host`, any torch.distributed backend) the collectives are host-staged callbacks."""

    def __init__(self, renderer, dist, world: int, rank: int, slot_groups: int, pass_slots=(0, 1),
                 group_capacity: int | None = None, list_capacity: int | None = None, overlap: bool = True,
                 stage_through_host: bool = False, loopback: bool = False, raster_depth: bool = False,
                 slot_runs: int | None = None, global_group_cap: int | None = None):
        """slot_runs: run entries a shard slot holds (shard_run_capacity of the largest shard; None = slot_groups).
        raster_depth: the frames rasterise their own depth (trhost_set_raster_depth): adds the cross-rank MAX of the
        depth buffer before every HZB build (one more communicator / process group)."""
        import ctypes as C

        import torch

        from . import host, rhi
        self.torch, self.dist, self.host, self.rhi = torch, dist, host, rhi
        self.world, self.rank, self.pass_slots = int(world), int(rank), tuple(pass_slots)
        self.list_capacity = int(list_capacity if list_capacity is not None else 32 * (group_capacity if group_capacity is not None else world * slot_groups))
        self.group_capacity = int(group_capacity if group_capacity is not None else world * slot_groups)
        L = host.load()
        self._keep = []
        d = host.ExchangeDesc()
        d.world, d.rank, d.slot_groups, d.group_capacity = self.world, self.rank, int(slot_groups), self.group_capacity
        d.list_capacity, d.overlap = self.list_capacity, int(bool(overlap))
        d.slot_runs = int(slot_groups if slot_runs is None else slot_runs)
        d.global_group_capacity = int(global_group_cap or 0)          # Q2 made global (module docstring)
        d.pass_slot_mask = sum(1 << s for s in self.pass_slots)
        # which id lists exist on SOME rank: every rank posts the in-frame late-count collective of exactly those buckets
        n_op, n_am = C.c_uint32(), C.c_uint32()
        host._check(L.trhost_scene_list_sizes(C.byref(n_op), C.byref(n_am)))
        present = [int(n_op.value > 0), int(n_am.value > 0)]
        if not loopback and dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.tensor(present, dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            present = [int(v) for v in t.cpu().tolist()]
        d.list_presence_mask = present[0] | (present[1] << 1)
        self.comms = []
        self.collective = "loopback" if loopback else "host-staged" if stage_through_host else None
        if loopback:
            # diagnostic (bench.py --emulate-ranks with TR_EMULATE_LOOPBACK=1): ONE process plays rank `rank` of `world`; a
            # collective copies this rank's contribution into every slot on the device, so the unpack sees `world` shards
            # (the volume of a real run, not its contents).
            # (ONE broadcasting copy kernel through torch: world hipMemcpyAsync calls were measured -- eight 5-us copy launches in a
            # row on the exchange stream, the same frame: 0.218 ms either way, the emulated frame is GPU-bound, host 0.16 ms)
            def fn(_user, send, recv, count, stream):
                try:
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream or 0))):
                        src = torch.as_tensor(_DevWords(int(send), int(count)), device="cuda")
                        torch.as_tensor(_DevWords(int(recv), int(count) * self.world), device="cuda").view(self.world, int(count)).copy_(src.unsqueeze(0).expand(self.world, -1))
                    return 0
                except Exception as e:          # a ctypes callback cannot propagate
                    import sys
                    print(f"loopback all-gather failed: {e}", file=sys.stderr, flush=True)
                    return 1
            cb = host.ALLGATHER_FN(fn)
            self._keep.append(cb)
            d.slots_allgather = d.late_allgather = C.cast(cb, C.c_void_p).value
        elif stage_through_host:
            groups = [dist.new_group(), dist.new_group()]

            def staged(group):
                def fn(_user, send, recv, count, stream):
                    try:
                        torch.cuda.ExternalStream(int(stream or 0)).synchronize()
                        src = torch.as_tensor(_DevWords(int(send), int(count)), device="cuda").cpu()
                        out = torch.empty(int(count) * self.world, dtype=torch.int32)
                        dist.all_gather_into_tensor(out, src, group=group)
                        torch.as_tensor(_DevWords(int(recv), int(count) * self.world), device="cuda").copy_(out)
                        torch.cuda.current_stream().synchronize()
                        return 0
                    except Exception as e:          # a ctypes callback cannot propagate
                        import sys
                        print(f"[rank {rank}] staged all-gather failed: {e}", file=sys.stderr, flush=True)
                        return 1
                cb = host.ALLGATHER_FN(fn)
                self._keep.append(cb)
                return C.cast(cb, C.c_void_p).value
            d.slots_allgather, d.late_allgather = staged(groups[0]), staged(groups[1])
            if raster_depth:
                dgroup = dist.new_group()

                def depth_fn(_user, words, count, stream):
                    try:
                        torch.cuda.ExternalStream(int(stream or 0)).synchronize()
                        dev = torch.as_tensor(_DevWords(int(words), int(count)), device="cuda")
                        h = dev.cpu()                       # non-negative floats order like their int32 bit patterns
                        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=dgroup)
                        dev.copy_(h)
                        torch.cuda.current_stream().synchronize()
                        return 0
                    except Exception as e:
                        import sys
                        print(f"[rank {rank}] staged depth all-reduce failed: {e}", file=sys.stderr, flush=True)
                        return 1
                dcb = host.DEPTH_ALLREDUCE_FN(depth_fn)
                self._keep.append(dcb)
                d.depth_allreduce_max = C.cast(dcb, C.c_void_p).value
        else:
            # Direct RCCL communicators; if any rank cannot create them (librccl not where torch keeps it, bootstrap
            # refused, ...) EVERY rank falls back to the process group's own all-gather on device tensors -- slower
            # (torch's stream hand-over per call) but the same data path, and said so on stderr.
            import os
            import sys

            def agree(ok: int) -> bool:
                flag = torch.tensor([int(ok)], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return bool(int(flag.item()))

            # Step 1: every rank checks its environment and the library LOCALLY (no communication), then all ranks agree.
            # Only then does anyone touch a communicator, so a rank that cannot use RCCL directly never leaves the others
            # waiting inside a broadcast or ncclCommInitRank.
            ok, why = 1, ""
            try:
                if os.environ.get("TR_NO_DIRECT_RCCL"):
                    raise RuntimeError("TR_NO_DIRECT_RCCL is set")
                R = RcclComm.lib(torch)
                fn_addr = C.cast(L.trhost_rccl_allgather, C.c_void_p).value
            except Exception as e:
                ok, why = 0, str(e)
            direct = agree(ok)
            if not direct and (why or rank == 0):
                print(f"[rank {rank}] direct RCCL communicator unavailable ({why or 'another rank cannot use it'}); using torch.distributed all_gather", file=sys.stderr, flush=True)
            if direct:
                # Step 2: all ranks construct the communicators in the same order and agree on the outcome AFTER EACH ONE: a
                # rank whose communicator k failed must not skip the broadcast + ncclCommInitRank of communicator k + 1 while
                # the others are inside it (mismatched collectives hang).  All ranks leave the loop together.
                for k in range(3 if raster_depth else 2):
                    try:
                        comm = RcclComm(dist, torch, self.world, self.rank)
                        user = (C.c_void_p * 2)(C.cast(R.ncclAllReduce if k == 2 else R.ncclAllGather, C.c_void_p).value, comm.comm.value)
                        self.comms.append(comm)
                        self._keep.append(user)
                    except Exception as e:
                        ok = 0
                        print(f"[rank {rank}] ncclCommInitRank failed ({e}); using torch.distributed all_gather", file=sys.stderr, flush=True)
                    direct = agree(ok)
                    if not direct:
                        break
            self.collective = "rccl-direct" if direct else "pg"
            if direct:
                d.slots_allgather, d.slots_user = fn_addr, C.cast(self._keep[0], C.c_void_p).value
                d.late_allgather, d.late_user = fn_addr, C.cast(self._keep[1], C.c_void_p).value
                if raster_depth:
                    d.depth_allreduce_max = C.cast(L.trhost_rccl_allreduce_max_u32, C.c_void_p).value
                    d.depth_user = C.cast(self._keep[2], C.c_void_p).value
            else:
                for c in self.comms:
                    c.destroy()
                self.comms, self._keep = [], []
                groups = [dist.new_group(), dist.new_group()]
                hand_over = [torch.cuda.Stream(), torch.cuda.Stream()]       # torch-side streams the process group is called from

                def through_torch(group, mine):
                    def fn(_user, send, recv, count, stream):
                        try:
                            ext = torch.cuda.ExternalStream(int(stream or 0))
                            with torch.cuda.stream(mine):
                                mine.wait_stream(ext)
                                src = torch.as_tensor(_DevWords(int(send), int(count)), device="cuda")
                                out = torch.as_tensor(_DevWords(int(recv), int(count) * self.world), device="cuda")
                                dist.all_gather_into_tensor(out, src, group=group)
                                ext.wait_stream(mine)
                            return 0
                        except Exception as e:          # a ctypes callback cannot propagate
                            print(f"[rank {rank}] all-gather through torch.distributed failed: {e}", file=sys.stderr, flush=True)
                            return 1
                    cb = host.ALLGATHER_FN(fn)
                    self._keep.append(cb)
                    return C.cast(cb, C.c_void_p).value
                d.slots_allgather, d.late_allgather = through_torch(groups[0], hand_over[0]), through_torch(groups[1], hand_over[1])
                if raster_depth:
                    dgroup, dstream = dist.new_group(), torch.cuda.Stream()

                    def depth_fn(_user, words, count, stream):
                        try:
                            ext = torch.cuda.ExternalStream(int(stream or 0))
                            with torch.cuda.stream(dstream):
                                dstream.wait_stream(ext)
                                dist.all_reduce(torch.as_tensor(_DevWords(int(words), int(count)), device="cuda"), op=dist.ReduceOp.MAX, group=dgroup)
                                ext.wait_stream(dstream)
                            return 0
                        except Exception as e:
                            print(f"[rank {rank}] depth all-reduce through torch.distributed failed: {e}", file=sys.stderr, flush=True)
                            return 1
                    dcb = host.DEPTH_ALLREDUCE_FN(depth_fn)
                    self._keep.append(dcb)
                    d.depth_allreduce_max = C.cast(dcb, C.c_void_p).value
        host._check(L.trhost_exchange_create(C.byref(d)))
        self._L = L

    def run(self):
        self.host._check(self._L.trhost_exchange_run())

    def wait(self):
        self.host._check(self._L.trhost_exchange_wait())

    def results(self, pass_slot: int):
        """(records[G,3], visible list[V]) of the whole scene for one pass slot, as u32 (host copy).
        Raises if a slot or the whole-scene buffers overflowed or a rank dropped groups."""
        import ctypes as C
        self.wait()
        h = [C.c_void_p() for _ in range(4)]
        self.host._check(self._L.trhost_exchange_outputs(int(pass_slot), *[C.byref(x) for x in h]))
        args = self.host._download(h[3].value, np.uint32, 8)
        G, V, status = min(int(args[0]), int(args[3])), int(args[4]), int(args[7])     # {X, 1, 1, validRecords}: X counts dropped groups too (Q2)
        if status:
            raise RuntimeError(f"shard exchange failed (status {status}): "
                               + ("a rank's groups exceed slot_groups; " if status & STATUS_SLOT_OVERFLOW else "")
                               + ("whole-scene capacity exceeded; " if status & STATUS_CAPACITY else "")
                               + ("corrupt slot header; " if status & STATUS_BAD_HEADER else "")
                               + ("a rank dropped groups at its capacity (Q2)" if status & STATUS_GROUPS_DROPPED else ""))
        if V > self.list_capacity:
            raise RuntimeError(f"whole-scene visible list holds {self.list_capacity} entries, frame produced {V}")
        return (self.host._download(h[0].value, np.uint32, 3 * G).reshape(-1, 3), self.host._download(h[2].value, np.uint32, V))

    def close(self):
        self.host._check(self._L.trhost_exchange_destroy())
        for c in self.comms:
            c.destroy()
        self.comms = []
