"""Multi-GPU exchange of the visibility results: one process per GPU, instances sharded by
contiguous ranges; every rank ends the frame with the WHOLE scene's amplification records and ordered
visible lists, bit-identical to a single-GPU frame (rank-major concatenation == single-GPU canonical
order).  Not in the reference (single GPU, GraphicRHI.cpp:165).

What crosses xGMI is the compact form of a rank's pass slots -- per group of 32 meshlets the 12-byte
record and the 4-byte lane mask (1 bit per tested meshlet instead of 4 bytes per visible meshlet: 28 MB
instead of 143 MB per frame on the 100 M-meshlet config) -- in one fixed-capacity SHARD SLOT per rank:

    words [0, 16)             header: {G_s, V_s} of pass slot s at words 2s, 2s+1; word 8 = overflow flag
    words [16, 16+3S)         records of pass slot 0, then 1, ... back to back (S = slot_groups)
    words [16+3S, 16+4S)      lane masks in the same order

Per frame:  pack (HIP kernel, compute stream)  ->  ONE equal-size all_gather_into_tensor (RCCL)  ->
unpack (HIP kernels: rank-major concatenation with device-side offsets from the headers, then the same
count/scan/expand list build the single-GPU path uses).  There is no host read-back, so nothing stalls
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
Group-capacity overflow (the reference's 65 535-group cap, Q2) is NOT made global: a sharded run equals
the single-GPU run only if no rank drops groups; a rank that does raises STATUS_GROUPS_DROPPED.
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


def slot_words(slot_groups: int) -> int:
    return HEADER_WORDS + 4 * int(slot_groups)


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
    the whole-scene outputs.  HipShardExchange implements them with the gfx950 kernels."""

    def __init__(self, dist, torch, world: int, rank: int, slot_groups: int, pass_slots=(0, 1),
                 group_capacity: int | None = None, list_capacity: int | None = None, device="cpu"):
        assert 1 <= len(pass_slots) <= MAX_PASS_SLOTS and all(0 <= s < MAX_PASS_SLOTS for s in pass_slots)
        self.dist, self.torch, self.world, self.rank = dist, torch, int(world), int(rank)
        self.slot_groups = int(slot_groups)
        self.slot_words = slot_words(slot_groups)
        assert self.world * self.slot_words < 2 ** 32, "gathered buffer exceeds 2^32 words"
        self.pass_slots = tuple(pass_slots)
        self.group_capacity = int(group_capacity if group_capacity is not None else self.world * self.slot_groups)
        assert self.group_capacity <= 1 << 27, "more than 2^27 groups cannot be encoded as (g << 5) | lane"
        self.list_capacity = int(list_capacity if list_capacity is not None else 32 * self.group_capacity)
        self.device = device
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
        G, V, status = int(args[0]), int(args[4]), int(args[7])
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


class HipShardExchange(ShardExchange):
    """GPU path over the C++ host mirror: packs the renderer's pass slots on its stream, gathers and
    unpacks on a second stream (overlap=True) so the next frame's culling runs meanwhile."""

    def __init__(self, renderer, dist, world: int, rank: int, slot_groups: int, pass_slots=(0, 1),
                 group_capacity: int | None = None, list_capacity: int | None = None, overlap: bool = True,
                 stage_through_host: bool = False):
        import torch

        from . import rhi
        super().__init__(dist, torch, world, rank, slot_groups, pass_slots, group_capacity, list_capacity, device="cuda")
        self.rhi, self.r = rhi, renderer
        self.stage_through_host = bool(stage_through_host)
        L = rhi.load()
        self.dev = rhi.Device(handle=renderer.device())                       # the renderer's device (compute stream)
        self.compute = torch.cuda.ExternalStream(int(L.trhip_device_stream(self.dev.h) or 0))
        self.overlap = bool(overlap)
        if self.overlap:
            self.comm = torch.cuda.Stream()
            self.comm_dev = rhi.Device(torch.cuda.current_device(), stream=self.comm.cuda_stream)
        else:
            self.comm, self.comm_dev = self.compute, self.dev
        self.packed = [torch.cuda.Event() for _ in range(2)]
        self.released = [torch.cuda.Event() for _ in range(2)]
        self.send_buf = [self.dev.wrap_buffer(t.data_ptr(), t.numel() * 4, f"ShardSlotSend{i}") for i, t in enumerate(self.send)]
        self.pack_cl = [self.dev.create_command_list() for _ in range(2)]
        self._pack_key = [None, None]
        # the unpack only touches buffers owned here: recorded once per buffer index
        self.unpack_cl = []
        push = np.array([self.world, self.slot_groups], np.uint32)
        self._wrapped = []
        for b in range(2):
            cl = self.comm_dev.create_command_list()
            binds = [rhi.PUSH(0), rhi.SRV(0, self._wrap(self.recv[b], f"ShardSlotsRecv{b}"))]
            for s in self.pass_slots:
                o = self.out[s]
                binds += [rhi.UAV(4 * s, self._wrap(o["records"], f"AllRecords{s}")), rhi.UAV(4 * s + 1, self._wrap(o["masks"], f"AllMasks{s}")),
                          rhi.UAV(4 * s + 2, self._wrap(o["list"], f"AllVisibleList{s}")), rhi.UAV(4 * s + 3, self._wrap(o["args"], f"AllArgs{s}"))]
            cl.open()
            cl.dispatch("visibility_CS_UnpackShards", binds, (1, 1, 1), push=push)
            cl.close()
            self.unpack_cl.append(cl)

        # in-frame exchange of the late-list lengths (module docstring)
        self.late_counts = [torch.zeros(self.world, dtype=torch.int32, device="cuda") for _ in range(2)]
        # its own communicator: the 4-byte in-frame collective must not queue behind the previous frame's slot
        # exchange on the process group's stream (every rank creates the group, in the same order)
        # RCCL proper: two communicators of our own (slot exchange on the comm stream, late counts on the auxiliary one).
        # Should their set-up fail on every rank alike (library / symbol trouble), the process group's collectives do
        # the job instead -- slower per call, same results; the ranks agree on that with one all-reduce.
        self.rccl_slots = self.rccl_late = None
        if not self.stage_through_host:
            ok = 1
            try:
                self.rccl_slots = RcclComm(dist, torch, self.world, self.rank)
                self.rccl_late = RcclComm(dist, torch, self.world, self.rank)
            except (OSError, AttributeError, RuntimeError) as e:
                import sys
                print(f"[rank {rank}] direct RCCL communicators unavailable ({e}); using the process group", file=sys.stderr, flush=True)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                for c in (self.rccl_slots, self.rccl_late):
                    if c is not None:
                        c.destroy()
                self.rccl_slots = self.rccl_late = None
        self.late_group = dist.new_group() if self.rccl_late is None else None
        self.aux = torch.cuda.Stream()
        self.late_posted = [torch.cuda.Event() for _ in range(2)]
        self.late_ready = [torch.cuda.Event() for _ in range(2)]
        self._ptr_views = {}
        self._hook_error = None
        renderer.set_shard_late_exchange(self.late_exchange)

    def late_exchange(self, hip_stream: int, late_count_ptr: int, shard_info_ptr: int, bucket: int, phase: int):
        """Runs inside renderer.frame() (include/trhost.h).  Phase 0, right after the early instance cull: the
        all-gather of the late counts and the {lower ranks, all ranks} kernel go to the auxiliary stream, behind an
        event on the compute stream -- they complete while the early meshlet cull runs.  Phase 1, right before the late
        instance cull: the compute stream waits for that result."""
        try:
            if phase == 1:
                self.compute.wait_event(self.late_ready[bucket])
                return
            t = self._ptr_views.get(late_count_ptr)
            if t is None:
                t = self._ptr_views[late_count_ptr] = self.torch.as_tensor(_DevWords(late_count_ptr, 1), device="cuda")
            self.late_posted[bucket].record(self.compute)
            with self.torch.cuda.stream(self.aux):
                self.aux.wait_event(self.late_posted[bucket])
                if self.rccl_late is not None:
                    self.rccl_late.all_gather_i32(late_count_ptr, self.late_counts[bucket].data_ptr(), 1, self.aux.cuda_stream)
                else:
                    self._all_gather(self.late_counts[bucket], t, group=self.late_group)
                rc = self.rhi.load().trhip_launch_shard_late_info(self.aux.cuda_stream, self.late_counts[bucket].data_ptr(), self.world, self.rank, shard_info_ptr)
                if rc != 0:
                    raise RuntimeError(self.rhi.load().trhip_last_error().decode(errors="replace"))
                self.late_ready[bucket].record(self.aux)
        except Exception as e:      # a ctypes callback cannot propagate: re-raised by run()
            self._hook_error = e

    def _wrap(self, t, name):
        buf = self.comm_dev.wrap_buffer(t.data_ptr(), t.numel() * 4, name)
        self._wrapped.append(buf)
        return buf

    def _begin(self, b):
        if self._hook_error is not None:
            e, self._hook_error = self._hook_error, None
            raise e
        self.compute.wait_event(self.released[b])

    def _pack(self, b):
        rhi = self.rhi
        pbs = [(s, self.r.pass_buffers(s)) for s in self.pass_slots]
        key = tuple((s, pb.ran, pb.records, pb.vis_mask, pb.dispatch_args, pb.draw_args) for s, pb in pbs)
        cl = self.pack_cl[b]
        if self._pack_key[b] != key:                    # the render graph hands out the same buffers frame after frame: record once
            binds = [rhi.PUSH(0), rhi.UAV(0, self.send_buf[b])]
            for s, pb in pbs:
                if not pb.ran:
                    continue
                for k, h in enumerate((pb.records, pb.vis_mask, pb.dispatch_args, pb.draw_args)):
                    x = rhi.bind(rhi.BIND_STRUCTURED_SRV, 4 * s + k)
                    x.resource = h
                    binds.append(x)
            cl.open()
            cl.dispatch("visibility_CS_PackShard", binds, (1, 1, 1), push=np.array([self.slot_groups], np.uint32))
            cl.close()
            self._pack_key[b] = key
        self.dev.execute(cl)
        self.packed[b].record(self.compute)

    @contextlib.contextmanager
    def _comm(self, b):
        self.comm.wait_event(self.packed[b])
        with self.torch.cuda.stream(self.comm):
            yield
            self.released[b].record(self.comm)

    def _exchange_slots(self, b):
        if self.rccl_slots is not None:
            self.rccl_slots.all_gather_i32(self.send[b].data_ptr(), self.recv[b].data_ptr(), self.slot_words, self.comm.cuda_stream)
        else:
            self._all_gather(self.recv[b], self.send[b])

    def _unpack(self, b):
        self.comm_dev.execute(self.unpack_cl[b])

    def wait(self):
        self.comm.synchronize()

    def close(self):
        self.wait()
        self.aux.synchronize()
        self.r.set_shard_late_exchange(None)
        for c in (self.rccl_slots, self.rccl_late):
            if c is not None:
                c.destroy()
        for cl in self.pack_cl + self.unpack_cl:
            cl.release()
        for buf in self.send_buf + self._wrapped:
            buf.release()
        if self.overlap:
            self.comm_dev.destroy()
