"""Multi-GPU exchange of the visibility results: one process per GPU, instances sharded by
contiguous ranges, per-rank visible lists + amplification records all-gathered over RCCL/xGMI so
that every rank ends the frame with the whole scene's lists in the single-GPU canonical order
(rank-major concatenation).  Not in the reference (single GPU, GraphicRHI.cpp:165).

Per frame and pass slot (early / late):
  1. all-gather of {groups G_r, visible V_r} per rank                        (tiny, fixed size)
  2. host reads the counts (one sync) and derives the offsets
  3. own list entries (g << 5 | lane) are rebased by sum_{p<r} G_p           (HIP kernel, in place)
  4. variable-length all-gather of lists (4 B x V_r) and records (12 B x G_r)
The 2-phase structure needs no collective between the phases (late lists stay rank-local).
torch.distributed is plumbing here (process group + RCCL calls on torch's current stream).
"""
from __future__ import annotations

import numpy as np


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


def all_gather_varlen(dist, out, local, counts, rank, uneven_ok: bool):
    """Gather `local[:counts[rank]]` of every rank into `out` at the exclusive offsets of `counts`
    (1-D tensors of one dtype).  RCCL: one coalesced uneven all_gather; gloo (CPU tests): broadcasts."""
    offs, total = exclusive_offsets(counts)
    assert out.numel() >= total, (out.numel(), total)
    views = [out[offs[p]:offs[p] + int(counts[p])] for p in range(len(counts))]
    if uneven_ok:
        dist.all_gather(views, local[:int(counts[rank])])
    else:
        views[rank].copy_(local[:int(counts[rank])])
        for p in range(len(counts)):
            if int(counts[p]):
                dist.broadcast(views[p], src=p)
    return offs, total


def exchange_counts(dist, torch, local_counts, world):
    """local_counts: 1-D int32 tensor -> numpy [world, n] on the host (the frame's one host sync)."""
    parts = [torch.empty_like(local_counts) for _ in range(world)]
    dist.all_gather(parts, local_counts)
    return torch.stack(parts).cpu().numpy().astype(np.int64)


def gather_slot(dist, rank, world, out_list, out_records, local_list, local_records, G, V, rebase, uneven_ok):
    """One pass slot.  G, V: per-rank group / visible counts (host, length world).  `rebase(add)` adds
    `add` to the first V[rank] entries of local_list in place (HIP kernel on the GPU path).
    Returns (total groups, total visible)."""
    gbase = int(sum(G[:rank]))
    if gbase and V[rank]:
        rebase(gbase << 5)
    _, v_tot = all_gather_varlen(dist, out_list, local_list, V, rank, uneven_ok)
    _, g3_tot = all_gather_varlen(dist, out_records, local_records, [3 * int(g) for g in G], rank, uneven_ok)
    return g3_tot // 3, v_tot


class _DevArray:
    """Device memory owned by the back end, viewed by torch through __cuda_array_interface__."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


class VisibleListGather:
    """GPU path: gathers the outputs of the C++ host mirror's pass slots 0 (early) and 1 (late)."""

    SLOTS = (0, 1)

    def __init__(self, renderer, dist, world: int, rank: int, global_record_cap: int):
        import torch

        from . import rhi
        self.torch, self.rhi, self.r, self.dist, self.world, self.rank = torch, rhi, renderer, dist, world, rank
        self.dev = rhi.Device(handle=renderer.device())
        self.all_lists = [torch.empty(global_record_cap * 32, dtype=torch.int32, device="cuda") for _ in self.SLOTS]
        self.all_records = [torch.empty(global_record_cap * 3, dtype=torch.int32, device="cuda") for _ in self.SLOTS]
        self.counts_local = torch.zeros(2 * len(self.SLOTS), dtype=torch.int32, device="cuda")
        self.counts_buf = self.dev.wrap_buffer(self.counts_local.data_ptr(), 16, "GatherCounts")
        self.cl = self.dev.create_command_list()
        self.last_counts = None
        self.totals = None
        self._views = {}

    def _tensor(self, handle, nwords):
        L = self.rhi.load()
        ptr = L.trhip_buffer_device_ptr(handle)
        key = (ptr, nwords)
        t = self._views.get(key)
        if t is None:
            t = self.torch.as_tensor(_DevArray(ptr, nwords), device="cuda")
            self._views[key] = t
        return t

    def run(self):
        torch, L, rhi = self.torch, self.rhi.load(), self.rhi
        pbs = [self.r.pass_buffers(s) for s in self.SLOTS]
        # 1. {G0, V0, G1, V1} with one tiny kernel into a torch-owned 16-byte tensor
        cl = self.cl
        cl.open()
        b = []
        for i, pb in enumerate(pbs):
            if pb.ran:
                x, y = rhi.bind(rhi.BIND_STRUCTURED_SRV, 2 * i), rhi.bind(rhi.BIND_STRUCTURED_SRV, 2 * i + 1)
                x.resource, y.resource = pb.dispatch_args, pb.draw_args
                b += [x, y]
        b.append(rhi.UAV(0, self.counts_buf))
        cl.dispatch("visibility_CS_PackCounts", b, (1, 1, 1))
        cl.close()
        self.dev.execute(cl)
        c = exchange_counts(self.dist, torch, self.counts_local, self.world)          # 2. the frame's one host sync
        self.last_counts = c
        self.totals = []
        for i, pb in enumerate(pbs):
            G, V = c[:, 2 * i], c[:, 2 * i + 1]
            if not pb.ran:
                self.totals.append((0, 0))
                continue
            lst = self._tensor(pb.visible_list, L.trhip_buffer_size(pb.visible_list) // 4)
            rec = self._tensor(pb.records, L.trhip_buffer_size(pb.records) // 4)

            def rebase(add, pb=pb):                                                    # 3. rebase own entries (HIP kernel)
                cl.open()
                bb = [rhi.PUSH(0), rhi.bind(rhi.BIND_STRUCTURED_UAV, 0), rhi.bind(rhi.BIND_STRUCTURED_SRV, 0)]
                bb[1].resource, bb[2].resource = pb.visible_list, pb.draw_args
                cl.dispatch("visibility_CS_RebaseVisibleList", bb, (1, 1, 1), push=np.array([add >> 5], np.uint32))
                cl.close()
                self.dev.execute(cl)
            self.totals.append(gather_slot(self.dist, self.rank, self.world, self.all_lists[i], self.all_records[i], lst, rec, G, V, rebase, True))   # 4.

    def results(self, slot_index: int):
        G, V = self.totals[slot_index]
        return (self.all_records[slot_index][:G * 3].cpu().numpy().view(np.uint32).reshape(-1, 3),
                self.all_lists[slot_index][:V].cpu().numpy().view(np.uint32))
