"""Multi-GPU exchange of the visibility results: one process per GPU, instances sharded by
contiguous ranges, per-rank visible lists + amplification records all-gathered over RCCL/xGMI so
that every rank ends the frame with the whole scene's lists in the single-GPU canonical order
(rank-major concatenation).  Not in the reference (single GPU, GraphicRHI.cpp:165).

Per frame and pass slot (early / late):
  1. all_gather_into_tensor of {groups G_r, visible V_r} per rank           (tiny, fixed size)
  2. host reads the counts (one sync) and derives the offsets
  3. own list entries (g << 5 | lane) are rebased by sum_{p<r} G_p          (HIP kernel, in place)
  4. variable-length all-gather of lists (4 B x V_r) and records (12 B x G_r)
The 2-phase structure needs no collective between the phases (late lists stay rank-local).
"""
from __future__ import annotations

import numpy as np

from . import rhi


def exclusive_offsets(counts):
    """counts[R] -> (offsets[R], total)."""
    offs, acc = [], 0
    for c in counts:
        offs.append(acc)
        acc += int(c)
    return offs, acc


def all_gather_varlen(dist, out, local, counts, rank, backend_uneven_ok: bool):
    """Gather `local[:counts[rank]]` of every rank into `out` at the exclusive offsets of `counts`
    (1-D tensors of one dtype).  RCCL: one coalesced uneven all_gather; gloo (CPU tests): broadcasts."""
    offs, total = exclusive_offsets(counts)
    assert out.numel() >= total
    views = [out[offs[p]:offs[p] + int(counts[p])] for p in range(len(counts))]
    if backend_uneven_ok:
        dist.all_gather(views, local[:int(counts[rank])])
    else:
        views[rank].copy_(local[:int(counts[rank])])
        for p in range(len(counts)):
            if int(counts[p]):
                dist.broadcast(views[p], src=p)
    return offs, total


class VisibleListGather:
    """Owns (as torch tensors) the per-rank output buffers handed to the frame driver and the
    gathered whole-scene lists."""

    SLOTS = (0, 1)   # early / late opaque; alpha-mask slots follow the same pattern when present

    def __init__(self, dev: rhi.Device, dist, world: int, rank: int, record_cap: int, list_cap: int, global_record_cap: int | None = None):
        import torch
        self.torch, self.dev, self.dist, self.world, self.rank = torch, dev, dist, world, rank
        self.tensors = {}
        g_rec = global_record_cap if global_record_cap is not None else record_cap * world
        self.all_lists = [torch.empty(g_rec * 32, dtype=torch.int32, device="cuda") for _ in self.SLOTS]
        self.all_records = [torch.empty(g_rec * 3, dtype=torch.int32, device="cuda") for _ in self.SLOTS]
        self.counts_local = torch.zeros(4, dtype=torch.int32, device="cuda")
        self.counts_all = torch.zeros(world * 4, dtype=torch.int32, device="cuda")
        self.cl = dev.create_command_list()
        self.last_counts = None

    def alloc(self, nbytes, name, stride, indirect):
        """FrameDriver hook: records / lists / args of slots 0,1 live in torch tensors (RCCL reads them)."""
        if not any(name.startswith(p) for p in ("MeshletAmplificationDataBuffer", "VisibleMeshletListBuffer",
                                                "MeshletDispatchArgumentsBuffer", "VisibleMeshletDrawArgsBuffer")):
            return None
        t = self.torch.zeros((nbytes + 3) // 4, dtype=self.torch.int32, device="cuda")
        self.tensors[name] = t
        b = self.dev.wrap_buffer(t.data_ptr(), nbytes, name, stride=stride)
        return b

    def run(self, drv):
        torch, dist = self.torch, self.dist
        T = self.tensors
        # 1. per-rank counts {G0, V0, G1, V1}; G = min(X, validRecords)
        for i, s in enumerate(self.SLOTS):
            a = T[f"MeshletDispatchArgumentsBuffer{s}"]
            self.counts_local[2 * i] = torch.minimum(a[0], a[3])
            self.counts_local[2 * i + 1] = T[f"VisibleMeshletDrawArgsBuffer{s}"][0]
        dist.all_gather_into_tensor(self.counts_all, self.counts_local)
        c = self.counts_all.cpu().numpy().reshape(self.world, 4).astype(np.int64)   # 2. one host sync
        self.last_counts = c
        for i, s in enumerate(self.SLOTS):
            G, V = c[:, 2 * i], c[:, 2 * i + 1]
            gbase = int(G[:self.rank].sum())
            if gbase and V[self.rank]:                                                 # 3. rebase own entries
                cl = self.cl
                cl.open()
                cl.dispatch("visibility_CS_RebaseVisibleList",
                            [rhi.PUSH(0), rhi.UAV(0, drv.visibleList[s]), rhi.SRV(0, drv.drawArgs[s])], (1, 1, 1),
                            push=np.array([gbase], np.uint32))
                cl.close()
                self.dev.execute(cl)
            # 4. lists and records
            all_gather_varlen(dist, self.all_lists[i], T[f"VisibleMeshletListBuffer{s}"], V, self.rank, True)
            all_gather_varlen(dist, self.all_records[i], T[f"MeshletAmplificationDataBuffer{s}"], G * 3, self.rank, True)

    def results(self, slot_index: int):
        c = self.last_counts
        G, V = int(c[:, 2 * slot_index].sum()), int(c[:, 2 * slot_index + 1].sum())
        return (self.all_records[slot_index][:G * 3].cpu().numpy().view(np.uint32).reshape(-1, 3),
                self.all_lists[slot_index][:V].cpu().numpy().view(np.uint32))
