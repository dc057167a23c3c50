"""numpy statement of the shard-slot exchange protocol of toyrenderer_amd/gather.py (test
infrastructure): what the HIP pack / unpack kernels must produce, word for word.  Used by the gloo
(CPU, world_size > 1) tests as the data movers and by the -m gpu tests as the checker."""
import numpy as np

from toyrenderer_amd.gather import HEADER_WORDS, MAX_PASS_SLOTS, ShardExchange, slot_words


def pack_shard_np(local: dict, slot_groups: int) -> np.ndarray:
    """local: pass slot -> (records u32[G,3], masks u32[G], V[, groups_counted]).  Returns the shard slot as
    u32 words.  groups_counted > G means the rank dropped groups at its capacity (header word 9)."""
    out = np.zeros(slot_words(slot_groups), np.uint32)
    start, overflow = 0, 0
    rec_out = out[HEADER_WORDS:HEADER_WORDS + 3 * slot_groups]
    mask_out = out[HEADER_WORDS + 3 * slot_groups:]
    for s in range(MAX_PASS_SLOTS):
        if s not in local:
            continue
        rec, masks, V = local[s][:3]
        rec = np.asarray(rec, np.uint32).reshape(-1, 3)
        g = len(rec)
        if len(local[s]) > 3 and local[s][3] != g:
            out[9] = 1
        if start + g > slot_groups:
            g, overflow = slot_groups - start, 1
        rec_out[3 * start:3 * (start + g)] = rec[:g].reshape(-1)
        mask_out[start:start + g] = np.asarray(masks, np.uint32)[:g]
        out[2 * s], out[2 * s + 1] = g, V
        start += g
    out[8] = overflow
    return out


def expand_masks_np(masks: np.ndarray) -> np.ndarray:
    """(g << 5) | lane for every set bit, groups ascending, lanes ascending."""
    masks = np.asarray(masks, np.uint32)
    bits = np.unpackbits(masks.view(np.uint8).reshape(-1, 4), axis=1, bitorder="little").astype(bool)
    g, lane = np.nonzero(bits)
    return ((g.astype(np.uint32) << np.uint32(5)) | lane.astype(np.uint32)).astype(np.uint32)


def unpack_shards_np(recv: np.ndarray, world: int, slot_groups: int, pass_slots, group_capacity: int) -> dict:
    """recv: world x slot words.  Returns pass slot -> dict(records[G,3], masks[G], list[V], G, V, status)."""
    recv = np.asarray(recv).view(np.uint32).reshape(world, slot_words(slot_groups))
    status = 0
    parts = {s: ([], []) for s in pass_slots}
    total = {s: 0 for s in pass_slots}
    for p in range(world):
        hdr = recv[p, :HEADER_WORDS]
        if hdr[8]:
            status |= 1
        if hdr[9]:
            status |= 8
        rec = recv[p, HEADER_WORDS:HEADER_WORDS + 3 * slot_groups].reshape(-1, 3)
        masks = recv[p, HEADER_WORDS + 3 * slot_groups:]
        in_slot = 0
        for s in range(MAX_PASS_SLOTS):
            g = int(hdr[2 * s]) if s in parts else 0
            src = in_slot
            in_slot += g
            if in_slot > slot_groups:
                status |= 4
                g = 0
            if s not in parts:
                continue
            if total[s] + g > group_capacity:
                g = group_capacity - total[s]
                status |= 2
            parts[s][0].append(rec[src:src + g])
            parts[s][1].append(masks[src:src + g])
            total[s] += g
    out = {}
    for s in pass_slots:
        r = np.concatenate(parts[s][0]) if parts[s][0] else np.zeros((0, 3), np.uint32)
        m = np.concatenate(parts[s][1]) if parts[s][1] else np.zeros(0, np.uint32)
        lst = expand_masks_np(m)
        out[s] = dict(records=r, masks=m, list=lst, G=len(r), V=len(lst), status=status)
    return out


class NumpyShardExchange(ShardExchange):
    """ShardExchange with the numpy movers (CPU tensors, any torch.distributed backend)."""

    def __init__(self, dist, torch, world, rank, slot_groups, pass_slots=(0, 1), **kw):
        super().__init__(dist, torch, world, rank, slot_groups, pass_slots, device="cpu", **kw)
        self.local = {}

    def set_local(self, local: dict):
        self.local = local

    def _pack(self, b):
        self.send[b].copy_(self.torch.from_numpy(pack_shard_np(self.local, self.slot_groups).view(np.int32)))

    def _unpack(self, b):
        res = unpack_shards_np(self.recv[b].numpy(), self.world, self.slot_groups, self.pass_slots, self.group_capacity)
        for s, r in res.items():
            o = self.out[s]
            o["records"][:3 * r["G"]] = self.torch.from_numpy(r["records"].reshape(-1).view(np.int32).copy())
            o["masks"][:r["G"]] = self.torch.from_numpy(r["masks"].view(np.int32).copy())
            n = min(r["V"], self.list_capacity)
            o["list"][:n] = self.torch.from_numpy(r["list"][:n].view(np.int32).copy())
            o["args"].copy_(self.torch.from_numpy(np.array([r["G"], 1, 1, r["G"], r["V"], 1, 1, r["status"]], np.uint32).view(np.int32)))
