"""numpy statement of the shard-slot exchange protocol of toyrenderer_amd/gather.py (test
infrastructure): what the HIP pack / unpack kernels must produce, word for word.  Used by the gloo
(CPU, world_size > 1) tests as the data movers and by the -m gpu tests as the checker."""
import numpy as np

from toyrenderer_amd.gather import HEADER_WORDS, MAX_PASS_SLOTS, ShardExchange, slot_words


def runs_of_records_np(rec: np.ndarray) -> np.ndarray:
    """u32[E,4] run entries {instance, lod, first group offset, index of the first record}: a record continues a run when
    it repeats instance and lod and its group offset is the previous one + 32 (mod 2^32)."""
    rec = np.asarray(rec, np.uint32).reshape(-1, 3)
    if len(rec) == 0:
        return np.zeros((0, 4), np.uint32)
    cont = (rec[1:, 0] == rec[:-1, 0]) & (rec[1:, 1] == rec[:-1, 1]) & (rec[1:, 2] == rec[:-1, 2] + np.uint32(32))
    first = np.concatenate([[0], np.nonzero(~cont)[0] + 1]).astype(np.uint32)
    return np.concatenate([rec[first], first[:, None]], axis=1).astype(np.uint32)


def records_of_runs_np(runs: np.ndarray, groups: int) -> np.ndarray:
    """Inverse of runs_of_records_np for a pass slot of `groups` records."""
    runs = np.asarray(runs, np.uint32).reshape(-1, 4)
    out = np.zeros((groups, 3), np.uint32)
    for e, (inst, lod, off0, first) in enumerate(runs):
        nxt = int(runs[e + 1][3]) if e + 1 < len(runs) else groups
        nxt, first = min(nxt, groups), int(first)
        n = max(nxt - first, 0)
        out[first:first + n, 0], out[first:first + n, 1] = inst, lod
        out[first:first + n, 2] = (np.uint64(off0) + np.uint64(32) * np.arange(n, dtype=np.uint64)).astype(np.uint32)
    return out


def pack_shard_np(local: dict, slot_groups: int, slot_runs: int | None = None) -> np.ndarray:
    """local: pass slot -> (records u32[G,3], masks u32[G], V[, groups_counted]).  Returns the shard slot as
    u32 words.  groups_counted (header word 2s + 1; default G) is the rank's dispatch counter, which also counts the groups
    of instances dropped at the capacity (Q2): > G means the rank dropped groups (header word 9)."""
    R = slot_groups if slot_runs is None else slot_runs
    out = np.zeros(slot_words(slot_groups, R), np.uint32)
    start, overflow, run_end = 0, 0, 0
    run_out = out[HEADER_WORDS:HEADER_WORDS + 4 * R]
    mask_out = out[HEADER_WORDS + 4 * R:]
    for s in range(MAX_PASS_SLOTS):
        if s in local:
            rec, masks, V = local[s][:3]
            rec = np.asarray(rec, np.uint32).reshape(-1, 3)
            g = len(rec)
            counted = int(local[s][3]) if len(local[s]) > 3 else g
            if counted != g:
                out[9] = 1
            if start + g > slot_groups:
                g, overflow = slot_groups - start, 1
            runs = runs_of_records_np(rec[:g])
            keep = max(min(len(runs), R - run_end), 0)
            run_out[4 * run_end:4 * (run_end + keep)] = runs[:keep].reshape(-1)
            run_end += len(runs)
            mask_out[start:start + g] = np.asarray(masks, np.uint32)[:g]
            out[2 * s], out[2 * s + 1] = g, counted
            start += g
        out[10 + s] = run_end
    out[8] = 1 if overflow or run_end > R else 0
    return out


def expand_masks_np(masks: np.ndarray) -> np.ndarray:
    """(g << 5) | lane for every set bit, groups ascending, lanes ascending."""
    masks = np.asarray(masks, np.uint32)
    bits = np.unpackbits(masks.view(np.uint8).reshape(-1, 4), axis=1, bitorder="little").astype(bool)
    g, lane = np.nonzero(bits)
    return ((g.astype(np.uint32) << np.uint32(5)) | lane.astype(np.uint32)).astype(np.uint32)


def global_cut_np(runs: np.ndarray, sent: int, counted: int, before: int, cap: int):
    """Q2 made global (gather.py): this rank's groups start at `before` in the rank-major whole-scene order, it counted
    `counted` groups and sent the first `sent` (its locally valid ones).  Returns (groups of this rank that lie in front of
    the first instance the single-GPU pass would drop at `cap`, whether that instance is this rank's)."""
    if before + counted < cap:
        return sent, False
    j = cap - before - 1                              # the record the dropped instance's run contains: off <= j < off + g, off + g >= cap - before
    if j >= sent:
        return sent, True                             # it is the instance the rank dropped itself (or lies behind it): everything sent is valid
    first = np.asarray(runs, np.uint32).reshape(-1, 4)[:, 3].astype(np.int64)
    return int(first[np.searchsorted(first, j, side="right") - 1]), True


def unpack_shards_np(recv: np.ndarray, world: int, slot_groups: int, pass_slots, group_capacity: int, slot_runs: int | None = None,
                     global_cap: int | None = None) -> dict:
    """recv: world x slot words.  Returns pass slot -> dict(records[G,3], masks[G], list[V], G, V, X, status).  global_cap:
    the group capacity of the single-GPU run this exchange reproduces (Q2 made global); None / 0: a rank that dropped groups
    raises status bit 8 instead."""
    R = slot_groups if slot_runs is None else slot_runs
    recv = np.asarray(recv).view(np.uint32).reshape(world, slot_words(slot_groups, R))
    status = 0
    parts = {s: ([], []) for s in pass_slots}
    total = {s: 0 for s in pass_slots}
    counted_before = {s: 0 for s in pass_slots}
    cut_done = {s: False for s in pass_slots}
    for p in range(world):
        hdr = recv[p, :HEADER_WORDS]
        if hdr[8]:
            status |= 1
        if hdr[9] and not global_cap:
            status |= 8
        runs = recv[p, HEADER_WORDS:HEADER_WORDS + 4 * R].reshape(-1, 4)
        masks = recv[p, HEADER_WORDS + 4 * R:]
        in_slot, run_begin, bad = 0, 0, False
        for s in range(MAX_PASS_SLOTS):
            sent = int(hdr[2 * s])
            g = sent if s in parts else 0
            src = in_slot
            in_slot += sent
            run_end = int(hdr[10 + s])
            if in_slot > slot_groups or run_end < run_begin or run_end > R:
                bad = True
            if bad:
                status |= 4
                g, run_end = 0, run_begin
            if s in parts:
                full = g
                if global_cap and not bad:
                    if cut_done[s]:
                        g = 0                         # behind the first dropped instance: undefined in the single-GPU run
                    else:
                        g, cut_done[s] = global_cut_np(runs[run_begin:run_end], g, int(hdr[2 * s + 1]), counted_before[s], int(global_cap))
                    counted_before[s] += int(hdr[2 * s + 1])
                if total[s] + g > group_capacity:
                    g = group_capacity - total[s]
                    status |= 2
                parts[s][0].append(records_of_runs_np(runs[run_begin:run_end], full)[:g])
                parts[s][1].append(masks[src:src + g])
                total[s] += g
            run_begin = run_end
    out = {}
    for s in pass_slots:
        r = np.concatenate(parts[s][0]) if parts[s][0] else np.zeros((0, 3), np.uint32)
        m = np.concatenate(parts[s][1]) if parts[s][1] else np.zeros(0, np.uint32)
        lst = expand_masks_np(m)
        out[s] = dict(records=r, masks=m, list=lst, G=len(r), V=len(lst), X=counted_before[s] if global_cap else len(r), status=status)
    return out


class NumpyShardExchange(ShardExchange):
    """ShardExchange with the numpy movers (CPU tensors, any torch.distributed backend)."""

    def __init__(self, dist, torch, world, rank, slot_groups, pass_slots=(0, 1), **kw):
        super().__init__(dist, torch, world, rank, slot_groups, pass_slots, device="cpu", **kw)
        self.local = {}

    def set_local(self, local: dict):
        self.local = local

    def _pack(self, b):
        self.send[b].copy_(self.torch.from_numpy(pack_shard_np(self.local, self.slot_groups, self.slot_runs).view(np.int32)))

    def _unpack(self, b):
        res = unpack_shards_np(self.recv[b].numpy(), self.world, self.slot_groups, self.pass_slots, self.group_capacity, self.slot_runs,
                               global_cap=self.global_group_cap)
        for s, r in res.items():
            o = self.out[s]
            o["records"][:3 * r["G"]] = self.torch.from_numpy(r["records"].reshape(-1).view(np.int32).copy())
            o["masks"][:r["G"]] = self.torch.from_numpy(r["masks"].view(np.int32).copy())
            n = min(r["V"], self.list_capacity)
            o["list"][:n] = self.torch.from_numpy(r["list"][:n].view(np.int32).copy())
            o["args"].copy_(self.torch.from_numpy(np.array([r["X"], 1, 1, r["G"], r["V"], 1, 1, r["status"]], np.uint32).view(np.int32)))
