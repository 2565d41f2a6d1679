"""Pair sharding over the GPUs of one node.

Every pair is independent (the reference runs batch size 1 per forward, main.py:122-127, no cross-pair
state, read-only weights), so the partition is the one a DistributedSampler would make (main.py:106):
pair i -> rank i mod W.  No data-path collective exists; the only exchange is ONE gather of the per-pair
result records (match scores) to rank 0 over RCCL/xGMI at the end of a run.

Record layout (int32 words, fixed size so that no size pre-exchange is needed):
    [pair_id, n_scores, aux0, aux1 | score bits ... padded to max_scores]
`pair_id` = -1 marks an unused slot, `n_scores` is the TRUE number of correspondences of the pair (a value above
`max_scores` means the tail was cut: with the reference's mutual top-k fine matching `num_corr * 64 * k` is an exact
upper bound, so 3DMatch records are never cut), aux0/aux1 carry two fp32 values (the tester puts IR / PIR there),
scores travel as their fp32 bit patterns.  Every rank contributes `slots` records; slots = ceil(n_pairs / world) is
known on every rank from the pair count alone.
"""
import torch
import torch.distributed as dist

HEADER = 4


def pairs_for_rank(n_pairs, rank, world):
    """Indices of the pairs rank `rank` of `world` processes."""
    return list(range(rank, n_pairs, world))


def slots_per_rank(n_pairs, world):
    return (n_pairs + world - 1) // world


def max_scores_per_pair(num_corr, point_limit, fine_topk, mutual=True):
    """Upper bound of correspondences one pair can emit (modules.py:259-266: row top-k AND/OR column top-k per patch)."""
    return int(num_corr) * int(point_limit) * int(fine_topk) * (1 if mutual else 2)


def _group_device():
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def gather_counts(value):
    """All ranks contribute one integer; every rank gets the list (kept for callers that only need a count)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [int(value)]
    world = dist.get_world_size()
    t = torch.tensor([int(value)], dtype=torch.int64, device=_group_device())
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]


def empty_records(slots, max_scores, device):
    buf = torch.zeros((slots, HEADER + max_scores), dtype=torch.int32, device=device)
    buf[:, 0] = -1
    return buf


def pack_records(pair_ids, starts, scores_flat, max_scores, aux=None):
    """Records of one engine batch, built on the device without a host round trip per pair.

    pair_ids: B global pair indices; starts: B+1 row offsets into `scores_flat` (the engine's fine_offsets + n_out, as
    finish_batch() already holds them on the host); scores_flat: the engine's out_scores; aux: optional (B,2) float tensor."""
    dev = scores_flat.device
    B = len(pair_ids)
    buf = empty_records(B, max_scores, dev)
    if B == 0:
        return buf
    st = torch.as_tensor(list(starts), dtype=torch.int64, device=dev)
    buf[:, 0] = torch.as_tensor(list(pair_ids), dtype=torch.int32, device=dev)
    buf[:, 1] = (st[1:] - st[:-1]).to(torch.int32)
    if aux is not None:
        buf[:, 2:4] = aux.to(device=dev, dtype=torch.float32).contiguous().view(torch.int32)
    lo, hi = int(starts[0]), int(starts[-1])
    if hi > lo:
        r = torch.arange(lo, hi, device=dev)
        pair = torch.searchsorted(st[1:], r, right=True)
        col = r - st[pair]
        keep = col < max_scores
        bits = scores_flat[lo:hi].contiguous().view(torch.int32)
        buf[pair[keep], HEADER + col[keep]] = bits[keep]
    return buf


def records_from_list(records, max_scores, device=None):
    """[(pair_id, 1-D float tensor)] or [(pair_id, tensor, (aux0, aux1))] -> packed record buffer."""
    if device is None:
        device = records[0][1].device if records else "cpu"
    buf = empty_records(len(records), max_scores, device)
    for i, rec in enumerate(records):
        pid, s = rec[0], rec[1]
        n = int(s.numel())
        buf[i, 0] = int(pid)
        buf[i, 1] = n
        if len(rec) > 2:
            buf[i, 2:4] = torch.tensor(list(rec[2]), dtype=torch.float32, device=device).view(torch.int32)
        k = min(n, max_scores)
        if k:
            buf[i, HEADER:HEADER + k] = s.detach().to(device, torch.float32).contiguous().view(torch.int32)[:k]
    return buf


class GatheredRecords:
    """Rank 0's view of the gathered record blocks: a read-only mapping {pair_id: scores (1-D float32 cpu tensor)}.
    Only the record headers are copied to the host eagerly; a pair's scores leave the device when they are read.
    .aux {pair_id: (aux0, aux1)}, .n_scores {pair_id: true count}, .truncated [pair ids whose tail was cut],
    .ranks_seen (ranks that contributed at least one record), .backend ('nccl' = RCCL, 'gloo', 'local' = no process group)."""

    def __init__(self, blocks, max_scores, backend):
        self.blocks, self.max_scores, self.backend = blocks, max_scores, backend
        self.aux, self.n_scores, self.truncated, self._where = {}, {}, [], {}
        heads = torch.stack([b[:, :HEADER] for b in blocks]).cpu() if blocks and blocks[0].shape[0] else torch.zeros((len(blocks), 0, HEADER), dtype=torch.int32)
        auxf = heads[:, :, 2:4].contiguous().view(torch.float32)
        seen = set()
        for r in range(heads.shape[0]):
            ids, ns = heads[r, :, 0].tolist(), heads[r, :, 1].tolist()
            ax = auxf[r].tolist()
            for i, pid in enumerate(ids):
                if pid < 0:
                    continue
                seen.add(r)
                self._where[pid] = (r, i)
                self.n_scores[pid] = ns[i]
                self.aux[pid] = (ax[i][0], ax[i][1])
                if ns[i] > max_scores:
                    self.truncated.append(pid)
        self.ranks_seen = len(seen)

    def __len__(self):
        return len(self._where)

    def __contains__(self, pid):
        return pid in self._where

    def __iter__(self):
        return iter(self._where)

    def keys(self):
        return self._where.keys()

    def __getitem__(self, pid):
        r, i = self._where[pid]
        k = min(self.n_scores[pid], self.max_scores)
        return self.blocks[r][i, HEADER:HEADER + k].cpu().view(torch.float32)

    def items(self):
        return [(pid, self[pid]) for pid in self._where]


def gather_result_records(records, slots, max_scores):
    """THE collective of the path: every rank sends its (slots, HEADER + max_scores) int32 record block to rank 0 in ONE
    `gather` (RCCL over xGMI with backend 'nccl', gloo in the CPU tests; nothing is exchanged without a process group).

    records: a packed buffer (pack_records / records_from_list; fewer than `slots` rows are padded with empty slots) or a
    list accepted by records_from_list.  Returns a GatheredRecords on rank 0, None elsewhere."""
    distributed = dist.is_available() and dist.is_initialized()
    dev = _group_device() if distributed else None
    if not torch.is_tensor(records):
        records = records_from_list(records, max_scores, dev)
    if records.shape[0] > slots:
        raise ValueError(f"{records.shape[0]} records for {slots} slots")
    if records.shape[1] != HEADER + max_scores:
        raise ValueError("record width does not match max_scores")
    if distributed and records.device.type != torch.device(dev).type:
        records = records.to(dev)
    if records.shape[0] < slots:
        records = torch.cat([records, empty_records(slots - records.shape[0], max_scores, records.device)], 0)
    records = records.contiguous()
    if not distributed:
        return GatheredRecords([records], max_scores, "local")
    world, rank = dist.get_world_size(), dist.get_rank()
    blocks = [torch.empty_like(records) for _ in range(world)] if rank == 0 else None
    dist.gather(records, blocks, dst=0)
    if rank != 0:
        return None
    return GatheredRecords(blocks, max_scores, dist.get_backend())
