"""Pair sharding over the GPUs of one node.

Every pair is independent (the reference runs batch size 1 per forward, main.py:122-127, no cross-pair
state, read-only weights), so the partition is the one a DistributedSampler would make (main.py:106):
pair i -> rank i mod W.  No data-path collective exists; the only exchange is ONE gather of the per-pair
result records (match scores) to rank 0 over RCCL/xGMI at the end of a run.

Block layout (one per rank, int32 words, fixed size so that no size pre-exchange is needed):
    [slots x (pair_id, n_scores, aux0, aux1)] [pool: slots * scores_per_pair score words]
The headers are one per slot; `pair_id` = -1 marks an unused slot, `n_scores` is the TRUE number of correspondences of the
pair, aux0 / aux1 carry two fp32 values (the tester puts IR / PIR there).  The scores of the used slots lie back to back in
the pool in slot order as fp32 bit patterns -- a pair may use more than `scores_per_pair` words as long as the rank's total
fits (`scores_per_pair` is the AVERAGE capacity: the exact upper bound of the mutual top-k fine matching, num_corr * 64 * k =
49 152 for the 3DMatch settings, is 100 MB per 512 pairs, three orders of magnitude above what travels).  When the pool is
full the tail is cut: the receiver sees it from the headers (sum of n_scores > pool) and lists the cut pairs in `.truncated`.
Default: 4092 scores per pair on average = 16 KB per slot, 8 MB per 512 pairs.
"""
import torch
import torch.distributed as dist

HEADER = 4
DEFAULT_SCORES_PER_PAIR = 4092


def pairs_for_rank(n_pairs, rank, world):
    """Indices of the pairs rank `rank` of `world` processes."""
    return list(range(rank, n_pairs, world))


def slots_per_rank(n_pairs, world):
    return (n_pairs + world - 1) // world


def max_scores_per_pair(num_corr, point_limit, fine_topk, mutual=True):
    """Upper bound of correspondences one pair can emit (modules.py:259-266: row top-k AND/OR column top-k per patch)."""
    return int(num_corr) * int(point_limit) * int(fine_topk) * (1 if mutual else 2)


def block_words(slots, scores_per_pair):
    return int(slots) * (HEADER + int(scores_per_pair))


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


class RecordBatch:
    """The records of one engine batch before they are laid into the rank's block: `headers` (B, 4) int32 ON THE HOST (pair id,
    true score count, 0, 0), `aux` an optional (B, 2) float tensor (device or host) for the two aux words, and the batch's scores
    (n,) float32, pair after pair, on the device the engine wrote them to.  Nothing here waits for the device."""

    def __init__(self, headers, scores, aux=None):
        self.headers, self.scores, self.aux = headers, scores, aux

    def __len__(self):
        return int(self.headers.shape[0])


def pack_records(pair_ids, starts, scores_flat, aux=None):
    """Records of one engine batch.  No per-pair work and no host <-> device traffic: the engine already emits the scores of a
    batch pair after pair, so the pool part is ONE contiguous slice of its `out_scores` (copied on the stream, so that the 100 MB
    output buffer is not kept alive), and the header is built on the host from numbers finish_batch() holds there anyway -- it
    travels to the device with the block, at the end.  (A host-to-device copy here would queue behind the NEXT batch's forward,
    which is already on the stream, and stall the host for its whole duration.)

    pair_ids: B global pair indices; starts: B+1 row offsets into `scores_flat` (the engine's fine_offsets + n_out);
    aux: optional (B, 2) float tensor."""
    B = len(pair_ids)
    head = torch.zeros((B, HEADER), dtype=torch.int32)
    if B:
        st = torch.as_tensor(list(starts), dtype=torch.int64)
        head[:, 0] = torch.as_tensor(list(pair_ids), dtype=torch.int32)
        head[:, 1] = (st[1:] - st[:-1]).to(torch.int32)
    lo, hi = (int(starts[0]), int(starts[-1])) if B else (0, 0)
    return RecordBatch(head, scores_flat[lo:hi].to(torch.float32).clone(), aux if B else None)


def records_from_list(records):
    """[(pair_id, 1-D float tensor)] or [(pair_id, tensor, (aux0, aux1))] -> RecordBatch (host-side helper of the tests)."""
    device = records[0][1].device if records else "cpu"
    head = torch.zeros((len(records), HEADER), dtype=torch.int32)
    for i, rec in enumerate(records):
        head[i, 0], head[i, 1] = int(rec[0]), int(rec[1].numel())
        if len(rec) > 2:
            head[i, 2:4] = torch.tensor(list(rec[2]), dtype=torch.float32).view(torch.int32)
    scores = torch.cat([r[1].detach().to(device, torch.float32).reshape(-1) for r in records]) if records else torch.zeros(0)
    return RecordBatch(head, scores)


def assemble_block(batches, slots, scores_per_pair=DEFAULT_SCORES_PER_PAIR, device=None):
    """Lay the RecordBatches of a rank (in order) into its fixed-size block; unused slots get pair_id -1."""
    batches = list(batches)
    if device is None:
        device = batches[0].scores.device if batches else "cpu"
    n = sum(len(b) for b in batches)
    if n > slots:
        raise ValueError(f"{n} records for {slots} slots")
    pool = int(slots) * int(scores_per_pair)
    block = torch.zeros(block_words(slots, scores_per_pair), dtype=torch.int32, device=device)
    heads = block[:slots * HEADER].view(slots, HEADER)
    heads[n:, 0] = -1
    if n:
        heads[:n] = torch.cat([b.headers for b in batches], 0).to(device)
        row = 0
        for b in batches:
            if b.aux is not None:
                heads[row:row + len(b), 2:4] = b.aux.to(device=device, dtype=torch.float32).contiguous().view(torch.int32)
            row += len(b)
        sc = torch.cat([b.scores.to(device) for b in batches])
        k = min(int(sc.numel()), pool)
        block[slots * HEADER:slots * HEADER + k] = sc[:k].contiguous().view(torch.int32)
    return block


class GatheredRecords:
    """Rank 0's view of the gathered blocks: a read-only mapping {pair_id: scores (1-D float32 cpu tensor)}.
    Only the headers are copied to the host eagerly; a pair's scores leave the device when they are read.
    .aux {pair_id: (aux0, aux1)}, .n_scores {pair_id: true count}, .truncated [pair ids whose scores were cut by a full
    pool], .ranks_seen (ranks that contributed at least one record), .backend ('nccl' = RCCL, 'gloo', 'local' = no group)."""

    def __init__(self, blocks, slots, scores_per_pair, backend):
        self.blocks, self.slots, self.scores_per_pair, self.backend = blocks, int(slots), int(scores_per_pair), backend
        self.aux, self.n_scores, self.truncated, self._where = {}, {}, [], {}
        pool = self.slots * self.scores_per_pair
        heads = (torch.stack([b[:self.slots * HEADER].view(self.slots, HEADER) for b in blocks]).cpu() if blocks and self.slots
                 else torch.zeros((len(blocks), 0, HEADER), dtype=torch.int32))
        auxf = heads[:, :, 2:4].contiguous().view(torch.float32)
        seen = set()
        for r in range(heads.shape[0]):
            ids, ns = heads[r, :, 0].tolist(), heads[r, :, 1].tolist()
            ax = auxf[r].tolist()
            cur = 0
            for i, pid in enumerate(ids):
                if pid < 0:
                    continue
                seen.add(r)
                have = max(0, min(ns[i], pool - cur))
                self._where[pid] = (r, cur, have)
                self.n_scores[pid] = ns[i]
                self.aux[pid] = (ax[i][0], ax[i][1])
                if have < ns[i]:
                    self.truncated.append(pid)
                cur += ns[i]
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
        r, off, k = self._where[pid]
        base = self.slots * HEADER + off
        return self.blocks[r][base:base + k].cpu().view(torch.float32)

    def items(self):
        return [(pid, self[pid]) for pid in self._where]


def gather_result_records(records, slots, scores_per_pair=DEFAULT_SCORES_PER_PAIR):
    """THE collective of the path: every rank sends its block (block_words(slots, scores_per_pair) int32) to rank 0 in ONE
    `gather` (RCCL over xGMI with backend 'nccl', gloo in the CPU tests; nothing is exchanged without a process group).

    records: an assembled block, a RecordBatch, a list of RecordBatches (laid out in order) or a list accepted by
    records_from_list.  Returns a GatheredRecords on rank 0, None elsewhere."""
    distributed = dist.is_available() and dist.is_initialized()
    dev = _group_device() if distributed else None
    if isinstance(records, RecordBatch):
        records = [records]
    if not torch.is_tensor(records):
        records = list(records)
        if records and not isinstance(records[0], RecordBatch):
            records = [records_from_list(records)]
        records = assemble_block(records, slots, scores_per_pair, dev)
    if records.numel() != block_words(slots, scores_per_pair):
        raise ValueError("block size does not match (slots, scores_per_pair)")
    if distributed and records.device.type != torch.device(dev).type:
        records = records.to(dev)
    records = records.contiguous()
    if not distributed:
        return GatheredRecords([records], slots, scores_per_pair, "local")
    world, rank = dist.get_world_size(), dist.get_rank()
    blocks = [torch.empty_like(records) for _ in range(world)] if rank == 0 else None
    dist.gather(records, blocks, dst=0)
    if rank != 0:
        return None
    return GatheredRecords(blocks, slots, scores_per_pair, dist.get_backend())
