"""Test-mode harness: the loop of lib/tester.py:19-69 (reference) on the MI355X engine.

Per pair it writes `{snapshot_dir}/{benchmark}/{idx}.pth` with exactly the keys the reference's tester saves
(lib/tester.py:56-69) so that registration/evaluate_registration_c2f.py can consume the files unchanged.
Differences by design: pairs are sharded over ranks by their GLOBAL index (pair i -> rank i mod W, the file name
keeps the global index -- the reference's DDP test mode would overwrite files), several
pairs go through the engine per forward (`pairs_per_forward`), and checkpoints load through the same
'module.'-stripping rule as lib/trainer.py:94-130.  At the end of the run ONE collective (shard.gather_result_records:
a gather of the per-pair records {pair id, #correspondences, IR, PIR, match scores}, packed into one fixed-size block per
rank, over RCCL) brings every rank's results to rank 0; `Tester.records` holds them there.  `test()` returns the per-rank
correspondence counts on rank 0 and None on every other rank (only rank 0 receives the records).
"""
import os

import torch

from .shard import assemble_block, gather_result_records, pairs_for_rank, slots_per_rank


def load_pretrain(model, path):
    """lib/trainer.py:94-130 `_load_pretrain`: state['state_dict'], 'module.' prefixes stripped, strict load."""
    state = torch.load(path, map_location="cpu")
    sd = state["state_dict"] if "state_dict" in state else state
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    model.load_state_dict(sd, strict=True)
    return model


class Tester:
    def __init__(self, config, model, dataset, snapshot_dir="snapshot", pairs_per_forward=8, rank=0, world=1, evaluate=False,
                 estimate_normals=False, view_point=(0.0, 0.0, 0.0)):
        """evaluate: also compute PIR / IR per pair on the device (lib/loss.py:169-213 Evaluator) and return their means.
        estimate_normals: ignore the dataset's normals and recompute them on the GPU from the points the way the
        reference's dataset code does (open3d estimate_normals(knn=33) + normal_redirect, dataset/tdmatch.py:120-127)."""
        self.config, self.model, self.dataset = config, model, dataset
        self.snapshot_dir = snapshot_dir
        self.pairs_per_forward = pairs_per_forward
        self.rank, self.world = rank, world
        self.evaluate, self.estimate_normals, self.view_point = evaluate, estimate_normals, view_point
        self.metrics = None
        self.records = None   # rank 0 after test(): shard.GatheredRecords {pair id: match scores}

    def _to_device(self, item, device):
        out = {}
        for k, v in item.items():
            out[k] = v.to(device) if torch.is_tensor(v) else v
        return out

    def test(self, limit=None):
        benchmark = self.config["benchmark"] if isinstance(self.config, dict) else self.config.benchmark
        out_dir = os.path.join(self.snapshot_dir, str(benchmark))
        os.makedirs(out_dir, exist_ok=True)
        n = len(self.dataset) if limit is None else min(limit, len(self.dataset))
        mine = pairs_for_rank(n, self.rank, self.world)
        device = next(self.model.parameters()).device
        self.model.eval()
        def load(s):
            ids = mine[s:s + self.pairs_per_forward]
            items = [self._to_device(self.dataset[i], device) for i in ids]
            if self.estimate_normals:   # one batched call for all clouds of this forward
                from .prep import estimate_normals
                clouds = [it["raw_src_pcd"] for it in items] + [it["tgt_points"] for it in items]
                off = torch.tensor([c.shape[0] for c in clouds], device=device).cumsum(0).to(torch.int32)
                nrm = estimate_normals(torch.cat(clouds).float(), off, 33, self.view_point)
                lo = [0] + off.tolist()
                for k, it in enumerate(items):
                    it["src_normals"] = nrm[lo[k]:lo[k + 1]]
                    it["tgt_normals"] = nrm[lo[len(items) + k]:lo[len(items) + k + 1]]
            pairs = [dict(src_pcd=it["src_points"].contiguous(), tgt_pcd=it["tgt_points"].contiguous(),
                          src_feats=it["src_feats"].contiguous(), tgt_feats=it["tgt_feats"].contiguous(),
                          src_normals=it["src_normals"].contiguous(), tgt_normals=it["tgt_normals"].contiguous(),
                          rot=it["rot"], trans=it["trans"], src_raw_pcd=it["raw_src_pcd"].contiguous()) for it in items]
            return ids, items, pairs, self.model.launch_batch(pairs)

        evaluator = None
        if self.evaluate:
            from .evaluate import Evaluator
            evaluator = Evaluator(self.config)
        blocks = []   # packed result records of every batch (device)
        with torch.no_grad():
            starts = list(range(0, len(mine), self.pairs_per_forward))
            nxt = load(starts[0]) if starts else None
            for k in range(len(starts)):
                ids, items, pairs, handle = nxt
                # the next batch is loaded and enqueued before this one is unpacked and written to disk
                nxt = load(starts[k + 1]) if k + 1 < len(starts) else None
                aux = None
                if evaluator is not None:
                    ir, pir, _, _ = evaluator.evaluate_batch(handle)
                    aux = torch.stack([ir.float(), pir.float()], 1)
                outs = self.model.finish_batch(handle)
                blocks.append(self.model.batch_records(handle, ids, aux))
                for idx, it, p, o in zip(ids, items, pairs, outs):
                    data = dict()  # lib/tester.py:56-69
                    data["src_raw_pcd"] = p["src_raw_pcd"].cpu()
                    data["src_pcd"], data["tgt_pcd"] = p["src_pcd"].cpu(), p["tgt_pcd"].cpu()
                    data["src_nodes"], data["tgt_nodes"] = o["src_nodes"].cpu(), o["tgt_nodes"].cpu()
                    data["src_node_desc"], data["tgt_node_desc"] = o["src_node_feats"].cpu(), o["tgt_node_feats"].cpu()
                    data["src_point_desc"], data["tgt_point_desc"] = o["src_point_feats"].cpu(), o["tgt_point_feats"].cpu()
                    data["src_corr_pts"], data["tgt_corr_pts"] = o["src_corr_points"].cpu(), o["tgt_corr_points"].cpu()
                    data["confidence"] = o["corr_scores"].cpu()
                    data["gt_tgt_node_occ"] = o["gt_tgt_node_occ"].cpu()
                    data["gt_src_node_occ"] = o["gt_src_node_occ"].cpu()
                    data["rot"], data["trans"] = p["rot"].cpu(), p["trans"].cpu()
                    if benchmark in ("4DMatch", "4DLoMatch") and "metric_index" in it:
                        data["metric_index_list"] = it["metric_index"]
                    torch.save(data, os.path.join(out_dir, f"{idx}.pth"))
        # ---- the one collective of the run: every rank's records -> rank 0 (RCCL over xGMI under torch.distributed.run)
        per_pair = self.model.record_scores_per_pair()
        self.records = gather_result_records(assemble_block(blocks, slots_per_rank(n, self.world), per_pair, device),
                                             slots_per_rank(n, self.world), per_pair)
        if self.records is None:     # ranks other than 0
            return None
        if self.records.truncated:
            import warnings
            warnings.warn(f"result records: the score pool of a rank was full, the scores of {len(self.records.truncated)} pair(s) were "
                          f"cut (first: {self.records.truncated[:8]}); raise config key record_scores_per_pair (now {per_pair}). "
                          "The .pth files on disk are complete.")
        counts = [0] * self.world
        for pid, cnt in self.records.n_scores.items():
            counts[pid % self.world] += cnt
        if evaluator is not None:
            # PIR of a pair without coarse correspondences is the mean of an empty tensor = nan in the reference
            # (lib/loss.py:191): such pairs are left out of the PIR mean (and counted) instead of poisoning it or counting as 0
            irs = [a[0] for a in self.records.aux.values()]
            pirs = [a[1] for a in self.records.aux.values() if a[1] == a[1]]
            self.metrics = {"IR": sum(irs) / max(len(irs), 1), "PIR": sum(pirs) / max(len(pirs), 1), "pairs": len(irs),
                            "pairs_without_coarse": len(irs) - len(pirs)}
        return counts


class SyntheticPairs(torch.utils.data.Dataset):
    """Stand-in for dataset/tdmatch.py (no 3DMatch data in this image): seeded synthetic pairs with the same keys."""

    def __init__(self, n_pairs, n_points=5000, config=2):
        self.n_pairs, self.n_points, self.config = n_pairs, n_points, config

    def __len__(self):
        return self.n_pairs

    def __getitem__(self, i):
        from .synthetic import make_pair
        return {k: torch.from_numpy(v) for k, v in make_pair(self.n_points, config=self.config, pair_index=i).items()}
