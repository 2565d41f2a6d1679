"""Correspondence-quality evaluators on the GPU.

`Evaluator` mirrors lib/loss.py:169-213 (same config keys eval_acceptance_overlap / eval_acceptance_radius, same
evaluate_coarse / evaluate_fine / forward -> {'PIR', 'IR'}); `get_inlier_ratio_correspondence` mirrors
registration/benchmark_utils.py:69-77.  `evaluate_batch` does the same for all pairs of an engine batch in two launches.
"""
import torch

from . import _lib as L


def _cfg(cfg, k, d=None):
    return cfg.get(k, d) if isinstance(cfg, dict) else getattr(cfg, k, d)


def _inlier_counts(starts, src_pts, tgt_pts, rot, trans, radius):
    pairs = int(starts.shape[0]) - 1
    out = torch.empty((pairs,), dtype=torch.int32, device=src_pts.device)
    L.check(L.lib().roitr_inlier_counts(pairs, L.ptr(starts), L.ptr(src_pts), L.ptr(tgt_pts), L.ptr(rot), L.ptr(trans),
                                        L.c_float(radius), L.ptr(out), L.stream_ptr()), "inlier_counts")
    return out


def get_inlier_ratio_correspondence(src_node, tgt_node, rot, trans, inlier_distance_threshold=0.1):
    """registration/benchmark_utils.py:69-77 (inliers / number of correspondences)."""
    n = int(src_node.shape[0])
    dev = src_node.device
    starts = torch.tensor([0, n], dtype=torch.int32, device=dev)
    cnt = _inlier_counts(starts, src_node.contiguous().float(), tgt_node.contiguous().float(), rot.reshape(1, 3, 3).contiguous().float(),
                         trans.reshape(1, 3).contiguous().float(), inlier_distance_threshold)
    return cnt[0].float() / n


class Evaluator(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.acceptance_overlap = float(_cfg(cfg, "eval_acceptance_overlap", 0.0))
        self.acceptance_radius = float(_cfg(cfg, "eval_acceptance_radius", 0.1))

    @torch.no_grad()
    def evaluate_coarse(self, output_dict):
        """lib/loss.py:175-193: fraction of predicted node pairs that are ground-truth pairs with enough overlap."""
        tgt_idx = output_dict["tgt_node_corr_indices"].to(torch.int32).contiguous().reshape(1, -1)
        src_idx = output_dict["src_node_corr_indices"].to(torch.int32).contiguous().reshape(1, -1)
        n = int(tgt_idx.shape[1])
        dev = tgt_idx.device
        gt_idx = output_dict["gt_node_corr_indices"].to(torch.int32).contiguous().reshape(1, -1, 2)
        gt_ov = output_dict["gt_node_corr_overlaps"].contiguous().float().reshape(1, -1)
        ng = int(gt_idx.shape[1])
        hits = torch.zeros((1,), dtype=torch.int32, device=dev)
        if n > 0 and ng > 0:
            n_corr = torch.tensor([n], dtype=torch.int32, device=dev)
            gt_cnt = torch.tensor([ng], dtype=torch.int32, device=dev)
            L.check(L.lib().roitr_coarse_hits(1, n, L.ptr(n_corr), L.ptr(tgt_idx), L.ptr(src_idx), ng, L.ptr(gt_idx), L.ptr(gt_ov),
                                              L.ptr(gt_cnt), L.c_float(self.acceptance_overlap), L.ptr(hits), L.stream_ptr()), "coarse_hits")
        return hits[0].float() / n if n > 0 else torch.tensor(float("nan"), device=dev)   # mean of an empty tensor is nan in the reference

    @torch.no_grad()
    def evaluate_fine(self, output_dict, data_dict):
        """lib/loss.py:195-206."""
        rot, trans = data_dict["rot"], data_dict["trans"]
        if rot.dim() == 3:
            rot, trans = rot[0], trans[0]
        src = output_dict["src_corr_points"]
        if src.shape[0] == 0:
            return 0.0
        return get_inlier_ratio_correspondence(src, output_dict["tgt_corr_points"], rot, trans, self.acceptance_radius)

    def forward(self, output_dict, data_dict):
        return {"PIR": self.evaluate_coarse(output_dict), "IR": self.evaluate_fine(output_dict, data_dict)}

    @torch.no_grad()
    def evaluate_batch(self, handle):
        """IR / PIR of every pair of a RIGA_v2.launch_batch() handle (needs rot/trans): two launches for the whole batch.
        Returns (ir (B,), pir (B,), n_corr_fine (B,), n_corr_coarse (B,)) as device tensors; empty sets give IR 0 / PIR nan."""
        out, B, P = handle["out"], handle["B"], handle["P"]
        dev = out["n_out"].device
        if not handle["have_gt"]:
            raise L.RoitrError("evaluate_batch needs ground-truth transforms (rot / trans) in the pairs")
        rot, trans = handle["keep"][4], handle["keep"][5]
        starts = torch.cat([out["fine_offsets"].view(B, P)[:, 0], out["n_out"]]).contiguous()
        inl = _inlier_counts(starts, out["out_src_pts"], out["out_tgt_pts"], rot, trans, self.acceptance_radius)
        n_fine = (starts[1:] - starts[:-1])
        hits = torch.empty((B,), dtype=torch.int32, device=dev)
        cap = int(out["gt_corr_idx"].shape[1])
        L.check(L.lib().roitr_coarse_hits(B, P, L.ptr(out["n_corr"]), L.ptr(out["tgt_corr"]), L.ptr(out["src_corr"]), cap,
                                          L.ptr(out["gt_corr_idx"]), L.ptr(out["gt_corr_overlaps"]), L.ptr(out["gt_corr_count"]),
                                          L.c_float(self.acceptance_overlap), L.ptr(hits), L.stream_ptr()), "coarse_hits")
        ir = torch.where(n_fine > 0, inl.float() / n_fine.clamp_min(1).float(), torch.zeros_like(inl, dtype=torch.float32))
        pir = hits.float() / out["n_corr"].float()
        return ir, pir, n_fine, out["n_corr"]
