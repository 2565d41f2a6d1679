"""`python -m roitr_amd.main <config.yaml> [--pretrain ckpt.pth] [--synthetic N_PAIRS] [--n-points N]`

Test-mode entry mirroring main.py:16-139 of the reference for `mode: test`: load the flattened YAML config,
build the model, load the checkpoint, run the tester.  Under torch.distributed.run every rank takes its share of
the pairs (one process per GPU, RCCL for the final gather)."""
import argparse
import os

import torch

from .config import Config, load_config
from .riga import create_model
from .tester import SyntheticPairs, Tester, load_pretrain


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--pretrain", default=None)
    ap.add_argument("--closed-form-weights", action="store_true",
                    help="ignore any checkpoint and fill the model with the deterministic closed-form weights of roitr_amd/weights.py "
                         "(parity / benchmark runs without a released checkpoint)")
    ap.add_argument("--synthetic", type=int, default=8, help="number of synthetic pairs (no datasets ship with this repo)")
    ap.add_argument("--n-points", type=int, default=5000)
    ap.add_argument("--snapshot-dir", default="snapshot")
    ap.add_argument("--pairs-per-forward", type=int, default=8)
    ap.add_argument("--evaluate", action="store_true", help="report mean PIR / IR (lib/loss.py Evaluator) computed on the device")
    ap.add_argument("--estimate-normals", action="store_true", help="recompute the normals on the GPU (open3d knn=33 + normal_redirect)")
    args = ap.parse_args()
    config = Config(load_config(args.config))
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl")
    model = create_model(config).cuda()
    ckpt = None if args.closed_form_weights else (args.pretrain or config.get("pretrain"))
    if ckpt:
        if not os.path.exists(ckpt):   # lib/trainer.py:128-130 _load_pretrain: raise ValueError('no checkpoint found')
            raise ValueError(f"=> no checkpoint found at '{ckpt}' (pass --closed-form-weights to run without one)")
        load_pretrain(model, ckpt)
    else:
        from .riga import state_dict_layout
        from .weights import closed_form_param
        sd = model.state_dict()
        for k, shape, kind in state_dict_layout(model.factor, model.architecture):
            if kind == "param":
                sd[k].copy_(torch.from_numpy(closed_form_param(k, tuple(shape))))
        print("[roitr_amd] no checkpoint given: using closed-form weights (roitr_amd/weights.py)")
    data = SyntheticPairs(args.synthetic, args.n_points)
    tester = Tester(config, model, data, args.snapshot_dir, args.pairs_per_forward, rank, world, evaluate=args.evaluate,
                    estimate_normals=args.estimate_normals)
    counts = tester.test()
    if rank == 0 and tester.metrics:
        print(f"[roitr_amd] PIR {tester.metrics['PIR']:.4f}  IR {tester.metrics['IR']:.4f}  over {tester.metrics['pairs']} pairs")
    if rank == 0:
        print(f"[roitr_amd] wrote {args.synthetic} result files under {args.snapshot_dir}/{config.benchmark}; "
              f"correspondences per rank: {counts}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
