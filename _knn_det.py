"""Repro hunt: is the 3-NN / kNN with tie replay deterministic run to run, also beside a busy second stream?"""
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from roitr_amd import pointops as P
from roitr_amd.synthetic import make_pair
B, N = 64, 8000
pairs = [make_pair(N, config=4, pair_index=i, normals="field") for i in range(B)]
fine = torch.from_numpy(np.concatenate([p["src_points"] for p in pairs] + [p["tgt_points"] for p in pairs])).cuda()
off_f = (torch.arange(1, 2 * B + 1, dtype=torch.int32) * N).cuda()
m = N // 4
idx = P.furthestsampling(fine, off_f, (torch.arange(1, 2 * B + 1, dtype=torch.int32) * m).cuda())
coarse = fine[idx.long()].contiguous()
off_c = (torch.arange(1, 2 * B + 1, dtype=torch.int32) * m).cuda()
side = torch.cuda.Stream()
big = torch.randn((8192, 8192), device="cuda")
for ns, (ref, roff, qry, qoff) in ((3, (coarse, off_c, fine, off_f)), (17, (fine, off_f, coarse, off_c)), (17, (coarse, off_c, coarse, off_c))):
    base = None
    bad = 0
    for it in range(12):
        if it % 2:
            with torch.cuda.stream(side):
                for _ in range(3): big @ big
        i, d = P.knnquery_raw(ns, ref, qry, roff, qoff)
        torch.cuda.synchronize()
        if base is None: base = (i.clone(), d.clone())
        else:
            ne = int((i != base[0]).sum()) + int((d != base[1]).sum())
            bad += ne > 0
            if ne: print("  run", it, "differs in", ne, "entries; rows", torch.nonzero((i != base[0]).any(1)).flatten()[:5].tolist())
    print("nsample", ns, "queries", qry.shape[0], "runs differing from the first:", bad)
