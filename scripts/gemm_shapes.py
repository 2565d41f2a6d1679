"""Per-shape GEMM time table of one B-pair forward (debug: ROITR_GEMM_SHAPES=1 makes every launch synchronous).
    python scripts/gemm_shapes.py [pairs] [3DMatch|4DMatch] [f32|bf16]"""
import os, sys
os.environ["ROITR_GEMM_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roitr_amd.synthetic import make_pair
from tests.gpu_util import build_model, pair_to_device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bench = sys.argv[2] if len(sys.argv) > 2 else "3DMatch"
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
fd = bench == "4DMatch"
model = build_model(bench, operand_dtype=dtype, weights="selective")
pool = [pair_to_device(make_pair(8000 if fd else 5000, config=4 if fd else 2, pair_index=i, normals="field")) for i in range(B)]
with torch.no_grad():
    model.forward_batch(pool, want_gt=True)
torch.cuda.synchronize()
