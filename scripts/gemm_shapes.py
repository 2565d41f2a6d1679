"""Per-shape GEMM time table of one B-pair forward (debug: ROITR_GEMM_SHAPES=1 makes every launch synchronous)."""
import os, sys
os.environ["ROITR_GEMM_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roitr_amd.synthetic import make_pair
from tests.gpu_util import build_model, pair_to_device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = build_model("3DMatch")
pool = [pair_to_device(make_pair(5000, config=2, pair_index=i)) for i in range(B)]
with torch.no_grad():
    model.forward_batch(pool, want_gt=True)
torch.cuda.synchronize()
