import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roitr_amd.synthetic import make_pair
from tests.gpu_util import build_model, pair_to_device
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
model = build_model("4DMatch")
pairs = [pair_to_device(make_pair(N, config=4, pair_index=i)) for i in range(B)]
with torch.no_grad():
    for _ in range(3):
        model.forward_batch(pairs)
torch.cuda.synchronize()
