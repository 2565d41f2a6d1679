"""One pair per call, two calls in flight: the per-forward weight-signature walk on / off (RIGA_v2.weights_frozen)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roitr_amd.synthetic import make_pair
from roitr_amd.harness import build_model, pair_to_device
model = build_model("3DMatch", weights="selective")
pool = [pair_to_device(make_pair(5000, config=2, pair_index=i, normals="field")) for i in range(16)]
torch.cuda.synchronize()
model.inputs_resident = True


def loop(n):
    h = model.launch_batch([pool[0]])
    for s in range(n):
        nx = model.launch_batch([pool[(s + 1) % 16]]) if s + 1 < n else None
        model.finish_batch(h); h = nx
    torch.cuda.synchronize()


with torch.no_grad():
    loop(50)
    for rep in range(3):
        for frozen in (False, True):
            model.weights_frozen = frozen
            t0 = time.perf_counter(); loop(300); dt = time.perf_counter() - t0
            print(f"weights_frozen={frozen}: {1e3 * dt / 300:.3f} ms per pair")
