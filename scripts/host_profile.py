"""Where the HOST time of a one-pair-per-call step goes (cProfile over the pipelined loop): python scripts/host_profile.py [pairs]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roitr_amd.synthetic import make_pair
from roitr_amd.harness import build_model, pair_to_device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model = build_model("3DMatch", weights="selective")
pool = [pair_to_device(make_pair(5000, config=2, pair_index=i, normals="field")) for i in range(max(16, 2 * B))]
torch.cuda.synchronize()
model.inputs_resident = True
batch = lambda s: [pool[(s * B + j) % len(pool)] for j in range(B)]


def loop(n):
    h = model.launch_batch(batch(0))
    for s in range(n):
        nx = model.launch_batch(batch(s + 1)) if s + 1 < n else None
        model.finish_batch(h)
        h = nx
    torch.cuda.synchronize()


with torch.no_grad():
    loop(30)
    t0 = time.perf_counter(); loop(200); dt = time.perf_counter() - t0
    print(f"B={B}: {1e3 * dt / 200:.3f} ms per step")
    pr = cProfile.Profile(); pr.enable(); loop(200); pr.disable()
    st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
