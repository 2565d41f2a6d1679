"""Upper bound of cross-forward overlap: two engines (own scratch arenas) on two streams, forwards alternating between them, against
one engine with two batches in flight (what bench.py runs).  Usage: python scripts/overlap_probe.py [pairs_per_step] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roitr_amd import harness, synthetic  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    pool = [harness.pair_to_device(synthetic.make_pair(5000, pair_index=i)) for i in range(B + 64)]
    batch = lambda s: [pool[(s * 7 + j) % len(pool)] for j in range(B)]
    models = [harness.build_model("3DMatch"), harness.build_model("3DMatch")]
    torch.cuda.synchronize()

    def one_engine(n):
        m = models[0]
        h = m.launch_batch(batch(0))
        for s in range(n):
            nxt = m.launch_batch(batch(s + 1)) if s + 1 < n else None
            m.finish_batch(h)
            h = nxt

    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def two_engines(n):
        hs = [None, None]
        for s in range(n + 2):
            k = s % 2
            if hs[k] is not None:
                models[k].finish_batch(hs[k])
                hs[k] = None
            if s < n:
                with torch.cuda.stream(streams[k]):
                    hs[k] = models[k].launch_batch(batch(s))

    for name, fn in (("one engine, two batches in flight", one_engine), ("two engines on two streams", two_engines)):
        fn(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name}: {1e3 * dt / steps:.2f} ms per {B}-pair step, {B * steps / dt:.0f} pairs/s", flush=True)


if __name__ == "__main__":
    main()
