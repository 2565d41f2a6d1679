"""CPU stand-in for the HIP engine, loaded by bench.py ONLY through ROITR_BENCH_TEST_ENGINE (tests/test_bench_launch_cpu.py).
It provides the three calls benchloop.run_steps makes on the model; a "pair" is a dict with a seed, its result is (seed mod 5)
scores all equal to the seed.  Nothing under roitr_amd/ knows this file."""
import time

import torch

from roitr_amd.shard import pack_records


class StubEngine:
    def __init__(self, rank):
        self.rank = rank

    def launch_batch(self, pairs, want_gt=True):
        return {"pairs": pairs}

    def finish_batch(self, h):
        time.sleep(0.001 * (self.rank + 1))
        res, starts, flat = [], [0], []
        for p in h["pairs"]:
            n = p["seed"] % 5
            sc = torch.full((n,), float(p["seed"]))
            res.append({"corr_scores": sc})
            flat.append(sc)
            starts.append(starts[-1] + n)
        h["starts"], h["flat"] = starts, torch.cat(flat) if flat else torch.zeros(0)
        return res

    def batch_records(self, h, ids, aux=None):
        return pack_records(ids, h["starts"], h["flat"], aux)


def make(rank, world):
    return StubEngine(rank), (lambda ids: [{"seed": int(i)} for i in ids])
