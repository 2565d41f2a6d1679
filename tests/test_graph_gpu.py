"""The forward replayed as a HIP graph (roitr_engine_forward_graph) must be bit-identical to the launch-by-launch forward,
for repeated shapes (warm-up -> capture -> replay), changing shapes, and across the persistent buffer ring."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KEYS = ("src_point_feats", "tgt_point_feats", "src_node_feats", "tgt_node_feats", "corr_scores", "src_corr_points", "tgt_corr_points",
        "matching_scores", "gt_src_node_occ", "gt_node_corr_overlaps")


def _same(a, b):
    for k in KEYS:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["src_node_corr_indices"], b["src_node_corr_indices"])


def test_graph_forward_equals_plain_forward():
    from roitr_amd.synthetic import make_pair
    from tests.gpu_util import build_model, pair_to_device
    model = build_model("3DMatch")
    pairs = [pair_to_device(make_pair(1024, config=1, pair_index=i)) for i in range(6)]
    other = [pair_to_device(make_pair(1500, config=1, pair_index=10 + i)) for i in range(2)]
    with torch.no_grad():
        ref = [model.forward_batch([p])[0] for p in pairs]
        ref_o = [model.forward_batch([p])[0] for p in other]
        ref2 = model.forward_batch(pairs[:2])
        assert model.graph_count() == 0
        # one pair per call: call 1 = warm-up, call 2 = capture, then replays; results copied out before the ring wraps
        for i, p in enumerate(pairs):
            got = model.forward_batch([p], graph=True)[0]
            _same(got, ref[i])
        assert model.graph_count() >= 1
        # a different shape in between, then the first shape again (its graphs are still valid)
        for i, p in enumerate(other):
            _same(model.forward_batch([p], graph=True)[0], ref_o[i])
        _same(model.forward_batch([pairs[3]], graph=True)[0], ref[3])
        # a two-pair batch through the graph path, three times (warm-up, capture, replay)
        for _ in range(4):
            got2 = model.forward_batch(pairs[:2], graph=True)
            _same(got2[0], ref2[0]); _same(got2[1], ref2[1])
        # two launches in flight
        h0 = model.launch_batch([pairs[4]], graph=True)
        h1 = model.launch_batch([pairs[5]], graph=True)
        _same(model.finish_batch(h0)[0], ref[4])
        _same(model.finish_batch(h1)[0], ref[5])


def test_forward_is_deterministic_and_batch_invariant():
    """Same pairs twice, and inside differently composed batches: every output bit-identical (atomics only hand out
    slots whose order never reaches the results; every kernel is row-local)."""
    from roitr_amd.synthetic import make_pair
    from tests.gpu_util import build_model, pair_to_device
    model = build_model("3DMatch")
    pairs = [pair_to_device(make_pair(n, config=2, pair_index=i)) for i, n in enumerate((2000, 1024, 3000, 1500))]
    with torch.no_grad():
        a = model.forward_batch(pairs)
        b = model.forward_batch(pairs)
        c = model.forward_batch([pairs[2], pairs[0]])
    for x, y in zip(a, b):
        _same(x, y)
    _same(c[0], a[2]); _same(c[1], a[0])


def test_batches_in_flight_do_not_interfere():
    """Two forwards enqueued back to back (what the tester and bench.py do) share the engine's scratch arena, and each of them
    runs its geometry chain -- FPS, grids, kNN, the embedding E, partition, ground-truth outputs -- on the engine's side stream
    beside its feature path: the second call's side stream must not touch the arena before the first call is through, and every
    side-stream product must be joined before it is read.  Results of calls that overlap other calls, in both orders and with
    ground-truth outputs on and off, are bit-identical to the same pairs run alone."""
    from roitr_amd.synthetic import make_pair
    from tests.gpu_util import build_model, pair_to_device
    model = build_model("3DMatch")
    small = [pair_to_device(make_pair(n, config=2, pair_index=i)) for i, n in enumerate((1024, 1500))]
    big = [pair_to_device(make_pair(n, config=2, pair_index=10 + i)) for i, n in enumerate((5000, 4000, 3000, 5000, 2048, 4500))]
    with torch.no_grad():
        ref_small = model.forward_batch(small)
        torch.cuda.synchronize()
        ref_big = model.forward_batch(big)
        torch.cuda.synchronize()
        for rounds in range(3):
            hs = [model.launch_batch(big), model.launch_batch(small), model.launch_batch(big, want_gt=False), model.launch_batch(small)]
            got = [model.finish_batch(h) for h in hs]
            for x, y in zip(got[0], ref_big):
                _same(x, y)
            for x, y in zip(got[1], ref_small):
                _same(x, y)
            for x, y in zip(got[3], ref_small):
                _same(x, y)
            for x, y in zip(got[2], ref_big):
                assert torch.equal(x["corr_scores"], y["corr_scores"]) and torch.equal(x["src_corr_points"], y["src_corr_points"])


def test_sampling_ahead_of_the_previous_forward_changes_nothing():
    """launch_batch(inputs_resident=True) hands the engine an event instead of stream order for the inputs
    (RoitrForwardIO::inputs_ready): the descriptors and the first sampling level of call s+1 then run on the geometry stream
    while call s is still on the main one, in scratch of their own.  Four calls in flight, the modes mixed, big and small
    batches alternating (so that a stale descriptor block, FPS scratch or pick list of the call before -- or of the call
    before that, which shares the alternating arena -- would be read): every output bit-identical to the pairs run alone."""
    from roitr_amd.synthetic import make_pair
    from tests.gpu_util import build_model, pair_to_device
    model = build_model("3DMatch")
    small = [pair_to_device(make_pair(n, config=2, pair_index=i)) for i, n in enumerate((1024, 1500))]
    big = [pair_to_device(make_pair(n, config=2, pair_index=10 + i)) for i, n in enumerate((5000, 4000, 3000, 5000, 2048, 4500))]
    one = [pair_to_device(make_pair(5000, config=2, pair_index=30))]
    with torch.no_grad():
        ref_small, ref_big, ref_one = model.forward_batch(small), model.forward_batch(big), model.forward_batch(one)
        torch.cuda.synchronize()
        for rounds in range(3):
            hs = [model.launch_batch(big, inputs_resident=True), model.launch_batch(small, inputs_resident=True),
                  model.launch_batch(one, inputs_resident=rounds != 1), model.launch_batch(big, inputs_resident=True),
                  model.launch_batch(one, inputs_resident=True), model.launch_batch(small, inputs_resident=rounds == 2)]
            got = [model.finish_batch(h) for h in hs]
            for g, ref in zip(got, (ref_big, ref_small, ref_one, ref_big, ref_one, ref_small)):
                for x, y in zip(g, ref):
                    _same(x, y)
        # the default of the model object (what bench.py sets)
        model.inputs_resident = True
        h0, h1 = model.launch_batch(one), model.launch_batch(one)
        _same(model.finish_batch(h0)[0], ref_one[0]); _same(model.finish_batch(h1)[0], ref_one[0])


def test_ahead_mode_switch_between_batch_sizes():
    """Calls of up to 128 pairs run their WHOLE geometry chain ahead of the previous call out of the alternating arenas, larger calls
    only their first sampling level (engine.cpp AHEAD_MAX_PAIRS): a different arena discipline on either side of the switch.  Calls of
    130 and 8 pairs mixed, four to six in flight, inputs ordered by event or by stream; 128 and 129 pairs back to back; a big call
    between two small ones that share an arena.  Every output bit-identical to the same call run alone."""
    from roitr_amd.synthetic import make_pair
    from tests.gpu_util import build_model, pair_to_device
    model = build_model("3DMatch", weights="selective")
    base = [pair_to_device(make_pair(1024 + 64 * (i % 3), config=2, pair_index=i, normals="field")) for i in range(13)]
    torch.cuda.synchronize()
    b8, b130 = base[:8], [base[i % 13] for i in range(130)]
    b128, b129 = b130[:128], b130[:129]
    with torch.no_grad():
        refs = {}
        for name, b in (("b8", b8), ("b130", b130), ("b128", b128), ("b129", b129)):
            refs[name] = model.forward_batch(b)
            torch.cuda.synchronize()
        plans = [
            [("b130", True), ("b8", True), ("b130", True), ("b8", True), ("b8", True), ("b130", False)],
            [("b8", True), ("b130", True), ("b8", True), ("b8", False), ("b130", True), ("b8", True)],
            [("b128", True), ("b129", True), ("b128", True), ("b129", True)],
            [("b129", True), ("b128", True), ("b8", True), ("b129", False), ("b128", True)],
        ]
        for plan in plans:
            hs = [(name, model.launch_batch({"b8": b8, "b130": b130, "b128": b128, "b129": b129}[name], inputs_resident=res)) for name, res in plan]
            for name, h in hs:
                got = model.finish_batch(h)
                assert len(got) == len(refs[name])
                for x, y in zip(got, refs[name]):
                    _same(x, y)


def test_stress_of_calls_in_flight():
    """scripts/stress_inflight.py as a test: 30 rounds of 8 calls in flight, batch sizes 1 - 12 of clouds of 1024 - 5000 points drawn at
    random, ground-truth outputs on / off, inputs by event / by stream -- a race between the two streams or a stale arena shows as a
    bitwise mismatch against the same batch run alone."""
    import random
    from roitr_amd.synthetic import make_pair
    from tests.gpu_util import build_model, pair_to_device
    keys = KEYS + ("src_node_corr_indices",)
    rng = random.Random(7)
    model = build_model("3DMatch", weights="selective")
    sizes = (1024, 1500, 2048, 3000, 4000, 5000)
    pool = [pair_to_device(make_pair(sizes[i % len(sizes)], config=2, pair_index=i, normals="field")) for i in range(24)]
    torch.cuda.synchronize()
    batches = [[pool[j] for j in rng.sample(range(24), rng.choice((1, 1, 2, 3, 6, 12)))] for _ in range(16)]
    with torch.no_grad():
        refs = []
        for b in batches:
            refs.append(model.forward_batch(b))
            torch.cuda.synchronize()
        bad = []
        for r in range(30):
            order = [rng.randrange(len(batches)) for _ in range(8)]
            hs = [(i, model.launch_batch(batches[i], want_gt=rng.random() < 0.8, inputs_resident=rng.random() < 0.7)) for i in order]
            for i, h in hs:
                got = model.finish_batch(h)
                for x, y in zip(got, refs[i]):
                    for k in keys:
                        if k.startswith("gt_") and not h["have_gt"]:
                            continue
                        if not torch.equal(x[k], y[k]):
                            bad.append((r, i, k))
    assert not bad, bad[:10]
