#!/bin/bash
# sampling-ahead (RoitrForwardIO::inputs_ready): parity of calls in flight, one-pair and headline A/B
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
out=gpurun_out/ahead; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_graph_gpu.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -5 $out/tests.log
B="--no-cpu-baseline --no-profile-pass --no-rccl-selftest --no-single-pair"
timeout 300 python bench.py --pairs-per-step 1 --steps 300 --warmup 30 $B > $out/b1_ahead.json 2> $out/b1_ahead.err
timeout 300 python bench.py --pairs-per-step 1 --steps 300 --warmup 30 $B --no-sampling-ahead > $out/b1_plain.json 2> $out/b1_plain.err
timeout 300 python bench.py --pairs-per-step 8 --steps 100 --warmup 10 $B > $out/b8_ahead.json 2> $out/b8_ahead.err
timeout 300 python bench.py --pairs-per-step 8 --steps 100 --warmup 10 $B --no-sampling-ahead > $out/b8_plain.json 2> $out/b8_plain.err
timeout 600 python bench.py --no-cpu-baseline --no-profile-pass --no-rccl-selftest > $out/bench_ahead.json 2> $out/bench_ahead.err
timeout 600 python bench.py $B --no-sampling-ahead > $out/bench_plain.json 2> $out/bench_plain.err
python - <<PY
import json
for f in ("b1_ahead","b1_plain","b8_ahead","b8_plain","bench_ahead","bench_plain"):
    try:
        j=json.loads(open("$out/%s.json"%f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j.get("single_pair_mode"))
    except Exception as e: print(f, "failed", e)
PY
