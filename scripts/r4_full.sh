#!/bin/bash
# the whole geometry chain of call s+1 beside call s (calls of up to 128 pairs): parity of calls in flight, batch curve A/B
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
out=gpurun_out/full; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_graph_gpu.py tests/test_model_gpu.py tests/test_correspondences_gpu.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -3 $out/tests.log
timeout 300 python scripts/stress_inflight.py 60 2>&1 | tail -2
B="--no-cpu-baseline --no-rccl-selftest --no-single-pair --no-profile-pass"
run() { name=$1; pp=$2; st=$3; shift 3; env "$@" timeout 300 python bench.py --pairs-per-step $pp --steps $st --warmup 6 $B > $out/$name.json 2> $out/$name.err; }
for b in 1 8 32 64 128; do
  steps=$(( 4096 / b )); [ $steps -gt 300 ] && steps=300; [ $steps -lt 16 ] && steps=16
  run on_$b $b $steps ROITR_X=0
  run off_$b $b $steps ROITR_KNN0_AHEAD=0
done
python - <<PY
import json
for b in (1,8,32,64,128):
    r=[]
    for m in ("on","off"):
        try:
            j=json.loads(open("$out/%s_%d.json"%(m,b)).read().strip().splitlines()[-1]); r.append((j["value"], j["ms_per_step"]))
        except Exception as e: r.append(("failed",str(e)[:60]))
    print(b, "full-ahead", r[0], "| partial", r[1])
PY
