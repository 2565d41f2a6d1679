#!/bin/bash
# level-1 grid + self kNN ahead on the geometry stream (ahead mode): parity of calls in flight, A/B at 512 / 8 / 1 pairs per call
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
out=gpurun_out/knn0; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_graph_gpu.py tests/test_model_gpu.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -3 $out/tests.log
timeout 300 python scripts/stress_inflight.py 40 2>&1 | tail -2
B="--no-cpu-baseline --no-rccl-selftest --no-single-pair"
run() { name=$1; pp=$2; st=$3; shift 3; env "$@" timeout 300 python bench.py --pairs-per-step $pp --steps $st --warmup 6 $B > $out/$name.json 2> $out/$name.err; }
run h_on 512 8 ROITR_X=0
run h_off 512 8 ROITR_KNN0_AHEAD=0
run h_on2 512 8 ROITR_X=0
run h_off2 512 8 ROITR_KNN0_AHEAD=0
run b8_on 8 100 ROITR_X=0
run b8_off 8 100 ROITR_KNN0_AHEAD=0
run b1_on 1 300 ROITR_X=0
run b1_off 1 300 ROITR_KNN0_AHEAD=0
run b64_on 64 30 ROITR_X=0
run b64_off 64 30 ROITR_KNN0_AHEAD=0
python - <<PY
import json
for f in ("h_on","h_off","h_on2","h_off2","b8_on","b8_off","b1_on","b1_off","b64_on","b64_off"):
    try:
        j=json.loads(open("$out/%s.json"%f).read().strip().splitlines()[-1]); k=j.get("kernel_ms_per_step",{})
        print(f, j["value"], j["ms_per_step"], {x:round(k.get(x,0),2) for x in ("phase.encoder","phase.matching","knn_query_kernel","fps_kernel")})
    except Exception as e: print(f, "failed", e)
PY
