#!/bin/bash
# FPS residency in large batches: block size x grid cap (clouds resident per CU), plus the med3 kNN insertion chain
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
out=gpurun_out/fpsgrid; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_pointops_gpu.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -3 $out/tests.log
B="--no-cpu-baseline --no-rccl-selftest --no-single-pair"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $out/$name.json 2> $out/$name.err; }
run A ROITR_X=0
run B ROITR_FPS_GRID=256
run C ROITR_FPS_BLOCK=512
run D ROITR_FPS_BLOCK=512 ROITR_FPS_GRID=256
run E ROITR_FPS_BLOCK=512 ROITR_FPS_GRID=512
run F ROITR_FPS_GRID=512
python - <<PY
import json
for f in "ABCDEF":
    try:
        j=json.loads(open("$out/%s.json"%f).read().strip().splitlines()[-1]); k=j.get("kernel_ms_per_step",{})
        print(f, j["value"], j["ms_per_step"], {x:k.get(x) for x in ("fps_kernel","knn_query_kernel","phase.encoder","gemm_kernel","phase.forward")})
    except Exception as e: print(f, "failed", e)
PY
