#!/bin/bash
# gemm_small_kernel (32x32 tiles, 16x16x4 MFMA) for small grids + the cheap weight signature: parity and one-pair A/B
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
out=gpurun_out/small; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_stages_gpu.py tests/test_model_gpu.py tests/test_graph_gpu.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -4 $out/tests.log
B="--no-cpu-baseline --no-profile-pass --no-rccl-selftest --no-single-pair"
run() { name=$1; pp=$2; st=$3; shift 3; env "$@" timeout 300 python bench.py --pairs-per-step $pp --steps $st --warmup 20 $B > $out/$name.json 2> $out/$name.err; }
run b1_small 1 300 ROITR_X=0
run b1_old 1 300 ROITR_GEMM_SMALL_MAX=0
run b1_256 1 300 ROITR_GEMM_SMALL_MAX=256
run b1_1024 1 300 ROITR_GEMM_SMALL_MAX=1024
run b8_small 8 100 ROITR_X=0
run b8_old 8 100 ROITR_GEMM_SMALL_MAX=0
run b8_1024 8 100 ROITR_GEMM_SMALL_MAX=1024
run b32_small 32 40 ROITR_X=0
run b32_old 32 40 ROITR_GEMM_SMALL_MAX=0
python - <<PY
import json
for f in ("b1_small","b1_old","b1_256","b1_1024","b8_small","b8_old","b8_1024","b32_small","b32_old"):
    try:
        j=json.loads(open("$out/%s.json"%f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"])
    except Exception as e: print(f, "failed", e)
PY
