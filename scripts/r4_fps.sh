#!/bin/bash
# round-4 FPS rewrite: parity, one-pair latency with 8- and 4-wave workgroups, headline, one-pair timeline, MFMA k-order probe
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
out=gpurun_out/fps; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_pointops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "fps or chain or forward" > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -3 $out/tests.log
B="--no-cpu-baseline --no-profile-pass --no-rccl-selftest"
timeout 300 python bench.py --pairs-per-step 1 --steps 200 --warmup 20 $B --no-single-pair > $out/b1_512.json 2> $out/b1_512.err
ROITR_FPS_SMALL_BATCH_BLOCK=256 timeout 300 python bench.py --pairs-per-step 1 --steps 200 --warmup 20 $B --no-single-pair > $out/b1_256.json 2> $out/b1_256.err
timeout 600 python bench.py $B --no-single-pair > $out/bench.json 2> $out/bench.err
python - <<PY
import json
for f in ("b1_512","b1_256","bench"):
    try:
        j=json.loads(open("$out/%s.json"%f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], {k:v for k,v in j.get("stage_ms_per_step",{}).items() if k in ("fps","knn","encoder")})
    except Exception as e: print(f, "failed", e)
PY
hipcc --offload-arch=gfx950 -O2 -Wno-everything scripts/micro/mfma_korder.hip -o /tmp/mfma_korder && /tmp/mfma_korder > $out/mfma_korder.txt 2>&1; cat $out/mfma_korder.txt
timeout 600 bash scripts/b1_timeline.sh $out/b1 > $out/b1_timeline.txt 2>&1; head -30 $out/b1_timeline.txt
