#!/bin/bash
# where the embedding E and the tail of the geometry chain are enqueued relative to the encoder levels
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
out=gpurun_out/sideorder; rm -rf $out; mkdir -p $out
B="--no-cpu-baseline --no-rccl-selftest --no-single-pair"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $out/$name.json 2> $out/$name.err; }
run A ROITR_X=0
run B ROITR_SIDE_TAIL_AFTER=1
run C ROITR_SIDE_TAIL_AFTER=2
run D ROITR_SIDE_GEO_AFTER=1 ROITR_SIDE_TAIL_AFTER=1
run E ROITR_SIDE_GEO_AFTER=1 ROITR_SIDE_TAIL_AFTER=2
run F ROITR_SIDE_GEO_AFTER=2 ROITR_SIDE_TAIL_AFTER=1
run G ROITR_SIDE_GEO_AFTER=2 ROITR_SIDE_TAIL_AFTER=2
run H ROITR_SIDE_GEO_AFTER=0
python - <<PY
import json
for f in "ABCDEFGH":
    try:
        j=json.loads(open("$out/%s.json"%f).read().strip().splitlines()[-1]); k=j.get("kernel_ms_per_step",{})
        print(f, j["value"], j["ms_per_step"], {x:round(k.get(x,0),2) for x in ("phase.encoder","phase.global_transformer","phase.decoder","phase.matching","gemm_kernel","local_block_kernel","geo_table_kernel")})
    except Exception as e: print(f, "failed", e)
PY
