#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
out=gpurun_out/b1host; rm -rf $out; mkdir -p $out
B="--no-cpu-baseline --no-profile-pass --no-rccl-selftest --no-single-pair"
ROITR_BENCH_TRACE=1 timeout 300 python bench.py --pairs-per-step 1 --steps 60 --warmup 20 $B > $out/b1.json 2> $out/b1.err
grep "bench trace" $out/b1.err | tail -12
timeout 300 python scripts/bench_graph.py > $out/graph.txt 2>&1; cat $out/graph.txt | tail -10
