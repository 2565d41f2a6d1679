#!/bin/bash
# Per-kernel time table of config 4 (bf16, 64 pairs per call): rocprofv3 --kernel-trace --stats of 4 steps.
#   bash scripts/c4_kernel_times.sh [outdir]
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
out=${1:-gpurun_out/c4k}; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python bench.py --config 4 --steps 4 --warmup 2 --no-cpu-baseline --no-single-pair --no-rccl-selftest --no-profile-pass > $out/log.txt 2>&1
tail -1 $out/log.txt | cut -c1-160
python scripts/prof_summary.py $out s 6 40
