#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh r01
# 1) bench.py JSON line, 2) rocprofv3 --kernel-trace --stats (csv), 3) two separate --pmc passes (FETCH_SIZE, WRITE_SIZE).
set -u
tag=${1:-r01}
export TMPDIR=/tmp
out=gpurun_out/$tag
rm -rf $out; mkdir -p $out
python bench.py > $out/bench.log 2>&1
tail -1 $out/bench.log > $out/${tag}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python bench.py --no-cpu-baseline --no-single-pair --steps 4 --warmup 1 > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python bench.py --no-cpu-baseline --no-single-pair --steps 1 --warmup 1 > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python bench.py --no-cpu-baseline --no-single-pair --steps 1 --warmup 1 > $out/pmc_write.log 2>&1
python scripts/pmc_summary.py $out/pmc_fetch/f_counter_collection.csv $out/pmc_write/w_counter_collection.csv $out/${tag}_pmc_traffic.json 512
python scripts/prof_summary.py $out/stats s 5 45 > $out/${tag}_kernel_summary.txt
cp $out/stats/s_kernel_stats.csv $out/${tag}_kernel_stats.csv
python scripts/hbm_table.py $out $tag > $out/${tag}_hbm_gbs.txt
cat $out/${tag}_bench.json
