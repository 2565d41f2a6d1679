#!/bin/bash
# re-run the bench lines that read roofline.traffic from profiles/pmc_traffic.json (after a change of how bench.py reads it)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
tag=${1:-r04}; out=gpurun_out/${tag}_lines; rm -rf $out; mkdir -p $out
python bench.py > $out/bench.log 2>&1; tail -1 $out/bench.log > $out/${tag}_bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.log 2>&1; tail -1 $out/bench_driver.log > $out/${tag}_bench_driver_cmd.json
python bench.py --weights plain --no-cpu-baseline --no-single-pair > $out/bench_plain.log 2>&1; tail -1 $out/bench_plain.log > $out/${tag}_bench_plain_weights.json
python bench.py --cloud surface --no-cpu-baseline --no-single-pair > $out/bench_surface.log 2>&1; tail -1 $out/bench_surface.log > $out/${tag}_bench_surface.json
for f in $out/${tag}_bench*.json; do python -c "import json,sys; j=json.load(open('$f')); r=j['roofline']; print('$f', j['value'], j['ms_per_step'], r['frac'], r['traffic'], r.get('traffic_over_algorithmic'), j.get('single_pair_mode',{}).get('ms_per_pair'))"; done
