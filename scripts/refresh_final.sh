#!/bin/bash
# after a late engine change: the GPU suite, then the bench lines / sweep / timeline that depend on it (profiles/<tag>_*)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
tag=${1:-r04}; out=gpurun_out/${tag}_refresh; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $out/suite.txt; cat $out/suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $out/bench.log 2>&1; tail -1 $out/bench.log > $out/${tag}_bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.log 2>&1; tail -1 $out/bench_driver.log > $out/${tag}_bench_driver_cmd.json
python bench.py --weights plain --no-cpu-baseline --no-single-pair > $out/bench_plain.log 2>&1; tail -1 $out/bench_plain.log > $out/${tag}_bench_plain_weights.json
python bench.py --cloud surface --no-cpu-baseline --no-single-pair > $out/bench_surface.log 2>&1; tail -1 $out/bench_surface.log > $out/${tag}_bench_surface.json
python bench.py --config 3 --no-cpu-baseline --no-single-pair > $out/bench_c3.log 2>&1; tail -1 $out/bench_c3.log > $out/${tag}_bench_config3.json
python bench.py --pairs-per-step 1 --steps 200 --warmup 20 --no-cpu-baseline > $out/bench_b1.log 2>&1; tail -1 $out/bench_b1.log > $out/${tag}_bench_pairs1.json
bash scripts/batch_sweep.sh $out/sweep > $out/${tag}_batch_sweep.txt 2>&1
bash scripts/b1_timeline.sh $out/b1 > $out/${tag}_b1_timeline.txt 2>&1
for f in $out/${tag}_bench*.json; do python -c "import json,sys; j=json.load(open('$f')); r=j['roofline']; print('$f'.split('/')[-1], j['value'], j['ms_per_step'], r['frac'], r.get('traffic'), (j.get('single_pair_mode') or {}).get('ms_per_pair'), (j.get('single_pair_mode') or {}).get('ms_per_pair_one_call_in_flight'))"; done
cat $out/${tag}_batch_sweep.txt; head -4 $out/${tag}_b1_timeline.txt
