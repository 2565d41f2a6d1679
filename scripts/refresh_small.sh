#!/bin/bash
# after the last engine change of round 4 (small-batch scheduling only): GPU suite + the lines it can move
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
tag=${1:-r04}; out=gpurun_out/${tag}_small; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $out/suite.txt; cat $out/suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $out/bench.log 2>&1; tail -1 $out/bench.log > $out/${tag}_bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.log 2>&1; tail -1 $out/bench_driver.log > $out/${tag}_bench_driver_cmd.json
python bench.py --pairs-per-step 1 --steps 200 --warmup 20 --no-cpu-baseline > $out/bench_b1.log 2>&1; tail -1 $out/bench_b1.log > $out/${tag}_bench_pairs1.json
python bench.py --config 4 --no-cpu-baseline --no-single-pair > $out/bench_c4.log 2>&1; tail -1 $out/bench_c4.log > $out/${tag}_bench_config4_bf16.json
python bench.py --config 4 --dtype f32 --no-cpu-baseline --no-single-pair > $out/bench_c4f.log 2>&1; tail -1 $out/bench_c4f.log > $out/${tag}_bench_config4_f32.json
bash scripts/batch_sweep.sh $out/sweep > $out/${tag}_batch_sweep.txt 2>&1
bash scripts/b1_timeline.sh $out/b1 > $out/${tag}_b1_timeline.txt 2>&1
for f in $out/${tag}_bench*.json; do python -c "import json,sys; j=json.load(open('$f')); r=j['roofline']; print('$f'.split('/')[-1], j['value'], j['ms_per_step'], r['frac'], (j.get('single_pair_mode') or {}).get('ms_per_pair'), (j.get('single_pair_mode') or {}).get('ms_per_pair_one_call_in_flight'))"; done
cat $out/${tag}_batch_sweep.txt; head -3 $out/${tag}_b1_timeline.txt
