#!/bin/bash
# pairs/s of the headline workload against the number of pairs per engine call (run through gpurun from the repo root)
out=${1:-gpurun_out/sweep}; mkdir -p $out
for b in 1 8 32 64 128 256 512; do
  steps=$(( 4096 / b )); [ $steps -gt 200 ] && steps=200; [ $steps -lt 8 ] && steps=8
  python bench.py --pairs-per-step $b --steps $steps --warmup 5 --no-cpu-baseline --no-single-pair --no-profile-pass 2>/dev/null | tail -1 > $out/b$b.json
  python -c "import json,sys; d=json.load(open('$out/b$b.json')); print('pairs_per_step', $b, 'pairs_per_s', d['value'], 'ms_per_step', d['ms_per_step'])"
done
