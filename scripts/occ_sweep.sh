export TMPDIR=/tmp
mkdir -p gpurun_out/h7
for occ in "6,6,6,6" "5,6,6,6" "8,6,6,6" "6,4,6,6" "6,9,6,6" "4,4,6,6"; do
  export ROITR_GRID_OCC=$occ
  python bench.py --no-cpu-baseline --no-single-pair --steps 6 --warmup 3 > gpurun_out/h7/bench_$occ.log 2>&1
  tail -1 gpurun_out/h7/bench_$occ.log | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('occ=$occ', o['value'], o['ms_per_step'], {k:o['kernel_ms_per_step'][k] for k in ('knn_query_kernel','grid_build_kernel','local_block_kernel','local_attn_kernel')})"
done
