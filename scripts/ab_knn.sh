export TMPDIR=/tmp
mkdir -p gpurun_out/h6
for v in 0 1; do
  export ROITR_KNN_CLOUD=$v
  python bench.py --no-cpu-baseline --no-single-pair > gpurun_out/h6/bench_$v.log 2>&1
  tail -1 gpurun_out/h6/bench_$v.log | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('cloud=$v', o['value'], o['ms_per_step'], {k:o['kernel_ms_per_step'][k] for k in ('knn_query_kernel','gemm_kernel','local_block_kernel','fps_kernel','grid_build_kernel')})"
done
export ROITR_KNN_CLOUD=0
rm -rf gpurun_out/h6/st; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h6/st -o s -- python bench.py --no-cpu-baseline --no-single-pair --no-rccl-selftest --no-profile-pass --steps 2 --warmup 1 > gpurun_out/h6/stats.log 2>&1
python scripts/prof_summary.py gpurun_out/h6/st s 3 60 | grep -i "knn\|total\|sort\|grid"
