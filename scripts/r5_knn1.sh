# round 5, kNN ball kernel: parity + A/B of the per-call times
export TMPDIR=/tmp
mkdir -p gpurun_out/k1
timeout 900 python -m pytest tests/test_pointops_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > gpurun_out/k1/pytest.log 2>&1; tail -5 gpurun_out/k1/pytest.log
for x in 0 11 15; do
  ROITR_KNN_X=$x timeout 300 python scripts/bench_knn_shapes.py > gpurun_out/k1/shapes_$x.log 2>&1; cat gpurun_out/k1/shapes_$x.log
done
ROITR_KNN_STATS=1 ROITR_KNN_X=15 timeout 300 python scripts/bench_knn_shapes.py 2>&1 | tail -3
