export TMPDIR=/tmp
mkdir -p gpurun_out/k3
timeout 900 python -m pytest tests/test_pointops_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > gpurun_out/k3/pytest.log 2>&1; tail -5 gpurun_out/k3/pytest.log
for cfg in "0 6" "56 6" "56 9.4" "120 9.4" "56 8" "56 11"; do
  set -- $cfg
  ROITR_KNN_X=$1 timeout 300 python scripts/bench_knn_shapes.py 1024 $2 > gpurun_out/k3/shapes_$1_$2.log 2>&1; grep -v amdgpu.ids gpurun_out/k3/shapes_$1_$2.log
done
ROITR_KNN_STATS=1 ROITR_KNN_X=120 timeout 300 python scripts/bench_knn_shapes.py 1024 9.4 2>&1 | tail -2
ROITR_KNN_X=56 bash scripts/sq_cmd.sh "knn|sort_q" gpurun_out/k3/sq python scripts/bench_knn_shapes.py 1024 9.4 > gpurun_out/k3/sq.txt 2>&1; cat gpurun_out/k3/sq.txt
