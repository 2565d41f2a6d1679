export TMPDIR=/tmp
mkdir -p gpurun_out/k2
for x in 0 11; do
  ROITR_KNN_X=$x bash scripts/sq_cmd.sh "knn|sort_q" gpurun_out/k2/sq_$x python scripts/bench_knn_shapes.py > gpurun_out/k2/sq_$x.txt 2>&1; cat gpurun_out/k2/sq_$x.txt
done
tail -3 gpurun_out/k2/sq_0/c.log gpurun_out/k2/sq_0/d.log
