export TMPDIR=/tmp
mkdir -p gpurun_out/h5
for abl in 0 1 3 7; do
  export ROITR_KNN_ABL=$abl
  rm -rf gpurun_out/h5/st; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h5/st -o s -- python bench.py --no-cpu-baseline --no-single-pair --no-rccl-selftest --no-profile-pass --steps 1 --warmup 1 > gpurun_out/h5/stats_$abl.log 2>&1
  echo "== abl $abl"; python scripts/prof_summary.py gpurun_out/h5/st s 2 60 | grep -i "knn_cloud"
done
