# kNN development loop on the GPU box: bit-exact suite, per-call statistics, SQ counters, rocprofv3 kernel stats
export TMPDIR=/tmp
mkdir -p gpurun_out/h4
python -m pytest tests/test_pointops_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > gpurun_out/h4/pytest.log 2>&1; tail -3 gpurun_out/h4/pytest.log
ROITR_KNN_STATS=1 ROITR_KNN_STATS_VERBOSE=1 python bench.py --no-cpu-baseline --no-single-pair --no-rccl-selftest --no-profile-pass --steps 1 --warmup 1 2>&1 | grep KNN | sort | uniq -c | sort -rn | head -24
bash scripts/sq_kernels.sh "knn|grid" gpurun_out/sqk2 2>&1 | tail -14
rm -rf gpurun_out/h4/st; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h4/st -o s -- python bench.py --no-cpu-baseline --no-single-pair --no-rccl-selftest --no-profile-pass --steps 2 --warmup 1 > gpurun_out/h4/stats.log 2>&1
python scripts/prof_summary.py gpurun_out/h4/st s 3 60 | grep -i "knn\|total\|sort\|grid"
