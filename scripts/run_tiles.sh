cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_stages_gpu.py tests/test_fdmatch_gpu.py -q -m gpu -x 2>&1 | tail -3
for m in 64 128 256; do
  export ROITR_LN_FUSE_MAX=$m
  python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse_max $m', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_ms_per_step']['gemm_kernel'])"
done
