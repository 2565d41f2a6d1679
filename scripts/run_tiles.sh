cd $GRAFT_REPO_ROOT
for pf in 0 1; do
  if [ $pf = 1 ]; then export ROITR_GEMM_PF2=1; fi
  for s in "79872 256 256" "79872 256 2048" "1280000 64 64" "19968 256 256"; do python scripts/bench_gemm.py $s 30; done
  python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pf2 $pf', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_ms_per_step']['gemm_kernel'])"
done
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_stages_gpu.py -q -m gpu -x 2>&1 | tail -3
