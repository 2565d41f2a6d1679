cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_stages_gpu.py -q -m gpu 2>&1 | tail -4
python bench.py --no-cpu-baseline --pairs-per-step 128 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['pairs_per_step'], d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
