# full GPU suite + smoke + the driver's bench command
export TMPDIR=/tmp
mkdir -p gpurun_out/s
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/s/pytest.log 2>&1; grep -v "^\[descriptor\]" gpurun_out/s/pytest.log | tail -6; grep "^\[descriptor\]" gpurun_out/s/pytest.log | sort -t' ' -k8 -g | tail -8
timeout 300 python -c "import __graft_entry__ as G; G.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s/bench.log 2>&1; tail -1 gpurun_out/s/bench.log | cut -c1-900
