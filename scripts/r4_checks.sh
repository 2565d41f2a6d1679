export TMPDIR=/tmp
mkdir -p gpurun_out/h8
python -m pytest tests/test_stages_gpu.py tests/test_correspondences_gpu.py -m gpu -x -q -k "function_table or scan_like or engine_uses_the_table" > gpurun_out/h8/pytest.log 2>&1; tail -4 gpurun_out/h8/pytest.log
python bench.py --no-cpu-baseline --no-single-pair --cloud surface > gpurun_out/h8/bench_surface.log 2>&1
tail -1 gpurun_out/h8/bench_surface.log | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('surface', o['value'], o['ms_per_step'], o['config']['correspondences_found'], o['result_gather']['truncated_pairs'], {k:o['kernel_ms_per_step'][k] for k in ('knn_query_kernel','grid_build_kernel','fps_kernel','gemm_kernel','ot_kernel','phase.matching')})"
ROITR_GEO_TABLE_H=1 python bench.py --no-cpu-baseline --no-single-pair > gpurun_out/h8/bench_h1.log 2>&1
tail -1 gpurun_out/h8/bench_h1.log | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('table h=1', o['value'], o['ms_per_step'], o['config']['geometric_embedding_table'], o['kernel_ms_per_step']['geo_table_kernel'])"
python bench.py --no-cpu-baseline --no-single-pair > gpurun_out/h8/bench_h2.log 2>&1
tail -1 gpurun_out/h8/bench_h2.log | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('table default', o['value'], o['ms_per_step'], o['config']['geometric_embedding_table'], o['kernel_ms_per_step']['geo_table_kernel'])"
