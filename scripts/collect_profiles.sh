#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh r04
# 1) timeout 900 rocprofv3 --kernel-trace --stats (csv) of the default bench,
# 2) two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) of the default bench -> profiles/pmc_traffic.json,
# 3) bench.py JSON lines (config 2 default, one pair per step, config 3, config 4 bf16 / f32, the driver's command),
# 4) the SQ counter passes (scripts/knn_config5_sq.sh, scripts/sq_pass.sh) BEFORE the bench lines that attach them, bench.py --config 5,
# 5) the batch-size curve (scripts/batch_sweep.sh) and the one-pair device timeline (scripts/b1_timeline.sh).
set -u
tag=${1:-r02}
export TMPDIR=/tmp
out=gpurun_out/$tag
rm -rf $out; mkdir -p $out
P="python bench.py --no-cpu-baseline --no-single-pair --no-rccl-selftest --no-profile-pass"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- $P --steps 4 --warmup 1 > $out/stats.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- $P --steps 1 --warmup 1 > $out/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- $P --steps 1 --warmup 1 > $out/pmc_write.log 2>&1
python scripts/pmc_summary.py $out/pmc_fetch/f_counter_collection.csv $out/pmc_write/w_counter_collection.csv $out/${tag}_pmc_traffic.json 512 2 2
# the bench lines below read roofline.traffic from THIS round's counter passes
cp $out/${tag}_pmc_traffic.json profiles/pmc_traffic.json
python scripts/prof_summary.py $out/stats s 5 45 > $out/${tag}_kernel_summary.txt
cp $out/stats/s_kernel_stats.csv $out/${tag}_kernel_stats.csv
python scripts/hbm_table.py $out $tag > $out/${tag}_hbm_gbs.txt
bash scripts/knn_config5_sq.sh $out/knn5sq > $out/${tag}_knn_config5_sq.txt 2>&1
cp $out/knn5sq/sq_knn_config5.json $out/sq_knn_config5.json
bash scripts/sq_pass.sh $out/sq > $out/${tag}_sq_pass.txt 2>&1
python scripts/sq_forward_json.py $out/sq/a $out/sq_forward.json > /dev/null 2>&1
# the bench lines below attach these counters only when their kernel-source stamp matches the build they time (bench.py attach_traffic)
cp $out/sq_forward.json profiles/sq_forward.json
cp $out/sq_knn_config5.json profiles/sq_knn_config5.json
python bench.py > $out/bench.log 2>&1
tail -1 $out/bench.log > $out/${tag}_bench.json
python bench.py --pairs-per-step 1 --steps 200 --warmup 20 --no-cpu-baseline > $out/bench_b1.log 2>&1
tail -1 $out/bench_b1.log > $out/${tag}_bench_pairs1.json
python bench.py --cloud surface --no-cpu-baseline --no-single-pair > $out/bench_surface.log 2>&1
tail -1 $out/bench_surface.log > $out/${tag}_bench_surface.json
python bench.py --weights plain --no-cpu-baseline --no-single-pair > $out/bench_plain.log 2>&1
tail -1 $out/bench_plain.log > $out/${tag}_bench_plain_weights.json
python bench.py --config 3 --no-cpu-baseline --no-single-pair > $out/bench_c3.log 2>&1
tail -1 $out/bench_c3.log > $out/${tag}_bench_config3.json
python bench.py --config 4 --no-single-pair > $out/bench_c4.log 2>&1
tail -1 $out/bench_c4.log > $out/${tag}_bench_config4_bf16.json
python bench.py --config 4 --dtype f32 --no-cpu-baseline --no-single-pair > $out/bench_c4f.log 2>&1
tail -1 $out/bench_c4f.log > $out/${tag}_bench_config4_f32.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.log 2>&1
tail -1 $out/bench_driver.log > $out/${tag}_bench_driver_cmd.json
python bench.py --config 5 > $out/bench_c5.log 2>&1
tail -1 $out/bench_c5.log > $out/${tag}_bench_config5.json
# round 6: the three-way-split engine mode beside the fp32 headline, the split GEMM alone, the host share with eight rank processes
python bench.py --dtype f32x3 --no-cpu-baseline > $out/bench_x3.log 2>&1
tail -1 $out/bench_x3.log > $out/${tag}_bench_f32x3.json
timeout 300 python scripts/bench_gemm_x3.py > $out/${tag}_gemm_x3.txt 2>&1
PAIRS=128 timeout 600 bash scripts/host_share_8ranks.sh $out/host8 > $out/host8.log 2>&1
cp $out/host8/host_share.json $out/${tag}_host_share_8ranks.json
bash scripts/batch_sweep.sh $out/sweep > $out/${tag}_batch_sweep.txt 2>&1
bash scripts/b1_timeline.sh $out/b1 > $out/${tag}_b1_timeline.txt 2>&1
for f in $out/${tag}_bench*.json; do echo $f; cut -c1-400 $f; done
