#!/bin/bash
# BASELINE.json configs[4] micro-benchmark (N = 30000, k = 64 kNN + fused PPF): old wave-per-query selection kernel vs the
# workgroup-per-cell kernel, timing + SQ counters (separate rocprofv3 --pmc pass, kernel-trace only).  Run through gpurun:
#   bash scripts/knn_config5.sh <outdir>
set -u
out=${1:-gpurun_out/knn5}
export TMPDIR=/tmp
mkdir -p $out
{
  echo "== general selection kernel (ROITR_KNN_NO_CELL=1)"
  for c in 1 4 32; do ROITR_KNN_NO_CELL=1 python scripts/bench_knn.py 30000 64 $c; done
  echo "== workgroup-per-cell kernel (default)"
  for c in 1 4 32; do ROITR_KNN_STATS=1 python scripts/bench_knn.py 30000 64 $c; done
  for rho in 18 26 30; do echo "== workgroup-per-cell kernel, grid occupancy $rho points per cell"; ROITR_KNN_RHO=$rho ROITR_KNN_STATS=1 python scripts/bench_knn.py 30000 64 32; done
} > $out/timing.txt 2>&1
for mode in old new; do
  if [ $mode = old ]; then export ROITR_KNN_NO_CELL=1; else unset ROITR_KNN_NO_CELL; fi
  rm -rf $out/pmc_$mode
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES \
     --output-format csv -d $out/pmc_$mode -o g -- python scripts/bench_knn.py 30000 64 32 > $out/pmc_$mode.log 2>&1
  python - $out/pmc_$mode/g_counter_collection.csv $mode <<'PY' >> $out/sq_counters.txt
import csv, collections, re, sys
t = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    m_ = re.search(r"(knn_\w+|grid_build\w*)", r["Kernel_Name"])
    if not m_: continue
    k = m_.group(1)
    t[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
    t[k]["_ns"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 8.0   # 8 counters per dispatch row
queries = 30000 * 32
print("== mode", sys.argv[2], "(32 clouds x 30000 points, k = 64; per launch averages)")
for k, v in t.items():
    if "knn" in k and v["SQ_WAVES"] > 0:
        w, L = v["SQ_WAVE_CYCLES"], n[k]
        print("%-22s launches %3d  avg %.0f us (profiled)  wait_any %.3f wait_inst %.3f active %.3f | VALU/query %.0f  LDS/query %.0f  SALU/query %.0f" % (
            k, L, v["_ns"] / L / 1e3, v["SQ_WAIT_ANY"] / w, v["SQ_WAIT_INST_ANY"] / w, v["SQ_ACTIVE_INST_ANY"] / w,
            v["SQ_INSTS_VALU"] / L / queries, v["SQ_INSTS_LDS"] / L / queries, v["SQ_INSTS_SALU"] / L / queries))
PY
done
cat $out/timing.txt $out/sq_counters.txt
