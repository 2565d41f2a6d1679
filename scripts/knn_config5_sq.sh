#!/bin/bash
# SQ counter pass of `bench.py --config 5` (kNN(64) + PPF, 32 clouds x 30000 points): own rocprofv3 run, kernel-trace + PMC only.
#   bash scripts/knn_config5_sq.sh [outdir]      -> <outdir>/sq_knn_config5.json (copy to profiles/)
# VALU-issue fraction of a kernel = SQ_INSTS_VALU x 2 cycles (a wave64 VALU instruction occupies its SIMD-32 for 2 cycles,
# MI355X_MICROARCH.md "Execution model") / (kernel duration x 2.4 GHz x 1024 SIMDs): the share of the chip's VALU issue slots
# the launch actually used -- an exact kNN is bound by that, not by the 39 MB per cloud it moves.
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
out=${1:-gpurun_out/knn5sq}; rm -rf $out; mkdir -p $out
P="python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES \
    --output-format csv -d $out/a -o s -- $P > $out/a.log 2>&1
python - $out <<'PY'
import csv, collections, glob, json, os, re, sys
sys.path.insert(0, os.getcwd())
from roitr_amd.build import source_hash
out = sys.argv[1]
t = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int); dur = collections.defaultdict(float)
for f in glob.glob(out + "/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k).split("(")[0]
        t[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            n[k] += 1
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
clouds, npts = 32, 30000
res = {"clouds": clouds, "n_points": npts, "k": 64, "kernel_source_sha16": source_hash(), "source": "rocprofv3 --kernel-trace --pmc SQ_* (scripts/knn_config5_sq.sh), per launch averages",
       "definition": "valu_issue_frac = SQ_INSTS_VALU * 2 cycles / (launch duration * 2.4 GHz * 1024 SIMDs)", "kernels": {}}
for k, v in t.items():
    if not (k.startswith("knn_") or k.startswith("grid_build")) or not n[k]:
        continue
    L = n[k]; w = v["SQ_WAVE_CYCLES"] or 1.0; d = dur[k] / L
    res["kernels"][k] = {"launches": L, "avg_us_profiled": round(d * 1e6, 1), "valu_per_query": round(v["SQ_INSTS_VALU"] / L / (clouds * npts), 1),
                         "salu_per_query": round(v["SQ_INSTS_SALU"] / L / (clouds * npts), 1), "lds_per_query": round(v["SQ_INSTS_LDS"] / L / (clouds * npts), 1),
                         "wave_cycles_issuing": round(v["SQ_ACTIVE_INST_ANY"] / w, 3), "wave_cycles_waiting": round(v["SQ_WAIT_ANY"] / w, 3),
                         "valu_issue_frac": round(v["SQ_INSTS_VALU"] / L * 2.0 / (d * 2.4e9 * 1024), 4)}
json.dump(res, open(out + "/sq_knn_config5.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
