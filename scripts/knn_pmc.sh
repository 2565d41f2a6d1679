cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/kp; rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/kp -o g -- python scripts/bench_knn.py 30000 64 32 > gpurun_out/kp.log 2>&1
python - <<PY
import csv,collections
t=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("gpurun_out/kp/g_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0][-28:]
    t[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in t.items():
    if "knn" in k:
        w=v["SQ_WAVE_CYCLES"]
        print(k, {c: f"{x:.3e}" for c,x in v.items()})
        print("  wait_any %.3f wait_inst %.3f active %.3f | per wave: valu %.0f lds %.0f salu %.0f  wave-cycles(quad) %.0f" % (v["SQ_WAIT_ANY"]/w, v["SQ_WAIT_INST_ANY"]/w, v["SQ_ACTIVE_INST_ANY"]/w, v["SQ_INSTS_VALU"]/v["SQ_WAVES"], v["SQ_INSTS_LDS"]/v["SQ_WAVES"], v["SQ_INSTS_SALU"]/v["SQ_WAVES"], w/v["SQ_WAVES"]))
PY
