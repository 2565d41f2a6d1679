cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for s in "79872 256 256" "79872 256 2048" "8192 8192 1024"; do python scripts/bench_gemm.py $s 30; done
rm -rf gpurun_out/gp; rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d gpurun_out/gp -o g -- python scripts/bench_gemm.py 79872 256 256 5 > gpurun_out/gp.log 2>&1
python - <<PY
import csv,collections
t=collections.defaultdict(float); n=0
for r in csv.DictReader(open("gpurun_out/gp/g_counter_collection.csv")):
    if "gemm_kernel" in r["Kernel_Name"]:
        t[r["Counter_Name"]]+=float(r["Counter_Value"])
print({k: f"{v:.3e}" for k,v in t.items()})
w=t["SQ_WAVE_CYCLES"]
for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_LDS"): print(k, round(t[k]/w,3))
print("MFMA busy / busy cycles", t["SQ_VALU_MFMA_BUSY_CYCLES"]/max(t["SQ_BUSY_CYCLES"],1))
PY
