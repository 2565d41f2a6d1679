#!/bin/bash
# SQ counter pass of the headline forward (own rocprofv3 run, kernel-trace only): per kernel family, share of wave cycles spent
# waiting / issuing and instructions per wave.   bash scripts/sq_pass.sh [outdir]
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
out=${1:-gpurun_out/sq}; rm -rf $out; mkdir -p $out
P="python bench.py --no-cpu-baseline --no-single-pair --no-rccl-selftest --no-profile-pass --steps 1 --warmup 1"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/a -o s -- $P > $out/a.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/b -o s -- $P > $out/b.log 2>&1
python - <<PY
import csv,collections,re,glob
def load(d):
    t=collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); k=re.sub(r"^void ","",k).split("(")[0]
            t[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    return t
a=load("$out/a"); b=load("$out/b")
keys=sorted(a, key=lambda k:-a[k]["SQ_WAVE_CYCLES"])[:22]
print("%-44s %9s %6s %6s %6s | per wave: %7s %6s %6s | %7s %7s %6s" % ("kernel","wavecyc","wait","w_inst","active","valu","lds","salu","vmem_rd","vmem_wr","mfma"))
for k in keys:
    v=a[k]; w=v["SQ_WAVE_CYCLES"] or 1; n=v["SQ_WAVES"] or 1; u=b.get(k,{}); nb=1
    print("%-44s %9.3e %6.3f %6.3f %6.3f | %7.0f %6.0f %6.0f | %7.1f %7.1f %6.0f" % (k[:44], w, v["SQ_WAIT_ANY"]/w, v["SQ_WAIT_INST_ANY"]/w, v["SQ_ACTIVE_INST_ANY"]/w,
          v["SQ_INSTS_VALU"]/n, v["SQ_INSTS_LDS"]/n, v["SQ_INSTS_SALU"]/n, u.get("SQ_INSTS_VMEM_RD",0)/n, u.get("SQ_INSTS_VMEM_WR",0)/n, u.get("SQ_INSTS_MFMA",0)/n))
PY
