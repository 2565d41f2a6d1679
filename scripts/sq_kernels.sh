#!/bin/bash
# SQ counter passes (own rocprofv3 runs, kernel-trace only) of one short bench step, rows filtered by a kernel-name pattern:
#   bash scripts/sq_kernels.sh <pattern> <outdir> [bench args...]
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
pat=${1:-knn}; out=${2:-gpurun_out/sqk}; shift 2
rm -rf $out; mkdir -p $out
P="python bench.py --no-cpu-baseline --no-single-pair --no-rccl-selftest --no-profile-pass --steps 1 --warmup 1 $*"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/a -o s -- $P > $out/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $out/b -o s -- $P > $out/b.log 2>&1
python3 - <<PY
import csv,collections,re,glob
def load(d):
    t=collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); k=re.sub(r"^void ","",k).split("(")[0]
            t[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    return t
a=load("$out/a"); b=load("$out/b")
print("%-40s %9s %6s %6s %6s | per wave: %7s %6s %6s %7s %6s | lds: %6s %6s %6s" % ("kernel","wavecyc","wait","w_inst","active","valu","lds","salu","vmem_rd","vm_wr","w_lds","act","confl"))
for k in sorted(a, key=lambda k:-a[k]["SQ_WAVE_CYCLES"]):
    if not re.search("$pat", k): continue
    v=a[k]; w=v["SQ_WAVE_CYCLES"] or 1; n=v["SQ_WAVES"] or 1; u=b.get(k,{})
    print("%-40s %9.3e %6.3f %6.3f %6.3f | %17.0f %6.0f %6.0f %7.1f %6.1f | %6.3f %6.3f %6.3f" % (k[:40], w, v["SQ_WAIT_ANY"]/w, v["SQ_WAIT_INST_ANY"]/w, v["SQ_ACTIVE_INST_ANY"]/w,
          v["SQ_INSTS_VALU"]/n, v["SQ_INSTS_LDS"]/n, v["SQ_INSTS_SALU"]/n, u.get("SQ_INSTS_VMEM_RD",0)/n, u.get("SQ_INSTS_VMEM_WR",0)/n,
          u.get("SQ_WAIT_INST_LDS",0)/w, u.get("SQ_ACTIVE_INST_LDS",0)/w, (u.get("SQ_LDS_BANK_CONFLICT",0)/u["SQ_LDS_IDX_ACTIVE"]) if u.get("SQ_LDS_IDX_ACTIVE") else 0))
PY
