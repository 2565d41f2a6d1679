#!/bin/bash
# SQ / cache counter passes (own rocprofv3 runs, kernel-trace only) of an arbitrary command, rows filtered by a kernel-name pattern:
#   bash scripts/sq_cmd.sh <pattern> <outdir> <command...>
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
pat=${1:-knn}; out=${2:-gpurun_out/sqc}; shift 2
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/a -o s -- "$@" > $out/a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $out/b -o s -- "$@" > $out/b.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/c -o s -- "$@" > $out/c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $out/d -o s -- "$@" > $out/d.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/e -o s -- "$@" > $out/e.log 2>&1
python3 - <<PY
import csv,collections,re,glob
def load(d):
    t=collections.defaultdict(lambda: collections.defaultdict(float)); c=collections.Counter()
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); k=re.sub(r"^void ","",k).split("(")[0]
            t[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    return t
a=load("$out/a"); b=load("$out/b"); c=load("$out/c"); d=load("$out/d"); e5=load("$out/e")
print("%-34s %9s %6s %6s %6s | per wave: %7s %6s %6s %6s %7s %6s | actVALU | per wave: L1acc  L1->L2  L2hit  | TA_BUSY_avr TCPstall/GUI" % ("kernel","wavecyc","wait","w_inst","active","valu","lds","salu","smem","vmem_rd","vm_wr"))
for k in sorted(a, key=lambda k:-a[k]["SQ_WAVE_CYCLES"]):
    if not re.search("$pat", k): continue
    v=a[k]; w=v["SQ_WAVE_CYCLES"] or 1; n=v["SQ_WAVES"] or 1; u=b.get(k,{}); x=c.get(k,{}); y=d.get(k,{})
    hit=x.get("TCC_HIT_sum",0); mis=x.get("TCC_MISS_sum",0)
    print("%-34s %9.3e %6.3f %6.3f %6.3f | %17.0f %6.0f %6.0f %6.0f %7.1f %6.1f | %7.3f | %15.0f %7.0f %6.3f | %11.3g %8.3f" % (k[:34], w, v["SQ_WAIT_ANY"]/w, v["SQ_WAIT_INST_ANY"]/w, v["SQ_ACTIVE_INST_ANY"]/w,
          v["SQ_INSTS_VALU"]/n, v["SQ_INSTS_LDS"]/n, v["SQ_INSTS_SALU"]/n, u.get("SQ_INSTS_SMEM",0)/n, u.get("SQ_INSTS_VMEM_RD",0)/n, u.get("SQ_INSTS_VMEM_WR",0)/n,
          u.get("SQ_ACTIVE_INST_VALU",0)/w, x.get("TCP_TOTAL_CACHE_ACCESSES_sum",0)/n, x.get("TCP_TCC_READ_REQ_sum",0)/n, hit/(hit+mis) if hit+mis else 0,
          y.get("TA_BUSY_avr",0), (y.get("TCP_PENDING_STALL_CYCLES_sum",0)/y["GRBM_GUI_ACTIVE"]) if y.get("GRBM_GUI_ACTIVE") else 0))
    z=e5.get(k,{})
    if z: print("    mfma/wave %.0f  MFMA_BUSY_CYCLES/BUSY_CU_CYCLES %.3f  GUI_ACTIVE %.3g  TA_BUSY_sum/GUI %.1f" % (z.get("SQ_INSTS_MFMA",0)/n, z.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/max(z.get("SQ_BUSY_CU_CYCLES",1),1), y.get("GRBM_GUI_ACTIVE",0), y.get("TA_TA_BUSY_sum",0)/max(y.get("GRBM_GUI_ACTIVE",1),1)))
PY
