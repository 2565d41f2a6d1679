#!/bin/bash
# SQ / GRBM counter pass of the GEMM probe (own rocprofv3 runs, kernel-trace only): effective clock and matrix-pipe busy share.
#   bash scripts/micro/gemm_gen2_pmc.sh <shape index> <outdir>
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
shape=${1:-0}; out=${2:-gpurun_out/g2pmc}; rm -rf $out; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I roitr_amd/csrc -o /tmp/gemm_gen2 scripts/micro/gemm_gen2.hip -ldl 2>/dev/null || exit 1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $out/a -o s -- /tmp/gemm_gen2 roitr_amd/lib/libroitr_hip.so rnd $shape > $out/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/b -o s -- /tmp/gemm_gen2 roitr_amd/lib/libroitr_hip.so rnd $shape > $out/b.log 2>&1
python3 - <<PY
import csv,collections,glob,re
def load(d):
    t=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); k=re.sub(r"^void ","",k).split("(")[0]
            t[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            if r["Counter_Name"]=="SQ_WAVES" or r["Counter_Name"]=="SQ_INSTS_VALU": n[k]+=1
    return t,n
def durations(d):
    t=collections.defaultdict(list)
    for f in glob.glob(d+"/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); k=re.sub(r"^void ","",k).split("(")[0]
            t[k].append((float(r["End_Timestamp"])-float(r["Start_Timestamp"]))*1e-3)
    return t
a,na=load("$out/a"); b,nb=load("$out/b"); du=durations("$out/a")
print("%-34s %5s %8s %7s | %6s %6s %6s %6s | per wave: %6s %6s %6s %6s | lds: %6s %6s"%("kernel","n","us","GHz","mfma%","wait","w_inst","active","mfma","valu","salu","lds","w_lds","confl"))
for k in sorted(a, key=lambda k:-a[k]["SQ_WAVE_CYCLES"]):
    v=a[k]; n=na[k] or 1; u=b.get(k,{}); w=v["SQ_WAVE_CYCLES"] or 1; wv=v["SQ_WAVES"] or 1
    us=sum(du[k])/max(len(du[k]),1)
    ghz=v["GRBM_GUI_ACTIVE"]/n/(us*1e3) if us else 0
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over SIMDs?  report against GRBM_GUI_ACTIVE x 1024 SIMDs and x 256 CUs
    mf=v["SQ_VALU_MFMA_BUSY_CYCLES"]/(v["GRBM_GUI_ACTIVE"]*1024) if v["GRBM_GUI_ACTIVE"] else 0
    bw=b.get(k,{}); wb=1
    print("%-34s %5d %8.1f %7.3f | %6.3f %6.3f %6.3f %6.3f | %16.0f %6.0f %6.0f %6.0f | %6.3f %6.3f"%(k[:34],n,us,ghz,mf,v["SQ_WAIT_ANY"]/w,v["SQ_WAIT_INST_ANY"]/w,v["SQ_ACTIVE_INST_ANY"]/w,
          v["SQ_INSTS_MFMA"]/wv,u.get("SQ_INSTS_VALU",0)/wv,u.get("SQ_INSTS_SALU",0)/wv,u.get("SQ_INSTS_LDS",0)/wv, u.get("SQ_WAIT_INST_LDS",0)/w, (u.get("SQ_LDS_BANK_CONFLICT",0)/u.get("SQ_LDS_IDX_ACTIVE",1)) if u.get("SQ_LDS_IDX_ACTIVE",0) else 0))
PY
