#!/bin/bash
# one-pair-per-call forward: kernel time vs gaps on the device timeline (rocprofv3 kernel trace of bench.py --pairs-per-step 1)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
out=${1:-gpurun_out/b1}; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/t -o k -- python bench.py --pairs-per-step 1 --steps 40 --warmup 10 --no-cpu-baseline --no-single-pair --no-profile-pass --no-rccl-selftest > $out/log.txt 2>&1
python - <<PY
import csv,glob,collections,re
rows=[]
for f in glob.glob("$out/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]).split("(")[0][-40:], r.get("Queue_Id","")))
rows.sort()
# take the last 20 forwards: the first transformer of the network (one launch per forward) marks them
marks=[i for i,r in enumerate(rows) if "local_first_kernel" in r[2]]
if len(marks)>22:
    a,b=marks[-21],marks[-1]
    seg=rows[a:b]; nf=20
    wall=(seg[-1][1]-seg[0][0])/1e3
    busy=0; cur_s,cur_e=seg[0][0],seg[0][1]
    for s,e,n,q in seg[1:]:
        if s>cur_e: busy+=cur_e-cur_s; cur_s,cur_e=s,e
        else: cur_e=max(cur_e,e)
    busy+=cur_e-cur_s
    print("forwards",nf,"wall per forward %.1f us, device busy (union of kernels) %.1f us, kernels per forward %.0f, sum of kernel durations %.1f us"%(wall/nf,busy/1e3/nf,len(seg)/nf,sum(e-s for s,e,_,_ in seg)/1e3/nf))
    t=collections.defaultdict(lambda:[0,0])
    for s,e,n,q in seg: t[n][0]+=e-s; t[n][1]+=1
    for n,(d,c) in sorted(t.items(),key=lambda kv:-kv[1][0])[:25]:
        print("%-42s calls/fwd %5.1f  us/fwd %7.1f  avg %6.1f"%(n,c/nf,d/1e3/nf,d/1e3/c))
    # one forward, kernel by kernel: start offset, duration, gap to the previous end on the same queue, queue id
    a,b=marks[-2],marks[-1]
    t0=rows[a][0]; last={}
    print("--- one forward (us): start  dur  gap_same_queue  queue  kernel")
    for s,e,n,q in rows[a:b]:
        g=(s-last[q])/1e3 if q in last else 0.0
        last[q]=e
        print("%8.1f %7.1f %6.1f  q%-3s %s"%((s-t0)/1e3,(e-s)/1e3,g,q,n))
PY
