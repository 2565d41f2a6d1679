#!/bin/bash
# Host-side cost of the bench loop with EIGHT rank processes on one host: the pool offers one GPU per box, so the
# eight processes share cuda:0 (the device time per step is then ~8x a real rank's and says nothing) -- what is read is the HOST side:
# `host_ms_per_step` of every process (launch = packing + allocation + the engine call's ~350 launches, unpack = finish_batch after the
# device answered), each process pinned to its own 8 cores like one rank of an 8-GPU node, against the same process running alone.
#   PAIRS=128 bash scripts/host_share_8ranks.sh [outdir]     (eight 512-pair processes do not fit one GPU's 288 GB: 128 pairs per call; the host cost is per pair)
cd ${GRAFT_REPO_ROOT:-.}
out=${1:-gpurun_out/host8}; rm -rf $out; mkdir -p $out
ARGS="--pairs-per-step ${PAIRS:-128} --steps 6 --warmup 3 --no-cpu-baseline --no-single-pair --no-profile-pass --no-rccl-selftest"
nc=$(nproc)
taskset -c 0-7 timeout 300 python bench.py $ARGS > $out/alone.json 2> $out/alone.err
for r in 0 1 2 3 4 5 6 7; do
  lo=$(( (r * 8) % nc )); hi=$(( lo + 7 < nc ? lo + 7 : nc - 1 ))
  taskset -c $lo-$hi timeout 600 python bench.py $ARGS > $out/rank$r.json 2> $out/rank$r.err &
done
wait
python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
def line(f):
    try:
        return json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        return None
a = line(out + "/alone.json")
rows = [line(f"{out}/rank{r}.json") for r in range(8)]
res = {"cores_on_host": os.cpu_count(), "alone": None if not a else {"ms_per_step": a["ms_per_step"], "host_ms_per_step": a.get("host_ms_per_step")},
       "eight_processes_one_gpu": [None if not d else {"ms_per_step": d["ms_per_step"], "host_ms_per_step": d.get("host_ms_per_step")} for d in rows]}
ok = [d for d in rows if d]
if ok:
    res["mean_host_ms_per_step_at_8"] = {k: round(sum(d["host_ms_per_step"][k] for d in ok) / len(ok), 3) for k in ("launch", "unpack")}
json.dump(res, open(out + "/host_share.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
