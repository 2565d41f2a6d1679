"""Per-kernel VALU-issue share of the headline forward from the SQ counter pass of scripts/sq_pass.sh (kernels run serialised there):
    python scripts/sq_forward_json.py gpurun_out/<tag>/sq/a profiles/sq_forward.json
valu_issue_frac = SQ_INSTS_VALU x 2 cycles (a wave64 VALU instruction occupies its SIMD for 2 cycles, MI355X_MICROARCH.md "Execution
model") / (kernel time x 2.4 GHz x 1024 SIMDs): the share of the chip's VALU issue slots the launches of a kernel used -- the bound of
the selection kernels (kNN, FPS, OT), which move few bytes.  bench.py attaches the kNN rows to its `knn+ppf` roofline entry."""
import collections
import csv
import glob
import json
import re
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roitr_amd.build import source_hash


def main(src, dst):
    t = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    dur = collections.defaultdict(float)
    for f in set(glob.glob(src + "/**/*counter_collection.csv", recursive=True) + glob.glob(src + "/*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"^void ", "", k).split("(")[0]
            t[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES":
                n[k] += 1
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    res = {"kernel_source_sha16": source_hash(), "source": "rocprofv3 --kernel-trace --pmc SQ_* (scripts/sq_pass.sh pass a; kernels serialised by the counter collection)",
           "definition": "valu_issue_frac = SQ_INSTS_VALU * 2 cycles / (kernel time * 2.4 GHz * 1024 SIMDs)", "kernels": {}}
    for k, v in sorted(t.items(), key=lambda kv: -dur[kv[0]]):
        if not n[k] or dur[k] <= 0 or "rocclr" in k or "at::native" in k:
            continue
        w = v["SQ_WAVE_CYCLES"] or 1.0
        res["kernels"][k] = {"launches": n[k], "total_ms": round(dur[k] * 1e3, 3), "avg_us": round(dur[k] / n[k] * 1e6, 1),
                             "valu_per_wave": round(v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1.0), 1),
                             "wave_cycles_issuing": round(v["SQ_ACTIVE_INST_ANY"] / w, 3), "wave_cycles_waiting": round(v["SQ_WAIT_ANY"] / w, 3),
                             "valu_issue_frac": round(v["SQ_INSTS_VALU"] * 2.0 / (dur[k] * 2.4e9 * 1024), 4)}
    knn = {k: r for k, r in res["kernels"].items() if k.startswith("knn_")}
    if knn:
        tot = sum(r["total_ms"] for r in knn.values())
        res["knn_family"] = {"total_ms": round(tot, 3),
                             "valu_issue_frac": round(sum(r["valu_issue_frac"] * r["total_ms"] for r in knn.values()) / tot, 4)}
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res.get("knn_family"), indent=1))
    for k, r in list(res["kernels"].items())[:16]:
        print("%-46s %8.1f us  valu/wave %8.0f  issue %.3f" % (k[:46], r["avg_us"], r["valu_per_wave"], r["valu_issue_frac"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
