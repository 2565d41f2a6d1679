"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per kernel family.

    python scripts/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [pairs_per_step] [steps] [config]

`steps` = engine forwards the profiled command ran (warm-up included): `total_hbm_bytes_per_step` = every kernel's bytes / steps.

Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md "HBM [CDNA4]": FETCH_SIZE and WRITE_SIZE are
reported in KiB-ish units of 1024 B; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is DOUBLED for wide
coalesced reads (all our streaming kernels use 16 B/lane loads).  WRITE_SIZE is taken as reported (uncalibrated)."""
import collections, csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roitr_amd.build import source_hash

def load(path, counter):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"[<(].*", "", name)
        tot[name] += float(r["Counter_Value"]); n[name] += 1
    return tot, n

f, fn = load(sys.argv[1], "FETCH_SIZE")
w, wn = load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[5]) if len(sys.argv) > 5 else None
out = {"pairs_per_step": int(sys.argv[4]) if len(sys.argv) > 4 else None, "baseline_config": int(sys.argv[6]) if len(sys.argv) > 6 else 2,
       "forwards_profiled": steps, "kernel_source_sha16": source_hash(),
       "note": "bytes per launch; fetch = FETCH_SIZE*1024*2 (gfx950 half-count correction), write = WRITE_SIZE*1024", "kernels": {}}
for k in sorted(f, key=lambda k: -f[k]):
    if k.startswith("__amd") or "at::" in k:
        continue
    fb = f[k] * 1024 * 2 / max(fn[k], 1)
    wb = w.get(k, 0.0) * 1024 / max(wn.get(k, 1), 1)
    out["kernels"][k] = {"launches": fn[k], "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                         "hbm_bytes_per_launch": round(fb + wb)}
if steps:
    tot = sum(v * 1024 * 2 for k, v in f.items() if not k.startswith("__amd")) + sum(v * 1024 for k, v in w.items() if not k.startswith("__amd"))
    out["total_hbm_bytes_per_step"] = round(tot / steps)
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out["kernels"].items())[:14]:
    print(f"{k:28s} launches {v['launches']:5d}  fetch {v['fetch_bytes_per_launch']/1e6:9.2f} MB  write {v['write_bytes_per_launch']/1e6:9.2f} MB")
