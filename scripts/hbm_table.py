"""Per-kernel HBM GB/s = PMC bytes per launch (profiles/r01_pmc_traffic.json) / average launch duration (profiles/r01_kernel_stats.csv).

    python scripts/hbm_table.py profiles r01 > profiles/r01_hbm_gbs.txt
"""
import collections, csv, json, re, sys
d, tag = sys.argv[1], sys.argv[2]
pmc = json.load(open(f"{d}/{tag}_pmc_traffic.json"))["kernels"]
dur = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f"{d}/{tag}_kernel_stats.csv")):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"[<(].*", "", n)
    dur[n][0] += float(r["TotalDurationNs"]); dur[n][1] += int(r["Calls"])
print(f"{'kernel':28s} {'avg launch us':>14s} {'fetch MB':>10s} {'write MB':>10s} {'HBM GB/s':>10s} {'% of 8 TB/s':>12s}")
rows = []
for k, v in pmc.items():
    if k in dur and dur[k][1]:
        us = dur[k][0] / dur[k][1] / 1e3
        gbs = v["hbm_bytes_per_launch"] / (us * 1e-6) / 1e9
        rows.append((dur[k][0], k, us, v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6, gbs))
for _, k, us, f, w, gbs in sorted(rows, reverse=True)[:24]:
    print(f"{k:28s} {us:14.1f} {f:10.1f} {w:10.1f} {gbs:10.0f} {100 * gbs / 8000:11.1f}%")
