"""Summarise a rocprofv3 --kernel-trace --stats CSV pair: per-kernel totals (per forward) and GEMM shapes."""
import collections, csv, re, sys
d, prefix, nfwd = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = list(csv.DictReader(open(f"{d}/{prefix}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms over {nfwd} forwards = {tot/1e6/nfwd:.3f} ms/forward")
for r in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 24]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"\(.*", "", n)[:48]
    print(f"{n:48s} calls={int(r['Calls']):5d} ms/fwd={float(r['TotalDurationNs'])/1e6/nfwd:7.3f} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={float(r['Percentage']):5.1f}")
