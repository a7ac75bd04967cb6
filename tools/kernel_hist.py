# histogram of one kernel's durations from a rocprofv3 kernel trace CSV: python tools/kernel_hist.py <csv> <name-substring>
import csv, sys, collections
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if sys.argv[2] in r["Kernel_Name"]]) / 1e3
print(f"{sys.argv[2]}: n={d.size} mean={d.mean():.1f} us median={np.median(d):.1f} p90={np.percentile(d, 90):.1f} max={d.max():.1f} sum={d.sum() / 1e3:.1f} ms")
edges = [0, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1e9]
h, _ = np.histogram(d, edges)
for lo, hi, n in zip(edges[:-1], edges[1:], h):
    if n: print(f"  {lo:6.0f}..{hi:<6.0f} us: {n:7d}  ({d[(d >= lo) & (d < hi)].sum() / 1e3:8.1f} ms)")
