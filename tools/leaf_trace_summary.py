"""Digest of a rocprofv3 --kernel-trace of `tools/kernel_cases.py dyn_nuts` (the dynamic model's chain on
the device: evaluation + the wide leaf's two launches per leapfrog): duration distribution per kernel,
so that the launches that also ADVANCE the chain (end of a doubling / of a transition) show up beside
the common ones.   python tools/leaf_trace_summary.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
import numpy as np

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, d in sorted(by.items()):
    d = np.array(d)
    print(f"{name:28s} n={d.size:5d}  median {np.median(d):7.1f}  mean {d.mean():7.1f}  p90 {np.percentile(d, 90):7.1f}  max {d.max():7.1f} us")
b = np.array(by.get("nd::kw_leaf", [0.0]))
slow = b[b > 3 * np.median(b)]
print(f"kw_leaf launches that advance the chain (> 3 x median): {slow.size} of {b.size}, median {np.median(slow) if slow.size else 0:.1f} us, "
      f"{slow.sum() / max(b.sum(), 1e-9):.0%} of its total time")
ts = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
print(f"span {(ts[-1][1] - ts[0][0]) / 1e3:.0f} us, busy {sum(e - s for s, e in ts) / 1e3:.0f} us")
