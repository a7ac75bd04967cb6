# kernel durations and inter-kernel gaps from a rocprofv3 --kernel-trace CSV
import sys, csv, glob, collections
import numpy as np
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = collections.Counter(); dur = collections.defaultdict(list)
gaps = collections.defaultdict(list)
for i, r in enumerate(rows):
    n = r['Kernel_Name'][:60]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    dur[n].append(e - s)
    if i: gaps[n].append(s - int(rows[i - 1]['End_Timestamp']))
for n in dur:
    d = np.array(dur[n]); g = np.array(gaps[n]) if gaps[n] else np.array([0])
    print(f"{n:60s} n={len(d):7d} dur avg {d.mean()/1e3:8.2f} us med {np.median(d)/1e3:8.2f}  gap-before avg {g.mean()/1e3:8.2f} med {np.median(g)/1e3:8.2f}")
