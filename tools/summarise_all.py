"""Digest of tools/profile_all.sh: one row per kernel of every case -- calls, average / minimum
duration (rocprofv3 --kernel-trace --stats), FETCH_SIZE / WRITE_SIZE per launch (separate PMC
passes), HBM-side bytes = 2 x FETCH + WRITE (gfx950 reports half of a wide coalesced read stream:
MI355X_MICROARCH.md, HBM section), and the two roofline fractions: on the ALGORITHMIC bytes (the
reference's storage format: SURVEY.md section 8d) and on the bytes the kernel actually has to read.
Writes kernels.json next to the markdown it prints."""
import csv, glob, json, os, sys

d = sys.argv[1]
PEAK = 8000.0  # GB/s
rows = []
print("| case | kernel | calls | avg us | min us | FETCH KB | WRITE KB | HBM-side KB (2F+W) | algorithmic KB | frac (algorithmic) | needed KB | frac (needed) |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for case_dir in sorted(glob.glob(os.path.join(d, "*/"))):
    case = os.path.basename(case_dir.rstrip("/"))
    meta = {}
    try:
        for ln in open(os.path.join(case_dir, "trace.log")):
            if ln.startswith("{"):
                meta = json.loads(ln)
    except OSError:
        pass
    ks = glob.glob(os.path.join(case_dir, "trace/**/*kernel_stats.csv"), recursive=True)
    if not ks:
        continue

    def pmc(sub, name):
        out = {}
        for f in glob.glob(os.path.join(case_dir, sub, "**/*counter_collection.csv"), recursive=True):
            acc = {}
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != name:
                    continue
                k = r["Kernel_Name"]
                a = acc.setdefault(k, [0.0, 0])
                a[0] += float(r["Counter_Value"]); a[1] += 1
            out = {k: v[0] / v[1] for k, v in acc.items() if v[1]}
        return out

    fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
    for r in csv.DictReader(open(ks[0])):
        name = r["Name"]
        if float(r["Percentage"]) < 1.0 or name.startswith("__amd_rocclr") or "at::native" in name:
            continue
        avg, mn = float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3
        f = next((v for k, v in fetch.items() if k.split("(")[0] == name.split("(")[0]), None)
        w = next((v for k, v in write.items() if k.split("(")[0] == name.split("(")[0]), None)
        hbm = (2 * f + w) if f is not None and w is not None else None
        alg = meta.get("algorithmic_bytes_per_launch")
        need = meta.get("library_copy_bytes_per_launch")
        main = any(s in name for s in ("dc_eval", "dc_vec_stream", "dyn_pass", "dyn_fused", "neu_fused", "neu_big", "predict_score_grid"))
        fa = alg / 1e3 / avg / PEAK if (alg and main) else None
        fn = need / 1e3 / avg / PEAK if (need and main) else None
        rows.append({"case": case, "kernel": name, "calls": int(r["Calls"]), "avg_us": avg, "min_us": mn,
                     "fetch_kb": f, "write_kb": w, "hbm_kb": hbm,
                     "algorithmic_kb": alg / 1024 if (alg and main) else None, "frac_algorithmic": fa,
                     "needed_kb": need / 1024 if (need and main) else None, "frac_needed": fn, "meta": meta})
        fmt = lambda v, p=1: "" if v is None else f"{v:.{p}f}"
        print(f"| {case} | `{name[:70]}` | {r['Calls']} | {avg:.2f} | {mn:.2f} | {fmt(f)} | {fmt(w)} | {fmt(hbm)} | "
              f"{fmt(rows[-1]['algorithmic_kb'])} | {fmt(fa, 3)} | {fmt(rows[-1]['needed_kb'])} | {fmt(fn, 3)} |")
json.dump(rows, open(os.path.join(d, "kernels.json"), "w"), indent=1)
