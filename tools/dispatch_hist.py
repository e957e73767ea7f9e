"""Per-dispatch view of a rocprofv3 --kernel-trace csv: for kernels whose name contains a pattern, group the
dispatches by grid size and print count / total time.  python tools/dispatch_hist.py <dir> <pattern> [steps]"""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))
        agg[g][0] += 1
        agg[g][1] += dur
tot = sum(v[1] for v in agg.values())
print(f"# {pat}: {sum(v[0] for v in agg.values())/steps:.1f} dispatches/step, {tot/steps:.1f} us/step")
for g, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"grid {g:>10d}  {n/steps:6.1f}/step  avg {t/n:8.1f} us  total/step {t/steps:8.1f} us")
