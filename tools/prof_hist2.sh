cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_h
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_h -o h -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-raster-only --no-codec > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
rows = []
for f in glob.glob("/tmp/prof_h/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append(r)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for k, r in enumerate(rows):
    if "reduce_kernel" in r["Kernel_Name"] and int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) <= 8:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if dur > 100:
            nm = r["Kernel_Name"]
            i = nm.find("ReduceOp<"); 
            key = nm[i:i+110]
            prev = rows[k-1]["Kernel_Name"][:60]; nxt = rows[k+1]["Kernel_Name"][:60] if k+1 < len(rows) else ""
            agg[(key, prev, nxt)][0] += 1; agg[(key, prev, nxt)][1] += dur
for (key, prev, nxt), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print(n, round(t/n,1), key); print("    prev:", prev); print("    next:", nxt)
PY
