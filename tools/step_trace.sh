# Ordered kernel list of ONE timed step (the second-to-last) of a short bench run: start offset, duration, gap before, name
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -o g -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-codec --no-image-loss --no-raster-only --no-heavy --no-eval-fps "$@" > gpurun_out/step_trace.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/prof_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
short = lambda n: n.replace("void ", "").replace("at::native::", "")[:110]
starts = [i for i, r in enumerate(rows) if r[2].startswith("void preprocess_kernel<true>")]
a, b = starts[-3], starts[-2]
seg = rows[a:b]
t0 = seg[0][0]
prev_end = t0
with open("gpurun_out/step_trace.txt", "w") as o:
    o.write(f"# one step: {len(seg)} launches, {(rows[b][0]-t0)/1e6:.3f} ms wall, busy {sum(e-s for s,e,_ in seg)/1e6:.3f} ms\n")
    for s, e, n in seg:
        o.write(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  gap {(s-prev_end)/1e3:6.1f}  {short(n)}\n")
        prev_end = max(prev_end, e)
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in seg:
        k = short(n).split("(")[0][:80]
        agg[k][0] += 1; agg[k][1] += e - s
    o.write("\n# by kernel\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write(f"{c:4d} {t/1e3:9.1f} us  {k}\n")
print(open("gpurun_out/step_trace.txt").read()[:300])
PY
