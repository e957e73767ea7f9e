# GPU idle time inside the timed steps: kernel trace of a short bench run, then busy fraction and the largest gaps
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_gaps
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gaps -o g -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-codec --no-image-loss --no-raster-only > gpurun_out/gaps.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_gaps/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in csv.DictReader(open(f))))
# steps are delimited by the filter kernel (prefilter_voxel): take the last 5 full steps
starts = [i for i, r in enumerate(rows) if r[2].startswith("void preprocess_kernel<true>") or "filter" in r[2]]
starts = starts[-7:-1] if len(starts) >= 7 else starts
out = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    t0, t1 = seg[0][0], rows[b][0]
    busy, gaps, cur_end = 0, [], seg[0][0]
    for s, e, n in seg:
        if s > cur_end:
            gaps.append((s - cur_end, n))
        busy += max(0, e - max(s, cur_end))
        cur_end = max(cur_end, e)
    gaps.sort(reverse=True)
    out.append((t1 - t0, busy, len(seg), gaps[:6], sum(g for g, _ in gaps), sum(1 for g, _ in gaps if g > 20000)))
with open("gpurun_out/gaps.txt", "w") as o:
    for w, busy, n, top, tot, big in out:
        o.write(f"step {w/1e6:.3f} ms  busy {busy/1e6:.3f} ms ({busy/w:.1%})  {n} launches  idle {tot/1e6:.3f} ms  gaps>20us: {big}\n")
        for g, name in top:
            o.write(f"      gap {g/1e3:7.1f} us before {name}\n")
print(open("gpurun_out/gaps.txt").read())
PY
