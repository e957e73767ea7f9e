# per-launch durations of the MLP-family kernels in one training step (kernel trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_mlp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_mlp -o m -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-codec --no-image-loss --no-raster-only > gpurun_out/mlp_launches.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_mlp/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", ""))) for r in csv.DictReader(open(f))))
starts = [i for i, r in enumerate(rows) if "filter_voxel_kernel" in r[2]]
a, b = starts[-2], starts[-1]
with open("gpurun_out/mlp_launches.txt", "w") as o:
    for s, e, n, g, w in rows[a:b]:
        if any(k in n for k in ("mlp2", "mlp3", "wgrad", "mlp_small")):
            o.write(f"{(e - s) / 1e3:9.1f} us  grid {g:>8s} wg {w:>5s}  {n.split('(')[0][:70]}\n")
print(open("gpurun_out/mlp_launches.txt").read())
PY
