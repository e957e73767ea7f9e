set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_codec
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_codec -o c -- python tools/codec_prof.py > gpurun_out/prof_codec.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_codec/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gaussian_" in r["Kernel_Name"] or "compact" in r["Kernel_Name"]]
with open("gpurun_out/codec_kernels.txt", "w") as o:
    for r in rows:
        o.write(f'{r["Kernel_Name"].split("(")[0]:28s} grid {r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size")} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6:9.3f} ms\n')
print(open("gpurun_out/codec_kernels.txt").read())
PY
