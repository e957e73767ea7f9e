# instruction mix of the device coders (one PMC pass, no trace flags)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/pmc_codec
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/pmc_codec -o c -- python tools/codec_prof.py > gpurun_out/pmc_codec.log 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/pmc_codec/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "gaussian_" not in k: continue
        key = (k, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""), r["Dispatch_Id"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
with open("gpurun_out/pmc_codec_summary.txt", "w") as o:
    for key, a in sorted(agg.items(), key=lambda kv: int(kv[0][2])):
        waves = a["SQ_WAVES"] or 1
        o.write(f"{key[0]:26s} grid {key[1]:>8s} waves {waves:7.0f}  per-wave: cycles {a['SQ_WAVE_CYCLES']/waves:12.0f} valu {a['SQ_INSTS_VALU']/waves:10.0f} salu {a['SQ_INSTS_SALU']/waves:10.0f} smem {a['SQ_INSTS_SMEM']/waves:8.0f} lds {a['SQ_INSTS_LDS']/waves:8.0f} vmem_rd {a['SQ_INSTS_VMEM_RD']/waves:8.0f}\n")
print(open("gpurun_out/pmc_codec_summary.txt").read())
PY
