#!/bin/bash
# VALU issue of the blend kernels at HEAD -> gpurun_out/pmc_valu.json (copy to profiles/): two rocprofv3 --pmc passes (no trace
# flags) of a short headline bench; valu_busy_frac = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel duration),
# duration = GRBM_GUI_ACTIVE / 8 XCDs.  The file records the sha256 of the blend kernels' sources (bench.py attaches the figure
# only while they are unchanged).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps"
rm -rf /tmp/pmc_v1 /tmp/pmc_v2
timeout -k 5 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc_v1 -o s -- $CMD > /dev/null 2>&1
timeout -k 5 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_v2 -o s -- $CMD > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, json, hashlib, os
def collect(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
            if "blend" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    return {k: {c: v / max(1, cnt[(k, c)]) for c, v in a.items()} for k, a in agg.items()}
a, b = collect("/tmp/pmc_v1"), collect("/tmp/pmc_v2")
names = {"blend_bwd_rows_kernel": "blend_bwd", "blend_fwd_rows_kernel": "blend_fwd"}
out, raw = {}, {}
for k, n in names.items():
    if k in a and k in b and b[k].get("GRBM_GUI_ACTIVE"):
        dur = b[k]["GRBM_GUI_ACTIVE"] / 8.0
        out[n] = round(a[k]["SQ_INSTS_VALU"] * 4.0 / (1024.0 * dur), 3)
        raw[n] = {"SQ_INSTS_VALU": a[k]["SQ_INSTS_VALU"], "SQ_INSTS_LDS": a[k].get("SQ_INSTS_LDS"), "duration_cycles": dur,
                  "SQ_LDS_BANK_CONFLICT": b[k].get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": b[k].get("SQ_LDS_IDX_ACTIVE")}
csrc = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "contextgs_amd", "csrc")
dig = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest() for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h", ".cpp"))}
json.dump({"_source": "tools/pmc_valu_gen.sh: rocprofv3 --pmc passes on the headline step of `python bench.py` (1 M anchors, 1920x1080, MI355X), "
                      "the kernels at the commit whose sources hash to file_digests; valu_busy_frac = SQ_INSTS_VALU x 4 cycles / "
                      "(1024 SIMDs x kernel duration), duration = GRBM_GUI_ACTIVE / 8 XCDs",
           "valu_busy_frac": out, "per_dispatch": raw,
           "_note": "NOMINAL figure: instructions x 4 cycles over SIMD time, not a measured pipe occupancy", "file_digests": dig},
          open("gpurun_out/pmc_valu.json", "w"), indent=1)
print(json.dumps({"valu_busy_frac": out, "per_dispatch": raw}))
PY
