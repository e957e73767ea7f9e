#!/bin/bash
# SQ counters of the one-wave-per-SIMD kernels (mlp3_bwd_wg, ctxl_bwd) from a short headline bench (one --pmc pass, no trace flags)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/pm_ow
(cd /tmp && timeout -k 5 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pm_ow -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps > /dev/null 2>&1)
python - <<'PY' | tee gpurun_out/pmc_onewave.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/pm_ow/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0][:40]
        if not any(s in k for s in ("mlp3_bwd_wg", "ctxl_bwd", "ctxl_fwd", "mlp3_fwd")): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, a in sorted(agg.items()):
    n = max(1, cnt[k]); wc = a["SQ_WAVE_CYCLES"] / n
    print(f"{k:42s} dispatches {cnt[k]:3d}  wave cycles {wc:14.0f}  " + "  ".join(f"{c[3:]} {a[c] / n / wc:.3f}" for c in sorted(a) if c != "SQ_WAVE_CYCLES"))
PY
