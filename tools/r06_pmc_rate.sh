#!/bin/bash
# SQ counters of the fused rate-subset kernels from the micro-benchmark (one --pmc pass, no trace flags)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/pm_rs
(cd /tmp && timeout -k 5 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pm_rs -o s -- python $GRAFT_REPO_ROOT/tools/rate_sub_micro.py 807417 121000 71 > /dev/null 2>&1)
rm -rf /tmp/pm_rs2
(cd /tmp && timeout -k 5 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pm_rs2 -o s -- python $GRAFT_REPO_ROOT/tools/rate_sub_micro.py 807417 121000 71 > /dev/null 2>&1)
python - <<'PY' | tee gpurun_out/r06_pmc_rate.txt
import csv, glob, collections
for d in ("/tmp/pm_rs", "/tmp/pm_rs2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    first = None
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0][:40]
            if not any(s in k for s in ("rs_main", "rs_wgrad", "level_rate", "mlp2_")): continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            first = first or r["Counter_Name"]
            if r["Counter_Name"] == first: cnt[k] += 1
    for k, a in sorted(agg.items()):
        n = max(1, cnt[k])
        if "SQ_WAVE_CYCLES" in a:
            wc = a["SQ_WAVE_CYCLES"] / n
            print(f"{k:42s} dispatches {cnt[k]:3d}  wave cycles {wc:14.0f}  " + "  ".join(f"{c[3:]} {a[c] / n / wc:.3f}" for c in sorted(a) if c != "SQ_WAVE_CYCLES"))
        else:
            print(f"{k:42s} dispatches {cnt[k]:3d}  " + "  ".join(f"{c[3:]} {a[c] / n:.0f}" for c in sorted(a)))
PY
