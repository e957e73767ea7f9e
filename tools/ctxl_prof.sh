#!/bin/bash
# micro-benchmark + SQ counters of the fused level kernels.  Output gpurun_out/ctxl_micro.txt, ctxl_pmc.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
N=${1:-800000}
python tools/ctxl_micro.py $N 20 71 2>&1 | tee gpurun_out/ctxl_micro.txt
python tools/ctxl_micro.py 160000 20 71 2>&1 | tee -a gpurun_out/ctxl_micro.txt
python tools/ctxl_micro.py 40000 20 15 2>&1 | tee -a gpurun_out/ctxl_micro.txt
run() { rm -rf /tmp/pm_$1; (cd /tmp && rocprofv3 --pmc $2 --output-format csv -d /tmp/pm_$1 -o s -- python $GRAFT_REPO_ROOT/tools/ctxl_micro.py $N 2 71 > /dev/null 2>&1); }
run a "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS"
run b "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"
run c "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_WAVES"
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for p in "abc":
    for f in glob.glob(f"/tmp/pm_{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if p == "a" and r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
names = sorted(set(c for a in agg.values() for c in a))
with open("gpurun_out/ctxl_pmc.txt", "w") as f:
    for k, a in sorted(agg.items()):
        if "ctxl" not in k: continue
        n = max(1, cnt[k])
        f.write(f"{k}  dispatches={cnt[k]}\n")
        for c in names:
            if c in a: f.write(f"    {c:40s} {a[c]/n:16.0f} per dispatch\n")
print(open("gpurun_out/ctxl_pmc.txt").read()[:6000])
PY
