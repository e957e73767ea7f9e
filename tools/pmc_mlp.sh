#!/bin/bash
# PMC passes for the MFMA kernels (mlp3_*, mlp2_*, wgrad_multi): MFMA pipe busy, waits, memory-instruction cycles, L2->EA
# write stalls.  Output gpurun_out/pmc_mlp.txt.  No trace flags beside --pmc (gpurun rule).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { rm -rf /tmp/pm_$1; (cd /tmp && rocprofv3 --pmc $2 --output-format csv -d /tmp/pm_$1 -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps > /dev/null 2>&1); }
run a "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS"
run b "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"
run c "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_IB_STALL"
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for p in "abc":
    seen = collections.Counter()
    for f in glob.glob(f"/tmp/pm_{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            k = k[:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if p == "a" and r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
names = sorted(set(c for a in agg.values() for c in a))
with open("gpurun_out/pmc_mlp.txt", "w") as f:
    for k, a in sorted(agg.items()):
        if not any(s in k for s in ("mlp3", "mlp2", "wgrad_multi_kernel", "expand_bwd", "expand_write")): continue
        n = max(1, cnt[k])
        f.write(f"{k}  dispatches={cnt[k]}\n")
        for c in names:
            if c in a: f.write(f"    {c:40s} {a[c]/n:16.0f} per dispatch\n")
print(open("gpurun_out/pmc_mlp.txt").read()[:6000])
PY
