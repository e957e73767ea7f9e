# SQ counters of the blend kernels (one PMC pass per counter group, no trace flags): usage tools/pmc_blend.sh [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL"; do
  rm -rf /tmp/pmc_b
  env "$@" rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_b -o s -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps > /dev/null 2>&1
  python - "$grp" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/pmc_b/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
        if "blend" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, a in agg.items():
    print(k, {c: (round(v / max(1, cnt[(k, c)])), ) for c, v in a.items()})
PY
done
