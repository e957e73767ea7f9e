# SQ issue/stall breakdown of the blend kernels (SURVEY §8d "honest ceiling note"): one PMC pass, no trace flags.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/pmc_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU \
  --output-format csv -d /tmp/pmc_sq -o s -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-raster-only --no-codec > gpurun_out/pmc_sq.log 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob("/tmp/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k] += 1
want = ["blend_bwd_rows_kernel", "blend_fwd_rows_kernel", "blend_bwd_kernel", "blend_fwd_kernel", "mlp3_bwd_kernel", "mlp3_fwd_kernel", "wgrad_multi_kernel", "mlp2_bwd_kernel",
        "expand_bwd_kernel", "preprocess_kernel", "radix_scatter_kernel", "noise_quant_fwd_kernel", "rowcat_fwd_kernel"]
with open("gpurun_out/pmc_sq_summary.txt", "w") as f:
    f.write("# rocprofv3 --pmc SQ_* (one pass), sums over the dispatches of a kernel; fractions of SQ_WAVE_CYCLES\n")
    f.write("# (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, MI355X_MICROARCH.md 'PMC slots')\n")
    f.write(f"{'kernel':28s} {'disp':>5s} {'active_any':>10s} {'act_valu':>9s} {'act_lds':>8s} {'wait_any':>9s} {'wait_inst':>9s} {'wait_lds':>9s} {'valu_inst/wave_cyc':>19s}\n")
    for k in want:
        a = agg.get(k)
        if not a or not a["SQ_WAVE_CYCLES"]:
            continue
        w = a["SQ_WAVE_CYCLES"]
        f.write(f"{k:28s} {cnt[k]:5d} {a['SQ_ACTIVE_INST_ANY']/w:10.3f} {a['SQ_ACTIVE_INST_VALU']/w:9.3f} {a['SQ_ACTIVE_INST_LDS']/w:8.3f} "
                f"{a['SQ_WAIT_ANY']/w:9.3f} {a['SQ_WAIT_INST_ANY']/w:9.3f} {a['SQ_WAIT_INST_LDS']/w:9.3f} {a['SQ_INSTS_VALU']/w:19.4f}\n")
print(open("gpurun_out/pmc_sq_summary.txt").read())
PY
