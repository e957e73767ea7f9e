# HBM traffic of the dominant kernels via PMC counters, collected in their OWN passes (no trace flags), as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots).
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CMD="timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-raster-only --no-codec --no-heavy --no-eval-fps --no-image-loss"
rm -rf /tmp/pmc_r /tmp/pmc_w
timeout -k 5 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_r -o r -- $CMD > gpurun_out/pmc_r.log 2>&1
timeout -k 5 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- $CMD > gpurun_out/pmc_w.log 2>&1
python - <<'PY'
import csv, glob, collections
out = []
for tag, d in (("FETCH_SIZE", "/tmp/pmc_r"), ("WRITE_SIZE", "/tmp/pmc_w")):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != tag:
                continue
            k = r["Kernel_Name"].split("(")[0]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    out.append((tag, agg))
with open("gpurun_out/pmc_hbm_summary.txt", "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per-dispatch averages in KiB-units as reported\n")
    f.write("# NOTE (MI355X_MICROARCH.md): on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x\n")
    for tag, agg in out:
        f.write(f"\n[{tag}]\n")
        for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
            f.write(f"{n:6d} dispatches  avg {tot/n:14.1f}  total {tot:16.1f}  {k[:90]}\n")
# bytes per launch for bench.py (profiles/pmc_traffic.json): FETCH_SIZE x 2 (gfx950 correction for wide
# streaming reads, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both reported in KiB
import json
names = {"blend_bwd_kernel": "blend_bwd", "blend_fwd_kernel": "blend_fwd",
         "blend_bwd_rows_kernel": "blend_bwd", "blend_fwd_rows_kernel": "blend_fwd", "preprocess_kernel": "preprocess_plain",
         "expand_preprocess_kernel": "preprocess",
         "preprocess_bwd_kernel": "preprocess_bwd", "expand_bwd_kernel": "expand_bwd", "expand_write_kernel": "expand_fwd",
         "mlp3_bwd_kernel": "mlp3_bwd_data_only", "mlp3_bwd_wg_kernel": "mlp3_bwd", "mlp3_fwd_kernel": "mlp3_fwd", "wgrad4_kernel": "wgrad4"}
rd, wr = dict(out)["FETCH_SIZE"], dict(out)["WRITE_SIZE"]
tr = {}
for k, (n, tot) in rd.items():
    base = k.replace("void ", "").split("<")[0].strip()
    if base in names and k in wr:
        tr[names[base]] = int((2 * tot / n + wr[k][1] / wr[k][0]) * 1024)
# the context model's level loop as a group: (2 FETCH + WRITE) of every dispatch of its kernels, per training step (one step =
# two dispatches of the level-0/1 backward kernel, or of noise_quant_bwd on the separate-launch path)
ctx_keys = ("ctxl_", "rs_main", "rs_wgrad", "level_rate_", "mlp2_", "wgrad_multi", "wgrad_reduce", "wgrad_tail", "mlp_small_reduce", "ctx_gather_bwd", "eb_bits_", "rowcat_",
            "rowgather4", "gather_rows_segmented", "hyper_noise", "ctx_choose", "noise_quant_", "means_finalize", "rate_finish")
tot_ctx, steps_ctx = 0.0, 0
for k, (n, tot) in rd.items():
    if any(s_ in k for s_ in ctx_keys) and k in wr:
        tot_ctx += (2 * tot + wr[k][1]) * 1024
    if "ctxl_bwd_kernel<71>" in k:
        steps_ctx = n // 2
import hashlib, os
csrc = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "contextgs_amd", "csrc")
digests = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest() for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h", ".cpp"))}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch",
           "workload": "bench.py default (1M anchors, 1920x1080, full training step)", "bytes_per_launch": tr,
           "ctx_group_bytes_per_step": (int(tot_ctx / steps_ctx) if steps_ctx else None), "ctx_group_steps": steps_ctx,
           "file_digests": digests},
          open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(open("gpurun_out/pmc_hbm_summary.txt").read()[:3000])
print(json.dumps(tr))
PY
