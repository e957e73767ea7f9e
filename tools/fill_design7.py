"""Refresh DESIGN.md section 7 from a bench line: python tools/fill_design7.py [gpurun_out/r06_bench_1m.json] [n_gpu_tests]
(tools/design7_template.md holds the text with %(name)s slots; %% is a literal percent sign)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_bench_1m.json")
ntests = sys.argv[2] if len(sys.argv) > 2 else "451"
j = json.loads(open(src).read().strip().splitlines()[-1])
c = j["codec"]; v2 = c["container_v2"]; k = j["kernels"]; r = j["roofline"]; c3 = c["c3_500k"]
us = lambda name: k[name]["avg_us"]
launches = j.get("launches_per_step")
if launches is None:
    try:
        launches = open(os.path.join(ROOT, "profiles", "r06_launch_attribution.txt")).readline().split()[1]
    except Exception:
        launches = "?"
vals = dict(value="%.1f" % j["value"], ms="%.2f" % j["ms_per_step"], smin="%.3f" % j["timing"]["ms_per_step_min"],
            smax="%.3f" % j["timing"]["ms_per_step_max"], ro="%.1f" % j["value_raster_only"], mid="%.1f" % j["value_mid_phase_noise"],
            loss="%.1f" % j["value_with_l1_ssim_loss"], heavy="%.1f" % j["value_heavy_pairs"], hip="%.2f" % j["hip_kernel_ms_per_step"],
            gap="%.2f" % (j["ms_per_step"] - j["hip_kernel_ms_per_step"]), launches=str(launches),
            ctx="%.2f" % j["ctx_group_roofline"]["ms_per_step"], ctxl="%d" % round(j["ctx_group_roofline"]["launches_per_step"]),
            mlpms="%.2f" % j["mlp_group_roofline"]["ms_per_step"], mlpfrac="%.1f" % (100 * j["mlp_group_roofline"]["frac"]),
            dsort="%.0f" % us("depth_sort"), oscan="%.0f" % us("offsets_scan"), tbin="%.0f" % (us("emit_pairs") + us("tile_sort")),
            bms="%.2f" % (us("blend_bwd") / 1e3), fms="%.2f" % (us("blend_fwd") / 1e3),
            bfrac="%.1f" % (100 * r["frac"]), bgbs="%.0f" % r["achieved"], bus="%.0f" % r["avg_launch_us"],
            bvalu="%.2f" % (r.get("valu_nominal_frac") or float("nan")), ntests=ntests,
            e1="%.1f" % c["encode_Manchors_per_s"], d1="%.1f" % c["decode_Manchors_per_s"],
            e2="%.1f" % v2["encode_Manchors_per_s"], d2="%.1f" % v2["decode_Manchors_per_s"],
            e2ms="%.1f" % (v2["encode_s"] * 1e3), d2ms="%.1f" % (v2["decode_s"] * 1e3),
            e2runs=" / ".join("%.1f" % (x * 1e3) for x in v2["encode_s_runs"]),
            c3e1="%.1f" % c3["container_v1"]["encode_Manchors_per_s"], c3d1="%.1f" % c3["container_v1"]["decode_Manchors_per_s"],
            c3e2="%.1f" % c3["container_v2"]["encode_Manchors_per_s"], c3d2="%.1f" % c3["container_v2"]["decode_Manchors_per_s"],
            fps_d="%.0f" % c["test_fps"]["decoded_views_per_s"], fps_n="%.0f" % c["test_fps"]["not_decoded_views_per_s"],
            cpu="%.3f" % j["cpu_baseline"]["value"])
t = open(os.path.join(ROOT, "tools", "design7_template.md")).read()
out = re.sub(r"%\((\w+)\)s", lambda m: vals[m.group(1)], t).replace("%%", "%")
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a, b = s.index("## 7. Measured (MI355X, 1 GPU;"), s.index("## 8. What comes next")
open(p, "w").write(s[:a] + out + s[b:])
print({k_: vals[k_] for k_ in ("value", "ms", "heavy", "ctx", "e2", "d2", "launches", "gap")})
