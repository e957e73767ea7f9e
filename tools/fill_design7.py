"""Refresh DESIGN.md section 7 from a bench line: python tools/fill_design7.py [gpurun_out/r05_bench_1m.json]
(tools/design7_template.md holds the text with %(name)s slots)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_bench_1m.json")
j = json.loads(open(src).read().strip().splitlines()[-1])
c = j["codec"]; v2 = c["container_v2"]
vals = dict(value="%.1f" % j["value"], ms="%.2f" % j["ms_per_step"], smin="%.3f" % j["timing"]["ms_per_step_min"],
            smax="%.3f" % j["timing"]["ms_per_step_max"], ro="%.1f" % j["value_raster_only"], mid="%.1f" % j["value_mid_phase_noise"],
            loss="%.1f" % j["value_with_l1_ssim_loss"], heavy="%.1f" % j["value_heavy_pairs"], hip="%.2f" % j["hip_kernel_ms_per_step"],
            ctx="%.2f" % j["ctx_group_roofline"]["ms_per_step"], mlpms="%.2f" % j["mlp_group_roofline"]["ms_per_step"],
            mlpfrac="%.1f" % (100 * j["mlp_group_roofline"]["frac"]), e1="%.1f" % c["encode_Manchors_per_s"],
            d1="%.1f" % c["decode_Manchors_per_s"], e2="%.1f" % v2["encode_Manchors_per_s"], d2="%.1f" % v2["decode_Manchors_per_s"],
            e2ms="%.1f" % (v2["encode_s"] * 1e3), d2ms="%.1f" % (v2["decode_s"] * 1e3),
            fps_d="%.0f" % c["test_fps"]["decoded_views_per_s"], fps_n="%.0f" % c["test_fps"]["not_decoded_views_per_s"],
            cpu="%.3f" % j["cpu_baseline"]["value"])
t = open(os.path.join(ROOT, "tools", "design7_template.md")).read()
out = re.sub(r"%\((\w+)\)s", lambda m: vals[m.group(1)], t)
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a, b = s.index("## 7. Measured (MI355X, 1 GPU;"), s.index("## 8. What comes next")
open(p, "w").write(s[:a] + out + s[b:])
print({k: vals[k] for k in ("value", "ms", "heavy", "ctx", "e2", "d2")})
