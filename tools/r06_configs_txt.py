"""The one-line-per-workload summary of a directory of bench lines (profiles/r06_configs.txt): python tools/r06_configs_txt.py <dir> > <dir>/r06_configs.txt"""
import json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "profiles"
WL = "headline cfg1 cfg2 cfg3 cfg4 cfg5 area knot ref_bunny ref_cow ref_teapot ref_sphere headline_nocull".split()
for w in WL:
    f = os.path.join(d, "r06_bench_%s.json" % w)
    if not os.path.exists(f) or not os.path.getsize(f): continue
    j = json.loads(open(f).read()); c = j["config"]; r = j["roofline"]; b = j.get("cpu_baseline") or {}; p = b.get("parity") or {}
    print(w, c["workload"], c.get("flags") or "", "|", j["value"], "Mrays/s walked (", j["value_counted"], "counted )", j["ms_per_step"], "ms/frame |", c["frame"], "| pass1", c["pass1_ms"], "sobel", c.get("sobel_ms"),
          "ssaa", c.get("ssaa_ms"), "frame kernel", c.get("frame_kernel_ms"), "| pipelined", c.get("pipelined_ms_per_frame"),
          "| first frame of a new view", c.get("new_view_first_frame_ms"), "cold scene", c.get("cold_frame_gpu_busy_before_ms"), "| rays", c["rays_per_frame"],
          "| roofline", r.get("kernel"), "frac", r.get("frac"), "useful", r.get("useful_frac"), "hbm_frac", r.get("hbm_frac"),
          "| reference CPU", b.get("value"), "Mrays/s on", b.get("cores"), "cores; whole frame == reference: pass 1", p.get("pass1_equals_reference_full_frame"), "frame", p.get("frame_equals_reference_full_frame"),
          ("(the reference differed from itself in %d pixels between two runs)" % p["reference_pixels_differing_between_its_own_two_runs"]) if "reference_pixels_differing_between_its_own_two_runs" in p else "")
