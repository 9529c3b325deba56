#!/bin/bash
# GPU box: HBM write / fetch traffic of pass 1 of the headline for the variants under rendering_amd/_variants (the framebuffer-store experiments):
# separate rocprofv3 --pmc passes (counters only) of tools/time_stages.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
cp rendering_amd/librtx_hip.so /tmp/librtx_orig.so
for v in rendering_amd/_variants/librtx_*.so; do
  cp $v rendering_amd/librtx_hip.so
  for cnt in WRITE_SIZE FETCH_SIZE; do
    OUT=$R/gpurun_out/wt/$(basename $v .so)_$cnt; rm -rf $OUT; mkdir -p $OUT
    (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $cnt --output-format csv -d $OUT -o q -- python $R/tools/time_stages.py scenes/cfg2_smooth_250k.scene 4096 4096 6 > $OUT/log.txt 2>&1)
  done
  python - "$R/gpurun_out/wt/$(basename $v .so)" "$(basename $v)" <<'PY'
import csv, glob, sys
base, name = sys.argv[1], sys.argv[2]
out = {}
for cnt in ("WRITE_SIZE", "FETCH_SIZE"):
    v = [float(r["Counter_Value"]) for f in glob.glob(base + "_" + cnt + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "Pass1" in r["Kernel_Name"] and r["Counter_Name"] == cnt]
    out[cnt] = v[-1] if v else float("nan")
t = [l for l in open(base + "_WRITE_SIZE/log.txt") if "frames" in l]
print("%-22s pass 1 per launch: WRITE_SIZE %.1f MB, FETCH_SIZE %.1f MB (as reported, KB x 1024; x 2 for 16-byte streams: MI355X_MICROARCH.md) | %s" % (name, out["WRITE_SIZE"] / 1024, out["FETCH_SIZE"] / 1024, t[-1].strip()[:110] if t else ""))
PY
done
cp /tmp/librtx_orig.so rendering_amd/librtx_hip.so
