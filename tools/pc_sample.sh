#!/bin/bash
# GPU box: PC sampling (rocprofv3, host-trap) of the pass-1 / SSAA kernels of one bench.py workload, aggregated per source line
# (the library is built with -gline-tables-only: tools/build_variants.sh "pcs:-gline-tables-only") -> gpurun_out/pcs/<cfg>_lines.txt
# Usage: tools/pc_sample.sh [config] [interval us]
CFG=${1:-headline}; IV=${2:-1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pcs; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -f $R/rendering_amd/_variants/librtx_pcs.so ] && { cp $R/rendering_amd/librtx_hip.so /tmp/librtx_orig.so; cp $R/rendering_amd/_variants/librtx_pcs.so $R/rendering_amd/librtx_hip.so; }
rocprofv3 -L > $OUT/list_avail.txt 2>&1
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval $IV --output-format csv -d /tmp/pcs_$CFG -o pcs -- \
  python $R/bench.py --config $CFG --steps 30 --warmup 2 --no-cpu-baseline > $OUT/$CFG.log 2>&1
echo "rc=$?" >> $OUT/$CFG.log
[ -f /tmp/librtx_orig.so ] && cp /tmp/librtx_orig.so $R/rendering_amd/librtx_hip.so
find /tmp/pcs_$CFG -type f | head -20 >> $OUT/$CFG.log
python - "$OUT" "$CFG" <<'PY'
import csv, glob, sys, collections, os
out, cfg = sys.argv[1], sys.argv[2]
files = glob.glob("/tmp/pcs_%s/**/*pc_sampling*.csv" % cfg, recursive=True)
kern = {}
for f in glob.glob("/tmp/pcs_%s/**/*kernel_trace*.csv" % cfg, recursive=True):
    for row in csv.DictReader(open(f)):
        kern[row.get("Dispatch_Id")] = row.get("Kernel_Name", "?").split("(")[0]
byline = collections.Counter(); byinst = collections.Counter(); total = 0; bydisp = collections.Counter()
hdr = None
for f in files:
    rd = csv.DictReader(open(f))
    hdr = rd.fieldnames
    for row in rd:
        total += 1
        ins = row.get("Instruction", ""); com = row.get("Instruction_Comment", "")
        byline[com] += 1
        byinst[(com, ins)] += 1
        bydisp[row.get("Dispatch_Id")] += 1
with open(os.path.join(out, cfg + "_lines.txt"), "w") as o:
    o.write("files %s\nheader %s\nsamples %d\n" % (files, hdr, total))
    o.write("dispatch kernels: %s\n" % collections.Counter(kern.get(d, "?") for d in bydisp.elements()).most_common(12))
    for k, n in byline.most_common(400):
        o.write("%7d %5.2f%% %s\n" % (n, 100.0 * n / max(total, 1), k))
    o.write("\n== by instruction\n")
    for (c, i), n in byinst.most_common(600):
        o.write("%7d %5.2f%% %-60s %s\n" % (n, 100.0 * n / max(total, 1), i, c))
print("samples", total)
PY
