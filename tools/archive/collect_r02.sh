#!/bin/bash
# Here (after gpurun merged gpurun_out/ back): copies the round's measured summaries into profiles/ (tracked).
cd "$(dirname "$0")/.."
O=gpurun_out/r02
for f in r02_pass1_pmc.json r02_bench_default.json r02_bench_cfg2.json r02_kernel_stats.csv r02_kernel_stats_cfg2.csv r02_bench_under_rocprof.json r02_bench_cfg2_under_rocprof.json r02_configs.txt r02_shard_emulation.txt r02_bvh_build_time.txt r02_small_frames.txt; do
  [ -s $O/$f ] && cp $O/$f profiles/$f
done
python tools/isa_mix.py r02 > /dev/null
echo "sources $(python tools/srchash.py); pmc $(grep -o '"source_hash": "[0-9a-f]*"' profiles/r02_pass1_pmc.json); isa $(grep -o '"source_hash": "[0-9a-f]*"' profiles/r02_pass1_isa.json)"
