#!/bin/bash
# Here (after gpurun merged gpurun_out/ back): copies the round's measured summaries into profiles/ (tracked).
cd "$(dirname "$0")/.."
O=gpurun_out/r05
cp gpurun_out/pmc_r05/r05_pass1_pmc.json profiles/r05_pass1_pmc.json
for f in r05_bench_default.json r05_bench_cfg1.json r05_bench_cfg2.json r05_bench_cfg3.json r05_bench_cfg4.json r05_bench_cfg5.json r05_bench_area.json r05_kernel_stats.csv r05_kernel_stats_cfg1.csv r05_kernel_stats_cfg2.csv \
         r05_kernel_stats_cfg3.csv r05_kernel_stats_cfg4.csv r05_kernel_stats_cfg5.csv r05_kernel_stats_area.csv r05_bench_under_rocprof.json r05_configs.txt r05_shard_emulation.txt r05_shard_stages.txt \
         r05_new_view_probe.txt r05_cold_probe.txt r05_bvh_build_time.txt r05_cost_fit.txt r05_dbg_counts.txt; do
  [ -s $O/$f ] && cp $O/$f profiles/$f
done
python tools/isa_mix.py r05 > /dev/null
python tools/issue_account.py r05 > profiles/r05_issue_account.txt
echo "sources $(python tools/srchash.py); pmc $(grep -o '"source_hash": "[0-9a-f]*"' profiles/r05_pass1_pmc.json | head -1); isa $(grep -o '"source_hash": "[0-9a-f]*"' profiles/r05_pass1_isa.json | head -1)"
