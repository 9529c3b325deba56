#!/bin/bash
# Here (after gpurun merged gpurun_out/ back): copies the round's measured summaries into profiles/ (tracked).
cd "$(dirname "$0")/.."
O=gpurun_out/r03
cp gpurun_out/pmc_r03/r03_pass1_pmc.json profiles/r03_pass1_pmc.json
for f in r03_bench_default.json r03_bench_cfg2.json r03_kernel_stats.csv r03_kernel_stats_cfg2.csv r03_bench_under_rocprof.json r03_bench_cfg2_under_rocprof.json r03_configs.txt r03_shard_emulation.txt r03_dbg_counts.txt r03_cold_probe.txt r03_overhead_probe.txt r03_cost_fit.txt; do
  [ -s $O/$f ] && cp $O/$f profiles/$f
done
python tools/isa_mix.py r03 > /dev/null
python tools/issue_account.py > profiles/r03_issue_account.txt
echo "sources $(python tools/srchash.py); pmc $(grep -o '"source_hash": "[0-9a-f]*"' profiles/r03_pass1_pmc.json | head -1); isa $(grep -o '"source_hash": "[0-9a-f]*"' profiles/r03_pass1_isa.json | head -1)"
