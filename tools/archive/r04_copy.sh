#!/bin/bash
# Here (after gpurun merged gpurun_out/ back): copies the round's measured summaries into profiles/ (tracked).
cd "$(dirname "$0")/.."
O=gpurun_out/r04
cp gpurun_out/pmc_r04/r04_pass1_pmc.json profiles/r04_pass1_pmc.json
for f in r04_fuzz.txt r04_bench_default.json r04_bench_cfg2.json r04_kernel_stats.csv r04_kernel_stats_cfg2.csv r04_bench_under_rocprof.json r04_bench_cfg2_under_rocprof.json r04_configs.txt r04_shard_emulation.txt r04_dbg_counts.txt r04_cold_probe.txt r04_overhead_probe.txt r04_cost_fit.txt; do
  [ -s $O/$f ] && cp $O/$f profiles/$f
done
python tools/isa_mix.py r04 > /dev/null
python tools/issue_account.py > profiles/r04_issue_account.txt
echo "sources $(python tools/srchash.py); pmc $(grep -o '"source_hash": "[0-9a-f]*"' profiles/r04_pass1_pmc.json | head -1); isa $(grep -o '"source_hash": "[0-9a-f]*"' profiles/r04_pass1_isa.json | head -1)"
