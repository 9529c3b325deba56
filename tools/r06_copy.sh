#!/bin/bash
# Here (after gpurun merged gpurun_out/ back): copies the round's measured summaries into profiles/ (tracked).  Stage 1 / 2: see tools/r06_collect.sh.
cd "$(dirname "$0")/.."
O=gpurun_out/r06
if [ "$1" = 1 ]; then
  cp gpurun_out/pmc_r06/r06_pass1_pmc.json profiles/r06_pass1_pmc.json
  cp $O/r06_dbg_counts.txt profiles/r06_dbg_counts.txt
  python tools/isa_mix.py r06 > /dev/null
  python tools/issue_account.py r06 > profiles/r06_issue_account.txt
  cat profiles/r06_issue_account.txt
else
  for f in $(cd $O; ls r06_*.json r06_*.csv r06_*.txt 2>/dev/null); do [ -s $O/$f ] && cp $O/$f profiles/$f; done
fi
echo "sources $(python tools/srchash.py); pmc $(grep -o '"source_hash": "[0-9a-f]*"' profiles/r06_pass1_pmc.json | head -1); isa $(grep -o '"source_hash": "[0-9a-f]*"' profiles/r06_pass1_isa.json | head -1)"
