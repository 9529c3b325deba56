#!/bin/bash
# GPU box: the big fuzz on the CURRENT sources (tools/fuzz_many.py: random scenes of every object / material / light type, culling on and off, big meshes from random places;
# tools/bvh_fuzz.py: the device builder against the host builder), several seed ranges side by side.  -> gpurun_out/r06/r06_fuzz.txt
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
T=${1:-1500}
for base in ${BASES:-300000 310000 320000 330000 340000 350000}; do ( timeout $T python tools/fuzz_many.py $base 4000 2>/dev/null | grep -E "seeds|MISMATCH|progress|differs" | tail -4 > $O/fuzz_$base.txt ) & done
( timeout $T python tools/bvh_fuzz.py ${BVH0:-5000} 1500 2>/dev/null | tail -2 > $O/bvh_fuzz.txt ) &
wait
( for f in $O/fuzz_[0-9]*.txt; do tail -1 $f; done; grep -h MISMATCH $O/fuzz_[0-9]*.txt; echo "sources $(python tools/srchash.py)" ) > $O/r06_fuzz.txt
cat $O/bvh_fuzz.txt > $O/r06_bvh_fuzz.txt
cat $O/r06_fuzz.txt $O/r06_bvh_fuzz.txt
