python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
python tools/bvh_build_time.py 2>&1 | grep -v amdgpu > gpurun_out/r02_bvh_build_time.txt; cat gpurun_out/r02_bvh_build_time.txt
python tools/shard_time.py 2 4 8 2>&1 | grep -v amdgpu > gpurun_out/r02_shard_emulation.txt
python tools/shard_time.py 2 4 8 --size 8192 2>&1 | grep -v amdgpu >> gpurun_out/r02_shard_emulation.txt; cat gpurun_out/r02_shard_emulation.txt
