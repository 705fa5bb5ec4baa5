#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
python3 -m pytest tests/test_sq_gpu.py -x -q -m gpu 2>&1 | tail -2
(for rep in 1 2; do for sp in 4 8 16; do
  ODAM_CONFIG=sq.split=$sp ODAM_AMD_LIB=$R/odam_amd/libodam_amd_r4.so python3 tests/native/perf_sq_fit30.py
  ODAM_CONFIG=sq.split=$sp python3 tests/native/perf_sq_fit30.py
done; done
for sp in 8 16; do echo "== round 4 kernel, split $sp"; ODAM_CONFIG=sq.split=$sp python3 tests/native/perf_sq_stamps.py tests/native/_abl/libstamps_r4.so; echo "== this kernel, split $sp"; ODAM_CONFIG=sq.split=$sp python3 tests/native/perf_sq_stamps.py; done) 2>&1 | grep -v amdgpu > $O/r5_sq_fit30.txt; cat $O/r5_sq_fit30.txt
