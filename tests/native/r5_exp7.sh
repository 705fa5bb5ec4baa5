#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
(echo "== round 4 kernel, no view split"; ODAM_CONFIG=sq.split=1 python3 tests/native/perf_sq_stamps.py tests/native/_abl/libstamps_r4.so; echo "== this kernel, no view split"; ODAM_CONFIG=sq.split=1 python3 tests/native/perf_sq_stamps.py
 echo "== round 4 kernel, split 2"; ODAM_CONFIG=sq.split=2 python3 tests/native/perf_sq_stamps.py tests/native/_abl/libstamps_r4.so; echo "== this kernel, split 2"; ODAM_CONFIG=sq.split=2 python3 tests/native/perf_sq_stamps.py) 2>&1 | grep -v amdgpu > $O/r5_sq_phase_stamps_split.txt; cat $O/r5_sq_phase_stamps_split.txt
