#!/bin/bash
# round 5, experiment 8: sq_fit_kernel -- extent scan with fewer instructions per point (shared reciprocal, no branch), view split up to 32
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
python3 -m pytest tests/test_sq_gpu.py tests/test_sq_survey.py tests/test_e2e.py -m gpu -x -q 2>&1 | tail -5 > $O/r5_sq8_tests.txt; cat $O/r5_sq8_tests.txt
(for rep in 1 2; do for lib in odam_amd/libodam_amd_sqprev.so odam_amd/libodam_amd.so; do
   for cfg in sq.split=16 sq.split=32; do
     ODAM_AMD_LIB=$R/$lib ODAM_CONFIG=$cfg python3 tests/native/perf_sq_fit30.py 2>&1 | grep -v amdgpu
     ODAM_AMD_LIB=$R/$lib ODAM_CONFIG=$cfg python3 tests/native/perf_sq_fit30.py 8 2>&1 | grep -v amdgpu
   done; done; done
 for lib in odam_amd/libodam_amd_sqprev.so odam_amd/libodam_amd.so; do echo "== $lib"; ODAM_AMD_LIB=$R/$lib python3 tests/native/perf_sq.py 2>&1 | grep -v amdgpu; done) > $O/r5_sq8_ab.txt 2>&1; cat $O/r5_sq8_ab.txt
(for cfg in sq.split=16 sq.split=32; do echo "== this kernel, $cfg"; ODAM_CONFIG=$cfg bash tests/native/sq_stamps.sh; done) > $O/r5_sq8_stamps.txt 2>&1; cat $O/r5_sq8_stamps.txt
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/r5_gpu_tests_c.txt; cat $O/r5_gpu_tests_c.txt
