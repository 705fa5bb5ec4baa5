#!/bin/bash
# round 5, experiment 6: the persistent 16x16x32 kernel after the register fixes, layer by layer and as a forward
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
(echo "== ring kernel (cg.persist 0)"; ./tests/native/ring_stamps0 x16 np; echo "== persistent (cg.persist 1)"; ./tests/native/ring_stamps0 x16; echo "== ring again"; ./tests/native/ring_stamps0 x16 np; echo "== persistent again"; ./tests/native/ring_stamps0 x16) > $O/r5_ring_persist.txt 2>&1
cat $O/r5_ring_persist.txt
python3 tests/native/cmp_config.py cg.persist 0 1 32 > $O/r5_cmp_persist.txt 2>&1; cat $O/r5_cmp_persist.txt
python3 tests/native/sweep_config.py cg.persist 0,1 > $O/r5_sweep_persist.txt 2>&1; cat $O/r5_sweep_persist.txt
(echo "== round 4 kernel"; python3 tests/native/perf_sq_stamps.py tests/native/_abl/libstamps_r4.so; echo "== this kernel"; python3 tests/native/perf_sq_stamps.py; echo "== round 4 kernel"; python3 tests/native/perf_sq_stamps.py tests/native/_abl/libstamps_r4.so; echo "== this kernel"; python3 tests/native/perf_sq_stamps.py) 2>&1 | grep -v amdgpu > $O/r5_sq_phase_stamps.txt; cat $O/r5_sq_phase_stamps.txt
python3 -m pytest tests/test_sq_gpu.py tests/test_sq_survey.py -x -q -m gpu > $O/r5_pytest_sq.txt 2>&1; tail -3 $O/r5_pytest_sq.txt
(ODAM_AMD_LIB=$R/odam_amd/libodam_amd_r4.so python3 tests/native/perf_optim.py 2>&1 | head -2; python3 tests/native/perf_optim.py 2>&1 | head -2; ODAM_AMD_LIB=$R/odam_amd/libodam_amd_r4.so python3 tests/native/perf_optim.py 2>&1 | head -2; python3 tests/native/perf_optim.py 2>&1 | head -2) | grep -v amdgpu > $O/r5_perf_optim.txt; cat $O/r5_perf_optim.txt
