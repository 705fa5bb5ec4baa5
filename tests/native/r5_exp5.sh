#!/bin/bash
# round 5, experiment 5: SQ kernel with the dyadic-tree subdivision (bit-exactness + phase stamps); the persistent 16x16x32 kernel (bit identity + time)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
python3 -m pytest tests/test_sq_gpu.py tests/test_sq_survey.py tests/test_sq_math.py -x -q -m gpu > $O/r5_pytest_sq.txt 2>&1; tail -5 $O/r5_pytest_sq.txt
python3 tests/native/perf_sq_stamps.py 2>&1 | grep -v amdgpu > $O/r5_sq_phase_stamps.txt; cat $O/r5_sq_phase_stamps.txt
python3 tests/native/perf_optim.py 2>&1 | grep -v amdgpu | head -3 > $O/r5_perf_optim.txt; cat $O/r5_perf_optim.txt
python3 tests/native/cmp_config.py cg.persist 0 1 32 > $O/r5_cmp_persist.txt 2>&1; cat $O/r5_cmp_persist.txt
python3 tests/native/sweep_config.py cg.persist 0,1 > $O/r5_sweep_persist.txt 2>&1; cat $O/r5_sweep_persist.txt
python3 -m pytest tests/test_detr_gpu.py -x -q -m gpu > $O/r5_pytest_detr2.txt 2>&1; tail -5 $O/r5_pytest_detr2.txt
