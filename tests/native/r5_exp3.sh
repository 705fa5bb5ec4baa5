#!/bin/bash
# round 5, experiment 3: light tile stamps of the 16x16x32 loop; the branch-free epilogue on the bf16 path; the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
./tests/native/ring_stamps2 x16 > $O/r5_ring_stamps_light.txt 2>&1
python3 tests/native/ab_libs.py odam_amd/libodam_amd_r4.so odam_amd/libodam_amd.so resnet101 bf16 32 3 > $O/r5_ab_ep2_bf16.txt 2>&1
python3 tests/native/ab_libs.py odam_amd/libodam_amd_r4.so odam_amd/libodam_amd.so resnet50 fp32 8 2 > $O/r5_ab_ep2_f32_b8.txt 2>&1
python3 -m pytest tests -x -q -m gpu > $O/r5_pytest_all.txt 2>&1
cat $O/r5_ring_stamps_light.txt $O/r5_ab_ep2_bf16.txt $O/r5_ab_ep2_f32_b8.txt; tail -8 $O/r5_pytest_all.txt
