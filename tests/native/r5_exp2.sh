#!/bin/bash
# round 5, experiment 2: the 16x16x32 loop's epilogue without branches (counted waits, residual rows requested before the loop's last barrier)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
(echo "== round 4 build"; ./tests/native/ring_stamps0_r4 x16; echo "== this build"; ./tests/native/ring_stamps0 x16; echo "== round 4 build, stamps"; ./tests/native/ring_stamps1_r4 x16; echo "== this build, stamps"; ./tests/native/ring_stamps1 x16) > $O/r5_ring_stamps_ep2.txt 2>&1
python3 tests/native/ab_libs.py odam_amd/libodam_amd_r4.so odam_amd/libodam_amd.so resnet50 fp32 32 3 > $O/r5_ab_ep2_f32.txt 2>&1
python3 -m pytest tests/test_detr_gpu.py -x -q -m gpu > $O/r5_pytest_detr.txt 2>&1
cat $O/r5_ring_stamps_ep2.txt $O/r5_ab_ep2_f32.txt; tail -5 $O/r5_pytest_detr.txt
