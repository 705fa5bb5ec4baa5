#!/bin/bash
# round 5, experiment 7: the 256-column 16x16x32 tiles as eight waves of 32 x 256 (CG_X16_WM8)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
(echo "== this build"; ./tests/native/ring_stamps0_new x16; echo "== CG_X16_WM8"; ./tests/native/ring_stamps0_wm8 x16; echo "== this build"; ./tests/native/ring_stamps0_new x16; echo "== CG_X16_WM8"; ./tests/native/ring_stamps0_wm8 x16) > $O/r5_ring_wm8.txt 2>&1; cat $O/r5_ring_wm8.txt
python3 tests/native/ab_libs.py odam_amd/libodam_amd.so odam_amd/libodam_amd_wm8.so resnet50 fp32 32 3 > $O/r5_ab_wm8.txt 2>&1; cat $O/r5_ab_wm8.txt
