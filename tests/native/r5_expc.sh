#!/bin/bash
# round 5, experiment 6b: CG_X16_BREG in every 16x16x32 variant (128- and 64-column tiles, fused tails, the stem's 1024-thread kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
(echo "== round 4 order (CG_X16_BREG=0)"; ./tests/native/ring_stamps0 x16; echo "== CG_X16_BREG, all variants"; ./tests/native/ring_stamps0_breg2 x16; echo "== round 4 order (CG_X16_BREG=0)"; ./tests/native/ring_stamps0 x16; echo "== CG_X16_BREG, all variants"; ./tests/native/ring_stamps0_breg2 x16) > $O/r5_ring_breg2.txt 2>&1; cat $O/r5_ring_breg2.txt
python3 tests/native/ab_libs.py odam_amd/libodam_amd_prev.so odam_amd/libodam_amd_breg.so odam_amd/libodam_amd.so resnet50 fp32 32 3 > $O/r5_ab_breg2.txt 2>&1; cat $O/r5_ab_breg2.txt
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/r5_gpu_tests_breg.txt; cat $O/r5_gpu_tests_breg.txt
