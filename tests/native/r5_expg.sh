#!/bin/bash
# round 5, experiment 10: fused bottleneck tails with the two waves of a SIMD in opposite phase (CG_TAIL_STAGGER)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
python3 tests/native/ab_libs.py odam_amd/libodam_amd_nostagger.so odam_amd/libodam_amd.so resnet50 fp32 32 3 > $O/r5_ab_stagger.txt 2>&1; cat $O/r5_ab_stagger.txt
cd /tmp && export TMPDIR=/tmp
for v in nostagger stagger; do
  lib=$R/odam_amd/libodam_amd.so; [ $v = nostagger ] && lib=$R/odam_amd/libodam_amd_nostagger.so
  export ODAM_AMD_LIB=$lib
  rm -rf $O/r5_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_$v -o t -- python3 $R/tests/native/prof_detr1.py 32 fp32 resnet50 > $O/r5_$v.log 2>&1
  python3 $R/tests/native/analyze_trace.py $O/r5_$v/t_kernel_trace.csv 32 2 resnet50 > $O/r5_per_layer_$v.txt 2>&1
  rm -rf $O/r5_$v
  echo "== $v"; head -9 $O/r5_per_layer_$v.txt; tail -11 $O/r5_per_layer_$v.txt
done
unset ODAM_AMD_LIB
cd $R; python3 -m pytest tests/test_detr_gpu.py -m gpu -x -q 2>&1 | tail -4 > $O/r5_gpu_tests_stagger.txt; cat $O/r5_gpu_tests_stagger.txt
