#!/bin/bash
# per-layer table of the bf16 R101 forward (B=32) with the bf16-native kernel off (ODAM_CG_BIG=0) and on (1)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in ${MODES:-0 1}; do
  export ODAM_CG_BIG=$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof_bf16_big$mode -o t -- python3 $R/tests/native/prof_detr1.py 32 bf16 resnet101 > $R/gpurun_out/r2_prof_bf16_big$mode.log 2>&1
  python3 $R/tests/native/analyze_trace.py $R/gpurun_out/r2_prof_bf16_big$mode/t_kernel_trace.csv 32 2 resnet101 > $R/gpurun_out/r2_bf16_per_layer_big$mode.txt 2>&1
done
