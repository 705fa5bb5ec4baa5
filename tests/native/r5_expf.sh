#!/bin/bash
# round 5, experiment 9: do the early layers run faster per frame when a forward's intermediates fit the 256 MB memory-side cache?
# per-layer tables of the fp32 R50 forward at 4 / 8 / 16 / 32 frames, kernel choice pinned (cg.pin = 1: the same kernels at every size)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ODAM_CONFIG=cg.pin=1
for B in 4 8 16 32; do
  rm -rf $O/r5_b$B
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_b$B -o t -- python3 $R/tests/native/prof_detr1.py $B fp32 resnet50 > $O/r5_b$B.log 2>&1
  python3 $R/tests/native/analyze_trace.py $O/r5_b$B/t_kernel_trace.csv $B 2 resnet50 > $O/r5_per_layer_pinned_B$B.txt 2>&1
  rm -rf $O/r5_b$B
  echo "== B = $B"; head -9 $O/r5_per_layer_pinned_B$B.txt; tail -11 $O/r5_per_layer_pinned_B$B.txt
done
