#!/bin/bash
# round 5, experiment 11: layer2's fused launch with timing builds of its tail (WRONG results): CG_HACK = 4 without the tail's residual loads and
# stores, 8 without its products, 12 without both; per-layer traces.  Build first:
#   python3 -c "import __graft_entry__ as g; [g.build_library(lib='odam_amd/libodam_amd_hk%d.so' % h, extra_flags=('-DCG_HACK=%d' % h,)) for h in (4, 8, 12)]"
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in full hk4 hk8 hk12 full; do
  lib=$R/odam_amd/libodam_amd.so; [ $v != full ] && lib=$R/odam_amd/libodam_amd_$v.so
  export ODAM_AMD_LIB=$lib
  rm -rf $O/r5_h$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_h$v -o t -- python3 $R/tests/native/prof_detr1.py 32 fp32 resnet50 > $O/r5_h$v.log 2>&1
  python3 $R/tests/native/analyze_trace.py $O/r5_h$v/t_kernel_trace.csv 32 2 resnet50 > $O/r5_per_layer_h$v.txt 2>&1
  rm -rf $O/r5_h$v
  echo "== $v"; grep "^l2" $O/r5_per_layer_h$v.txt | head -12
done > $O/r5_l2_tail_hacks.txt 2>&1
cat $O/r5_l2_tail_hacks.txt
