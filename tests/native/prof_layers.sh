#!/bin/bash
# per-layer tables of one forward (B = 32): rocprofv3 kernel trace of tests/native/prof_detr1.py reduced by analyze_trace.py
#   bash tests/native/prof_layers.sh <tag> [fp32|bf16|both]      -> gpurun_out/<tag>_{f32split,r101_bf16}_per_layer.txt + kernel stats
#   PMC_DIR=<dir holding run_pmc.sh's pmc_*/ passes of THIS tree>: every row also gets compulsory bytes, PMC bytes and their ratio
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r3}
WHAT=${2:-both}
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift
  rm -rf $O/${TAG}_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_$name -o t -- python3 "$@" > $O/${TAG}_$name.log 2>&1
}
if [ "$WHAT" = "fp32" ] || [ "$WHAT" = "both" ]; then
  prof f32split $R/tests/native/prof_detr1.py 32 fp32 resnet50
  python3 $R/tests/native/analyze_trace.py $O/${TAG}_f32split/t_kernel_trace.csv 32 2 resnet50 ${PMC_DIR:+--pmc $PMC_DIR r50f32 4} > $O/${TAG}_detr_forward_B32_f32split_per_layer.txt 2>&1
  cp $O/${TAG}_f32split/t_kernel_stats.csv $O/${TAG}_detr_forward_B32_f32split_kernel_stats.csv 2>/dev/null
  tail -12 $O/${TAG}_detr_forward_B32_f32split_per_layer.txt
fi
if [ "$WHAT" = "bf16" ] || [ "$WHAT" = "both" ]; then
  prof r101bf16 $R/tests/native/prof_detr1.py 32 bf16 resnet101
  python3 $R/tests/native/analyze_trace.py $O/${TAG}_r101bf16/t_kernel_trace.csv 32 2 resnet101 ${PMC_DIR:+--pmc $PMC_DIR r101bf16 2} > $O/${TAG}_detr_r101_bf16_B32_per_layer.txt 2>&1
  cp $O/${TAG}_r101bf16/t_kernel_stats.csv $O/${TAG}_detr_r101_bf16_B32_kernel_stats.csv 2>/dev/null
  tail -12 $O/${TAG}_detr_r101_bf16_B32_per_layer.txt
fi
# the traces themselves are large: keep the summaries only
rm -rf $O/${TAG}_f32split $O/${TAG}_r101bf16
