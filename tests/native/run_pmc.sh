#!/bin/bash
# PMC passes behind profiles/pmc_summary.json (bench.py reads that file for roofline.traffic and sq_fit_kernel.valu_util).
# One counter group per rocprofv3 run (FETCH_SIZE and WRITE_SIZE do not fit one pass; no tracing flags beside --pmc).
#   bash tests/native/run_pmc.sh            on the GPU box; writes gpurun_out/pmc_*/ and gpurun_out/pmc_summary.json
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {   # name, counters, script args...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$name -o t -- python3 "$@" > $R/gpurun_out/pmc_$name.log 2>&1
}
for c in FETCH_SIZE WRITE_SIZE; do
  run r50f32_$c $c $R/tests/native/prof_detr1.py 32 fp32 resnet50
  run r101bf16_$c $c $R/tests/native/prof_detr1.py 32 bf16 resnet101
  run sq_$c $c $R/tests/native/prof_sq.py
done
run sq_valu "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" $R/tests/native/prof_sq.py
run sq500_valu "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE" $R/tests/native/prof_sq.py 500 256
run r50f32_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" $R/tests/native/prof_detr1.py 32 fp32 resnet50
run r101bf16_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" $R/tests/native/prof_detr1.py 32 bf16 resnet101
python3 $R/tests/native/sum_pmc.py --summary $R/gpurun_out > $R/gpurun_out/pmc_summary.json
