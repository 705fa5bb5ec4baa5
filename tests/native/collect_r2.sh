#!/bin/bash
# round-2 profile artefacts (copied from gpurun_out/r2c_* into profiles/ by hand afterwards)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
prof() {  # name, then the python script + args
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2c_$name -o t -- python3 "$@" > $O/r2c_$name.log 2>&1
}
prof bench $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config4
prof detr_f32 $R/tests/native/prof_detr1.py 32 fp32 resnet50
python3 $R/tests/native/analyze_trace.py $O/r2c_detr_f32/t_kernel_trace.csv 32 2 resnet50 > $O/r2c_detr_f32_per_layer.txt 2>&1
export ODAM_CG_BIG_F32=0
prof detr_f32_instr $R/tests/native/prof_detr1.py 32 fp32 resnet50
python3 $R/tests/native/analyze_trace.py $O/r2c_detr_f32_instr/t_kernel_trace.csv 32 2 resnet50 > $O/r2c_detr_f32_instr_per_layer.txt 2>&1
unset ODAM_CG_BIG_F32
prof detr_bf16 $R/tests/native/prof_detr1.py 32 bf16 resnet101
python3 $R/tests/native/analyze_trace.py $O/r2c_detr_bf16/t_kernel_trace.csv 32 2 resnet101 > $O/r2c_detr_bf16_per_layer.txt 2>&1
export ODAM_CG_BIG=0
prof detr_bf16_old $R/tests/native/prof_detr1.py 32 bf16 resnet101
python3 $R/tests/native/analyze_trace.py $O/r2c_detr_bf16_old/t_kernel_trace.csv 32 2 resnet101 > $O/r2c_detr_bf16_old_per_layer.txt 2>&1
unset ODAM_CG_BIG
prof sq $R/tests/native/prof_sq.py
prof assoc $R/tests/native/perf_assoc.py
cd $R
python3 tests/native/perf_conv_shape.py fp32 > $O/r2c_conv_shape_f32_split.txt 2>&1
ODAM_CG_BIG_F32=0 python3 tests/native/perf_conv_shape.py fp32 > $O/r2c_conv_shape_f32_instr.txt 2>&1
python3 tests/native/perf_conv_shape.py bf16 > $O/r2c_conv_shape_bf16.txt 2>&1
python3 tests/native/perf_detr_bf16.py > $O/r2c_perf_detr.txt 2>&1
python3 tests/native/perf_assoc.py > $O/r2c_perf_assoc.txt 2>&1
ODAM_ASSOC_PERSIST=0 python3 tests/native/perf_assoc.py > $O/r2c_perf_assoc_launches.txt 2>&1
./tests/native/ring_stamps0 > $O/r2c_ring_plain.txt 2>&1
./tests/native/ring_stamps1 > $O/r2c_ring_stamps.txt 2>&1
bash tests/native/run_pmc.sh
cp $O/pmc_summary.json $R/profiles/pmc_summary.json 2>/dev/null
python3 bench.py --steps 3 --warmup 1 > $O/r2c_bench_line.json 2> $O/r2c_bench_line.err
