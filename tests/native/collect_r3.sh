#!/bin/bash
# round-3 profile artefacts (gpurun_out/r3_* -> copied into profiles/ afterwards)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
bash tests/native/prof_layers.sh r3 both > $O/r3_prof_layers.log 2>&1
bash tests/native/run_pmc.sh > $O/r3_run_pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_bench_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_bench_prof -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config4 --no-secondary > $O/r3_bench_prof.log 2>&1
cp $O/r3_bench_prof/t_kernel_stats.csv $O/r3_bench_steps1_kernel_stats.csv 2>/dev/null
rm -rf $O/r3_bench_prof
cd $R
python3 tests/native/perf_assoc_loop.py 200 2>&1 | grep -v amdgpu | head -24 > $O/r3_perf_assoc_loop.txt
python3 tests/native/perf_conv_shape.py bf16 2>&1 | grep -v amdgpu > $O/r3_conv_shape_bf16.txt
python3 bench.py --steps 3 --warmup 1 > $O/r3_bench_line.json 2> $O/r3_bench_line.err
tail -c 600 $O/r3_bench_line.json
