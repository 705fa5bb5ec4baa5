#!/bin/bash
# round-4 profile artefacts (gpurun_out/r4_* -> copied into profiles/ afterwards): per-layer tables (fp32 R50, bf16 R101), PMC
# summary, kernel stats of the bench command, the bench line, SQ phase stamps
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
bash tests/native/prof_layers.sh r4 both > $O/r4_prof_layers.log 2>&1
bash tests/native/run_pmc.sh > $O/r4_run_pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r4_bench_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_bench_prof -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config4 --no-secondary > $O/r4_bench_prof.log 2>&1
cp $O/r4_bench_prof/t_kernel_stats.csv $O/r4_bench_steps1_kernel_stats.csv 2>/dev/null
rm -rf $O/r4_bench_prof
cd $R
bash tests/native/sq_stamps.sh > $O/r4_sq_phase_stamps.txt 2>&1
python3 bench.py --steps 3 --warmup 1 > $O/r4_bench_line.json 2> $O/r4_bench_line.err
tail -c 600 $O/r4_bench_line.json
