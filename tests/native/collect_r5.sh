#!/bin/bash
# round-5 profile artefacts (gpurun_out/r5_* -> copied into profiles/ afterwards): per-layer tables (fp32 R50, bf16 R101), EVERY block of
# pmc_summary.json (stamped with ODAM_COMMIT), kernel stats of the bench command, SQ phase stamps, the bench line
#   ODAM_COMMIT=$(git rev-parse --short HEAD) gpurun --timeout 3000 -- "ODAM_COMMIT=$ODAM_COMMIT bash tests/native/collect_r5.sh"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
bash tests/native/prof_layers.sh r5 both > $O/r5_prof_layers.log 2>&1
bash tests/native/run_pmc.sh > $O/r5_run_pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r5_bench_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_bench_prof -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config4 --no-secondary > $O/r5_bench_prof.log 2>&1
cp $O/r5_bench_prof/t_kernel_stats.csv $O/r5_bench_steps1_kernel_stats.csv 2>/dev/null
rm -rf $O/r5_bench_prof
cd $R
bash tests/native/sq_stamps.sh > $O/r5_sq_phase_stamps.txt 2>&1      # (builds the stamped library from this tree first: a stale one measures another kernel)
python3 bench.py --steps 3 --warmup 1 > $O/r5_bench_line.json 2> $O/r5_bench_line.err
tail -c 600 $O/r5_bench_line.json
