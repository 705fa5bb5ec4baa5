#!/bin/bash
# round-6 profile artefacts (gpurun_out/r6_* -> copied into profiles/ afterwards): EVERY block of pmc_summary.json (stamped with ODAM_COMMIT;
# new: the fit kernel's VALU-issue fraction at 64 x 256 and at configs[4]'s 500 x 256), per-layer tables (fp32 R50, bf16 R101) with compulsory
# vs PMC bytes per launch, kernel stats of the bench command, the association loop three ways, the bench line
#   ODAM_COMMIT=$(git rev-parse --short HEAD) gpurun --timeout 3000 -- "ODAM_COMMIT=$ODAM_COMMIT bash tests/native/collect_r6.sh"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
bash tests/native/run_pmc.sh > $O/r6_run_pmc.log 2>&1
PMC_DIR=$O bash tests/native/prof_layers.sh r6 both > $O/r6_prof_layers.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r6_bench_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r6_bench_prof -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config4 --no-secondary > $O/r6_bench_prof.log 2>&1
cp $O/r6_bench_prof/t_kernel_stats.csv $O/r6_bench_steps1_kernel_stats.csv 2>/dev/null
rm -rf $O/r6_bench_prof
cd $R
python3 tests/native/perf_assoc_fast.py 200 > $O/r6_assoc_loop.txt 2>&1
grep -h "sq fit ok" $O/pmc_sq_valu.log $O/pmc_sq500_valu.log > $O/r6_sq_launch_ms.txt 2>/dev/null
python3 bench.py --steps 3 --warmup 1 > $O/r6_bench_line.json 2> $O/r6_bench_line.err
tail -c 600 $O/r6_bench_line.json
