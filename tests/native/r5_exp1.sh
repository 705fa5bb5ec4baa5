#!/bin/bash
# round 5, experiment 1: first-round stagger of the ring kernel's workgroups (cg.stagger) -- does dephasing neighbouring CUs overlap epilogue traffic with products?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
python3 tests/native/sweep_config.py cg.stagger 0,16,32,48,64,96 > $O/r5_stagger_sweep_f32.txt 2>&1
python3 tests/native/sweep_config.py cg.stagger 0,16,32,48,64 resnet101 bf16 > $O/r5_stagger_sweep_bf16.txt 2>&1
bash tests/native/prof_layers.sh r5s0 fp32 > $O/r5s0.log 2>&1
ODAM_CONFIG=cg.stagger=48 bash tests/native/prof_layers.sh r5s48 fp32 > $O/r5s48.log 2>&1
python3 bench.py --steps 3 --warmup 1 --no-config4 > $O/r5_bench_start.json 2> $O/r5_bench_start.err
cat $O/r5_stagger_sweep_f32.txt $O/r5_stagger_sweep_bf16.txt; tail -c 1500 $O/r5_bench_start.json
