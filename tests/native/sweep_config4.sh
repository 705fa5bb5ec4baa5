#!/bin/bash
# configs[3] block of the bench (R101 bf16) by frames per forward and streams:  bash tests/native/sweep_config4.sh  (on the box)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for bs in "76 3" "76 4" "77 3" "38 3" "38 4" "57 3" "76 2" "76 3"; do
  set -- $bs
  python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-full-chain --config4-batch $1 --config4-streams $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['config4']
print('batch $1 streams $2: detect %.0f frames/s, scene %.0f, frac %.4f' % (c['frames_per_s_detect'], c['frames_per_s_scene'], c['roofline']['frac']))"
done
