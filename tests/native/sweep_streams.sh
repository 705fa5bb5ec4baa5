#!/bin/bash
# bench.py over (streams, frames per forward): is the default (3, 32) still on the plateau after the round's kernel changes?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
(for sb in "3 32" "3 38" "3 48" "4 32" "2 48" "3 64" "3 32"; do set -- $sb
   python3 bench.py --streams $1 --batch $2 --no-cpu-baseline --no-config4 --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('streams $1 batch $2 value %.1f detect-only %.1f frac %.4f' % (d['value'], d['detect_only_frames_per_s_per_gpu'], d['roofline']['frac']))"
 done) > $O/r5_stream_batch_sweep.txt 2>&1
cat $O/r5_stream_batch_sweep.txt
