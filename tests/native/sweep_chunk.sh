cd $GRAFT_REPO_ROOT
for c in 32 64 96 128 192; do python3 bench.py --chunk $c --steps 3 --warmup 1 --no-cpu-baseline --no-config4 --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $c: %.1f frames/s, step %.1f ms, detect-only %.1f' % (d['value'], d['ms_per_step'], d['detect_only_frames_per_s_per_gpu']))"; done
