# batch / stream / chunk choice per shard size (frames per rank at 8, 4, 2 GPUs; objects = 30 / ranks)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "125 32 3 4 128" "125 42 3 4 126" "125 21 3 4 126" "125 25 3 4 125" "125 16 3 4 128" "125 32 2 4 128" "125 63 2 4 126" "250 32 3 8 128" "250 42 3 8 126" "250 28 3 8 112" "500 32 3 15 128" "500 38 3 15 152"; do
set -- $cfg
python3 bench.py --frames $1 --batch $2 --streams $3 --objects $4 --chunk $5 --no-cpu-baseline --no-config4 --no-secondary --steps 3 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames $1 B $2 streams $3 chunk $5: %.1f frames/s, step %.1f ms, detect-only %.1f frames/s' % (d['value'], d['ms_per_step'], d['detect_only_frames_per_s_per_gpu']))"
done
