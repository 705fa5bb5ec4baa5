# batch / stream / chunk choice per shard size (frames per rank at 8, 4, 2, 1 GPUs; objects = 30 / ranks)
for cfg in "125 32 2 4 32" "125 32 2 4 64" "125 32 2 4 128" "1000 32 2 30 128" "1000 32 2 30 256" "1000 32 2 30 512"; do
set -- $cfg
python bench.py --frames $1 --batch $2 --streams $3 --objects $4 --chunk $5 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames $1 B $2 streams $3 chunk $5: %.1f frames/s, step %.1f ms, detect %.1f ms' % (d['value'], d['ms_per_step'], d['roofline']['detection_phase_ms_per_step']))"
done
