for cfg in "32 2" "48 2" "64 2" "32 3" "64 3"; do set -- $cfg
python bench.py --batch $1 --streams $2 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B $1 streams $2: %.1f frames/s, detect %.1f ms' % (d['value'], d['roofline']['detection_phase_ms_per_step']))"
done
