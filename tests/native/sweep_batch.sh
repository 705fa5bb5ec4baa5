cd $GRAFT_REPO_ROOT
python3 tests/native/perf_batch_sweep.py fp32 resnet50 32 38 48 64 76 77 2>&1 | grep -v amdgpu
python3 tests/native/perf_batch_sweep.py bf16 resnet101 32 38 64 76 77 2>&1 | grep -v amdgpu
for b in 32 38 77; do python3 bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-config4 --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench batch', d['config'].get('detector_batch'), d['value'], d['detect_only_frames_per_s_per_gpu'], d['roofline']['frac'])"; done
