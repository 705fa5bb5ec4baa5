# BASELINE configs[4] (10 k frames, 500 objects) priced on ONE GPU: the whole job, and one rank's share of an 8-way split
cd ${GRAFT_REPO_ROOT:-/root/repo}
P='import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["config"]["workload"][:110], "| value %.1f frames/s, step %.1f ms, detect-only %.1f frames/s, sq_fit launch %.1f ms" % (d["value"], d["ms_per_step"], d["detect_only_frames_per_s_per_gpu"], d["sq_fit_kernel"]["ms_per_launch"]))'
python3 bench.py --frames 1250 --objects 63 --steps 2 --warmup 1 --no-cpu-baseline --no-config4 --no-secondary 2>/dev/null | python3 -c "$P"
python3 bench.py --frames 10000 --objects 500 --steps 1 --warmup 1 --no-cpu-baseline --no-config4 --no-secondary 2>/dev/null | python3 -c "$P"
