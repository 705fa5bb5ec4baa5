#!/bin/bash
# frames/s of the bench step against the number of HIP streams (the ring kernels hold a CU's whole LDS: streams mix at workgroup
# granularity, so more streams = more CUs in different phases of different layers at any moment)
cd ${GRAFT_REPO_ROOT:-/root/repo}
P='import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d.get("config4") or {}
print("streams", d["config"].get("detector_streams"), "batch", d["config"].get("detector_batch"), "value %.1f detect-only %.1f frac %.4f" % (d["value"], d["detect_only_frames_per_s_per_gpu"], d["roofline"]["frac"]), "| config4:", c.get("workload", "")[:0], c.get("frames_per_s_detect"), (c.get("roofline") or {}).get("frac"))'
for s in 2 3 4; do python3 bench.py --streams $s --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --config4-streams $s 2>/dev/null | python3 -c "$P"; done
python3 bench.py --streams 3 --batch 38 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --config4-streams 3 --config4-batch 38 2>/dev/null | python3 -c "$P"
